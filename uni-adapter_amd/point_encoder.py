"""Uni3D point-cloud encoder on MI355X: drop-in for the reference `models/uni3d.py:Uni3D` /
`models/point_encoder.py:PointcloudEncoder` call surface (`model.eval()`,
`model.encode_pc(pc[B,N,6]) -> (B,embed_dim)`), weights taken from a reference-layout
state_dict (keys after stripping `module.`, utils/load_models.py:56-59).

Host code only prepares weights once (fusing q/k/v, interleaving the SwiGLU halves, padding K to
the GEMM granularity) and enqueues kernels of libuniadapter_hip.so; all arithmetic of the forward
runs in hand-written HIP (see include/uniadapter.h).  No CPU path.

Token layout used on device: patch token g of cloud b is row b*G+g, the CLS token of cloud b is
row Bc*G+b, so every big GEMM sees Bc*G (a multiple of 128) rows plus one thin tile of CLS rows.
"""
import ctypes as C
import math
import os
import warnings
from typing import Dict, Optional

import torch
from torch import nn

from . import lib
from .lib import GemmEpilogue

ACT_NONE, ACT_RELU, ACT_GELU, ACT_SWIGLU = 0, 1, 2, 3


class GemmProfiler:
    """Optional per-launch HIP-event timing of ua_gemm (bench.py roofline leg).  Events are
    recorded on the stream the kernel is launched on; read with `.summary()` after a sync."""

    def __init__(self):
        self.records = []

    def summary(self, kind=None):
        recs = [r for r in self.records if kind is None or r[3] == kind]
        flops = sum(r[2] for r in recs)
        ms = sum(r[0].elapsed_time(r[1]) for r in recs)
        return dict(launches=len(recs), flops=flops, ms=ms, bytes=sum(r[4] for r in recs))


PROFILER = None          # set to a GemmProfiler to time every GEMM launch
X_SPLIT_SCALE = 1.0      # power-of-two scale of the residual stream's split copy (folded LayerNorms): fp16 'hi' overflows at |x| > 65504


class SplitW:
    """A weight matrix (N,K) pre-split into the H32L32 format of ua_gemm_split (include/uniadapter.h),
    K zero-padded to a multiple of 32, scaled by a per-matrix power of two so that the fp16 'hi'
    part uses the fp16 range and the 'lo' part is a normal number."""

    def __init__(self, w):
        N, K = w.shape
        self.N, self.K = N, _round_up(K, 32)
        amax = float(w.abs().max())
        self.scale = float(min(4096.0, 2.0 ** math.floor(math.log2(30000.0 / max(amax, 1e-30)))))
        w = w.contiguous()
        self.data = torch.empty(N, self.K, device=w.device, dtype=torch.float32)      # same bytes as fp32
        lib.call('ua_split_f16', w.data_ptr(), w.stride(0), self.data.data_ptr(), self.K, N, K, self.K, self.scale,
                 lib.stream_ptr())

    def rows(self, a, b):
        """Output rows [a, b) of the matrix as a SplitW view (same storage, same scale)."""
        v = object.__new__(SplitW)
        v.N, v.K, v.scale, v.data = b - a, self.K, self.scale, self.data[a:b]
        return v


def _fold_layernorm(w, bias, gamma, beta):
    """LayerNorm folded into the Linear that consumes it (include/uniadapter.h, ua_gemm_epilogue):
    LN(a) @ w^T + bias = rstd * (a @ (gamma * w)^T - mean * colsum) + (bias + w @ beta).
    w (N, K >= len(gamma)) with zero columns beyond len(gamma).  Returns (SplitW(gamma * w), folded bias, colsum);
    colsum is taken over the values the matrix cores multiply (hi + lo of the scaled weights), so the mean term cancels
    against the product exactly as computed."""
    k0 = gamma.numel()
    wf = w.clone()
    wf[:, :k0] *= gamma
    bf = (bias.double() + w[:, :k0].double() @ beta.double()).float().contiguous()
    sw = SplitW(wf)
    q = wf * sw.scale
    hi = q.half()
    lo = (q - hi.float()).half()
    colsum = ((hi.double() + lo.double()).sum(1) / sw.scale).float().contiguous()
    return sw, bf, colsum


def _act_scale(bound):
    """Power-of-two scale for a split activation whose magnitude is bounded by `bound`."""
    return float(min(16.0, 2.0 ** math.floor(math.log2(30000.0 / max(bound, 1e-30)))))


def _gemm(A, lda, W, out, ldc, M, N, K, bias=None, scale=None, shift=None, rowgroup_add=None, ldg=0,
          group_rows=0, act=0, residual=None, ldr=0, pool_out=None, ldp=0, pool_rows=0, alg_flops=None,
          split_out=None, ld_split=0, split_scale=1.0, a_scale=None, stats_out=None, ln_mr=None, ln_colsum=None,
          ln_stats=None, ln_n=0, ln_eps=0.0):
    """C (M,N) = A (M,K) @ W (N,K)^T + epilogue; all tensors already on device.
    W is an fp32 tensor (exact-f32 MFMA kernel, A fp32) or a SplitW (3xf16-split MFMA kernel, A an
    H32L32 split matrix with scale `a_scale`).  `split_out`: also/only emit the output in split form.
    `alg_flops`: algorithmic FLOPs of the launch when K/N carry zero padding (default 2MNK)."""
    e = GemmEpilogue()
    e.split_out = split_out.data_ptr() if split_out is not None else None
    e.ld_split, e.split_scale = ld_split, split_scale
    e.bias = bias.data_ptr() if bias is not None else None
    e.scale = scale.data_ptr() if scale is not None else None
    e.shift = shift.data_ptr() if shift is not None else None
    e.rowgroup_add = rowgroup_add.data_ptr() if rowgroup_add is not None else None
    e.ldg, e.group_rows, e.act = ldg, group_rows, act
    e.residual = residual.data_ptr() if residual is not None else None
    e.ldr = ldr
    e.pool_out = pool_out.data_ptr() if pool_out is not None else None
    e.ldp, e.pool_rows = ldp, pool_rows
    if stats_out is not None:                      # (parts, rows, 2) partial row statistics of the output
        e.row_stats_out, e.ld_stats = stats_out.data_ptr(), stats_out.stride(0) // 2
    if ln_mr is not None:                          # folded LayerNorm on the A side: final (mean, rstd) per row ...
        e.ln_mean_rstd, e.ln_colsum = ln_mr.data_ptr(), ln_colsum.data_ptr()
    elif ln_stats is not None:                     # ... or the producer's partials (at most 4 tiles), reduced by the consumer
        e.ln_stats_in, e.ln_stats_ld, e.ln_parts, e.ln_n, e.ln_eps = ln_stats.data_ptr(), ln_stats.stride(0) // 2, ln_stats.shape[0], ln_n, ln_eps
        e.ln_colsum = ln_colsum.data_ptr()
    prof = PROFILER
    if prof is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    if isinstance(W, SplitW):
        lib.call('ua_gemm_split', A.data_ptr(), lda, a_scale, W.data.data_ptr(), W.K, W.scale,
                 out.data_ptr() if out is not None else None, ldc, M, N, W.K, C.byref(e), lib.stream_ptr())
        kind = 'split'
    else:
        lib.call('ua_gemm', A.data_ptr(), lda, W.data_ptr(), W.stride(0), out.data_ptr() if out is not None else None,
                 ldc, M, N, K, C.byref(e), lib.stream_ptr())
        kind = 'f32'
    if prof is not None:
        ev1.record()
        # algorithmic bytes of the launch: both operands once, every output array once, the residual once (4 bytes per element in
        # either storage form; a SwiGLU launch writes N / 2 columns)
        n_out = N // 2 if act == ACT_SWIGLU else N
        outs = (out is not None) + (split_out is not None) + (residual is not None)
        alg_bytes = 4.0 * (M * K + N * K + outs * M * n_out) + (4.0 * (M // max(1, pool_rows)) * n_out if pool_out is not None else 0.0)
        prof.records.append((ev0, ev1, alg_flops if alg_flops is not None else 2.0 * M * N * K, kind, alg_bytes))


def _row_stats(stats, n_parts, rows, n, eps, out):
    lib.call('ua_row_stats_finalize', stats.data_ptr(), n_parts, stats.stride(0) // 2, rows, n, eps, out.data_ptr(), lib.stream_ptr())


def _ln(x, ldx, w, b, eps, y, ldy, rows, cols, cols_pad, split_scale=None):
    """LayerNorm; with `split_scale` the output is written as an H32L32 split matrix."""
    if split_scale is None:
        lib.call('ua_layernorm', x.data_ptr(), ldx, w.data_ptr(), b.data_ptr(), eps, y.data_ptr(), ldy,
                 rows, cols, cols_pad, lib.stream_ptr())
    else:
        lib.call('ua_layernorm_split', x.data_ptr(), ldx, w.data_ptr(), b.data_ptr(), eps, y.data_ptr(), ldy,
                 rows, cols, cols_pad, split_scale, lib.stream_ptr())


def _round_up(x, m):
    return (x + m - 1) // m * m


class _Block:
    """Prepared weights of one transformer block (timm EvaBlock / ViT Block layouts)."""
    pass


class PointcloudEncoder(nn.Module):
    """Mirror of models/point_encoder.py:161-223 with the timm blocks folded in."""

    def __init__(self, state_dict: Dict[str, torch.Tensor], num_group: int, group_size: int, num_heads: int,
                 device=None, prefix: str = 'point_encoder.', ln_eps: float = 1e-6, bn_eps: float = 1e-5,
                 fps_start: int = 0, fps_skip_eps: float = 1e-3, pe_clouds: int = 8, precision: str = 'f32',
                 variant: str = 'uni3d', projection: Optional[torch.Tensor] = None, ball_radius: float = 0.2,
                 fold_layernorm: Optional[bool] = None):
        super().__init__()
        if not torch.cuda.is_available():
            raise RuntimeError('uni_adapter_amd.PointcloudEncoder needs a HIP device (MI355X); no CPU path')
        self.dev = torch.device(device) if device is not None else torch.device('cuda', torch.cuda.current_device())
        self.num_group, self.group_size, self.num_heads = int(num_group), int(group_size), int(num_heads)
        self.ln_eps, self.fps_start, self.fps_skip_eps = float(ln_eps), int(fps_start), float(fps_skip_eps)
        self.pe_clouds = int(os.environ.get('UA_PE_CLOUDS', pe_clouds))      # clouds per mini-PointNet slice (its per-point activations are 64 MB per cloud)
        if precision not in ('f32', 'split'):
            raise ValueError("precision must be 'f32' (exact-f32 MFMA) or 'split' (3xf16-split MFMA, fp32-accurate)")
        self.precision = precision
        sp = precision == 'split'
        # split mode: the blocks' LayerNorms are folded into the GEMMs either side of them (no LN pass over HBM);
        # UA_FOLD_LN=0 / fold_layernorm=False keeps the LN kernels (A/B aid)
        if fold_layernorm is None:
            fold_layernorm = os.environ.get('UA_FOLD_LN', '1') != '0'
        if os.environ.get('UA_SPLIT_SCHEDULE', '16') != '16':
            fold_layernorm = False      # the A/B main loops (schedules 8 and 2) have no fold prologue: ua_gemm_split returns UA_EUNSUPPORTED
        self.fold_ln = bool(fold_layernorm) and sp
        self._fold_checked = False
        self.last_block_cls_only = os.environ.get('UA_LAST_BLOCK_CLS_ONLY', '1') != '0'      # see _run_blocks
        if self.group_size not in (32, 64):
            raise NotImplementedError('patch-embed max-pool epilogue is built for group sizes 64 (Uni3D) and 32 (PointBERT)')
        if variant not in ('uni3d', 'ulip', 'openshape'):
            raise ValueError("variant must be 'uni3d', 'ulip' or 'openshape'")
        self.variant = variant
        sd = {k[len(prefix):]: v for k, v in state_dict.items() if k.startswith(prefix)}
        if variant == 'ulip':
            # PointBERT layout (models/ulip/pointbert/point_encoder.py:101-135) -> the internal key names
            ren = {}
            for k, v in sd.items():
                k2 = k.replace('reduce_dim.', 'encoder2trans.').replace('blocks.blocks.', 'visual.blocks.')
                if k2 in ('norm.weight', 'norm.bias'):
                    k2 = 'ulip_' + k2
                ren[k2] = v
            sd = ren
        if variant == 'openshape':
            # OpenShape PPAT layout (models/openshape/ppta.py:83-147) -> the internal block key names; the
            # PreNorm/Attention/FeedForward layer is a pre-LN ViT block with bias-free fused qkv and a GELU MLP
            ren = {}
            for k, v in sd.items():
                k2 = k
                if k.startswith('ppat.transformer.layers.'):
                    l, j, rest = k[len('ppat.transformer.layers.'):].split('.', 2)
                    rest = {'norm.weight': f'norm{int(j) + 1}.weight', 'norm.bias': f'norm{int(j) + 1}.bias',
                            'fn.to_qkv.weight': 'attn.qkv.weight', 'fn.to_out.0.weight': 'attn.proj.weight',
                            'fn.to_out.0.bias': 'attn.proj.bias', 'fn.net.0.weight': 'mlp.fc1.weight', 'fn.net.0.bias': 'mlp.fc1.bias',
                            'fn.net.3.weight': 'mlp.fc2.weight', 'fn.net.3.bias': 'mlp.fc2.bias'}[rest]
                    k2 = f'visual.blocks.{l}.{rest}'
                ren[k2] = v
            sd = ren
        d = lambda t: t.detach().to(self.dev, torch.float32).contiguous()

        def bn_affine(name):                       # eval BatchNorm1d -> per-channel scale/shift
            w, b = sd[name + '.weight'].double(), sd[name + '.bias'].double()
            rm, rv = sd[name + '.running_mean'].double(), sd[name + '.running_var'].double()
            s = w / torch.sqrt(rv + bn_eps)
            return d(s), d(b - rm * s)

        if variant == 'openshape':
            # ---- set abstraction MLP (pointnet_util.py:169-211) + lift (ppta.py:94) + head (ppta.py:139-147) ----
            self.ball_radius = float(ball_radius)
            self.sa = []
            i = 0
            while f'ppat.sa.mlp_convs.{i}.weight' in sd:
                wc = sd[f'ppat.sa.mlp_convs.{i}.weight'].reshape(sd[f'ppat.sa.mlp_convs.{i}.weight'].shape[0], -1).float()
                kin = wc.shape[1]
                wc = torch.nn.functional.pad(wc, (0, _round_up(kin, 4) - kin))
                sc, sh = bn_affine(f'ppat.sa.mlp_bns.{i}')
                self.sa.append((d(wc), d(sd[f'ppat.sa.mlp_convs.{i}.bias']), sc, sh))
                i += 1
            self.in_ch = sd['ppat.sa.mlp_convs.0.weight'].shape[1]                 # 3 + F
            self.sa_dim = self.sa[-1][0].shape[0]
            wl = sd['ppat.lift.0.weight'][:, :, 0].float()                          # (dim, 3 + sa_dim) on cat[centroids, feature]
            self.lift_k = _round_up(self.sa_dim + 3, 4)
            wl = torch.cat([wl[:, 3:], wl[:, :3]], 1)                               # our row layout: [feature | centroid]
            self.w_lift = d(torch.nn.functional.pad(wl, (0, self.lift_k - wl.shape[1]))); self.b_lift = d(sd['ppat.lift.0.bias'])
            self.lift_ln = (d(sd['ppat.lift.2.weight']), d(sd['ppat.lift.2.bias']))
            self.cls_token = d(sd['ppat.cls_token'].reshape(-1))
            self.trans_dim = self.w_lift.shape[0]
            self.w_t2e = d(sd['proj.weight']); self.b_t2e = d(sd['proj.bias'])
            self.embed_dim = self.w_t2e.shape[0]
            self.enc_dim = self.sa_dim
        else:
            # ---- mini-PointNet (models/point_encoder.py:133-144) ----
            w1 = sd['encoder.first_conv.0.weight'][:, :, 0]
            self.in_ch = w1.shape[1]
            self.w1 = d(torch.nn.functional.pad(w1.float(), (0, 8 - self.in_ch)))             # K padded 6 -> 8
            self.b1 = d(sd['encoder.first_conv.0.bias'])
            self.bn1_s, self.bn1_t = bn_affine('encoder.first_conv.1')
            self.w2 = d(sd['encoder.first_conv.3.weight'][:, :, 0]); self.b2 = d(sd['encoder.first_conv.3.bias'])
            self.c_loc = self.w2.shape[0]                                             # 256
            w3 = sd['encoder.second_conv.0.weight'][:, :, 0]
            self.w3g = d(w3[:, :self.c_loc]); self.w3l = d(w3[:, self.c_loc:])        # [global | local] halves
            self.b3 = d(sd['encoder.second_conv.0.bias'])
            self.bn2_s, self.bn2_t = bn_affine('encoder.second_conv.1')
            self.w4 = d(sd['encoder.second_conv.3.weight'][:, :, 0]); self.b4 = d(sd['encoder.second_conv.3.bias'])
            self.enc_dim = self.w4.shape[0]
            self.n3 = self.w3l.shape[0]
            if sp:                                    # the three big convs go through the split kernel
                self.w2, self.w3l, self.w4 = SplitW(self.w2), SplitW(self.w3l), SplitW(self.w4)
            # ---- bridges / positional MLP (:175-186) ----
            self.w_e2t = d(sd['encoder2trans.weight']); self.b_e2t = d(sd['encoder2trans.bias'])
            self.trans_dim = self.w_e2t.shape[0]
            if variant == 'uni3d':
                self.w_t2e = d(sd['trans2embed.weight']); self.b_t2e = d(sd['trans2embed.bias'])
                self.embed_dim = self.w_t2e.shape[0]
            else:
                # ULIP: pc_embed = cat[cls, max(tokens)] @ pc_projection (models/ulip/ulip_model.py:15-18)
                if projection is None:
                    raise ValueError("variant 'ulip' needs the pc_projection matrix (2*trans_dim, embed)")
                self.w_t2e = d(projection.t()); self.b_t2e = None
                self.embed_dim = self.w_t2e.shape[0]
                self.ulip_norm = (d(sd['ulip_norm.weight']), d(sd['ulip_norm.bias']))
            self.w_p0 = d(torch.nn.functional.pad(sd['pos_embed.0.weight'], (0, 1))); self.b_p0 = d(sd['pos_embed.0.bias'])
            self.w_p2 = d(sd['pos_embed.2.weight']); self.b_p2 = d(sd['pos_embed.2.bias'])
            self.cls_token = d(sd['cls_token'].reshape(-1)); self.cls_pos = d(sd['cls_pos'].reshape(-1))
        Cd = self.trans_dim
        if Cd % self.num_heads or Cd // self.num_heads != 64:
            raise NotImplementedError('attention kernel is built for head_dim 64')
        # ---- transformer blocks (timm EvaBlock, key-driven) ----
        n_blocks = 1 + max([int(k.split('.')[2]) for k in sd if k.startswith('visual.blocks.')], default=-1)
        self.blocks = []
        for i in range(n_blocks):
            p = f'visual.blocks.{i}.'
            has = lambda k: (p + k) in sd
            g = lambda k: sd[p + k].detach().to(self.dev, torch.float32)
            b = _Block()
            b.n1w, b.n1b, b.n2w, b.n2b = d(g('norm1.weight')), d(g('norm1.bias')), d(g('norm2.weight')), d(g('norm2.bias'))
            zeros = torch.zeros(Cd, device=self.dev)
            if has('attn.q_proj.weight'):
                wq, wk, wv = g('attn.q_proj.weight'), g('attn.k_proj.weight'), g('attn.v_proj.weight')
                bq = g('attn.q_proj.bias') if has('attn.q_proj.bias') else zeros
                bk = g('attn.k_proj.bias') if has('attn.k_proj.bias') else zeros
                bv = g('attn.v_proj.bias') if has('attn.v_proj.bias') else zeros
                wqkv, bqkv = torch.cat([wq, wk, wv], 0), torch.cat([bq, bk, bv], 0)
            else:
                wqkv = g('attn.qkv.weight')
                bqkv = g('attn.qkv.bias') if has('attn.qkv.bias') else torch.zeros(3 * Cd, device=self.dev)
            if has('attn.q_bias'):
                bqkv = bqkv + torch.cat([g('attn.q_bias'), zeros, g('attn.v_bias') if has('attn.v_bias') else zeros], 0)
            b.wqkv, b.bqkv = d(wqkv), d(bqkv)
            b.ln1_scale = _act_scale(math.sqrt(Cd) * float(b.n1w.abs().max()) + float(b.n1b.abs().max()))
            b.ln2_scale = _act_scale(math.sqrt(Cd) * float(b.n2w.abs().max()) + float(b.n2b.abs().max()))
            b.an_w, b.an_b = (d(g('attn.norm.weight')), d(g('attn.norm.bias'))) if has('attn.norm.weight') else (None, None)
            b.wproj, b.bproj = d(g('attn.proj.weight')), d(g('attn.proj.bias'))
            b.gamma1 = d(g('gamma_1')) if has('gamma_1') else None
            b.gamma2 = d(g('gamma_2')) if has('gamma_2') else None
            if has('mlp.fc1_g.weight'):                                  # SwiGLU: interleave 32-col blocks
                wg, wx = g('mlp.fc1_g.weight'), g('mlp.fc1_x.weight')
                bg, bx = g('mlp.fc1_g.bias'), g('mlp.fc1_x.bias')
                H = wg.shape[0]; Hp = _round_up(H, 32)
                padr = lambda t: torch.nn.functional.pad(t, (0, 0, 0, Hp - H)) if t.dim() == 2 else torch.nn.functional.pad(t, (0, Hp - H))
                wg, wx, bg, bx = padr(wg), padr(wx), padr(bg), padr(bx)
                b.w1 = d(torch.stack([wg.reshape(Hp // 32, 32, Cd), wx.reshape(Hp // 32, 32, Cd)], 1).reshape(2 * Hp, Cd))
                b.b1 = d(torch.stack([bg.reshape(Hp // 32, 32), bx.reshape(Hp // 32, 32)], 1).reshape(2 * Hp))
                b.act = ACT_SWIGLU
            else:                                                         # plain GELU MLP
                w1_, b1_ = g('mlp.fc1.weight'), g('mlp.fc1.bias')
                H = w1_.shape[0]; Hp = _round_up(H, 32)
                b.w1 = d(torch.nn.functional.pad(w1_, (0, 0, 0, Hp - H))); b.b1 = d(torch.nn.functional.pad(b1_, (0, Hp - H)))
                b.act = ACT_GELU
            b.H, b.Hp = H, Hp
            b.mn_w, b.mn_b = (d(g('mlp.norm.weight')), d(g('mlp.norm.bias'))) if has('mlp.norm.weight') else (None, None)
            b.w2 = d(torch.nn.functional.pad(g('mlp.fc2.weight'), (0, Hp - H))); b.b2 = d(g('mlp.fc2.bias'))
            b.mn_scale = _act_scale(math.sqrt(H) * float(b.mn_w.abs().max()) + float(b.mn_b.abs().max())) if b.mn_w is not None else 1.0
            if self.fold_ln:
                # norm1 -> qkv (not for block 0, whose input no GEMM produced, nor for ULIP, which adds pos in between),
                # norm2 -> fc1, ffn sub-LN -> fc2
                b.f_qkv = _fold_layernorm(b.wqkv, b.bqkv, b.n1w, b.n1b) if (i > 0 and variant != 'ulip') else None
                b.f_w1 = _fold_layernorm(b.w1, b.b1, b.n2w, b.n2b)
                b.f_w2 = _fold_layernorm(b.w2, b.b2, b.mn_w, b.mn_b) if b.mn_w is not None else None
                # static bound of the un-normalised hidden activation act(a)[*b]: |a_n| <= |LN(x)|_2 |w_n|_2 + |bias_n|
                lnb = math.sqrt(Cd) * float(b.n2w.abs().max()) + float(b.n2b.norm())
                ab = lnb * float(b.w1.norm(dim=1).max()) + float(b.b1.abs().max())
                b.hraw_scale = _act_scale(ab * ab if b.act == ACT_SWIGLU else ab)
            if sp:
                b.wqkv, b.wproj, b.w1, b.w2 = SplitW(b.wqkv), SplitW(b.wproj), SplitW(b.w1), SplitW(b.w2)
            self.blocks.append(b)
        self.norm = (d(sd['visual.norm.weight']), d(sd['visual.norm.bias'])) if 'visual.norm.weight' in sd else None
        self.fc_norm = (d(sd['visual.fc_norm.weight']), d(sd['visual.fc_norm.bias'])) if 'visual.fc_norm.weight' in sd else None
        self._ws = {}
        self.ws_slot = 0          # forwards enqueued on different streams must use different workspace slots

    # ------------------------------------------------------------------ workspace
    def _workspace(self, Bc):
        key = (Bc, self.ws_slot)
        ws = self._ws.get(key)
        if ws is not None:
            return ws
        G, S, Cd = self.num_group, self.group_size, self.trans_dim
        pc = min(self.pe_clouds, Bc)
        Mp = pc * G * S
        rows = Bc * G + Bc
        Hp = max(b.Hp for b in self.blocks) if self.blocks else 32
        e = lambda *s: torch.empty(*s, device=self.dev, dtype=torch.float32)
        if self.variant == 'openshape':
            kin = self.sa[0][0].shape[1]
            ws = dict(centers=e(Bc, G, 3), feat=e(Mp, kin), liftin=torch.zeros(Bc * G, self.lift_k, device=self.dev),
                      tokc=e(Bc * G, Cd), x=e(rows, Cd), h=e(rows, Cd), qkv=e(rows, 3 * Cd), att=e(rows, Cd),
                      hid=e(rows, Hp), hidn=e(rows, Hp))
            for i, (wc, _, _, _) in enumerate(self.sa[:-1]):
                ws[f'sa{i}'] = e(Mp, wc.shape[0])
            self._fold_workspace(ws, rows, Cd, Hp)
            self._ws[key] = ws
            return ws
        ws = dict(
            centers=e(Bc, G, 3), centers4=torch.zeros(Bc * G, 4, device=self.dev), feat=e(Mp, 8),
            h1=e(Mp, 128), loc=e(Mp, self.c_loc), glob=e(pc * G, self.c_loc), g3=e(pc * G, self.n3),
            h3=e(Mp, self.n3), tok=e(Bc * G, self.enc_dim), tokc=e(Bc * G, Cd), posh=e(Bc * G, 128),
            pos=e(Bc * G, Cd), x=e(rows, Cd), h=e(rows, Cd), qkv=e(rows, 3 * Cd), att=e(rows, Cd),
            hid=e(rows, Hp), hidn=e(rows, Hp), cls=e(Bc, Cd), cls2=e(Bc, Cd), out=e(Bc, self.embed_dim))
        self._fold_workspace(ws, rows, Cd, Hp)
        self._ws[key] = ws
        return ws

    def _fold_workspace(self, ws, rows, Cd, Hp):
        """Buffers of the folded LayerNorms: the residual stream in split form, partial and final row statistics."""
        if not self.fold_ln:
            return
        e = lambda *s: torch.empty(*s, device=self.dev, dtype=torch.float32)
        ws.update(xs=e(rows, Cd), st_x=e((Cd + 255) // 256, rows, 2), st_h=e((2 * Hp + 255) // 256, rows, 2),
                  mr_x=e(rows, 2), mr_h=e(rows, 2))

    FOLD_MAX_ABS = 3.0e4          # fp16 'hi' of the residual stream's split copy overflows at 65504; keep a factor 2 of headroom
    FOLD_MAX_MEAN2_OVER_VAR = 1.0e4   # one-pass (sum, sumsq) statistics in fp32 lose log10(1 + mean^2/var) of their 7.2 digits

    @torch.no_grad()
    def _fold_probe(self, w, rows, Cd, b, x, have_x, have_h, r0=0):
        """First forward only: magnitude of the residual stream after block `b` and mean^2/var of the token rows whose one-pass
        statistics this block wrote (x after proj / fc2, the hidden activation after fc1).  Returns (amax |x| * scale, worst ratio)."""
        worst = 0.0
        for key, n, have in (('st_x', Cd, have_x), ('st_h', b.H, have_h)):
            st = w.get(key)
            if st is None or not have:
                continue
            parts = (Cd + 255) // 256 if key == 'st_x' else ((2 if b.act == ACT_SWIGLU else 1) * b.Hp + 255) // 256
            tot = st[:parts, (r0 if key == 'st_h' else 0):rows].double().sum(0)      # (the last block may have produced only its CLS rows)
            mean = tot[:, 0] / n
            var = (tot[:, 1] / n - mean * mean).clamp_min(1e-30)
            worst = max(worst, float((mean * mean / var).max()))
        return float(x[r0:rows].abs().max()) * X_SPLIT_SCALE, worst

    def _fold_verdict(self, amax, worst):
        """Once per encoder, from the statistics of EVERY block of the first forward: keep the folded LayerNorms only if the split
        copy of the residual stream cannot overflow fp16 and no token row is so far from zero mean that its one-pass variance is
        noise; otherwise fall back to the LayerNorm kernels.  Returns True if the fold stays on."""
        self._fold_checked = True
        self.fold_mean2_over_var, self.fold_max_abs = worst, amax
        if worst > 100.0:
            warnings.warn(f'folded LayerNorm: a token row has mean^2/var = {worst:.3g} (> 100); its one-pass statistics keep '
                          f'{7.2 - math.log10(1 + worst):.1f} of 7.2 digits')
        ok = amax < self.FOLD_MAX_ABS and worst <= self.FOLD_MAX_MEAN2_OVER_VAR and math.isfinite(amax) and math.isfinite(worst)
        if not ok:
            warnings.warn(f'folded LayerNorm switched off for this encoder: max |x| * scale = {amax:.3g} (limit {self.FOLD_MAX_ABS:.0e}), '
                          f'mean^2/var = {worst:.3g} (limit {self.FOLD_MAX_MEAN2_OVER_VAR:.0e}); running the LayerNorm kernels instead')
            self.fold_ln = False
        return ok

    def _run_blocks(self, w, Bc, cls_only_last=False):
        """The transformer blocks on the token rows of w['x'] (in place); returns w['x'].
        cls_only_last: the caller reads only the CLS rows of the result (Uni3D: `norm(x[:, 0])`, models/point_encoder.py:219; OpenShape:
        the CLS token).  The last block's outputs for the G patch tokens of a cloud are then dead values - the reference computes and
        drops them - so that block projects K and V for all rows (the CLS query attends to every token) but q, the attention, proj and
        the MLP only for the Bc CLS rows: 11.8 of the 338 GFLOP of a Uni3D-L forward are not executed, the CLS features keep their bits
        (a GEMM output element does not depend on where its tile lies).  Both GEMM modes; UA_LAST_BLOCK_CLS_ONLY=0 runs the block in full."""
        G, S, Cd, Hh = self.num_group, self.group_size, self.trans_dim, self.num_heads
        sp = self.precision == 'split'
        st = lib.stream_ptr()
        Mg = Bc * G
        # ---- transformer blocks (:217-218) ----
        rows = Mg + Bc
        x = w['x']
        first = True
        nb = len(self.blocks)
        probing = self.fold_ln and not self._fold_checked and nb > 0
        if probing:                                   # first forward of a folded encoder: keep the input for a possible re-run
            x_in, amax, worst = x[:rows].clone(), 0.0, 0.0
        for bi, b in enumerate(self.blocks):
            if self.variant == 'ulip' and not first:      # x = block(x + pos) in every block (pointbert/point_encoder.py:95-98)
                lib.call('ua_add_pos', x.data_ptr(), w['pos'].data_ptr(), self.cls_pos.data_ptr(), Bc, G, Cd, st)
            first = False
            fold = self.fold_ln
            # r0 / nr: first row and number of rows this block produces (everything after the K / V projection works on row views)
            cls_only = bool(cls_only_last) and bi == nb - 1 and self.variant != 'ulip' and self.last_block_cls_only
            r0 = Mg if cls_only else 0
            nr = rows - r0
            R = (lambda t: t[r0:]) if cls_only else (lambda t: t)            # row view of a (rows, ...) workspace array
            P = (lambda t: t[:, r0:]) if cls_only else (lambda t: t)         # row view of a (parts, rows, 2) statistics array
            xr = R(x)
            # split mode: every GEMM operand is an H32L32 pair written directly by its producer (attention, a GEMM epilogue or -
            # without the folded LayerNorms - the LN kernel); x stays fp32, q/k/v and the hidden activation exist only in split form.
            # Folded LayerNorms: the GEMM that produces x (or the hidden activation) also writes it in split form with per-row
            # (sum, sumsq) partials; the consuming GEMM multiplies the un-normalised rows by gamma (.) W and applies mean / rstd
            # in its epilogue (include/uniadapter.h, ua_gemm_epilogue), so no LayerNorm pass over HBM is left.
            fq = b.f_qkv if fold else None
            # statistics of x: up to 4 column tiles are reduced by the consuming GEMM itself, more by ua_row_stats_finalize
            x_parts = (Cd + 255) // 256
            xstat_all = xstat = {}
            if fold:
                if x_parts <= 4:
                    xstat_all, xstat = dict(ln_stats=w['st_x'], ln_n=Cd, ln_eps=self.ln_eps), dict(ln_stats=P(w['st_x']), ln_n=Cd, ln_eps=self.ln_eps)
                else:
                    xstat_all, xstat = dict(ln_mr=w['mr_x']), dict(ln_mr=R(w['mr_x']))
            if fq is None:
                s1 = b.ln1_scale if sp else None
                _ln(x, Cd, b.n1w, b.n1b, self.ln_eps, w['h'], Cd, rows, Cd, Cd, split_scale=s1)     # (all rows: K and V need them)
            if not sp:
                # (cls_only: k and v - output columns Cd..3Cd - for all rows, q - columns 0..Cd - for the CLS rows)
                for (ra, nrow, c0, c1) in (((0, rows, Cd, 3 * Cd), (Mg, Bc, 0, Cd)) if cls_only else ((0, rows, 0, 3 * Cd),)):
                    _gemm(w['h'][ra:], Cd, b.wqkv[c0:c1], w['qkv'][ra:, c0:], 3 * Cd, nrow, c1 - c0, Cd, bias=b.bqkv[c0:c1])
                lib.call('ua_attention_cls' if cls_only else 'ua_attention', w['qkv'].data_ptr(), 3 * Cd, w['att'].data_ptr(), Cd, Bc, G, Hh, Cd // Hh, st)
                att, s_att = w['att'], None
                if b.an_w is not None:
                    _ln(R(att), Cd, b.an_w, b.an_b, self.ln_eps, R(w['h']), Cd, nr, Cd, Cd)
                    att = w['h']
            else:
                # q/k/v only ever exist as a split matrix; attention writes its result in split form too
                # (cls_only: k and v - output columns Cd..3Cd - for all rows, q - columns 0..Cd - for the CLS rows)
                jobs = ((0, rows, Cd, 3 * Cd), (Mg, Bc, 0, Cd)) if cls_only else ((0, rows, 0, 3 * Cd),)
                for (ra, nrow, c0, c1) in jobs:
                    qv = w['qkv'][ra:, c0:]
                    if fq is None:
                        _gemm(w['h'][ra:], Cd, b.wqkv.rows(c0, c1), None, 3 * Cd, nrow, c1 - c0, Cd, bias=b.bqkv[c0:c1], a_scale=s1,
                              split_out=qv, ld_split=3 * Cd, split_scale=1.0)
                    else:
                        xs_kw = xstat_all if ra == 0 else xstat
                        _gemm(w['xs'][ra:], Cd, fq[0].rows(c0, c1), None, 3 * Cd, nrow, c1 - c0, Cd, bias=fq[1][c0:c1], a_scale=X_SPLIT_SCALE,
                              split_out=qv, ld_split=3 * Cd, split_scale=1.0, ln_colsum=fq[2][c0:c1], **xs_kw)
                attn_entry = 'ua_attention_split_cls' if cls_only else 'ua_attention_split'
                if b.an_w is None:
                    lib.call(attn_entry, w['qkv'].data_ptr(), 3 * Cd, 1.0, None, 0, w['h'].data_ptr(), Cd, 1.0,
                             Bc, G, Hh, Cd // Hh, st)
                    att, s_att = w['h'], 1.0
                else:
                    lib.call(attn_entry, w['qkv'].data_ptr(), 3 * Cd, 1.0, w['att'].data_ptr(), Cd, None, 0, 1.0,
                             Bc, G, Hh, Cd // Hh, st)
                    s_att = _act_scale(math.sqrt(Cd) * float(b.an_w.abs().max()) + float(b.an_b.abs().max()))
                    _ln(R(w['att']), Cd, b.an_w, b.an_b, self.ln_eps, R(w['h']), Cd, nr, Cd, Cd, split_scale=s_att)
                    att = w['h']
            Hp = b.Hp
            ld_h = w['hid'].stride(0)
            if fold:
                _gemm(R(att), Cd, b.wproj, xr, Cd, nr, Cd, Cd, bias=b.bproj, scale=b.gamma1, residual=xr, ldr=Cd, a_scale=s_att,
                      split_out=R(w['xs']), ld_split=Cd, split_scale=X_SPLIT_SCALE, stats_out=P(w['st_x']))
                if x_parts > 4:
                    _row_stats(P(w['st_x']), x_parts, nr, Cd, self.ln_eps, R(w['mr_x']))
                n1 = 2 * Hp if b.act == ACT_SWIGLU else Hp
                fl1 = (4.0 if b.act == ACT_SWIGLU else 2.0) * nr * b.H * Cd
                f2 = b.f_w2
                if f2 is not None:
                    # the hidden activation leaves the fc1 GEMM un-normalised, in split form, with its row statistics
                    _gemm(R(w['xs']), Cd, b.f_w1[0], None, ld_h, nr, n1, Cd, bias=b.f_w1[1], act=b.act, alg_flops=fl1, a_scale=X_SPLIT_SCALE,
                          ln_colsum=b.f_w1[2], split_out=R(w['hidn']), ld_split=ld_h, split_scale=b.hraw_scale,
                          stats_out=P(w['st_h']), **xstat)
                    _row_stats(P(w['st_h']), (n1 + 255) // 256, nr, b.H, self.ln_eps, R(w['mr_h']))
                    w2, b2, kw = f2[0], f2[1], dict(ln_mr=R(w['mr_h']), ln_colsum=f2[2])
                else:
                    _gemm(R(w['xs']), Cd, b.f_w1[0], None, ld_h, nr, n1, Cd, bias=b.f_w1[1], act=b.act, alg_flops=fl1, a_scale=X_SPLIT_SCALE,
                          ln_colsum=b.f_w1[2], split_out=R(w['hidn']), ld_split=ld_h, split_scale=b.hraw_scale, **xstat)
                    w2, b2, kw = b.w2, b.b2, {}
                last = bi == nb - 1 or self.blocks[bi + 1].f_qkv is None      # the next block runs its own norm1 kernel on x
                if not last:
                    kw.update(split_out=w['xs'], ld_split=Cd, split_scale=X_SPLIT_SCALE, stats_out=w['st_x'])
                _gemm(R(w['hidn']), ld_h, w2, xr, Cd, nr, Cd, Hp, bias=b2, scale=b.gamma2, residual=xr, ldr=Cd,
                      alg_flops=2.0 * nr * Cd * b.H, a_scale=b.hraw_scale, **kw)
                if not last and x_parts > 4:
                    _row_stats(w['st_x'], x_parts, rows, Cd, self.ln_eps, w['mr_x'])
                if probing:
                    a, r = self._fold_probe(w, rows, Cd, b, x, not last, f2 is not None, r0)
                    amax, worst = max(amax, a), max(worst, r)
                continue
            _gemm(R(att), Cd, b.wproj, xr, Cd, nr, Cd, Cd, bias=b.bproj, scale=b.gamma1, residual=xr, ldr=Cd, a_scale=s_att)
            s2 = b.ln2_scale if sp else None
            _ln(xr, Cd, b.n2w, b.n2b, self.ln_eps, R(w['h']), Cd, nr, Cd, Cd, split_scale=s2)
            if b.act == ACT_SWIGLU:
                _gemm(R(w['h']), Cd, b.w1, R(w['hid']), ld_h, nr, 2 * Hp, Cd, bias=b.b1, act=ACT_SWIGLU, alg_flops=4.0 * nr * b.H * Cd, a_scale=s2)
            else:
                _gemm(R(w['h']), Cd, b.w1, R(w['hid']), ld_h, nr, Hp, Cd, bias=b.b1, act=ACT_GELU, alg_flops=2.0 * nr * b.H * Cd, a_scale=s2)
            hid, s_h = w['hid'], None
            if b.mn_w is not None:
                s_h = b.mn_scale if sp else None
                _ln(R(w['hid']), ld_h, b.mn_w, b.mn_b, self.ln_eps, R(w['hidn']), ld_h, nr, b.H, Hp, split_scale=s_h)
                hid = w['hidn']
            elif sp:
                lib.call('ua_split_f16', R(hid).data_ptr(), ld_h, R(w['hidn']).data_ptr(), ld_h, nr, b.H, Hp, 1.0, st)
                hid, s_h = w['hidn'], 1.0
            _gemm(R(hid), ld_h, b.w2, xr, Cd, nr, Cd, Hp, bias=b.b2, scale=b.gamma2, residual=xr, ldr=Cd, alg_flops=2.0 * nr * Cd * b.H, a_scale=s_h)
        if probing and not self._fold_verdict(amax, worst):
            x[:rows].copy_(x_in)                        # the fold is off now: run the blocks again on the LayerNorm kernels
            return self._run_blocks(w, Bc, cls_only_last)
        return x

    @torch.no_grad()
    def _forward_openshape(self, pts, feats, w, Bc, N, return_tokens):
        """OpenShape PPAT (models/openshape/ppta.py:113-147): ball-query set abstraction -> lift + LayerNorm ->
        [tokens, cls] -> pre-LN blocks -> Linear on the CLS token."""
        G, S, Cd = self.num_group, self.group_size, self.trans_dim
        st = lib.stream_ptr()
        F_ = feats.shape[2]
        if 3 + F_ != self.in_ch:
            raise ValueError(f'point features have {F_} channels, the SA layer expects {self.in_ch - 3}')
        kin = self.sa[0][0].shape[1]
        pc = min(self.pe_clouds, Bc)
        for c0 in range(0, Bc, pc):
            nb = min(pc, Bc - c0)
            Mp = nb * G * S
            lib.call('ua_ball_group', pts[c0:].data_ptr(), feats[c0:].data_ptr(), w['centers'][c0:].data_ptr(), nb, N, G, S, F_,
                     self.ball_radius, w['feat'].data_ptr(), kin, None, st)
            cur, ldcur, kcur = w['feat'], kin, kin
            for i, (wc, bc, sc, sh) in enumerate(self.sa):                    # conv1x1 + BN + ReLU (pointnet_util.py:203-205)
                co = wc.shape[0]
                if i + 1 < len(self.sa):
                    _gemm(cur, ldcur, wc, w[f'sa{i}'], co, Mp, co, kcur, bias=bc, scale=sc, shift=sh, act=ACT_RELU)
                    cur, ldcur, kcur = w[f'sa{i}'], co, co
                else:                                                          # last layer: only the max over the group is kept (:207)
                    _gemm(cur, ldcur, wc, None, co, Mp, co, kcur, bias=bc, scale=sc, shift=sh, act=ACT_RELU,
                          pool_out=w['liftin'][c0 * G:], ldp=self.lift_k, pool_rows=S)
        Mg = Bc * G
        w['liftin'][:, self.sa_dim:self.sa_dim + 3].copy_(w['centers'].view(-1, 3))          # rows = [feature | centroid]
        _gemm(w['liftin'], self.lift_k, self.w_lift, w['tokc'], Cd, Mg, Cd, self.lift_k, bias=self.b_lift,
              alg_flops=2.0 * Mg * Cd * (self.sa_dim + 3))
        x = w['x']
        _ln(w['tokc'], Cd, self.lift_ln[0], self.lift_ln[1], 1e-5, x, Cd, Mg, Cd, Cd)        # nn.LayerNorm([dim]) of the lift
        x[Mg:].copy_(self.cls_token.expand(Bc, Cd))                                           # cls token rows (no positional term)
        self._run_blocks(w, Bc, cls_only_last=not return_tokens)
        out = torch.empty(Bc, self.embed_dim, device=self.dev, dtype=torch.float32)
        _gemm(x[Mg:], Cd, self.w_t2e, out, self.embed_dim, Bc, self.embed_dim, Cd, bias=self.b_t2e)      # proj(cls) (:141-142)
        if return_tokens:
            return out, torch.cat([x[Mg:].view(Bc, 1, Cd), x[:Mg].view(Bc, G, Cd)], 1)
        return out

    # ------------------------------------------------------------------ forward
    @torch.no_grad()
    def forward(self, pts, colors=None, return_tokens=False):
        """pts, colors: (Bc,N,3) fp32 on device -> (Bc, embed_dim)  (models/point_encoder.py:192-223;
        variant 'ulip': xyz only, models/ulip/pointbert/point_encoder.py:173-192)."""
        Bc, N, _ = pts.shape
        G, S, Cd, Hh = self.num_group, self.group_size, self.trans_dim, self.num_heads
        pts = pts.to(self.dev, torch.float32).contiguous()
        if self.variant in ('uni3d', 'openshape'):
            colors = colors.to(self.dev, torch.float32).contiguous()        # openshape: the (Bc,N,F) point features
        else:
            colors = None
        w = self._workspace(Bc)
        st = lib.stream_ptr()
        # ---- grouping (:108-127) ----
        lib.call('ua_fps', pts.data_ptr(), Bc, N, G, self.fps_start, self.fps_skip_eps, None, w['centers'].data_ptr(), st)
        if self.variant == 'openshape':
            return self._forward_openshape(pts, colors, w, Bc, N, return_tokens)
        w['centers4'][:, :3].copy_(w['centers'].view(-1, 3))
        # ---- mini-PointNet per chunk of clouds (:145-159) ----
        sp = self.precision == 'split'
        pc = min(self.pe_clouds, Bc)
        for c0 in range(0, Bc, pc):
            nb = min(pc, Bc - c0)
            Mp, Mg = nb * G * S, nb * G
            lib.call('ua_knn_group', pts[c0:].data_ptr(), colors[c0:].data_ptr() if colors is not None else None, w['centers'][c0:].data_ptr(),
                     nb, N, G, S, w['feat'].data_ptr(), 8, None, st)
            n3 = self.n3
            if not sp:
                _gemm(w['feat'], 8, self.w1, w['h1'], 128, Mp, 128, 8, bias=self.b1, scale=self.bn1_s, shift=self.bn1_t, act=ACT_RELU,
                      alg_flops=2.0 * Mp * 128 * self.in_ch)
                _gemm(w['h1'], 128, self.w2, w['loc'], self.c_loc, Mp, self.c_loc, 128, bias=self.b2,
                      pool_out=w['glob'], ldp=self.c_loc, pool_rows=S)
                _gemm(w['glob'], self.c_loc, self.w3g, w['g3'], n3, Mg, n3, self.c_loc)
                _gemm(w['loc'], self.c_loc, self.w3l, w['h3'], n3, Mp, n3, self.c_loc, bias=self.b3, rowgroup_add=w['g3'], ldg=n3,
                      group_rows=S, scale=self.bn2_s, shift=self.bn2_t, act=ACT_RELU)
                _gemm(w['h3'], n3, self.w4, None, self.enc_dim, Mp, self.enc_dim, n3, bias=self.b4,
                      pool_out=w['tok'][c0 * G:], ldp=self.enc_dim, pool_rows=S)
            else:
                # same chain; the per-point activations h1 / loc / h3 only ever exist in split form
                _gemm(w['feat'], 8, self.w1, None, 128, Mp, 128, 8, bias=self.b1, scale=self.bn1_s, shift=self.bn1_t, act=ACT_RELU,
                      alg_flops=2.0 * Mp * 128 * self.in_ch, split_out=w['h1'], ld_split=128)
                _gemm(w['h1'], 128, self.w2, None, self.c_loc, Mp, self.c_loc, 128, bias=self.b2, a_scale=1.0,
                      pool_out=w['glob'], ldp=self.c_loc, pool_rows=S, split_out=w['loc'], ld_split=self.c_loc)
                _gemm(w['glob'], self.c_loc, self.w3g, w['g3'], n3, Mg, n3, self.c_loc)
                _gemm(w['loc'], self.c_loc, self.w3l, None, n3, Mp, n3, self.c_loc, bias=self.b3, rowgroup_add=w['g3'], ldg=n3,
                      group_rows=S, scale=self.bn2_s, shift=self.bn2_t, act=ACT_RELU, a_scale=1.0, split_out=w['h3'], ld_split=n3)
                _gemm(w['h3'], n3, self.w4, None, self.enc_dim, Mp, self.enc_dim, n3, bias=self.b4, a_scale=1.0,
                      pool_out=w['tok'][c0 * G:], ldp=self.enc_dim, pool_rows=S)
        # ---- token assembly (:198-208) ----
        Mg = Bc * G
        _gemm(w['tok'], self.enc_dim, self.w_e2t, w['tokc'], Cd, Mg, Cd, self.enc_dim, bias=self.b_e2t)
        _gemm(w['centers4'], 4, self.w_p0, w['posh'], 128, Mg, 128, 4, bias=self.b_p0, act=ACT_GELU, alg_flops=2.0 * Mg * 128 * 3)
        _gemm(w['posh'], 128, self.w_p2, w['pos'], Cd, Mg, Cd, 128, bias=self.b_p2)
        lib.call('ua_assemble_tokens', w['tokc'].data_ptr(), w['pos'].data_ptr(), self.cls_token.data_ptr(),
                 self.cls_pos.data_ptr(), w['x'].data_ptr(), Bc, G, Cd, st)
        x = self._run_blocks(w, Bc, cls_only_last=not return_tokens)
        Mg = Bc * G
        rows = Mg + Bc
        out = torch.empty(Bc, self.embed_dim, device=self.dev, dtype=torch.float32)
        if self.variant == 'uni3d':
            # ---- head (:219-222) on the CLS rows ----
            cls = x[Mg:]
            cur, nxt = cls, w['cls']
            for nrm in (self.norm, self.fc_norm):
                if nrm is not None:
                    _ln(cur, Cd, nrm[0], nrm[1], self.ln_eps, nxt, Cd, Bc, Cd, Cd)
                    cur, nxt = nxt, (w['cls2'] if nxt is w['cls'] else w['cls'])
            _gemm(cur, Cd, self.w_t2e, out, self.embed_dim, Bc, self.embed_dim, Cd, bias=self.b_t2e)
        else:
            # ---- PointBERT head: LN on all tokens, cat[cls, max over patch tokens], @ pc_projection ----
            _ln(x, Cd, self.ulip_norm[0], self.ulip_norm[1], self.ln_eps, w['att'], Cd, rows, Cd, Cd)
            cat = torch.empty(Bc, 2 * Cd, device=self.dev, dtype=torch.float32)
            lib.call('ua_cls_max_concat', w['att'].data_ptr(), cat.data_ptr(), Bc, G, Cd, st)
            _gemm(cat, 2 * Cd, self.w_t2e, out, self.embed_dim, Bc, self.embed_dim, 2 * Cd)
        if return_tokens:
            # reference token order: [cls, patch tokens] per cloud
            tokens = torch.cat([x[Mg:].view(Bc, 1, Cd), x[:Mg].view(Bc, G, Cd)], 1)
            return out, tokens
        return out

    def flops_per_cloud(self):
        """Algorithmic FLOPs of one forward (SURVEY section 8d bookkeeping, reference formulation)."""
        G, S, Cd = self.num_group, self.group_size, self.trans_dim
        T = G + 1
        if self.variant == 'openshape':
            chans = [self.in_ch] + [wc.shape[0] for wc, _, _, _ in self.sa]
            f = G * S * 2 * sum(a * b for a, b in zip(chans[:-1], chans[1:])) + G * 2 * (self.sa_dim + 3) * Cd
        else:
            f = G * S * 2 * (self.in_ch * 128 + 128 * self.c_loc + (2 * self.c_loc) * self.n3 + self.n3 * self.enc_dim)
            f += G * 2 * (self.enc_dim * Cd + 3 * 128 + 128 * Cd)
        for b in self.blocks:
            f += 2 * T * Cd * Cd * 4 + 4 * T * T * Cd
            f += 2 * T * Cd * b.H * (3 if b.act == ACT_SWIGLU else 2)
        f += 2 * Cd * self.embed_dim * (2 if self.variant == 'ulip' else 1)
        return f


class Uni3D(nn.Module):
    """Mirror of models/uni3d.py:9-28 (`encode_pc`)."""

    def __init__(self, point_encoder: PointcloudEncoder, logit_scale: Optional[torch.Tensor] = None):
        super().__init__()
        self.point_encoder = point_encoder
        self.logit_scale = logit_scale if logit_scale is not None else torch.ones([]) * math.log(1 / 0.07)

    def encode_pc(self, pc):
        xyz = pc[:, :, :3].contiguous()
        color = pc[:, :, 3:].contiguous()
        return self.point_encoder(xyz, color)

    def forward(self, pc, text, image):
        return {'text_embed': text, 'pc_embed': self.encode_pc(pc), 'image_embed': image,
                'logit_scale': self.logit_scale.exp()}


def create_uni3d_from_state_dict(state_dict, args=None, num_group=None, group_size=None, num_heads=None, **kw):
    """Build the model from a Uni3D checkpoint's `module` dict (utils/load_models.py:53-61) or a
    synthetic state_dict of the same layout.  `args` follows utils/params.py:35-43."""
    sd = {(k[len('module.'):] if k.startswith('module.') else k): v for k, v in state_dict.items()}
    num_group = num_group or getattr(args, 'num_group', None) or getattr(args, 'num_group_uni3d', 512)
    group_size = group_size or getattr(args, 'group_size', None) or getattr(args, 'group_size_uni3d', 64)
    trans_dim = sd['point_encoder.encoder2trans.weight'].shape[0]
    num_heads = num_heads or trans_dim // 64
    enc = PointcloudEncoder(sd, num_group, group_size, num_heads, **kw)
    return Uni3D(enc, sd.get('logit_scale'))


class ULIP(nn.Module):
    """Mirror of models/ulip/ulip_model.py:6-18: `model(xyz[B,N,3]) -> (B, embed)`."""

    def __init__(self, point_encoder: PointcloudEncoder):
        super().__init__()
        self.point_encoder = point_encoder

    def forward(self, pc):
        return self.point_encoder(pc[:, :, :3].contiguous())

    def encode_pc(self, pc):
        """(B,N,6) harness batches: xyz only, as Uni_Adapter.py:58-60 slices them (lets TtaEngine drive any encoder)."""
        return self.forward(pc)


def create_ulip_from_state_dict(state_dict, num_group=512, group_size=32, num_heads=6, **kw):
    """ULIP-2 PointBERT (`models/ulip/pointbert/PointTransformer_8192point.yaml:15-25`) from a
    checkpoint-layout dict: `point_encoder.*` keys of PointTransformer + `pc_projection`."""
    sd = {(k[len('module.'):] if k.startswith('module.') else k): v for k, v in state_dict.items()}
    kw.setdefault('ln_eps', 1e-5)                  # nn.LayerNorm default (point_encoder.py:40,135)
    kw.setdefault('fps_skip_eps', -1.0)            # pure-torch FPS of models/ulip/pointbert/misc.py:40-60
    enc = PointcloudEncoder(sd, num_group, group_size, num_heads, variant='ulip', projection=sd['pc_projection'], **kw)
    return ULIP(enc)


class OpenShape(nn.Module):
    """Mirror of models/openshape/ppta.py:135-142 (`Projected`): `model(xyz[B,N,3], features[B,N,F]) -> (B, out_channel)`
    as called at Uni_Adapter.py:63-66."""

    def __init__(self, point_encoder: PointcloudEncoder):
        super().__init__()
        self.point_encoder = point_encoder

    def forward(self, xyz, features):
        return self.point_encoder(xyz[:, :, :3].contiguous(), features.contiguous())

    def encode_pc(self, pc):
        """(B,N,6) harness batches: model(xyz, feature) as at Uni_Adapter.py:63-65."""
        return self.forward(pc[:, :, :3], pc)


def create_openshape_from_state_dict(state_dict, patches=384, nsample=64, radius=0.2, num_heads=8, **kw):
    """OpenShape PointBERT 'scaling 4' (models/openshape/ppta.py:172-177, config.yaml) from a state dict in the key layout of
    `Projected(PointPatchTransformer, nn.Linear)`: `ppat.sa.*`, `ppat.lift.*`, `ppat.cls_token`, `ppat.transformer.layers.*`,
    `proj.*`.  The reference's own loader is a stub (utils/load_models.py:108-127), so this factory defines the wiring."""
    sd = {(k[len('module.'):] if k.startswith('module.') else k): v for k, v in state_dict.items()}
    kw.setdefault('ln_eps', 1e-5)                  # nn.LayerNorm default
    kw.setdefault('fps_skip_eps', -1.0)            # pure-torch FPS (pointnet_util.py:64-86; random start: pass fps_start)
    enc = PointcloudEncoder(sd, patches, nsample, num_heads, variant='openshape', prefix='', ball_radius=radius, **kw)
    return OpenShape(enc)
