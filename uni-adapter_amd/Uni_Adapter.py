"""Test-time adaptation harness on MI355X: drop-in for the reference `Uni_Adapter.py`
(`test_zeroshot_3d_core`, `get_logits_wrapper`, `softmax_entropy`, `compute_text_alignment_loss`)
for the `--use-mode-dota` path (the only one reachable from the reference CLI).

How the loop is organised on the device (same arithmetic per sample as Uni_Adapter.py:368-579):
  * the encoder is stateless, so the clouds of a CHUNK of samples (original + augmented) are
    encoded together - big, tile-aligned GEMMs instead of T=513-row ones;
  * the cache step is sequential in the stream (sample i sees the state left by samples < i), so it
    is replayed per sample, in order, from the chunk's features: clip logits -> fused
    predict+fit+fit -> residual learning (10 Adam steps) -> fusion.  Nothing in that replay touches
    the host: predictions, ranks and logits stay in HBM until the chunk (or the stream) ends;
  * with torch.distributed initialised (one process per GPU, RCCL), rank r encodes samples
    i = r (mod W) of every chunk, features are exchanged with ONE all_gather per chunk and every
    rank replays the identical cache sequence on its replica of the state ("exact mode": results
    are independent of the GPU count).
"""
import logging
import os
from types import SimpleNamespace

import torch
import torch.nn.functional as F

from . import lib
from .dota_mixture import DOTA_mix
from .stream import (REPLAY_COST_RATIO, agree_on_ratio, broadcast_from_owner, gather_deltas, gather_feature_blocks, gather_features, shard_block,
                     shard_counts, shard_indices)

logger = logging.getLogger(__name__)

LOGIT_SCALE = 100.0          # Uni_Adapter.py:57
NOISE_STD = 0.05             # Uni_Adapter.py:420
RES_STEPS = 10               # Uni_Adapter.py:467
RES_LR = 1e-3                # Uni_Adapter.py:350-351 (start == end)


# ----------------------------------------------------------------------------- small mirrors
def softmax_entropy(x, enable_softmax=True, temperature=1.0):
    """Uni_Adapter.py:21-26 (host-side helper kept for API compatibility; the hot loop uses the
    fused kernels)."""
    probs = torch.softmax(x / temperature, dim=1) if enable_softmax else x
    return -(probs * torch.log(probs + 1e-10)).sum(dim=1)


def get_logits_wrapper(args, model, feature, clip_weights):
    """Uni_Adapter.py:53-75 for vlm3d in {'uni3d', 'ulip', 'openshape'}, batch 1.  `clip_weights` is (D,K) as in
    the reference.  Returns (pc_features, logits, loss, prob_map, pred:int)."""
    vlm = getattr(args, 'vlm3d', 'uni3d')
    if vlm == 'uni3d':
        feat = model.encode_pc(feature)                       # :54-57
    elif vlm == 'ulip':
        feat = model(feature[:, :, :3].contiguous())          # :58-62 (xyz only)
    elif vlm == 'openshape':
        feat = model(feature[:, :, :3].contiguous(), feature)   # :63-66 (xyz, and xyz+rgb as point features)
    else:
        raise NotImplementedError(f"--vlm3d {vlm}")
    if feat.shape[0] != 1:
        raise ValueError('batch size must be 1 (reference defect 6)')
    text = clip_weights.t().contiguous()
    K, D = text.shape
    dev = feat.device
    fh = torch.empty(1, D, device=dev); z = torch.empty(1, K, device=dev); p = torch.empty(1, K, device=dev)
    ent = torch.empty(1, device=dev); pred = torch.empty(1, dtype=torch.int32, device=dev)
    lib.call('ua_clip_logits', lib.ptr(feat.contiguous()), lib.ptr(text), K, D, LOGIT_SCALE, 1, lib.ptr(fh), lib.ptr(z),
             lib.ptr(p), lib.ptr(ent), lib.ptr(pred, torch.int32), lib.stream_ptr())
    return fh, z, ent, p, int(pred.item())


class _AlignLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, e, mdl):
        K, M, D = mdl.num_classes, mdl.num_modes, mdl.input_shape
        e = e.contiguous()
        L, joint = mdl.joint(e)
        loss = torch.empty(1, device=e.device); gll = torch.empty(K, K, M, device=e.device)
        lib.call('ua_align_loss', lib.ptr(L), lib.ptr(joint), K, M, lib.ptr(loss), lib.ptr(gll), lib.stream_ptr())
        ge = torch.empty(K, D, device=e.device)
        lib.call('ua_gmm_loglik_bwd', lib.ptr(mdl.mu), lib.ptr(mdl.var), lib.ptr(e), lib.ptr(gll), K, K, M, D, mdl.epsilon,
                 lib.ptr(ge), lib.stream_ptr())
        ctx.save_for_backward(ge)
        ctx.mark_non_differentiable(L)
        return loss.reshape(()), L

    @staticmethod
    def backward(ctx, g_loss, g_L):
        (ge,) = ctx.saved_tensors
        return ge * g_loss, None


def compute_text_alignment_loss(class_embeddings, mode_dota_model):
    """Uni_Adapter.py:191-270: returns (loss, likelihood_matrix); loss is differentiable w.r.t.
    class_embeddings (K,D)."""
    if not class_embeddings.requires_grad:
        raise RuntimeError("class_embeddings must require gradients for optimization")
    return _AlignLoss.apply(class_embeddings.float(), mode_dota_model)


# ----------------------------------------------------------------------------- engine
class TextResiduals:
    """text_residuals + torch.optim.Adam state + the current normalised text matrix
    (Uni_Adapter.py:346-352, 388-393), all resident on the device."""

    def __init__(self, text0):
        self.text0 = text0.contiguous()
        K, D = text0.shape
        dev = text0.device
        self.res = torch.zeros(K, D, device=dev)
        self.m = torch.zeros(K, D, device=dev)
        self.v = torch.zeros(K, D, device=dev)
        self.e = torch.empty(K, D, device=dev)
        self.norm = torch.empty(K, device=dev)
        self.t = self.text0.clone()                 # text0 + res, unnormalised (the MFMA residual kernels work on it)
        self.step = 0
        lib.call('ua_text_normalize', lib.ptr(self.text0), lib.ptr(self.res), K, D, 0.0, lib.ptr(self.e), lib.ptr(self.norm),
                 lib.stream_ptr())


class TtaEngine:
    """Per-corruption state + the chunked loop body.  One instance per call of
    test_zeroshot_3d_core (the reference rebuilds all state per corruption, Uni_Adapter.py:336-352)."""

    def __init__(self, model, text_features, dota_cfg, num_modes, res_learning=True, chunk=16, device=None,
                 log_alignment_every=0, dist_group=None, composite=True, overlap=False, replay_mode=None,
                 replay_cost_ratio=REPLAY_COST_RATIO, two_stream_encode=False, emulated_rank=None, align_path=None, replay_lean=None):
        self.model = model
        self.composite = bool(composite) and not log_alignment_every
        # overlap=True: the in-order cache replay of chunk c runs on a second HIP stream while the
        # caller's stream already encodes chunk c+1 (the encoder does not depend on the cache state).
        # Results are complete after `synchronize()` (or the next `accuracy()`); same arithmetic.
        self.overlap = bool(overlap) and self.composite
        # (HIP exposes two stream priorities here; giving the replay stream either one changes nothing measurable: 451 / 451 / 450 pc/s)
        self.replay_stream = torch.cuda.Stream(device=device) if self.overlap else None
        # the original and the augmented clouds of a chunk are independent encoder batches: enqueued on two HIP streams
        # (separate workspaces) the tail of one batch's kernels fills under the other's: same results, 2.5 % faster for the
        # encoder alone (tools/two_stream_probe.py) but no gain once the cache replay already fills those tails - off by default
        self.two_stream_encode = bool(two_stream_encode) and hasattr(model, 'point_encoder')
        self.enc_stream = torch.cuda.Stream(device=device) if self.two_stream_encode else None
        self.dev = torch.device(device) if device is not None else torch.device('cuda', torch.cuda.current_device())
        self.text = text_features.detach().to(self.dev, torch.float32).contiguous()          # (K,D)
        self.K, self.D = self.text.shape
        self.M = int(num_modes)
        self.cfg = dict(dota_cfg)
        self.res_learning = bool(res_learning)
        self.chunk = int(chunk)
        self.log_every = int(log_alignment_every)
        self.mode_dota_model = DOTA_mix(self.cfg, self.D, self.K, self.text.t(), num_modes=self.M, device=self.dev)
        self.mode_dota_model.eval()
        self.residuals = TextResiduals(self.text) if self.res_learning else None
        self.i = 0                                                                           # samples seen
        self.world, self.rank, self.group = 1, 0, dist_group
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            self.world = torch.distributed.get_world_size(dist_group)
            self.rank = torch.distributed.get_rank(dist_group)
        if emulated_rank is not None:
            # (rank, world): act as one rank of a W-rank job without a process group - the caller moves the
            # buffers between engines itself (lock-step tests of the 'delta' mode, tools/emulate_scaling.py)
            self.rank, self.world = int(emulated_rank[0]), int(emulated_rank[1])
        self.emulated = emulated_rank is not None
        # where the sequential cache replay runs when W > 1 (stream.py): 'owner' = rank 0 only, with a
        # smaller encoder share for rank 0; 'replicated' = every rank replays everything
        self.replay_mode = replay_mode or os.environ.get('UA_REPLAY_MODE', 'owner')
        if self.replay_mode not in ('owner', 'replicated', 'delta', 'class', 'streams'):
            raise ValueError(f"replay_mode must be 'owner', 'replicated', 'class', 'delta' or 'streams', got {self.replay_mode!r}")
        if self.replay_mode == 'streams':
            # every rank adapts on its OWN stream (the driver deals whole corruption streams to ranks: their states are
            # independent, Uni_Adapter.py:336-352): this engine is a single-process engine whatever the job size
            self.world, self.rank = 1, 0
        # 'class' (exact, for large K): features are gathered as in 'replicated', but rank r keeps and
        # updates only classes [r*per, (r+1)*per) of the GMM state; ONE all_gather per chunk re-assembles the scores of its samples
        self.class_mode = self.replay_mode == 'class' and self.world > 1
        if self.class_mode and self.res_learning:
            raise ValueError("replay_mode='class' shards the GMM state by class; residual learning couples all classes "
                             "(K x K alignment matrix) and is not available in this mode - pass res_learning=False")
        # 'delta' (SURVEY section 8e "prototype deltas", NOT exact): every rank replays only its own block from the common
        # state and one all_gather of sufficient-statistic deltas merges the W replicas after each step
        self.delta_mode = self.replay_mode == 'delta' and self.world > 1
        if self.delta_mode and not self.composite:
            raise ValueError("replay_mode='delta' needs the composite replay (log_alignment_every=0)")
        self.replay_cost_ratio = float(replay_cost_ratio)
        self.replays = self.world == 1 or self.replay_mode in ('replicated', 'delta', 'class') or self.rank == 0
        if self.class_mode:
            self.class_per = -(-self.K // self.world)
            self.class_k0 = min(self.rank * self.class_per, self.K)
            self.class_k1 = min(self.class_k0 + self.class_per, self.K)
            self.dz_local = torch.zeros(max(1, self.class_per), device=self.dev)
        self._delta = None
        K, D, M, dev = self.K, self.D, self.M, self.dev
        e = lambda *s, dt=torch.float32: torch.empty(*s, device=dev, dtype=dt)
        self.fh, self.fa = e(1, D), e(1, D)
        self.z, self.prob, self.ent, self.dz = e(1, K), e(1, K), e(1), e(1, K)      # (dz: unused in 'class' mode - scores live in the exchange buffer)
        # UA_ALIGN_PATH=mfma runs the residual loop on the matrix cores (csrc/ua_align_mfma.hip) when the shape is inside its
        # envelope.  Opt-in: at K = 40 the two GEMMs are 0.05 GFLOP - the step is bound by kernel floors and per-CU L2
        # bandwidth, not arithmetic, and the VALU kernels (work spread over 160-400 workgroups) are faster
        self.align_mfma = (self.res_learning and lib.align_mfma_supported(K, M, D)
                           and (align_path or os.environ.get('UA_ALIGN_PATH', 'valu')) == 'mfma')
        if self.res_learning:               # K x K work arrays of the residual loop (1 GB each at K = 4096, M = 16: only when used)
            self.L, self.joint, self.gll, self.ge, self.loss = e(K, K), e(K, K, M), e(K, K, M), e(K, D), e(1)
            self.inv_var, self.row_const = e(K, M, D), e(K, M)
            if self.align_mfma:
                self.U = e(K, K, M)
                self.align_ws = torch.empty(lib.align_mfma_workspace_bytes(K, M, D), dtype=torch.uint8, device=dev)
        else:
            self.L = self.joint = self.gll = self.ge = self.inv_var = self.row_const = None
            self.loss = e(1)
        # Residual step in three launches (loss | backward | Adam) instead of the one-launch ua_align_step: bit-identical, ~45 % less
        # CU-time (the K x K loss statistics are computed once, not by each of the 4 K workgroups), two more dependent launches per
        # Adam step.  Beside this rank's own encoder work the replay is paid in CU-time, so this is +1.5 % at chunk 31 - and -1.5 %
        # at chunk 124, where the longer chain under contention becomes the critical path: off by default.
        env_lean = os.environ.get('UA_REPLAY_LEAN')
        if replay_lean is None:
            replay_lean = env_lean is not None and env_lean != '0'
        self.replay_lean = bool(replay_lean)
        self.tickets = torch.zeros(K, dtype=torch.int32, device=self.dev)      # hand-off counters of ua_align_step
        self.correct = torch.zeros(3, device=dev, dtype=torch.int64)                          # top-1/3/5 hits
        self.count = 0
        self.last_loss = None
        self._state = self._make_state()

    def _make_state(self):
        """ua_tta_state for the composite entry point; all pointers are fixed for the engine's life."""
        s, m, r = lib.TtaState(), self.mode_dota_model, self.residuals
        s.K, s.M, s.D = self.K, self.M, self.D
        s.eps, s.rho, s.eta, s.logit_scale = m.epsilon, float(self.cfg['rho']), float(self.cfg['eta']), LOGIT_SCALE
        s.res_learning, s.res_steps = int(self.res_learning), RES_STEPS
        s.lr, s.beta1, s.beta2, s.adam_eps = RES_LR, 0.9, 0.999, 1e-8
        s.mu, s.var, s.pi, s.c, s.class_counts = (t.data_ptr() for t in (m.mu, m.var, m.pi, m.c, m.class_counts))
        s.text0 = self.text.data_ptr()
        if r is not None:
            s.res, s.adam_m, s.adam_v, s.e, s.norm = (t.data_ptr() for t in (r.res, r.m, r.v, r.e, r.norm))
        s.fhat, s.faug, s.clip_logits, s.prob, s.dota_scores = (t.data_ptr() for t in (self.fh, self.fa, self.z, self.prob, self.dz))
        s.loss = self.loss.data_ptr()
        if self.res_learning:
            s.L, s.joint, s.grad_ll, s.grad_e = (t.data_ptr() for t in (self.L, self.joint, self.gll, self.ge))
            s.inv_var, s.row_const = self.inv_var.data_ptr(), self.row_const.data_ptr()
        s.tickets = self.tickets.data_ptr()
        s.replay_lean = int(self.replay_lean)
        if self.res_learning and self.align_mfma:
            s.align_mfma = 1
            s.t, s.U = r.t.data_ptr(), self.U.data_ptr()
            s.align_ws, s.align_ws_bytes = self.align_ws.data_ptr(), self.align_ws.numel()
        return s

    # -- 'owner' mode: measure rho = replay cost / encoder cost on this machine instead of assuming it ----------------
    @torch.no_grad()
    def calibrate_replay_cost(self, pc, rgb, noise, reps=2):
        """Times the encoder on the given few samples (every rank) and the in-order replay of their features (on a scratch
        copy of the adaptation state, which is restored), and sets `replay_cost_ratio` to rank 0's ratio on all ranks.
        Call once before the stream; a no-op unless W > 1 in 'owner' mode.  Returns the ratio."""
        if not (self.world > 1 and self.replay_mode == 'owner'):
            return self.replay_cost_ratio
        import ctypes
        pc, rgb, noise = (t.to(self.dev, torch.float32) for t in (pc, rgb, noise))
        n = pc.shape[0]
        clouds = torch.cat((torch.cat((pc, rgb), -1), torch.cat((pc + NOISE_STD * noise, rgb), -1)), 0)
        ev = lambda: torch.cuda.Event(enable_timing=True)
        feats = self.model.encode_pc(clouds)                                   # warm (allocations, code objects)
        e0, e1 = ev(), ev()
        e0.record()
        for _ in range(reps):
            feats = self.model.encode_pc(clouds)
        e1.record(); torch.cuda.synchronize()
        enc_ms = e0.elapsed_time(e1) / reps / n
        rep_ms = 0.0
        if self.rank == 0:
            feats2 = torch.stack((feats[:n], feats[n:]), 1).contiguous()
            saved = self.state_dict()
            final = torch.empty(n, self.K, device=self.dev); preds = torch.empty(n, dtype=torch.int32, device=self.dev)
            st = torch.cuda.current_stream().cuda_stream
            run = lambda: lib.call('ua_tta_chunk', ctypes.byref(self._state), lib.ptr(feats2), None, n, 1, self.residuals.step if self.res_learning else 0,
                                   lib.ptr(final), None, lib.ptr(preds, torch.int32), st)
            run()
            r0, r1 = ev(), ev()
            r0.record()
            for _ in range(reps):
                run()
            r1.record(); torch.cuda.synchronize()
            rep_ms = r0.elapsed_time(r1) / reps / n
            self.load_state_dict(saved)
            self.tickets.zero_()                       # (every ua_align_step leaves its hand-off counters at zero; stated, not assumed)
            self.last_loss = None
        # (the replay measured alone; beside rank 0's own encoder share it runs ~15 % slower - tools/emulate_scaling.py)
        self.replay_cost_ratio = agree_on_ratio(1.15 * rep_ms, enc_ms, 1 if self.emulated else self.world, self.group, self.dev)
        if self.rank == 0:
            logger.info('owner-mode replay / encoder cost ratio rho = %.4f (replay %.3f ms, encoder %.3f ms per sample)',
                        self.replay_cost_ratio, rep_ms, enc_ms)
        return self.replay_cost_ratio

    # -- encoder part: this rank's share of the chunk -------------------------------------
    def encode_chunk(self, pc, rgb, noise):
        """pc, rgb (n,N,3); noise (n,N,3) = the randn draw of Uni_Adapter.py:421.  Returns
        features (n,2,D): [:,0] original, [:,1] augmented cloud."""
        n = pc.shape[0]
        owner_mode = self.world > 1 and self.replay_mode == 'owner'
        if self.delta_mode:
            mine = shard_block(n, self.rank, self.world, 0.0)
        else:
            mine = shard_block(n, self.rank, self.world, self.replay_cost_ratio) if owner_mode else shard_indices(n, self.rank, self.world)
        D = self.D
        if len(mine):
            p, c, z = pc[mine], rgb[mine], noise[mine]
            orig, aug = torch.cat((p, c), -1), torch.cat((p + NOISE_STD * z, c), -1)                    # :385, :421-422
            if self.two_stream_encode and len(mine) >= 4:
                enc, cur = self.model.point_encoder, torch.cuda.current_stream()
                self.enc_stream.wait_stream(cur)
                with torch.cuda.stream(self.enc_stream):
                    enc.ws_slot = 1
                    fb = self.model.encode_pc(aug)
                enc.ws_slot = 0
                fa = self.model.encode_pc(orig)
                cur.wait_stream(self.enc_stream)
                fb.record_stream(cur)
                local = torch.stack([fa, fb], 1)
            else:
                clouds = torch.stack([orig, aug], 1)
                local = self.model.encode_pc(clouds.view(-1, clouds.shape[2], 6)).view(len(mine), 2, D)
        else:
            local = torch.zeros(0, 2, D, device=self.dev)
        if self.delta_mode:
            return local                          # features stay on the rank; the exchange is delta_exchange()
        # one exchange per chunk (RCCL all_gather over xGMI); a no-op at world 1
        if owner_mode:
            return gather_feature_blocks(local, n, self.rank, self.world, self.group, self.replay_cost_ratio)
        return gather_features(local, n, self.rank, self.world, self.group)

    # -- sequential cache step of ONE sample ----------------------------------------------
    def cache_step(self, feat, feat_aug, target, out_final, out_rank, out_pred):
        """feat, feat_aug: (1,D) encoder outputs; target: int32 (1,) device tensor or None.
        Writes final logits (1,K), rank (1,), pred (1,) into the given device views."""
        K, M, D = self.K, self.M, self.D
        st = lib.stream_ptr()
        mdl = self.mode_dota_model
        text = self.residuals.e if self.res_learning else self.text                         # :388-401
        lib.call('ua_clip_logits', lib.ptr(feat), lib.ptr(text), K, D, LOGIT_SCALE, 1, lib.ptr(self.fh), lib.ptr(self.z),
                 lib.ptr(self.prob), lib.ptr(self.ent), None, st)                            # :406
        lib.call('ua_clip_logits', lib.ptr(feat_aug), lib.ptr(text), K, D, LOGIT_SCALE, 2, lib.ptr(self.fa), None, None, None,
                 None, st)                                                                   # :424,:429
        mdl.step_fused(self.fh, self.fh, self.fa, self.prob, scores=self.dz)                 # :416,:417,:430
        if self.i > 0 and self.res_learning and self.align_mfma:                             # :444-476 on the matrix cores
            r, ws = self.residuals, self.align_ws
            lib.call('ua_align_mfma_prepare', lib.ptr(mdl.mu), lib.ptr(mdl.var), lib.ptr(mdl.pi), K, M, D, mdl.epsilon, ws.data_ptr(), ws.numel(), st)
            for _ in range(RES_STEPS):
                lib.call('ua_align_forward_mfma', lib.ptr(r.t), K, M, D, ws.data_ptr(), ws.numel(), lib.ptr(self.L), lib.ptr(self.joint),
                         lib.ptr(self.U), lib.ptr(r.norm), st)
                r.step += 1
                lib.call('ua_align_step_mfma', lib.ptr(self.L), lib.ptr(self.joint), lib.ptr(self.U), lib.ptr(r.norm), lib.ptr(r.text0),
                         lib.ptr(r.res), lib.ptr(r.m), lib.ptr(r.v), lib.ptr(r.t), K, M, D, r.step, RES_LR, 0.9, 0.999, 1e-8,
                         ws.data_ptr(), ws.numel(), lib.ptr(self.loss), None, st)
            lib.call('ua_text_normalize', lib.ptr(r.text0), lib.ptr(r.res), K, D, 0.0, lib.ptr(r.e), lib.ptr(r.norm), st)
            if self.log_every and self.i % self.log_every == 0:                              # :480-485 (the 11th loss evaluation)
                lib.call('ua_align_forward_mfma', lib.ptr(r.t), K, M, D, ws.data_ptr(), ws.numel(), lib.ptr(self.L), lib.ptr(self.joint),
                         lib.ptr(self.U), lib.ptr(r.norm), st)
                lib.call('ua_align_loss', lib.ptr(self.L), lib.ptr(self.joint), K, M, lib.ptr(self.loss), None, st)
                self.last_loss = self.loss.clone()
        elif self.i > 0 and self.res_learning:                                               # :444-476
            r = self.residuals
            lib.call('ua_gmm_prepare_alignment', lib.ptr(mdl.var), lib.ptr(mdl.pi), K, M, D, mdl.epsilon, lib.ptr(self.inv_var),
                     lib.ptr(self.row_const), st)
            for _ in range(RES_STEPS):
                self._align_eval(r, with_grad=True)
                r.step += 1
                lib.call('ua_residual_adam', lib.ptr(r.text0), lib.ptr(r.res), lib.ptr(r.m), lib.ptr(r.v), lib.ptr(self.ge),
                         lib.ptr(r.e), lib.ptr(r.norm), K, D, r.step, RES_LR, 0.9, 0.999, 1e-8, st)
            if self.log_every and self.i % self.log_every == 0:                              # :480-485
                self._align_eval(r, with_grad=False)
                self.last_loss = self.loss.clone()
        lib.call('ua_fuse_logits', lib.ptr(self.z), lib.ptr(self.dz), lib.ptr(mdl.c), K, M, float(self.cfg['rho']),
                 float(self.cfg['eta']), 1, lib.ptr(target, torch.int32) if target is not None else None,
                 lib.ptr(out_final), lib.ptr(out_rank, torch.int32) if target is not None else None,
                 lib.ptr(out_pred, torch.int32), st)                                         # :491-521
        self.i += 1

    def _align_eval(self, r, with_grad):
        K, M, D = self.K, self.M, self.D
        mdl = self.mode_dota_model
        st = lib.stream_ptr()
        lib.call('ua_align_forward', lib.ptr(mdl.mu), lib.ptr(self.inv_var), lib.ptr(self.row_const), lib.ptr(r.e), K, K, M, D,
                 lib.ptr(self.L), lib.ptr(self.joint), st)
        lib.call('ua_align_loss', lib.ptr(self.L), lib.ptr(self.joint), K, M, lib.ptr(self.loss),
                 lib.ptr(self.gll) if with_grad else None, st)
        if with_grad:
            lib.call('ua_align_backward', lib.ptr(mdl.mu), lib.ptr(self.inv_var), lib.ptr(r.e), lib.ptr(self.gll), K, K, M, D,
                     lib.ptr(self.ge), st)

    # -- one chunk ---------------------------------------------------------------------------
    @torch.no_grad()
    def process_chunk(self, pc, rgb, target=None, noise=None, generator=None):
        """pc, rgb: (n,N,3); target: (n,) or None; noise: (n,N,3) or None (drawn on device).
        Returns (final_logits (n,K), ranks (n,) int32, preds (n,) int32) - device tensors."""
        pc = pc.to(self.dev, torch.float32, non_blocking=True)
        rgb = rgb.to(self.dev, torch.float32, non_blocking=True)
        n = pc.shape[0]
        if noise is None:
            noise = torch.randn(pc.shape, device=self.dev, generator=generator)              # :421
        else:
            noise = noise.to(self.dev, torch.float32)
        feats = self.encode_chunk(pc, rgb, noise)
        if self.delta_mode:
            return self.delta_step(feats, target, n)
        final = torch.empty(n, self.K, device=self.dev)
        ranks = torch.full((n,), -1, device=self.dev, dtype=torch.int32)
        preds = torch.empty(n, device=self.dev, dtype=torch.int32)
        tgt = target.to(self.dev, torch.int32) if target is not None else None
        if not self.replays:
            # 'owner' mode, rank > 0: this rank only encodes; the state and the results live on rank 0
            if self.res_learning:
                self.residuals.step += RES_STEPS * (n - (1 if self.i == 0 else 0))
            self.i += n
            self.mode_dota_model.t += 2 * n
            return final.zero_(), ranks, preds.fill_(-1)
        if self.class_mode:
            main = torch.cuda.current_stream()
            rs = self.replay_stream if self.overlap else main
            if self.overlap:
                rs.wait_stream(main)
                for t in (feats, final, ranks, preds, tgt):
                    if t is not None:
                        t.record_stream(rs)
            with torch.cuda.stream(rs):
                # this rank's class slice stepped through the WHOLE chunk, then one collective, then the fusion of all n samples
                buf = self.class_local_chunk(feats)
                self.class_fuse_chunk(gather_deltas(buf, self.world, self.group), tgt, final, ranks, preds)
                if tgt is not None:
                    self.correct += torch.stack([(ranks < k).sum() for k in (1, 3, 5)])
                    self.count += n
            return final, ranks, preds
        if self.composite:
            # one host call enqueues the whole in-order replay of the chunk (csrc/ua_tta.hip)
            import ctypes
            feats = feats.contiguous()
            main = torch.cuda.current_stream()
            rs = self.replay_stream if self.overlap else main
            if self.overlap:
                rs.wait_stream(main)                       # features (and everything allocated above) are ready
                for t in (feats, final, ranks, preds, tgt):
                    if t is not None:
                        t.record_stream(rs)
            with torch.cuda.stream(rs):
                lib.call('ua_tta_chunk', ctypes.byref(self._state), lib.ptr(feats), lib.ptr(tgt, torch.int32) if tgt is not None else None,
                         n, self.i, self.residuals.step if self.res_learning else 0, lib.ptr(final),
                         lib.ptr(ranks, torch.int32) if tgt is not None else None, lib.ptr(preds, torch.int32), rs.cuda_stream)
                if tgt is not None:                                                          # utils/utils.py:154-167
                    self.correct += torch.stack([(ranks < k).sum() for k in (1, 3, 5)])
                    self.count += n
            if self.res_learning:
                self.residuals.step += RES_STEPS * (n - (1 if self.i == 0 else 0))
            self.i += n
            self.mode_dota_model.t += 2 * n
            return final, ranks, preds
        for j in range(n):
            self.cache_step(feats[j, 0:1], feats[j, 1:2], tgt[j:j + 1] if tgt is not None else None,
                            final[j:j + 1], ranks[j:j + 1], preds[j:j + 1])
        if tgt is not None:                                                                  # utils/utils.py:154-167
            self.correct += torch.stack([(ranks < k).sum() for k in (1, 3, 5)])
            self.count += n
        return final, ranks, preds

    # -- 'class' replay mode (exact, large K): this rank's class slice of the step for every sample of the chunk, ONE exchange,
    #    then the fusion of the chunk from the W gathered buffers (two halves so that W ranks can be driven in lock-step from
    #    one process).  Exact because, with residual learning off, nothing the fusion produces feeds back into the state: fit
    #    takes the zero-shot prob_map (Uni_Adapter.py:417,430) and mean(c) is only read by the fusion weight (:491). ---------
    def _class_bufs(self, n, fresh=False):
        """Per-chunk buffers of the class-sharded replay, allocated on the stream that runs the replay (the caching allocator
        hands a block back only in that stream's order): the exchange buffer is read by the collective while the next chunk's
        pack kernels may already be enqueued, so no buffer is shared between chunks.  `fresh` starts a chunk; the fusion half
        of the same chunk picks the clip-logit buffer up again."""
        if fresh or getattr(self, 'zbuf', None) is None or self.zbuf.shape[0] != n:
            self.zbuf = torch.empty(n, self.K, device=self.dev)
            self.xbuf = torch.zeros(n, self.class_per + 2, device=self.dev)
        return self.zbuf, self.xbuf

    def class_local_chunk(self, feats):
        """feats (n,2,D).  Per sample: clip logits over all K (every rank), predict + fit + fit on classes [k0,k1) of the
        state, pack [score slice | partial sum of c after this sample].  Returns the (n, per+2) exchange buffer."""
        K, M, D = self.K, self.M, self.D
        n = feats.shape[0]
        k0, kl = self.class_k0, self.class_k1 - self.class_k0
        mdl, st = self.mode_dota_model, lib.stream_ptr()
        zbuf, xbuf = self._class_bufs(n, fresh=True)
        feats = feats.contiguous()
        row = 4 * k0 * M
        for j in range(n):
            f = feats.data_ptr() + 4 * (2 * j) * D
            lib.call('ua_clip_logits_pair', f, f + 4 * D, lib.ptr(self.text), K, D, LOGIT_SCALE, lib.ptr(self.fh),
                     lib.ptr(self.fa), zbuf.data_ptr() + 4 * j * K, lib.ptr(self.prob), st)                       # :406,:424,:429
            if kl:
                lib.call('ua_gmm_step_fused', mdl.mu.data_ptr() + row * D, mdl.var.data_ptr() + row * D, mdl.pi.data_ptr() + row,
                         mdl.c.data_ptr() + row, mdl.class_counts.data_ptr() + 4 * k0, lib.ptr(self.fh), 1, lib.ptr(self.fh),
                         lib.ptr(self.fa), self.prob.data_ptr() + 4 * k0, lib.ptr(self.dz_local), kl, M, D, mdl.epsilon, st)   # :416,:417,:430
            lib.call('ua_class_shard_pack', lib.ptr(self.dz_local), mdl.c.data_ptr() + 4 * k0 * M, kl, M, self.class_per,
                     xbuf.data_ptr() + 4 * j * (self.class_per + 2), st)
        return xbuf

    def class_fuse_chunk(self, gathered, target, out_final, out_rank, out_pred):
        """gathered: (W, n, per+2) = every rank's class_local_chunk buffer in rank order; target (n,) int32 or None."""
        gathered = gathered.contiguous()
        n = gathered.shape[1]
        assert gathered.shape == (self.world, n, self.class_per + 2)
        zbuf, _ = self._class_bufs(n)
        lib.call('ua_fuse_logits_sharded', lib.ptr(zbuf), lib.ptr(gathered), self.world, n, self.class_per, self.K, self.M,
                 float(self.cfg['rho']), float(self.cfg['eta']), 1, lib.ptr(target, torch.int32) if target is not None else None,
                 lib.ptr(out_final), lib.ptr(out_rank, torch.int32) if target is not None else None, lib.ptr(out_pred, torch.int32),
                 None, lib.stream_ptr())                                                                      # :491-521
        self.i += n
        self.mode_dota_model.t += 2 * n

    # -- 'delta' replay mode: local replay + pack, exchange, merge (three steps so that W ranks can also be driven in
    #    lock-step from one process: tests/test_gpu_harness.py, tools/emulate_scaling.py) ---------------------------
    def _delta_buffers(self, per):
        K, M, D = self.K, self.M, self.D
        nS = K * M * D
        o = {'S1': 0, 'S2': nS, 'C': 2 * nS, 'cc': 2 * nS + K * M}
        o['res'] = o['cc'] + K
        o['m'], o['v'] = o['res'] + K * D, o['res'] + 2 * K * D
        o['out'] = o['res'] + 3 * K * D                      # per x (K final logits | rank | pred) of this rank's block
        stride = (o['out'] + per * (K + 2) + 3) // 4 * 4
        d = self._delta
        if d is None or d['stride'] != stride:
            mdl, r = self.mode_dota_model, self.residuals
            d = {'stride': stride, 'off': o, 'per': per, 'buf': torch.zeros(stride, device=self.dev),
                 's0': {k: torch.empty_like(t) for k, t in (('mu', mdl.mu), ('var', mdl.var), ('c', mdl.c), ('cc', mdl.class_counts))}}
            if r is not None:
                d['s0'].update(res=torch.empty_like(r.res), m=torch.empty_like(r.m), v=torch.empty_like(r.v))
            self._delta = d
        return d

    def delta_local(self, feats, target, n):
        """Replay this rank's block of an n-sample step from the common state and pack the deltas.  feats: (mine,2,D)
        features of the block; target: (n,) labels of the WHOLE step or None.  Returns the packed fp32 buffer."""
        import ctypes
        K, M, D = self.K, self.M, self.D
        counts = shard_counts(n, self.world, 0.0)
        start, mine, per = sum(counts[:self.rank]), counts[self.rank], max(counts)
        d = self._delta_buffers(per)
        o, buf, s0 = d['off'], d['buf'], d['s0']
        d['targets'] = target is not None
        mdl, r = self.mode_dota_model, self.residuals
        st = lib.stream_ptr()
        live = {'mu': mdl.mu, 'var': mdl.var, 'c': mdl.c, 'cc': mdl.class_counts}
        if r is not None:
            live.update(res=r.res, m=r.m, v=r.v)
        for k, t in live.items():
            s0[k].copy_(t)                                                   # S0 (device-to-device)
        res_out = buf[o['out']:o['out'] + per * (K + 2)]
        res_out.zero_()
        if mine:
            final = torch.empty(mine, K, device=self.dev)
            ranks = torch.full((mine,), -1, device=self.dev, dtype=torch.int32)
            preds = torch.empty(mine, device=self.dev, dtype=torch.int32)
            tgt = target.to(self.dev, torch.int32)[start:start + mine].contiguous() if target is not None else None
            feats = feats.contiguous()
            lib.call('ua_tta_chunk', ctypes.byref(self._state), lib.ptr(feats), lib.ptr(tgt, torch.int32) if tgt is not None else None,
                     mine, self.i + start, r.step if r is not None else 0, lib.ptr(final),
                     lib.ptr(ranks, torch.int32) if tgt is not None else None, lib.ptr(preds, torch.int32), st)
            out = res_out.view(per, K + 2)
            out[:mine, :K] = final
            out[:mine, K] = ranks.float()
            out[:mine, K + 1] = preds.float()
        lib.call('ua_gmm_delta_pack', lib.ptr(mdl.mu), lib.ptr(mdl.var), lib.ptr(mdl.c), lib.ptr(s0['mu']), lib.ptr(s0['var']),
                 lib.ptr(s0['c']), K, M, D, lib.ptr(buf), st)
        lib.call('ua_vec_delta', lib.ptr(mdl.class_counts), lib.ptr(s0['cc']), K, buf.data_ptr() + 4 * o['cc'], st)
        if r is not None:
            for key, t in (('res', r.res), ('m', r.m), ('v', r.v)):
                lib.call('ua_vec_delta', lib.ptr(t), lib.ptr(s0[key]), K * D, buf.data_ptr() + 4 * o[key], st)
        return buf

    def delta_step(self, feats, target, n, exchange=None):
        """Local replay + pack, the one collective, merge - on the replay stream when overlap=True, so that the caller's
        stream already encodes the next step meanwhile (results complete after `synchronize()`, as in the exact modes)."""
        main = torch.cuda.current_stream()
        rs = self.replay_stream if self.overlap else main
        if self.overlap:
            rs.wait_stream(main)
            feats.record_stream(rs)
        with torch.cuda.stream(rs):
            buf = self.delta_local(feats, target, n)
            return self.delta_merge((exchange or self.delta_exchange)(buf), n)

    def delta_exchange(self, buf):
        """ONE collective per step: all_gather of the packed deltas (RCCL over xGMI).  Returns (W, stride)."""
        return gather_deltas(buf, self.world, self.group)

    def delta_merge(self, gathered, n):
        """State <- S0 + the W ranks' deltas (rank order); returns the step's (final (n,K), ranks, preds) in stream order."""
        K, M, D, W = self.K, self.M, self.D, self.world
        d = self._delta
        o, s0, stride, per = d['off'], d['s0'], d['stride'], d['per']
        gathered = gathered.contiguous()
        assert gathered.shape == (W, stride)
        mdl, r = self.mode_dota_model, self.residuals
        st = lib.stream_ptr()
        g = gathered.data_ptr()
        lib.call('ua_gmm_delta_merge', lib.ptr(mdl.mu), lib.ptr(mdl.var), lib.ptr(mdl.pi), lib.ptr(mdl.c), lib.ptr(s0['mu']),
                 lib.ptr(s0['var']), lib.ptr(s0['c']), g, W, stride, K, M, D, st)
        lib.call('ua_vec_merge', lib.ptr(s0['cc']), g + 4 * o['cc'], W, stride, K, 1.0, lib.ptr(mdl.class_counts), st)
        counts = shard_counts(n, W, 0.0)
        if r is not None:
            for key, t in (('res', r.res), ('m', r.m), ('v', r.v)):
                lib.call('ua_vec_merge', lib.ptr(s0[key]), g + 4 * o[key], W, stride, K * D, 1.0 / W, lib.ptr(t), st)
            lib.call('ua_text_normalize', lib.ptr(r.text0), lib.ptr(r.res), K, D, 0.0, lib.ptr(r.e), lib.ptr(r.norm), st)
            torch.add(r.text0, r.res, out=r.t)
            starts = [sum(counts[:q]) for q in range(W)]
            r.step += max(RES_STEPS * (counts[q] - (1 if self.i + starts[q] == 0 and counts[q] else 0)) for q in range(W))
        out = gathered[:, o['out']:o['out'] + per * (K + 2)].view(W, per, K + 2)
        rows = torch.cat([out[q, :counts[q]] for q in range(W)], 0)                       # stream order
        final, ranks, preds = rows[:, :K].contiguous(), rows[:, K].to(torch.int32), rows[:, K + 1].to(torch.int32)
        if d['targets']:
            # every rank holds every rank's results, so the accuracy counters need no final all_reduce (Uni_Adapter.py:582)
            self.correct += torch.stack([(ranks < k).sum() for k in (1, 3, 5)])
            self.count += n
        self.i += n
        mdl.t += 2 * n
        return final, ranks, preds

    # -- checkpoint / resume of the adaptation state (absent in the reference) --
    def state_dict(self):
        """Everything the stream carries from sample to sample: GMM state, text residuals and their Adam
        moments, counters.  Plain tensors / ints (loadable with torch.load(weights_only=True))."""
        self.synchronize()
        if self.class_mode:
            raise NotImplementedError("state_dict() in replay_mode='class': each rank holds only its class slice of the state")
        m, r = self.mode_dota_model, self.residuals
        sd = {'version': 1, 'K': self.K, 'M': self.M, 'D': self.D, 'i': int(self.i), 't': int(m.t), 'count': int(self.count),
              'correct': self.correct.detach().cpu().clone(),
              'mu': m.mu.detach().cpu().clone(), 'var': m.var.detach().cpu().clone(), 'pi': m.pi.detach().cpu().clone(),
              'c': m.c.detach().cpu().clone(), 'class_counts': m.class_counts.detach().cpu().clone()}
        if r is not None:
            sd.update(res=r.res.detach().cpu().clone(), adam_m=r.m.detach().cpu().clone(), adam_v=r.v.detach().cpu().clone(),
                      e=r.e.detach().cpu().clone(), norm=r.norm.detach().cpu().clone(), adam_step=int(r.step))
        return sd

    def load_state_dict(self, sd):
        """Resume from `state_dict()`; tensors are copied INTO the existing device buffers (the C-ABI state
        holds their addresses)."""
        if (sd['K'], sd['M'], sd['D']) != (self.K, self.M, self.D):
            raise ValueError(f"state is for K,M,D = {(sd['K'], sd['M'], sd['D'])}, engine has {(self.K, self.M, self.D)}")
        if ('res' in sd) != (self.residuals is not None):
            raise ValueError('residual-learning state present in one of checkpoint / engine only')
        self.synchronize()
        m, r = self.mode_dota_model, self.residuals
        for dst, key in ((m.mu, 'mu'), (m.var, 'var'), (m.pi, 'pi'), (m.c, 'c'), (m.class_counts, 'class_counts')):
            dst.copy_(sd[key].to(dst.device, dst.dtype))
        m.t = int(sd['t'])
        if r is not None:
            for dst, key in ((r.res, 'res'), (r.m, 'adam_m'), (r.v, 'adam_v'), (r.e, 'e'), (r.norm, 'norm')):
                dst.copy_(sd[key].to(dst.device, dst.dtype))
            r.step = int(sd['adam_step'])
            r.t.copy_(r.text0 + r.res)
        self.i, self.count = int(sd['i']), int(sd['count'])
        self.correct.copy_(sd['correct'].to(self.correct.device, self.correct.dtype))

    def synchronize(self):
        """Wait for all enqueued work of this engine (both streams)."""
        if self.replay_stream is not None:
            self.replay_stream.synchronize()
        torch.cuda.current_stream().synchronize()

    def accuracy(self):
        """{'acc1','acc3','acc5'} in percent over everything processed so far (host sync)."""
        self.synchronize()
        if self.world > 1 and self.replay_mode == 'owner':
            t = torch.cat([self.correct, torch.tensor([self.count], device=self.dev, dtype=torch.int64)])
            broadcast_from_owner(t, self.group)
            self.correct.copy_(t[:3]); self.count = int(t[3])
        c = self.correct.tolist()
        d = max(1, self.count)
        return {'acc1': 100.0 * c[0] / d, 'acc3': 100.0 * c[1] / d, 'acc5': 100.0 * c[2] / d}


def _run_full_covariance_dota(test_loader, model, args):
    """`--use-dota` without `--use-mode-dota` (Uni_Adapter.py:328-332,409-412): full-covariance DOTA, one sample at a
    time.  The reference never assigns `final_logits` on this branch (it cannot run it); the fusion here is the rule
    of the usage note dota_mixture.py:288-293, final = clip + clamp(rho * mean(c) / batch, max=eta) * dota."""
    from .dota import DOTA
    dev = torch.device(args.device if getattr(args, 'device', None) else 'cuda')
    ptf = getattr(args, 'precomputed_text_features', None)
    text = ptf if isinstance(ptf, torch.Tensor) else torch.load(ptf, map_location='cpu', weights_only=True)
    text = text.to(dev, torch.float32).contiguous()                                    # (K,D)
    K, D = text.shape
    cfg = {'epsilon': args.dota_epsilon, 'sigma': args.dota_sigma, 'eta': args.dota_eta, 'rho': args.dota_rho}
    dota = DOTA(cfg, D, K, torch.full((D, K), 0.001), prior_pre_steps=getattr(args, 'dota_prior_pre_steps', None), device=dev)   # :329-330
    model.eval()
    correct = torch.zeros(3, dtype=torch.int64, device=dev)
    count = 0
    fh = torch.empty(1, D, device=dev); z = torch.empty(1, K, device=dev); prob = torch.empty(1, K, device=dev)
    for pc, target, _, rgb in test_loader:
        pc, rgb, target = torch.as_tensor(pc), torch.as_tensor(rgb), torch.as_tensor(target).reshape(-1)
        feats = model.encode_pc(torch.cat((pc.float(), rgb.float()), -1).to(dev))
        for j in range(feats.shape[0]):
            lib.call('ua_clip_logits', lib.ptr(feats[j:j + 1].contiguous()), lib.ptr(text), K, D, LOGIT_SCALE, 1, lib.ptr(fh), lib.ptr(z),
                     lib.ptr(prob), None, None, lib.stream_ptr())                       # :406
            dz = dota.predict(fh.half())                                              # :410
            dota.fit(fh, prob); dota.update()                                         # :411-412
            final = dota.fuse(z, dz, 1, cfg['rho'], cfg['eta'])
            rank = (final[0] > final[0, int(target[j])]).sum()
            correct += torch.stack([(rank < k) for k in (1, 3, 5)]).to(torch.int64)
            count += 1
    c = correct.tolist()
    d = max(1, count)
    acc = {'acc1': 100.0 * c[0] / d, 'acc3': 100.0 * c[1] / d, 'acc5': 100.0 * c[2] / d}
    logging.info(f"Final Results: Acc@1 {acc['acc1']:.3f} Acc@3 {acc['acc3']:.3f} Acc@5 {acc['acc5']:.3f}")
    return acc


def get_hyperparams(dataset_name):
    """utils/hyperparams.py:1-38 (values of the paper's settings)."""
    params = {'shot_capacity': 30, 'beta': 150, 'threshold': 0.5, 'lambda_reg': 0.11, 'use_new_approximation': True}
    name = dataset_name.lower()
    if 'scanobject' in name:
        params['lambda_reg'] = 0.20
    elif 'shapenet' in name:
        params.update(lambda_reg=0.07, threshold=0.45, use_new_approximation=False)
    return params


def _run_cluster_cache(test_loader, model, args, hp):
    """Neither --use-dota nor --use-mode-dota: the original Uni-Adapter cluster cache (Uni_Adapter.py:357-361,541-574), one
    sample at a time.  The refinement follows `args.use_new_approximation` (default True: conjugate gradients), as the
    reference's loop does (:557), not the per-dataset `hp` entry."""
    from .uni_cache import ClusterCache
    dev = torch.device(args.device if getattr(args, 'device', None) else 'cuda')
    ptf = getattr(args, 'precomputed_text_features', None)
    text = ptf if isinstance(ptf, torch.Tensor) else torch.load(ptf, map_location='cpu', weights_only=True)
    hp = dict(hp if hp is not None else get_hyperparams(getattr(args, 'dataset_name', 'modelnet')))
    hp['use_new_approximation'] = bool(getattr(args, 'use_new_approximation', True))
    cache = ClusterCache(text, hp, device=dev)
    model.eval()
    correct = torch.zeros(3, dtype=torch.int64, device=dev)
    count = 0
    for pc, target, _, rgb in test_loader:
        pc, rgb, target = torch.as_tensor(pc), torch.as_tensor(rgb), torch.as_tensor(target).reshape(-1)
        feats = model.encode_pc(torch.cat((pc.float(), rgb.float()), -1).to(dev))
        for j in range(feats.shape[0]):
            final, _ = cache.step(feats[j:j + 1])
            rank = (final[0] > final[0, int(target[j])]).sum()
            correct += torch.stack([(rank < k) for k in (1, 3, 5)]).to(torch.int64)
            count += 1
    c = correct.tolist()
    d = max(1, count)
    acc = {'acc1': 100.0 * c[0] / d, 'acc3': 100.0 * c[1] / d, 'acc5': 100.0 * c[2] / d}
    logging.info(f"Final Results: Acc@1 {acc['acc1']:.3f} Acc@3 {acc['acc3']:.3f} Acc@5 {acc['acc5']:.3f}")
    return acc


# ----------------------------------------------------------------------------- entry point
def test_zeroshot_3d_core(test_loader, validate_dataset_name, model, clip_model, tokenizer, args, hp):
    """Same signature and return value as Uni_Adapter.py:272-595.  `test_loader` yields
    (pc, target, target_name, rgb) with any batch size; samples are consumed in stream order.
    Extra optional attributes read from `args` (never required):
      chunk_size (16), aug_noise (callable(sample_index0, pc) -> noise, for reproducible streams)."""
    if not getattr(args, 'use_mode_dota', True):
        if getattr(args, 'use_dota', False):
            return _run_full_covariance_dota(test_loader, model, args)
        return _run_cluster_cache(test_loader, model, args, hp)
    model.eval()
    dota_cfg = {'epsilon': args.dota_epsilon, 'sigma': args.dota_sigma, 'eta': args.dota_eta, 'rho': args.dota_rho}
    logging.info(f"MODE DOTA Hyperparameters: {dota_cfg}")
    logging.info(f"num MODES for MODE DOTA: {args.mode_M}")
    dev = torch.device(args.device if getattr(args, 'device', None) else 'cuda')
    ptf = getattr(args, 'precomputed_text_features', None)
    if isinstance(ptf, torch.Tensor):
        text_features = ptf
    elif ptf and os.path.exists(ptf):
        logging.info(f"Loading precomputed text features from {ptf}")
        text_features = torch.load(ptf, map_location='cpu', weights_only=True)               # :298-300
    else:
        raise FileNotFoundError('precomputed text features are required (the text tower is out of scope): '
                                '--precomputed-text-features')
    chunk = int(getattr(args, 'chunk_size', 16))
    eng = TtaEngine(model, text_features, dota_cfg, args.mode_M, res_learning=getattr(args, 'res_learning', True),
                    chunk=chunk, device=dev, log_alignment_every=getattr(args, 'log_alignment_every', 0),
                    overlap=getattr(args, 'overlap_replay', True))
    logging.info(f"Initialized MODE-DOTA model with M={args.mode_M}.")
    noise_fn = getattr(args, 'aug_noise', None)
    start_event, end_event = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start_event.record()
    buf_pc, buf_rgb, buf_t = [], [], []
    seen = 0

    def flush():
        nonlocal seen
        if not buf_pc:
            return
        pc, rgb, t = torch.cat(buf_pc), torch.cat(buf_rgb), torch.cat(buf_t)
        noise = torch.cat([noise_fn(seen + j, pc[j:j + 1]) for j in range(pc.shape[0])]) if noise_fn else None
        if seen == 0 and eng.world > 1 and eng.replay_mode == 'owner':      # measure the replay / encoder cost ratio on the first clouds
            m = min(64, pc.shape[0])           # the whole first chunk up to 64 samples: the encoder's per-sample cost depends on its batch
            rho = eng.calibrate_replay_cost(pc[:m], rgb[:m], noise[:m] if noise is not None else torch.randn(m, pc.shape[1], 3))
            logging.info(f"owner replay mode: measured replay/encoder cost ratio {rho:.3f}")
        eng.process_chunk(pc, rgb, t, noise)
        seen += pc.shape[0]
        buf_pc.clear(); buf_rgb.clear(); buf_t.clear()

    for pc, target, target_name, rgb in test_loader:
        pc, rgb, target = torch.as_tensor(pc), torch.as_tensor(rgb), torch.as_tensor(target).reshape(-1)
        for j in range(pc.shape[0]):
            buf_pc.append(pc[j:j + 1]); buf_rgb.append(rgb[j:j + 1]); buf_t.append(target[j:j + 1])
            if len(buf_pc) == chunk:
                flush()
    flush()
    if eng.replay_stream is not None:
        torch.cuda.current_stream().wait_stream(eng.replay_stream)
    end_event.record()
    torch.cuda.synchronize()
    acc = eng.accuracy()
    logging.info(f"Final Results: Acc@1 {acc['acc1']:.3f} Acc@3 {acc['acc3']:.3f} Acc@5 {acc['acc5']:.3f}")
    logging.info(f"Total time: {start_event.elapsed_time(end_event):.3f} ms")
    return acc
