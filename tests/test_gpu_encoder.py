"""GPU parity of the Uni3D encoder forward (HIP kernels through the C ABI) against the CPU oracle
on seeded synthetic weights of the reference's state-dict layout.  The timm EVA02 block structure
is third-party and unpinned (see oracle/ref_cpu.py header); the oracle itself is pinned for the
grouping / mini-PointNet parts by tests/golden/g5-g7."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ref_cpu as O


def _clouds(B, N, seed):
    g = torch.Generator().manual_seed(seed)
    xyz = torch.randn(B, N, 3, generator=g)
    xyz = xyz / xyz.norm(dim=-1, keepdim=True).max()
    rgb = torch.rand(B, N, 3, generator=g)
    return torch.cat([xyz, rgb], -1)


@pytest.mark.parametrize('precision', ['f32', 'split'])
@pytest.mark.parametrize('B,trans_dim,depth,G,embed', [(2, 128, 2, 128, 64), (3, 256, 1, 256, 128), (1, 128, 2, 128, 64)])
def test_encode_pc_small(B, trans_dim, depth, G, embed, precision):
    from uni_adapter_amd.point_encoder import create_uni3d_from_state_dict
    sd = O.make_uni3d_state_dict(7, trans_dim=trans_dim, embed_dim=embed, enc_dim=512, depth=depth)
    pc = _clouds(B, 1024, 3)
    model = create_uni3d_from_state_dict(sd, num_group=G, group_size=64, num_heads=trans_dim // 64, precision=precision)
    model.eval()
    out, tokens = model.point_encoder(pc[:, :, :3].cuda(), pc[:, :, 3:].cuda(), return_tokens=True)
    sd64 = {k: v.double() for k, v in sd.items()}
    ref, rtok = O.uni3d_encode_pc(pc.double(), sd64, G, 64, trans_dim // 64, return_tokens=True)
    ref32 = O.uni3d_encode_pc(pc, sd, G, 64, trans_dim // 64)
    err = (out.cpu().double() - ref).abs().max().item()
    base = (ref32.double() - ref).abs().max().item()
    scale = ref.abs().max().item()
    assert err <= max(5 * base, 2e-5 * scale), (err, base, scale)
    assert (tokens.cpu().double() - rtok).abs().max().item() <= 1e-4 * rtok.abs().max().item()
    # encode_pc == point_encoder(xyz, rgb) and cosine vs oracle
    out2 = model.encode_pc(pc.cuda())
    assert torch.equal(out2, out)
    cos = torch.nn.functional.cosine_similarity(out.cpu().double(), ref, dim=1)
    assert (1 - cos).max().item() < 1e-9


@pytest.mark.parametrize('precision', ['f32', 'split'])
def test_encode_pc_uni3d_large_shape_one_block(precision):
    """Full Uni3D-L width (1024, 16 heads, hidden 2730, G=512) with 2 of the 24 blocks: exercises
    the real GEMM shapes incl. the K=2730->2752 padding and the SwiGLU interleave, in both GEMM
    modes (exact-f32 MFMA / 3xf16-split MFMA) at the SAME tolerance."""
    from uni_adapter_amd.point_encoder import create_uni3d_from_state_dict
    sd = O.make_uni3d_state_dict(11, trans_dim=1024, embed_dim=1024, enc_dim=512, depth=2)
    pc = _clouds(2, 1024, 5)
    model = create_uni3d_from_state_dict(sd, num_group=512, group_size=64, num_heads=16, precision=precision)
    out = model.encode_pc(pc.cuda())
    ref32 = O.uni3d_encode_pc(pc, sd, 512, 64, 16)
    cos = torch.nn.functional.cosine_similarity(out.cpu().double(), ref32.double(), dim=1)
    assert (1 - cos).max().item() < 1e-8
    assert (out.cpu() - ref32).abs().max().item() <= 1e-4 * ref32.abs().max().item()


def test_encoder_batch_invariance():
    """A cloud's feature must not depend on what else is in the chunk (sample sharding relies on it)."""
    from uni_adapter_amd.point_encoder import create_uni3d_from_state_dict
    sd = O.make_uni3d_state_dict(7, trans_dim=128, embed_dim=64, enc_dim=512, depth=2)
    pc = _clouds(5, 1024, 9).cuda()
    model = create_uni3d_from_state_dict(sd, num_group=128, group_size=64, num_heads=2, pe_clouds=2)
    full = model.encode_pc(pc)
    one = torch.cat([model.encode_pc(pc[i:i + 1]) for i in range(5)])
    assert torch.equal(full, one)


def test_split_mode_tracks_f32_mode_full_depth():
    """24-block Uni3D-L: features of the split-GEMM mode vs the exact-f32 mode (both HIP) - the
    deviation must stay at fp32 round-off level (cosine 1 - 1e-9, 100*cos logits within 1e-3)."""
    from uni_adapter_amd.point_encoder import create_uni3d_from_state_dict
    from uni_adapter_amd import synthetic
    sd = synthetic.uni3d_state_dict(0, depth=24, device='cuda')
    pc = _clouds(4, 1024, 13).cuda()
    text = synthetic.text_bank(40, 1024, 0, device='cuda')
    outs = {}
    for prec in ('f32', 'split'):
        m = create_uni3d_from_state_dict(sd, num_group=512, group_size=64, num_heads=16, precision=prec)
        outs[prec] = torch.nn.functional.normalize(m.encode_pc(pc).double(), dim=1)
        del m
    cos = (outs['f32'] * outs['split']).sum(1)
    dl = (100.0 * (outs['f32'] - outs['split']) @ text.double().T).abs().max().item()
    print('1-cos', (1 - cos).max().item(), 'max |d logit|', dl)
    assert (1 - cos).max().item() < 1e-10
    assert dl < 1e-3


def test_folded_layernorms_track_the_layernorm_kernels_full_depth():
    """24-block Uni3D-L in split mode: LayerNorms folded into the GEMMs (default) vs run as kernels (fold_layernorm=False), and each
    against the exact-f32 mode.  The fold is an algebraic re-association: the features must agree at fp32
    round-off level, and the folded path must be no further from the exact-f32 mode than 3x the un-folded one."""
    from uni_adapter_amd.point_encoder import create_uni3d_from_state_dict
    from uni_adapter_amd import synthetic
    sd = synthetic.uni3d_state_dict(0, depth=24, device='cuda')
    pc = _clouds(5, 1024, 17).cuda()                                 # 5 clouds: a ragged last row panel in every GEMM
    text = synthetic.text_bank(40, 1024, 0, device='cuda')
    outs = {}
    for name, kw in (('f32', dict(precision='f32')), ('fold', dict(precision='split', fold_layernorm=True)),
                     ('kernels', dict(precision='split', fold_layernorm=False))):
        m = create_uni3d_from_state_dict(sd, num_group=512, group_size=64, num_heads=16, **kw)
        assert m.point_encoder.fold_ln == (name == 'fold')
        outs[name] = torch.nn.functional.normalize(m.encode_pc(pc).double(), dim=1)
        if name == 'fold':                                           # the one-time check of the row statistics ran and found benign rows
            assert m.point_encoder._fold_checked and 0.0 <= m.point_encoder.fold_mean2_over_var < 100.0, m.point_encoder.fold_mean2_over_var
        del m
    logit = lambda a, b: (100.0 * (outs[a] - outs[b]) @ text.double().T).abs().max().item()
    d_fk, d_f32_fold, d_f32_k = logit('fold', 'kernels'), logit('f32', 'fold'), logit('f32', 'kernels')
    print('max |d logit|: fold vs kernels', d_fk, '| f32 vs fold', d_f32_fold, '| f32 vs kernels', d_f32_k)
    assert (1 - (outs['fold'] * outs['kernels']).sum(1)).max().item() < 1e-10
    assert d_fk < 1e-3 and d_f32_fold < 1e-3
    assert d_f32_fold <= 3.0 * d_f32_k + 1e-5


def test_folded_layernorm_falls_back_to_the_kernels_when_its_first_forward_check_fails(monkeypatch):
    """The fold is checked on EVERY block of the first forward (max |x| against the fp16 range of the split residual stream,
    mean^2/var of the rows whose one-pass statistics were written).  A failing check switches the encoder to the LayerNorm
    kernels and re-runs the blocks: the output is then bit for bit the un-folded encoder's, first forward included."""
    import warnings
    from uni_adapter_amd.point_encoder import create_uni3d_from_state_dict, PointcloudEncoder
    sd = O.make_uni3d_state_dict(5, trans_dim=256, embed_dim=128, enc_dim=512, depth=3)
    pc = _clouds(3, 1024, 9).cuda()
    ref = create_uni3d_from_state_dict(sd, num_group=128, group_size=64, num_heads=4, precision='split', fold_layernorm=False)
    want = ref.encode_pc(pc)
    ok = create_uni3d_from_state_dict(sd, num_group=128, group_size=64, num_heads=4, precision='split', fold_layernorm=True)
    got_fold = ok.encode_pc(pc)
    pe = ok.point_encoder
    assert pe.fold_ln and pe._fold_checked and pe.fold_max_abs > 0.0 and pe.fold_max_abs < PointcloudEncoder.FOLD_MAX_ABS
    assert (got_fold - want).abs().max().item() <= 1e-4 * want.abs().max().item()
    m = create_uni3d_from_state_dict(sd, num_group=128, group_size=64, num_heads=4, precision='split', fold_layernorm=True)
    m.point_encoder.FOLD_MAX_ABS = 0.5 * pe.fold_max_abs             # the same stream now 'overflows' the split copy
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter('always')
        got = m.encode_pc(pc)
    assert not m.point_encoder.fold_ln and any('switched off' in str(r.message) for r in rec)
    assert torch.equal(got, want)
    assert torch.equal(m.encode_pc(pc), want)                        # and stays on the kernels


@pytest.mark.parametrize('family,fold', [('uni3d', True), ('uni3d', False), ('uni3d_f32', False), ('openshape', True)])
def test_last_block_for_the_cls_rows_only_keeps_the_bits(family, fold, monkeypatch):
    """The reference reads only the CLS token after the last block (models/point_encoder.py:219 `norm(x[:, 0])`; OpenShape
    projects the CLS token), so that block's q / attention / proj / MLP for the patch tokens are dead values.  The encoder then runs
    them for the CLS rows only (k and v still for all rows): the features must equal the full block's bit for bit, with and without
    the folded LayerNorms, and `return_tokens=True` must still return every token of the full block."""
    from uni_adapter_amd import synthetic
    from uni_adapter_amd.point_encoder import create_uni3d_from_state_dict, create_openshape_from_state_dict
    monkeypatch.delenv('UA_LAST_BLOCK_CLS_ONLY', raising=False)
    pc = _clouds(3, 1024, 23).cuda()
    if family.startswith('uni3d'):
        sd = O.make_uni3d_state_dict(9, trans_dim=256, embed_dim=128, enc_dim=512, depth=3)
        m = create_uni3d_from_state_dict(sd, num_group=256, group_size=64, num_heads=4, precision='f32' if family.endswith('f32') else 'split',
                                         fold_layernorm=fold)
        run = lambda: m.encode_pc(pc)
        run_tok = lambda: m.point_encoder(pc[:, :, :3].contiguous(), pc[:, :, 3:].contiguous(), return_tokens=True)
    else:
        m = create_openshape_from_state_dict(synthetic.openshape_state_dict(2, depth=2, out_dim=256, device='cuda'), patches=128, device='cuda',
                                             precision='split', fold_layernorm=fold)
        run = lambda: m(pc[:, :, :3].contiguous(), pc.contiguous())
        run_tok = None
    enc = m.point_encoder if hasattr(m, 'point_encoder') else m.ppat
    assert enc.last_block_cls_only
    lean = run().clone()
    enc.last_block_cls_only = False
    full = run().clone()
    assert torch.equal(lean, full)
    if run_tok is not None:
        enc.last_block_cls_only = True
        out_t, tokens = run_tok()                                   # tokens requested: the block runs in full whatever the switch says
        assert torch.equal(out_t, full) and torch.isfinite(tokens).all()
        enc.last_block_cls_only = False
        _, tokens_full = run_tok()
        assert torch.equal(tokens, tokens_full)


def test_split_schedule_ab_switch_disables_the_fold(monkeypatch):
    """UA_SPLIT_SCHEDULE=8|2 (A/B main loops without the fold prologue) must not leave the folded GEMM calls in place."""
    from uni_adapter_amd.point_encoder import create_uni3d_from_state_dict
    sd = O.make_uni3d_state_dict(5, trans_dim=128, embed_dim=64, enc_dim=512, depth=1)
    monkeypatch.delenv('UA_FOLD_LN', raising=False)                  # (the suite is also run under UA_FOLD_LN=0)
    monkeypatch.setenv('UA_SPLIT_SCHEDULE', '8')
    m = create_uni3d_from_state_dict(sd, num_group=128, group_size=64, num_heads=2, precision='split')
    assert not m.point_encoder.fold_ln
    monkeypatch.delenv('UA_SPLIT_SCHEDULE')
    m = create_uni3d_from_state_dict(sd, num_group=128, group_size=64, num_heads=2, precision='split')
    assert m.point_encoder.fold_ln


# ---------------------------------------------------------------- ULIP-2 PointBERT variant (section 8 A4b)
def _golden(name):
    import os
    return np.load(os.path.join(os.path.dirname(__file__), 'golden', name))


@pytest.mark.parametrize('precision', ['f32', 'split'])
def test_ulip_encoder_against_reference_fixture(precision):
    """tests/golden/g8_ulip.npz holds the reference PointTransformer's own output (2 blocks, real
    widths) on seeded weights; the HIP encoder must agree with it within fp32 noise."""
    from uni_adapter_amd.point_encoder import create_ulip_from_state_dict
    g = _golden('g8_ulip.npz')
    sd = O.make_ulip_state_dict(int(g['weights_seed']), trans_dim=384, depth=2)
    model = create_ulip_from_state_dict(sd, num_group=128, group_size=32, num_heads=6, precision=precision, fps_start=3)
    xyz = torch.from_numpy(g['xyz'])
    out = model(xyz.cuda())
    ref = torch.from_numpy(g['embed'])
    sd64 = {k: v.double() for k, v in sd.items()}
    truth = O.ulip_encode_pc(xyz.double(), sd64, 128, 32, 6, center_idx=torch.from_numpy(g['center_idx']).long())
    base = (ref.double() - truth).abs().max().item()
    err = (out.cpu().double() - truth).abs().max().item()
    scale = truth.abs().max().item()
    assert err <= max(5 * base, 2e-5 * scale), (err, base, scale)
    assert (out.cpu() - ref).abs().max().item() <= 1e-4 * scale


@pytest.mark.parametrize('precision', ['f32', 'split'])
def test_ulip_encoder_full_depth_and_logits_wrapper(precision):
    """ULIP-2 shape (384 wide, 12 blocks, G=512, S=32, 8192->first 1024 points not needed: N=1024 here)
    end to end through get_logits_wrapper(vlm3d='ulip') vs the oracle."""
    import types
    from uni_adapter_amd.point_encoder import create_ulip_from_state_dict
    from uni_adapter_amd.Uni_Adapter import get_logits_wrapper
    from uni_adapter_amd import synthetic
    sd = synthetic.ulip_state_dict(5, depth=12)
    pc = _clouds(3, 2048, 9)
    model = create_ulip_from_state_dict(sd, num_group=512, group_size=32, num_heads=6, precision=precision)
    out = model(pc[:, :, :3].contiguous().cuda())
    sdo = {k: v for k, v in sd.items()}
    ref = O.ulip_encode_pc(pc[:, :, :3], sdo, 512, 32, 6)
    cos = torch.nn.functional.cosine_similarity(out.cpu().double(), ref.double(), dim=1)
    assert (1 - cos).max().item() < 1e-8
    assert (out.cpu() - ref).abs().max().item() <= 2e-4 * ref.abs().max().item()
    text = synthetic.text_bank(40, 512, seed=2)
    args = types.SimpleNamespace(vlm3d='ulip')
    fh, z, ent, p, pred = get_logits_wrapper(args, model, pc[:1].cuda(), text.t().contiguous().cuda())
    rz = 100.0 * torch.nn.functional.normalize(ref[:1], dim=1) @ text.t()
    assert (z.cpu() - rz).abs().max().item() < 1e-3
    assert pred == int(rz.argmax())


# ---------------------------------------------------------------- OpenShape PPAT variant (section 8 F1)
def test_ball_group_kernel_vs_reference_fixture():
    """ua_ball_group against the reference's own query_ball_point / sample_and_group output (g9)."""
    import uni_adapter_amd  # noqa: F401
    from uni_adapter_amd import lib
    g = _golden('g9_ppat.npz')
    xyz, feats = torch.from_numpy(g['xyz']).cuda(), torch.from_numpy(g['feats']).cuda()
    cidx = torch.from_numpy(g['center_idx'])
    B, N, _ = xyz.shape; G, S, Fc = cidx.shape[1], 64, feats.shape[2]
    centers = xyz[torch.arange(B)[:, None], cidx.cuda()].contiguous()
    out = torch.full((B * G * S, 12), 7.0, device='cuda'); idx = torch.empty(B, G, S, dtype=torch.int32, device='cuda')
    lib.call('ua_ball_group', lib.ptr(xyz), lib.ptr(feats), lib.ptr(centers), B, N, G, S, Fc, 0.2, lib.ptr(out), 12, lib.ptr(idx, torch.int32),
             lib.stream_ptr())
    want = torch.from_numpy(g['ball_idx']).long()
    got = idx.cpu().long()
    same = (got == want).all(-1)
    assert same.float().mean() >= 0.995, same.float().mean()       # boundary points (d ~ r^2) may fall either side of a rounding
    ar = torch.arange(B)[:, None, None]
    ref_rows = torch.cat([xyz.cpu()[ar, want] - centers.cpu()[:, :, None, :], feats.cpu()[ar, want]], -1)
    rows = out.cpu().view(B, G, S, 12)
    assert torch.equal(rows[same][..., :9], ref_rows[same]) and bool((rows[..., 9:] == 0).all())
    # groups with fewer than S hits are padded with the first hit (pointnet_util.py:106-108)
    assert bool((want[..., -1] == want[..., 0]).any())


@pytest.mark.parametrize('precision', ['f32', 'split'])
def test_openshape_encoder_vs_oracle_and_reference_parts(precision):
    """PPAT forward (SA -> lift -> LN -> blocks -> proj) vs the oracle, whose SA layer and transformer blocks are
    pinned to the reference's own classes by g9; 2 blocks at the real widths (dim 512, 8 heads, mlp 1536)."""
    from uni_adapter_amd.point_encoder import create_openshape_from_state_dict
    g = _golden('g9_ppat.npz')
    sd = O.make_ppat_state_dict(int(g['weights_seed']), depth=2)
    xyz, feats = torch.from_numpy(g['xyz']), torch.from_numpy(g['feats'])
    model = create_openshape_from_state_dict(sd, patches=128, nsample=64, radius=0.2, num_heads=8, precision=precision, fps_start=1)
    out, tokens = model.point_encoder(xyz.cuda(), feats.cuda(), return_tokens=True)
    ref, cen, sa, tok, y = O.ppat_encode(xyz, feats, sd, 128, 0.2, 64, 8, center_idx=torch.from_numpy(g['center_idx']), return_parts=True)
    sd64 = {k: v.double() for k, v in sd.items()}
    truth = O.ppat_encode(xyz.double(), feats.double(), sd64, 128, 0.2, 64, 8, center_idx=torch.from_numpy(g['center_idx']))
    base = (ref.double() - truth).abs().max().item()
    err = (out.cpu().double() - truth).abs().max().item()
    scale = truth.abs().max().item()
    assert err <= max(5 * base, 5e-5 * scale), (err, base, scale)
    # token rows after the blocks against the reference Transformer's output (it was fed the oracle's token input)
    assert (tokens.cpu() - torch.from_numpy(g['tokens_out'])).abs().max().item() <= 2e-4 * float(np.abs(g['tokens_out']).max())


def test_openshape_full_depth_logits_wrapper():
    """'scaling 4' shape (384 patches, 12 blocks, out 1280) through get_logits_wrapper(vlm3d='openshape')."""
    import types
    from uni_adapter_amd import synthetic
    from uni_adapter_amd.point_encoder import create_openshape_from_state_dict
    from uni_adapter_amd.Uni_Adapter import get_logits_wrapper
    sd = synthetic.openshape_state_dict(6, depth=12)
    pc = _clouds(2, 1024, 13)
    model = create_openshape_from_state_dict(sd, precision='split')
    out = model(pc[:, :, :3].contiguous().cuda(), pc.cuda())
    ref = O.ppat_encode(pc[:, :, :3].contiguous(), pc, sd, 384, 0.2, 64, 8)
    cos = torch.nn.functional.cosine_similarity(out.cpu().double(), ref.double(), dim=1)
    assert (1 - cos).max().item() < 1e-8
    assert (out.cpu() - ref).abs().max().item() <= 2e-4 * ref.abs().max().item()
    text = synthetic.text_bank(15, 1280, seed=2)
    fh, z, ent, p, pred = get_logits_wrapper(types.SimpleNamespace(vlm3d='openshape'), model, pc[:1].cuda(), text.t().contiguous().cuda())
    rz = 100.0 * torch.nn.functional.normalize(ref[:1], dim=1) @ text.t()
    assert (z.cpu() - rz).abs().max().item() < 1e-3 and pred == int(rz.argmax())


@pytest.mark.parametrize('family', ['ulip', 'openshape'])
def test_other_encoders_are_batch_invariant_across_front_end_chunks(family):
    """11 clouds = two mini-PointNet / set-abstraction chunks (8 + 3): every row equals the single-cloud result."""
    from uni_adapter_amd import synthetic
    from uni_adapter_amd.point_encoder import create_ulip_from_state_dict, create_openshape_from_state_dict
    if family == 'ulip':
        model = create_ulip_from_state_dict(synthetic.ulip_state_dict(3, depth=2, device='cuda'), num_group=128, device='cuda', precision='split')
    else:
        model = create_openshape_from_state_dict(synthetic.openshape_state_dict(3, depth=2, device='cuda'), patches=128, device='cuda', precision='split')
    pc = _clouds(11, 1024, 21).cuda()
    full = model.encode_pc(pc)
    for i in (0, 7, 8, 10):
        one = model.encode_pc(pc[i:i + 1])
        assert torch.equal(one[0], full[i]), i
