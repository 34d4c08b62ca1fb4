"""GPU parity of the whole per-sample loop (uni_adapter_amd.Uni_Adapter) against (a) the golden
streams produced by the reference's own loop body and (b) the CPU oracle end to end."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ref_cpu as O


def _load(golden_dir, name):
    return {k: v for k, v in np.load(os.path.join(golden_dir, name)).items()}


class FeatureFeeder(torch.nn.Module):
    """Stands in for the encoder exactly like the golden generator did: returns queued features."""
    def __init__(self, feats):
        super().__init__()
        self.feats = feats          # (n,2,D) device
        self.pos = 0

    def encode_pc(self, clouds):
        n2 = clouds.shape[0]
        out = self.feats.view(-1, self.feats.shape[-1])[self.pos:self.pos + n2]
        self.pos += n2
        return out.contiguous()


def _run_hip_stream(g, n, chunk, res_learning=True, with_targets=True, align_path=None):
    from uni_adapter_amd.Uni_Adapter import TtaEngine
    text = torch.from_numpy(g['text'])
    feats = torch.stack([torch.from_numpy(g['feat'][:n]), torch.from_numpy(g['feat_aug'][:n])], 1).cuda()
    eng = TtaEngine(FeatureFeeder(feats), text, {'epsilon': 1e-4, 'sigma': 1e-4, 'eta': 0.1, 'rho': 0.02}, int(g['M']), res_learning=res_learning,
                    chunk=chunk, align_path=align_path)
    finals, preds = [], []
    tgt = torch.from_numpy(g['labels'][:n]).int()
    for s0 in range(0, n, chunk):
        m = min(chunk, n - s0)
        f, r, p = eng.process_chunk(torch.zeros(m, 8, 3), torch.ones(m, 8, 3), tgt[s0:s0 + m] if with_targets else None)
        finals.append(f.cpu()); preds.append(p.cpu())
    return torch.cat(finals).double().numpy(), torch.cat(preds).numpy(), eng


@pytest.mark.parametrize('name,chunk', [('modelnet40', 16), ('scanobjectnn', 5), ('shapenet55', 1)])
def test_loop_with_residual_learning_vs_reference_golden(golden_dir, name, chunk):
    """The full --use-mode-dota loop (clip logits, predict, fit x2, 10-step residual learning, fusion) on the HIP path,
    fed with the golden encoder outputs, against what the reference's own loop body produced on the same stream - ALL
    golden steps (48 / 24 / 16):
      * the prediction of every step equals the reference's;
      * the first five samples are within the north-star bar of 1e-3 on the final logits.
    Later samples of a residual-learning stream cannot be held to 1e-3 by ANY fp32 evaluation, the reference's own
    included: tests/golden/g12 holds the reference run against itself (other thread count, permuted summation orders,
    fp64) and those runs are 1.3 apart at sample 5 - see test_long_stream_against_the_references_own_spread, which
    asserts the envelope; here the distance is only reported."""
    g = _load(golden_dir, f'g2_stream_{name}.npz')
    n = g['feat'].shape[0]
    final, pred, eng = _run_hip_stream(g, n, chunk)
    gold = g['final'].astype(np.float64)
    e = np.abs(final - gold).max(1)
    print(name, 'HIP vs reference, max |dlogit| per step:', np.round(e, 5).tolist())
    assert np.array_equal(pred, g['pred']), np.nonzero(pred != g['pred'])[0]
    assert np.array_equal(pred, gold.argmax(1))
    assert (e[:5] <= 1e-3).all(), e[:5]
    acc = eng.accuracy()
    want1 = 100.0 * float((g['pred'] == g['labels']).mean())
    assert abs(acc['acc1'] - want1) < 1e-9


def _spread_stats(T, P, x, S):
    """Leave-one-out statistics of trajectory `x` against the set S of other trajectories (final logits T[v] (n,K), predictions P[v]):
    diam[i] = largest pairwise distance inside S at step i; nearest / farthest distance of x to S."""
    dist = lambda u, v: np.abs(T[u] - T[v]).max(1)
    n = T[x].shape[0]
    diam = np.zeros(n)
    for i, u in enumerate(S):
        for v in S[i + 1:]:
            diam = np.maximum(diam, dist(u, v))
    d = np.stack([dist(x, v) for v in S])
    unanimous = np.all(np.stack([P[v] for v in S]) == P[S[0]][None], 0)
    return dict(diam=diam, nearest=d.min(0), farthest=d.max(0),
                over_nearest=int((d.min(0) > np.maximum(diam, 1e-3)).sum()), over_farthest=int((d.max(0) > np.maximum(diam, 1e-3)).sum()),
                over_2x=int((d.min(0) > np.maximum(2 * diam, 1e-3)).sum()),
                unanimous=int(unanimous.sum()), differs_where_unanimous=int(((P[x] != P[S[0]]) & unanimous).sum()))


@pytest.mark.parametrize('bank', ['modelnet40', 'scanobjectnn'])
def test_long_stream_against_the_references_own_spread(golden_dir, bank):
    """256-sample residual-learning streams (ModelNet40 bank, K = 40; ScanObjectNN bank, K = 15 - there the 8-thread run equals the
    1-thread run bit for bit, the permuted and the fp64 runs agree with it on only 44-48 % of the predictions).  tests/golden/g12 holds the reference's OWN loop body run
    five ways on it (oracle/make_golden.py: a = as is, b = 8 threads, c = feature dimension permuted, d = class order
    permuted - all mathematically the same stream in fp32 - and e = fp64).  Those runs agree to < 1e-3 on samples 0-4,
    are 1.3 apart at sample 5, ~25 apart from sample 64 on, and disagree on 8-22 % of the predictions: the reference is
    chaotic in fp32 (gradient through likelihood_matrix.max(), Uni_Adapter.py:247-248, and sign-like first Adam steps).
    The HIP path is therefore held to what the reference holds against itself, every bound calibrated by leave-one-out
    over the five reference runs (no threshold chosen by this test except the stated slack factors):
      1. while the reference runs still agree (their diameter <= 1e-3) HIP is within 1e-3 of run a and predicts the same;
      2. HIP is no farther from the nearest reference run than twice the runs' own diameter (max(1e-3, .)) - at EVERY step on the
         ModelNet40 stream, as each run is against the other four; on the ScanObjectNN stream one run exceeds that at one step, and HIP
         may do so no more often than the worst run;
      3. the number of steps where HIP is farther from its nearest run than the diameter (1x) is within the leave-one-out
         range of the reference runs, and so is the literal "farther from SOME run than the diameter" count;
      4. where all five reference runs predict the same class HIP predicts it too, up to the leave-one-out disagreement
         rate of the runs themselves (x2 slack); top-1 accuracy lies in the runs' range (+-5 points)."""
    g = _load(golden_dir, f'g12_long_stream_{bank}.npz')
    n = g['feat'].shape[0]
    final, pred, eng = _run_hip_stream(g, n, 32)
    V = [str(v) for v in g['variants']]
    T = {v: g['final_' + v].astype(np.float64) for v in V}
    P = {v: g['pred_' + v] for v in V}
    T['hip'], P['hip'] = final, pred
    loo = {v: _spread_stats(T, P, v, [u for u in V if u != v]) for v in V}
    h = _spread_stats(T, P, 'hip', V)
    for v in V:
        print('reference run', v, {k: loo[v][k] for k in ('over_nearest', 'over_farthest', 'over_2x', 'unanimous', 'differs_where_unanimous')})
    print('HIP', {k: h[k] for k in ('over_nearest', 'over_farthest', 'over_2x', 'unanimous', 'differs_where_unanimous')})
    print('HIP vs run a, first 24 steps:', np.round(np.abs(final - T['a']).max(1)[:24], 4).tolist())
    print('diameter of the runs, first 24 steps:', np.round(h['diam'][:24], 4).tolist())
    calm = int(np.argmax(h['diam'] > 1e-3)) if (h['diam'] > 1e-3).any() else n                  # 1
    assert calm >= 3, calm
    assert (np.abs(final - T['a']).max(1)[:calm] <= 1e-3).all()
    assert np.array_equal(pred[:calm], P['a'][:calm])
    assert h['over_2x'] <= max(loo[v]['over_2x'] for v in V), (h['over_2x'], [loo[v]['over_2x'] for v in V])        # 2
    assert h['over_nearest'] <= max(loo[v]['over_nearest'] for v in V), (h['over_nearest'], [loo[v]['over_nearest'] for v in V])        # 3
    assert h['over_farthest'] <= max(loo[v]['over_farthest'] for v in V), (h['over_farthest'], [loo[v]['over_farthest'] for v in V])
    worst = max(loo[v]['differs_where_unanimous'] / max(1, loo[v]['unanimous']) for v in V)       # 4
    assert h['differs_where_unanimous'] / h['unanimous'] <= 2 * worst, (h['differs_where_unanimous'], h['unanimous'], worst)
    accs = [float((P[v] == g['labels']).mean()) for v in V]
    acc = float((pred == g['labels']).mean())
    assert min(accs) - 0.05 <= acc <= max(accs) + 0.05, (acc, accs)
    assert abs(eng.accuracy()['acc1'] - 100.0 * acc) < 1e-9


def test_mfma_alignment_path_through_the_engine(golden_dir):
    """TtaEngine(align_path='mfma'): the residual loop on the matrix cores (csrc/ua_align_mfma.hip) driven by the composite
    replay, on the first 64 samples of the long stream: within 1e-2 of the reference while its own runs agree (see below),
    identical predictions there, never farther from the nearest reference run than twice the runs' diameter, and the
    per-kernel (non-composite) path gives bit-identical results."""
    from uni_adapter_amd.Uni_Adapter import TtaEngine
    g = _load(golden_dir, 'g12_long_stream_modelnet40.npz')
    n = 64
    final, pred, eng = _run_hip_stream(g, n, 16, align_path='mfma')
    assert eng.align_mfma
    V = [str(v) for v in g['variants']]
    T = {v: g['final_' + v][:n].astype(np.float64) for v in V}
    P = {v: g['pred_' + v][:n] for v in V}
    T['hip'], P['hip'] = final, pred
    h = _spread_stats(T, P, 'hip', V)
    calm = int(np.argmax(h['diam'] > 1e-3))
    # The GEMM form evaluates grad_e = w.(mu/v) - e * w.(1/v) as a difference of two products, so a gradient element that is
    # nearly zero carries the fp32-class error of the (30 x larger) products; Adam's first steps are -lr * sign(g), and a
    # flipped sign moves a residual by 2e-3.  Measured on this stream: 1.2e-3 / 5.6e-3 at samples 2 / 3 (the VALU kernels,
    # which sum (mu - e)/v directly in fp64: 3e-4) - one reason the VALU path is the default.  Bound here: 1e-2.
    assert calm >= 3 and (np.abs(final - T['a']).max(1)[:calm] <= 1e-2).all() and np.array_equal(pred[:calm], P['a'][:calm])
    print('mfma alignment path vs run a, calm prefix:', np.abs(final - T['a']).max(1)[:calm].tolist())
    assert (h['nearest'] > np.maximum(2 * h['diam'], 1e-2)).sum() == 0
    # composite entry == per-kernel calls on this path too
    text = torch.from_numpy(g['text'])
    feats = torch.stack([torch.from_numpy(g['feat'][:8]), torch.from_numpy(g['feat_aug'][:8])], 1).cuda()
    e2 = TtaEngine(FeatureFeeder(feats), text, {'epsilon': 1e-4, 'sigma': 1e-4, 'eta': 0.1, 'rho': 0.02}, int(g['M']), chunk=8, composite=False, align_path='mfma')
    f2 = e2.process_chunk(torch.zeros(8, 8, 3), torch.ones(8, 8, 3))[0].cpu().double().numpy()
    assert np.array_equal(f2, final[:8])


def test_long_stream_without_residual_learning_holds_the_1e_3_bar(golden_dir):
    """The same 256-sample stream with residual learning off (no chaotic element): final logits within a plain 1e-3 of the
    CPU oracle in fp64 at every step, predictions identical at every step, and closer to the fp64 result than the fp32
    oracle is (the kernels sum in fp64)."""
    g = _load(golden_dir, 'g12_long_stream_modelnet40.npz')
    n = g['feat'].shape[0]
    final, pred, _ = _run_hip_stream(g, n, 32, res_learning=False, with_targets=False)
    text = torch.from_numpy(g['text'])
    s64 = O.TtaStream(text, O.TtaConfig(num_modes=int(g['M']), res_learning=False), dtype=torch.float64)
    s32 = O.TtaStream(text, O.TtaConfig(num_modes=int(g['M']), res_learning=False))
    e64, e32 = np.zeros(n), np.zeros(n)
    for i in range(n):
        f, fa = torch.from_numpy(g['feat'][i:i + 1]), torch.from_numpy(g['feat_aug'][i:i + 1])
        t64 = s64.step(f.double(), fa.double())[2][0].numpy()
        t32 = s32.step(f, fa)[2][0].double().numpy()
        e64[i] = np.abs(final[i] - t64).max(); e32[i] = np.abs(t32 - t64).max()
        assert int(pred[i]) == int(t64.argmax()) == int(t32.argmax()), i
    print('HIP vs fp64 oracle, max per 32 steps:', [round(float(e64[i:i + 32].max()), 6) for i in range(0, n, 32)])
    print('fp32 oracle vs fp64 oracle      :', [round(float(e32[i:i + 32].max()), 6) for i in range(0, n, 32)])
    assert e64.max() <= 1e-3, e64.max()
    assert e64.max() <= e32.max()


def test_end_to_end_vs_oracle_with_encoder():
    """test_zeroshot_3d_core (reference signature) with the HIP encoder + cache on a synthetic
    stream, against the oracle running the same clouds, noise and weights on the CPU."""
    from uni_adapter_amd.Uni_Adapter import test_zeroshot_3d_core
    from uni_adapter_amd.point_encoder import create_uni3d_from_state_dict
    K, D, G, n = 12, 256, 128, 7
    sd = O.make_uni3d_state_dict(21, trans_dim=128, embed_dim=D, enc_dim=512, depth=2)
    g = torch.Generator().manual_seed(1)
    text = torch.nn.functional.normalize(torch.randn(K, D, generator=g), dim=1)
    xyz = torch.randn(n, 1024, 3, generator=g); xyz = xyz / xyz.norm(dim=-1, keepdim=True).amax(1, keepdim=True)
    rgb = torch.ones(n, 1024, 3)
    noise = torch.randn(n, 1024, 3, generator=g)
    labels = torch.randint(0, K, (n,), generator=g)
    model = create_uni3d_from_state_dict(sd, num_group=G, group_size=64, num_heads=2)
    args = SimpleNamespace(use_mode_dota=True, res_learning=True, dota_epsilon=1e-4, dota_sigma=1e-4, dota_eta=0.1, dota_rho=0.02,
                           mode_M=4, device='cuda:0', precomputed_text_features=text, chunk_size=3, print_freq=0,
                           aug_noise=lambda i, pc: noise[i:i + 1], vlm3d='uni3d')
    loader = [(xyz[i:i + 1], labels[i:i + 1], ['x'], rgb[i:i + 1]) for i in range(n)]
    acc = test_zeroshot_3d_core(loader, 'synthetic', model, None, None, args, {})
    # oracle
    s = O.TtaStream(text, O.TtaConfig(num_modes=4))
    hits = np.zeros(3)
    for i in range(n):
        pc6 = torch.cat([xyz[i:i + 1], rgb[i:i + 1]], -1)
        pc6a = torch.cat([xyz[i:i + 1] + 0.05 * noise[i:i + 1], rgb[i:i + 1]], -1)
        f = O.uni3d_encode_pc(pc6, sd, G, 64, 2)
        fa = O.uni3d_encode_pc(pc6a, sd, G, 64, 2)
        z, dz, final, pred = s.step(f, fa)
        a = O.accuracy_topk(final, labels[i:i + 1])
        hits += np.array(a) / 100.0
    for k, v in zip(('acc1', 'acc3', 'acc5'), hits / n * 100.0):
        assert abs(acc[k] - v) < 1e-9, (k, acc[k], v)


def test_chunking_does_not_change_results(golden_dir):
    from uni_adapter_amd.Uni_Adapter import TtaEngine
    g = _load(golden_dir, 'g2_stream_scanobjectnn.npz')
    text = torch.from_numpy(g['text']); n = 12
    feats = torch.stack([torch.from_numpy(g['feat'][:n]), torch.from_numpy(g['feat_aug'][:n])], 1).cuda()
    outs = []
    for chunk in (1, 4, 12):
        eng = TtaEngine(FeatureFeeder(feats), text, {'epsilon': 1e-4, 'sigma': 1e-4, 'eta': 0.1, 'rho': 0.02}, 8, chunk=chunk)
        fs = [eng.process_chunk(torch.zeros(min(chunk, n - s), 8, 3), torch.ones(min(chunk, n - s), 8, 3))[0].cpu() for s in range(0, n, chunk)]
        outs.append(torch.cat(fs))
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


def test_compute_text_alignment_loss_api(golden_dir):
    from uni_adapter_amd.Uni_Adapter import compute_text_alignment_loss
    from uni_adapter_amd.dota_mixture import DOTA_mix
    g = _load(golden_dir, 'g3_alignment_scanobjectnn.npz')
    K, M = g['pi'].shape; D = g['mu'].shape[2]
    text = torch.from_numpy(g['text'])
    mdl = DOTA_mix({'epsilon': float(g['eps']), 'sigma': 1e-4}, D, K, text.T, num_modes=M)
    mdl.mu.copy_(torch.from_numpy(g['mu'])); mdl.var.copy_(torch.from_numpy(g['var'])); mdl.pi.copy_(torch.from_numpy(g['pi']))
    res = torch.from_numpy(g['res']).cuda().requires_grad_(True)
    t = text.cuda() + res
    e = t / t.norm(dim=1, keepdim=True)
    loss, L = compute_text_alignment_loss(e, mdl)
    loss.backward()
    assert abs(loss.item() - float(g['loss'])) < 2e-6
    np.testing.assert_allclose(L.detach().cpu().numpy(), g['L'], rtol=0, atol=3e-3)
    ref = torch.from_numpy(g['grad_res'])
    assert (res.grad.cpu() - ref).abs().max().item() <= 2e-3 * ref.abs().max().item()
    # property stated in the reference docstring (Uni_Adapter.py:205-206)
    row = mdl.predict(e[3:4].detach())
    np.testing.assert_allclose(row.cpu().numpy()[0], L[3].detach().cpu().numpy(), rtol=0, atol=1e-3)


def test_composite_entry_equals_per_kernel_calls(golden_dir):
    """ua_tta_chunk (one host call per chunk) must be bit-identical to driving the individual
    entry points from Python."""
    from uni_adapter_amd.Uni_Adapter import TtaEngine
    g = _load(golden_dir, 'g2_stream_scanobjectnn.npz')
    text = torch.from_numpy(g['text']); n = 9
    feats = torch.stack([torch.from_numpy(g['feat'][:n]), torch.from_numpy(g['feat_aug'][:n])], 1).cuda()
    tgt = torch.from_numpy(g['labels'][:n]).int()
    res = []
    for composite in (True, False):
        eng = TtaEngine(FeatureFeeder(feats), text, {'epsilon': 1e-4, 'sigma': 1e-4, 'eta': 0.1, 'rho': 0.02}, 8, chunk=4, composite=composite)
        outs = [eng.process_chunk(torch.zeros(min(4, n - s), 8, 3), torch.ones(min(4, n - s), 8, 3), tgt[s:s + 4]) for s in range(0, n, 4)]
        res.append((torch.cat([o[0] for o in outs]).cpu(), torch.cat([o[1] for o in outs]).cpu(), torch.cat([o[2] for o in outs]).cpu(),
                    eng.residuals.res.cpu().clone(), eng.residuals.step, eng.mode_dota_model.mu.cpu().clone(), eng.accuracy()))
    for a, b in zip(res[0], res[1]):
        assert (torch.equal(a, b) if isinstance(a, torch.Tensor) else a == b)


def test_replay_overlap_gives_identical_results(golden_dir):
    """overlap=True (cache replay on a second stream while the next chunk is encoded) must not
    change a single bit of the outputs."""
    from uni_adapter_amd.Uni_Adapter import TtaEngine
    from uni_adapter_amd.point_encoder import create_uni3d_from_state_dict
    from uni_adapter_amd import synthetic
    sd = synthetic.uni3d_state_dict(3, trans_dim=128, embed_dim=256, depth=2)
    text = synthetic.text_bank(12, 256, 1)
    xyz, rgb = synthetic.point_clouds(12, 1024, 5)
    noise = torch.randn(12, 1024, 3, generator=torch.Generator().manual_seed(2))
    tgt = torch.arange(12, dtype=torch.int32) % 12
    outs = []
    for overlap in (False, True):
        model = create_uni3d_from_state_dict(sd, num_group=128, group_size=64, num_heads=2, precision='split')
        eng = TtaEngine(model, text, {'epsilon': 1e-4, 'sigma': 1e-4, 'eta': 0.1, 'rho': 0.02}, 4, chunk=4, overlap=overlap)
        res = [eng.process_chunk(xyz[s:s + 4], rgb[s:s + 4], tgt[s:s + 4], noise[s:s + 4]) for s in range(0, 12, 4)]
        eng.synchronize()
        outs.append((torch.cat([r[0] for r in res]).cpu(), torch.cat([r[1] for r in res]).cpu(), eng.accuracy(), eng.residuals.res.cpu().clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert outs[0][2] == outs[1][2] and torch.equal(outs[0][3], outs[1][3])


def _owner_mode_worker(rank, world, port, out):
    """One rank of a 2-process job sharing cuda:0 (gloo): TtaEngine in 'owner' replay mode."""
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), UA_REPLAY_MODE='owner')
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.cuda.set_device(0)
    import uni_adapter_amd  # noqa: F401
    from uni_adapter_amd import synthetic
    from uni_adapter_amd.point_encoder import create_uni3d_from_state_dict
    from uni_adapter_amd.Uni_Adapter import TtaEngine
    finals, acc = _run_small_stream(TtaEngine, create_uni3d_from_state_dict, synthetic)
    if rank == 0:
        torch.save({'finals': finals.cpu(), 'acc': acc}, out)
    else:
        torch.save({'acc': acc}, out + '.r1')
    dist.destroy_process_group()


def _run_small_stream(TtaEngine, create, synthetic):
    K, D, n, chunk = 10, 256, 10, 5
    sd = synthetic.uni3d_state_dict(1, trans_dim=128, embed_dim=D, depth=1, device='cuda')
    model = create(sd, num_group=128, group_size=64, num_heads=2, device='cuda')
    text = synthetic.text_bank(K, D, 2)
    xyz, rgb = synthetic.point_clouds(n, 1024, 3, device='cuda')
    noise = torch.randn(n, 1024, 3, device='cuda', generator=torch.Generator(device='cuda').manual_seed(4))
    tgt = torch.arange(n, device='cuda') % K
    eng = TtaEngine(model, text, {'epsilon': 1e-4, 'sigma': 1e-4, 'eta': 0.1, 'rho': 0.02}, 2, res_learning=True, chunk=chunk,
                    device=torch.device('cuda'), replay_cost_ratio=0.3)
    outs = []
    for c0 in range(0, n, chunk):
        f, _, _ = eng.process_chunk(xyz[c0:c0 + chunk], rgb[c0:c0 + chunk], tgt[c0:c0 + chunk], noise[c0:c0 + chunk])
        outs.append(f)
    return torch.cat(outs), eng.accuracy()


def test_owner_replay_mode_two_ranks_on_one_gpu(tmp_path):
    """W = 2 with the cache replay on rank 0 only (uneven encoder shares, one all_gather per chunk, accuracy
    broadcast) reproduces the single-process stream bit for bit; both ranks report the same accuracy."""
    import socket
    import torch.multiprocessing as mp
    from uni_adapter_amd import synthetic
    from uni_adapter_amd.point_encoder import create_uni3d_from_state_dict
    from uni_adapter_amd.Uni_Adapter import TtaEngine
    ref, ref_acc = _run_small_stream(TtaEngine, create_uni3d_from_state_dict, synthetic)
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / 'owner.pt')
    mp.spawn(_owner_mode_worker, args=(2, port, out), nprocs=2, join=True)
    got = torch.load(out, weights_only=True)
    assert torch.equal(got['finals'], ref.cpu())
    assert got['acc'] == ref_acc and torch.load(out + '.r1', weights_only=True)['acc'] == ref_acc


def test_full_length_stream_conserves_the_count_mass_and_stays_finite():
    """A whole ModelNet40-C-sized stream (2 468 samples, `main_test-time.py:68`; K = 40, M = 8, D = 1024, residual learning on) through
    the composite replay: size-independent properties of the cache step - every sample adds sum_k gamma = 1 twice, so
    sum(c) = K + 2 n and sum(class_counts) = 2 n (dota_mixture.py:196,233); pi stays a distribution per class; variances stay
    >= 1e-8; residuals, Adam moments and logits stay finite; the Adam step counter is 10 (n - 1)."""
    from uni_adapter_amd.Uni_Adapter import TtaEngine
    g = torch.Generator().manual_seed(5)
    K, D, M, n, chunk = 40, 1024, 8, 2468, 617
    text = torch.nn.functional.normalize(torch.randn(K, D, generator=g), dim=1)
    lab = torch.randint(0, K, (n,), generator=g)
    f0 = torch.nn.functional.normalize(text[lab] + 0.05 * torch.randn(n, D, generator=g), dim=1)
    f1 = torch.nn.functional.normalize(f0 + 0.02 * torch.randn(n, D, generator=g), dim=1)
    feats = torch.stack([f0, f1], 1).cuda().contiguous()
    e = TtaEngine(None, text, {'epsilon': 1e-4, 'sigma': 1e-4, 'eta': 0.1, 'rho': 0.02}, M, res_learning=True, chunk=chunk)
    dummy = torch.zeros(chunk, 8, 3)
    hits = 0
    for c0 in range(0, n, chunk):
        e.encode_chunk = lambda *a, _f=feats[c0:c0 + chunk]: _f
        fin, ranks, pred = e.process_chunk(dummy, dummy, lab[c0:c0 + chunk].cuda().int(), dummy)
        assert torch.isfinite(fin).all() and int(pred.min()) >= 0 and int(pred.max()) < K
        hits += int((pred.cpu().long() == lab[c0:c0 + chunk]).sum())
    e.synchronize()
    m, r = e.mode_dota_model, e.residuals
    assert abs(float(m.c.double().sum()) - (K + 2 * n)) < 1e-3 * (K + 2 * n)
    assert abs(float(m.class_counts.double().sum()) - 2 * n) < 1e-3 * 2 * n
    assert (m.pi.double().sum(1) - 1).abs().max().item() < 1e-5 and float(m.pi.min()) >= 0.0
    assert float(m.var.min()) >= 1e-8 and torch.isfinite(m.mu).all() and torch.isfinite(m.var).all()
    for t in (r.res, r.m, r.v, r.e, r.norm):
        assert torch.isfinite(t).all()
    assert r.step == 10 * (n - 1) and e.i == n and m.t == 2 * n
    assert e.accuracy()['acc1'] == pytest.approx(100.0 * hits / n, abs=1e-3) and hits > 0.9 * n      # an easy synthetic stream


def test_lean_replay_is_bit_identical_to_the_one_launch_residual_step():
    """ua_tta_state.replay_lean: the residual step as ua_align_loss + ua_align_backward + ua_residual_adam (less CU-time, for a
    replay that shares the GPU with the encoder) against the one-launch ua_align_step (less latency): same final logits,
    predictions, GMM state, residuals and Adam moments, bit for bit, over a stream with residual learning."""
    from uni_adapter_amd.Uni_Adapter import TtaEngine
    g = torch.Generator().manual_seed(21)
    K, D, M, n = 40, 1024, 8, 12
    text = torch.nn.functional.normalize(torch.randn(K, D, generator=g), dim=1)
    lab = torch.randint(0, K, (n,), generator=g)
    f0 = torch.nn.functional.normalize(text[lab] + 0.05 * torch.randn(n, D, generator=g), dim=1)
    f1 = torch.nn.functional.normalize(f0 + 0.02 * torch.randn(n, D, generator=g), dim=1)
    feats = torch.stack([f0, f1], 1).cuda().contiguous()
    cfg = {'epsilon': 1e-4, 'sigma': 1e-4, 'eta': 0.1, 'rho': 0.02}
    outs = []
    for lean in (False, True):
        e = TtaEngine(None, text, cfg, M, res_learning=True, chunk=n, replay_lean=lean)
        assert e.replay_lean == lean and e._state.replay_lean == int(lean)
        e.encode_chunk = lambda *a, _f=feats: _f
        dummy = torch.zeros(n, 8, 3)
        fin, _, pred = e.process_chunk(dummy, dummy, lab.cuda().int(), dummy)
        e.synchronize()
        m, r = e.mode_dota_model, e.residuals
        outs.append([t.clone() for t in (fin, pred, m.mu, m.var, m.pi, m.c, r.res, r.m, r.v, r.e, r.norm)])
    for a, b in zip(*outs):
        assert torch.equal(a, b)


def test_owner_mode_calibration_leaves_the_engine_untouched_and_rho_does_not_change_results():
    """`TtaEngine.calibrate_replay_cost` (owner mode, W > 1) replays a few samples on the LIVE adaptation state and restores it:
    every tensor and counter the stream carries must be bit-identical afterwards, and the stream must continue exactly as if
    the calibration had never run.  The ratio it measures only moves the cut between rank 0's and the other ranks' encoder
    blocks: features assembled from blocks cut with two different ratios (encoded as separate batches, as two ranks would)
    give bit-identical final logits - the encoder is invariant to batch composition."""
    from uni_adapter_amd import synthetic
    from uni_adapter_amd.point_encoder import create_uni3d_from_state_dict
    from uni_adapter_amd.Uni_Adapter import TtaEngine, NOISE_STD
    from uni_adapter_amd.stream import shard_counts
    K, D, n, chunk = 10, 256, 12, 6
    sd = synthetic.uni3d_state_dict(1, trans_dim=128, embed_dim=D, depth=1, device='cuda')
    model = create_uni3d_from_state_dict(sd, num_group=128, group_size=64, num_heads=2, device='cuda')
    text = synthetic.text_bank(K, D, 2)
    xyz, rgb = synthetic.point_clouds(n, 1024, 3, device='cuda')
    noise = torch.randn(n, 1024, 3, device='cuda', generator=torch.Generator(device='cuda').manual_seed(4))
    tgt = torch.arange(n, device='cuda') % K
    cfg = {'epsilon': 1e-4, 'sigma': 1e-4, 'eta': 0.1, 'rho': 0.02}

    def snapshot(e):
        m, r = e.mode_dota_model, e.residuals
        ts = dict(mu=m.mu, var=m.var, pi=m.pi, c=m.c, cc=m.class_counts, res=r.res, am=r.m, av=r.v, e=r.e, norm=r.norm,
                  tickets=e.tickets, correct=e.correct)
        if e.align_mfma:                  # text0 + res, kept current only by the matrix-core residual kernels
            ts['t'] = r.t
        return {k: v.clone() for k, v in ts.items()}, (int(e.i), int(m.t), int(r.step), int(e.count))

    ref = TtaEngine(model, text, cfg, 2, res_learning=True, chunk=chunk)
    eng = TtaEngine(model, text, cfg, 2, res_learning=True, chunk=chunk)
    want = [ref.process_chunk(xyz[c:c + chunk], rgb[c:c + chunk], tgt[c:c + chunk], noise[c:c + chunk])[0] for c in range(0, n, chunk)]
    got = [eng.process_chunk(xyz[:chunk], rgb[:chunk], tgt[:chunk], noise[:chunk])[0]]
    before, counters = snapshot(eng)
    eng.world, eng.replay_mode, eng.emulated = 2, 'owner', True          # act as rank 0 of a 2-rank owner-mode job for the calibration only
    rho = eng.calibrate_replay_cost(xyz[:4], rgb[:4], noise[:4])
    eng.world, eng.replay_mode = 1, 'owner'
    assert 0.0 < rho <= 2.0 and rho == eng.replay_cost_ratio
    after, counters2 = snapshot(eng)
    assert counters == counters2
    for k in before:
        assert torch.equal(before[k], after[k]), k
    got.append(eng.process_chunk(xyz[chunk:], rgb[chunk:], tgt[chunk:], noise[chunk:])[0])
    assert torch.equal(torch.cat(got), torch.cat(want))

    outs = {}
    for r_ in (0.0, 0.45):
        e2 = TtaEngine(model, text, cfg, 2, res_learning=True, chunk=chunk)
        fin = []
        for c in range(0, n, chunk):
            counts = shard_counts(chunk, 2, r_)
            blocks, s0 = [], c
            for cnt in counts:                                       # one encoder batch per (emulated) rank
                if cnt:
                    p_, c_, z_ = xyz[s0:s0 + cnt], rgb[s0:s0 + cnt], noise[s0:s0 + cnt]
                    both = torch.stack([torch.cat((p_, c_), -1), torch.cat((p_ + NOISE_STD * z_, c_), -1)], 1)
                    blocks.append(model.encode_pc(both.view(-1, 1024, 6)).view(cnt, 2, D))
                s0 += cnt
            feats = torch.cat(blocks, 0)
            e2.encode_chunk = lambda *a, _f=feats: _f
            fin.append(e2.process_chunk(xyz[c:c + chunk], rgb[c:c + chunk], tgt[c:c + chunk], noise[c:c + chunk])[0])
        outs[r_] = torch.cat(fin)
    assert shard_counts(chunk, 2, 0.0) != shard_counts(chunk, 2, 0.45)
    assert torch.equal(outs[0.0], outs[0.45]) and torch.equal(outs[0.0], torch.cat(want))


@pytest.mark.parametrize('family', ['ulip', 'openshape'])
def test_harness_drives_the_other_encoder_families(family):
    """test_zeroshot_3d_core with a ULIP-2 / OpenShape model: TtaEngine only needs `encode_pc(B,N,6)`."""
    import types
    from uni_adapter_amd import synthetic
    from uni_adapter_amd.point_encoder import create_ulip_from_state_dict, create_openshape_from_state_dict
    from uni_adapter_amd.Uni_Adapter import test_zeroshot_3d_core as core
    if family == 'ulip':
        model = create_ulip_from_state_dict(synthetic.ulip_state_dict(2, depth=1, embed_dim=256, device='cuda'), num_group=128, device='cuda')
        D = 256
    else:
        model = create_openshape_from_state_dict(synthetic.openshape_state_dict(2, depth=1, out_dim=256, device='cuda'), patches=128, device='cuda')
        D = 256
    K, n = 7, 6
    xyz, rgb = synthetic.point_clouds(n, 1024, seed=5)
    loader = [(xyz[i:i + 2], torch.arange(i, i + 2) % K, ['x'] * 2, rgb[i:i + 2]) for i in range(0, n, 2)]
    args = types.SimpleNamespace(use_mode_dota=True, dota_epsilon=1e-4, dota_sigma=1e-4, dota_eta=0.1, dota_rho=0.02, mode_M=2, device='cuda',
                                 precomputed_text_features=synthetic.text_bank(K, D, seed=1), res_learning=True, chunk_size=4, vlm3d=family,
                                 aug_noise=lambda i, pc: torch.zeros_like(pc))
    acc = core(loader, 'x', model, None, None, args, None)
    assert set(acc) == {'acc1', 'acc3', 'acc5'} and 0.0 <= acc['acc1'] <= acc['acc5'] <= 100.0


def test_use_dota_branch_runs_the_full_covariance_cache():
    """--use-dota without --use-mode-dota (BASELINE configs[0]: ULIP-2 + DOTA): the harness against the oracle loop
    (clip logits -> DotaFull.predict/fit/update -> clip + w * dota)."""
    import types
    from uni_adapter_amd import synthetic
    from uni_adapter_amd.point_encoder import create_ulip_from_state_dict
    from uni_adapter_amd.Uni_Adapter import test_zeroshot_3d_core as core
    D, K, n = 256, 6, 5
    sd = synthetic.ulip_state_dict(4, depth=1, embed_dim=D, device='cuda')
    model = create_ulip_from_state_dict(sd, num_group=128, device='cuda')
    xyz, rgb = synthetic.point_clouds(n, 1024, seed=8)
    text = synthetic.text_bank(K, D, seed=3)
    tgt = torch.arange(n) % K
    loader = [(xyz[i:i + 1], tgt[i:i + 1], ['x'], rgb[i:i + 1]) for i in range(n)]
    args = types.SimpleNamespace(use_mode_dota=False, use_dota=True, dota_epsilon=1e-4, dota_sigma=1e-4, dota_eta=0.1, dota_rho=0.02,
                                 device='cuda', precomputed_text_features=text, vlm3d='ulip')
    acc = core(loader, 'x', model, None, None, args, None)
    # oracle loop on the HIP encoder's features
    feats = model.encode_pc(torch.cat([xyz, rgb], -1).cuda()).cpu()
    m = O.DotaFull({'epsilon': 1e-4, 'sigma': 1e-4}, D, K, torch.full((D, K), 0.001))
    hits = 0
    for i in range(n):
        fh, z, _, prob, _ = O.clip_logits(feats[i:i + 1], text)
        dz = m.predict(fh.half()).float()
        m.fit(fh, prob); m.update()
        w = torch.clamp(0.02 * m.c.mean() / 1, max=0.1)
        final = z + w * dz
        hits += int(final.argmax()) == int(tgt[i])
    assert abs(acc['acc1'] - 100.0 * hits / n) < 1e-6


def test_cluster_cache_branch_of_the_harness():
    """Neither --use-dota nor --use-mode-dota: the original cluster cache through test_zeroshot_3d_core vs the oracle loop."""
    import types
    from uni_adapter_amd import synthetic
    from uni_adapter_amd.point_encoder import create_uni3d_from_state_dict
    from uni_adapter_amd.Uni_Adapter import get_hyperparams, test_zeroshot_3d_core as core
    D, K, n = 256, 5, 12
    model = create_uni3d_from_state_dict(synthetic.uni3d_state_dict(9, trans_dim=128, embed_dim=D, depth=1, device='cuda'), num_group=128,
                                         group_size=64, num_heads=2, device='cuda')
    xyz, rgb = synthetic.point_clouds(n, 1024, seed=11)
    text = synthetic.text_bank(K, D, seed=5)
    tgt = torch.arange(n) % K
    loader = [(xyz[i:i + 3], tgt[i:i + 3], ['x'] * 3, rgb[i:i + 3]) for i in range(0, n, 3)]
    hp = dict(get_hyperparams('modelnet'), shot_capacity=2)
    assert get_hyperparams('ShapeNetCore')['use_new_approximation'] is False and get_hyperparams('scanobjectnn')['lambda_reg'] == 0.20
    args = types.SimpleNamespace(use_mode_dota=False, use_dota=False, device='cuda', precomputed_text_features=text, vlm3d='uni3d',
                                 dataset_name='modelnet')
    acc = core(loader, 'x', model, None, None, args, hp)
    feats = model.encode_pc(torch.cat([xyz, rgb], -1).cuda()).cpu()
    oc = O.ClusterCache(text, 2, 150.0, 0.5, 0.11)
    hits = sum(int(oc.step(feats[i:i + 1])[0].argmax()) == int(tgt[i]) for i in range(n))
    assert abs(acc['acc1'] - 100.0 * hits / n) < 1e-6


def test_two_stream_encode_equals_single_batch():
    """The original and augmented clouds encoded as two batches on two HIP streams give bit-identical features."""
    from uni_adapter_amd import synthetic
    from uni_adapter_amd.point_encoder import create_uni3d_from_state_dict
    from uni_adapter_amd.Uni_Adapter import TtaEngine
    sd = synthetic.uni3d_state_dict(1, trans_dim=128, embed_dim=256, depth=2, device='cuda')
    model = create_uni3d_from_state_dict(sd, num_group=128, group_size=64, num_heads=2, device='cuda')
    text = synthetic.text_bank(10, 256, 2)
    xyz, rgb = synthetic.point_clouds(6, 1024, 3, device='cuda')
    noise = torch.randn(6, 1024, 3, device='cuda', generator=torch.Generator(device='cuda').manual_seed(4))
    cfg = {'epsilon': 1e-4, 'sigma': 1e-4, 'eta': 0.1, 'rho': 0.02}
    a = TtaEngine(model, text, cfg, 2, chunk=6, device=torch.device('cuda'), two_stream_encode=True)
    b = TtaEngine(model, text, cfg, 2, chunk=6, device=torch.device('cuda'), two_stream_encode=False)
    fa = a.encode_chunk(xyz, rgb, noise); fb = b.encode_chunk(xyz, rgb, noise)
    torch.cuda.synchronize()
    assert a.two_stream_encode and not b.two_stream_encode and torch.equal(fa, fb)


def _delta_lockstep(text, feats, labels, world, step, M):
    """W engines driven in lock-step in ONE process: local replay + pack on each, the stacked buffers stand in for the
    all_gather, merge on each.  Returns (finals, preds, engines)."""
    from uni_adapter_amd.Uni_Adapter import TtaEngine
    from uni_adapter_amd.stream import shard_counts
    cfg = {'epsilon': 1e-4, 'sigma': 1e-4, 'eta': 0.1, 'rho': 0.02}
    engs = [TtaEngine(None, text, cfg, M, res_learning=True, chunk=step, replay_mode='delta', emulated_rank=(r, world)) for r in range(world)]
    finals, preds = [], []
    n_all = feats.shape[0]
    for s0 in range(0, n_all, step):
        n = min(step, n_all - s0)
        counts = shard_counts(n, world, 0.0)
        tgt = labels[s0:s0 + n]
        bufs = []
        for r, e in enumerate(engs):
            a = s0 + sum(counts[:r])
            bufs.append(e.delta_local(feats[a:a + counts[r]], tgt, n).clone())
        gathered = torch.stack(bufs)
        outs = [e.delta_merge(gathered, n) for e in engs]
        for o in outs[1:]:
            assert torch.equal(o[0], outs[0][0]) and torch.equal(o[2], outs[0][2])
        finals.append(outs[0][0].cpu()); preds.append(outs[0][2].cpu())
    return torch.cat(finals), torch.cat(preds), engs


@pytest.mark.parametrize('world,step', [(2, 4), (3, 7), (4, 3)])
def test_delta_replay_mode_vs_oracle_emulation(golden_dir, world, step):
    """'delta' multi-GPU mode (SURVEY section 8e "prototype deltas"; no reference counterpart, NOT exact): W ranks replay
    their own blocks from the common state, the packed sufficient-statistic deltas are merged in rank order.  Checked
    against the oracle's emulation of the same rule (oracle.TtaDeltaStream): predictions identical; logits within 1e-3
    of the fp32 emulation plus that emulation's own distance to its fp64 run while the stream is calm (same three-way
    argument as test_loop_with_residual_learning_vs_reference_golden); all replicas bit-identical after every merge;
    total count mass conserved (each fit adds exactly sum(prob) = 1)."""
    g = _load(golden_dir, 'g2_stream_scanobjectnn.npz')
    text = torch.from_numpy(g['text']); M = int(g['M'])
    n = 2 * step + 1                                            # two full steps and a ragged one
    f_cpu = torch.stack([torch.from_numpy(g['feat'][:n]), torch.from_numpy(g['feat_aug'][:n])], 1)
    labels = torch.from_numpy(g['labels'][:n]).int()
    final, pred, engs = _delta_lockstep(text, f_cpu.cuda(), labels, world, step, M)
    o32 = O.TtaDeltaStream(text, O.TtaConfig(num_modes=M), world)
    o64 = O.TtaDeltaStream(text, O.TtaConfig(num_modes=M), world, dtype=torch.float64)
    r32, r64, p32 = [], [], []
    for s0 in range(0, n, step):
        a, p = o32.step_chunk(f_cpu[s0:s0 + step]); r32.append(a); p32 += p
        r64.append(o64.step_chunk(f_cpu[s0:s0 + step].double())[0])
    r32, r64 = torch.cat(r32).double().numpy(), torch.cat(r64).numpy()
    final = final.double().numpy()
    assert np.array_equal(pred.numpy(), np.array(p32)) and np.array_equal(pred.numpy(), r64.argmax(1))
    e_hip, e_ref, e_truth = np.abs(final - r32).max(1), np.abs(r32 - r64).max(1), np.abs(final - r64).max(1)
    print(world, 'HIP vs fp32 emulation :', np.round(e_hip, 5).tolist())
    print(world, 'fp32 vs fp64 emulation:', np.round(e_ref, 5).tolist())
    events = np.maximum(e_ref, e_truth) >= 0.1
    first_event = int(np.argmax(events)) if events.any() else n
    assert first_event >= min(5, n), first_event
    pre = slice(0, first_event)
    noise = np.maximum(np.maximum.accumulate(e_ref), 1e-3)
    assert (e_hip[:3] <= 1e-3).all()
    assert (e_hip[pre] <= 1e-3 + 2 * noise[pre]).all() and (e_truth[pre] <= 3 * noise[pre]).all()
    # replicas identical; accuracy identical; mass conserved
    e0 = engs[0]
    for e in engs[1:]:
        for a, b in ((e.mode_dota_model.mu, e0.mode_dota_model.mu), (e.mode_dota_model.var, e0.mode_dota_model.var),
                     (e.mode_dota_model.pi, e0.mode_dota_model.pi), (e.residuals.res, e0.residuals.res), (e.residuals.e, e0.residuals.e)):
            assert torch.equal(a, b)
        assert e.accuracy() == e0.accuracy() and e.residuals.step == e0.residuals.step == o32.s.adam.step
    K = text.shape[0]
    assert abs(e0.mode_dota_model.c.sum().item() - (K + 2.0 * n)) < 1e-2
    assert abs(e0.accuracy()['acc1'] - 100.0 * float((pred.numpy() == g['labels'][:n]).mean())) < 1e-9
    # the merged state itself against the fp32 emulation (c exactly tracks; mu to fp32 rounding of the same formula)
    assert (e0.mode_dota_model.c.cpu() - o32.s.st.c).abs().max().item() < 1e-4
    if first_event == n:
        assert (e0.mode_dota_model.mu.cpu() - o32.s.st.mu).abs().max().item() < 1e-4


def _delta_mode_worker(rank, world, port, out):
    """One rank of a 2-process job sharing cuda:0 (gloo): TtaEngine.process_chunk in 'delta' replay mode."""
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), UA_REPLAY_MODE='delta')
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.cuda.set_device(0)
    import uni_adapter_amd  # noqa: F401
    from uni_adapter_amd import synthetic
    from uni_adapter_amd.point_encoder import create_uni3d_from_state_dict
    from uni_adapter_amd.Uni_Adapter import TtaEngine
    finals, acc = _run_small_stream(TtaEngine, create_uni3d_from_state_dict, synthetic)
    torch.save({'finals': finals.cpu(), 'acc': acc}, out + f'.r{rank}')
    dist.destroy_process_group()


def test_delta_replay_mode_two_processes_match_lockstep(tmp_path):
    """The real exchange path (process_chunk -> all_gather of the packed deltas, 2 ranks sharing cuda:0 over gloo) gives
    every rank the results of the lock-step emulation on the same encoder features, bit for bit."""
    import socket
    import torch.multiprocessing as mp
    from uni_adapter_amd import synthetic
    from uni_adapter_amd.point_encoder import create_uni3d_from_state_dict
    K, D, n, chunk = 10, 256, 10, 5
    sd = synthetic.uni3d_state_dict(1, trans_dim=128, embed_dim=D, depth=1, device='cuda')
    model = create_uni3d_from_state_dict(sd, num_group=128, group_size=64, num_heads=2, device='cuda')
    text = synthetic.text_bank(K, D, 2)
    xyz, rgb = synthetic.point_clouds(n, 1024, 3, device='cuda')
    noise = torch.randn(n, 1024, 3, device='cuda', generator=torch.Generator(device='cuda').manual_seed(4))
    feats = torch.stack([model.encode_pc(torch.cat((xyz, rgb), -1)), model.encode_pc(torch.cat((xyz + 0.05 * noise, rgb), -1))], 1)
    labels = (torch.arange(n) % K).int()
    want, _, engs = _delta_lockstep(text, feats, labels, 2, chunk, 2)
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / 'delta.pt')
    mp.spawn(_delta_mode_worker, args=(2, port, out), nprocs=2, join=True)
    for r in range(2):
        got = torch.load(out + f'.r{r}', weights_only=True)
        assert torch.equal(got['finals'], want)
        assert got['acc'] == engs[0].accuracy()


@pytest.mark.parametrize('name,world', [('scanobjectnn', 4), ('scanobjectnn', 16), ('modelnet40', 8), ('shapenet55', 3)])
def test_class_sharded_replay_is_bit_identical_to_one_gpu(golden_dir, name, world):
    """'class' replay mode (exact mode for large K): rank r holds classes [r*per,(r+1)*per) of the state,
    per CHUNK one (n, per+2) buffer per rank is exchanged.  W engines driven in lock-step in one process (the stacked
    buffers stand in for the all_gather) against ONE engine holding everything: final logits, predictions, accuracy and
    every rank's slice of the state are bit-identical (K not a multiple of W, ranks with no class at all included)."""
    from uni_adapter_amd.Uni_Adapter import TtaEngine
    g = _load(golden_dir, f'g2_stream_{name}.npz')
    text = torch.from_numpy(g['text']); M = int(g['M']); K = text.shape[0]
    n = min(12, g['feat'].shape[0])
    feats = torch.stack([torch.from_numpy(g['feat'][:n]), torch.from_numpy(g['feat_aug'][:n])], 1).cuda()
    tgt = torch.from_numpy(g['labels'][:n]).int().cuda()
    cfg = {'epsilon': 1e-4, 'sigma': 1e-4, 'eta': 0.1, 'rho': 0.02}
    one = TtaEngine(FeatureFeeder(feats), text, cfg, M, res_learning=False, chunk=n)
    want = one.process_chunk(torch.zeros(n, 8, 3), torch.ones(n, 8, 3), tgt.cpu())
    engs = [TtaEngine(None, text, cfg, M, res_learning=False, chunk=n, replay_mode='class', emulated_rank=(r, world)) for r in range(world)]
    final = torch.empty(world, n, K, device='cuda'); ranks = torch.empty(world, n, dtype=torch.int32, device='cuda')
    preds = torch.empty(world, n, dtype=torch.int32, device='cuda')
    for s0 in range(0, n, 5):                                                    # chunks of 5, 5, 2: one exchange per chunk
        s1 = min(n, s0 + 5)
        gathered = torch.stack([e.class_local_chunk(feats[s0:s1]).clone() for e in engs])
        for r, e in enumerate(engs):
            e.class_fuse_chunk(gathered, tgt[s0:s1], final[r, s0:s1], ranks[r, s0:s1], preds[r, s0:s1])
    torch.cuda.synchronize()
    for r, e in enumerate(engs):
        assert torch.equal(final[r], want[0]) and torch.equal(ranks[r], want[1]) and torch.equal(preds[r], want[2]), r
        k0, k1 = e.class_k0, e.class_k1
        for a, b in ((e.mode_dota_model.mu, one.mode_dota_model.mu), (e.mode_dota_model.var, one.mode_dota_model.var),
                     (e.mode_dota_model.pi, one.mode_dota_model.pi), (e.mode_dota_model.c, one.mode_dota_model.c),
                     (e.mode_dota_model.class_counts, one.mode_dota_model.class_counts)):
            assert torch.equal(a[k0:k1], b[k0:k1]), r
    assert sum(e.class_k1 - e.class_k0 for e in engs) == K
    with pytest.raises(ValueError, match='residual learning'):
        TtaEngine(None, text, cfg, M, res_learning=True, replay_mode='class', emulated_rank=(0, 2))


@pytest.mark.parametrize('K,M', [(1156, 8), (4096, 16)])
def test_class_sharded_replay_at_baseline_bank_sizes(K, M):
    """BASELINE configs[3] (1156 classes, M = 8) and configs[4] (4096 x 16) at D = 1024, W = 8 ranks in lock-step, no residual
    learning (infeasible at such K in the reference too, SURVEY defect 11): the class-sharded replay (a) equals ONE engine
    holding the whole state bit for bit and (b) matches the CPU oracle's loop body on the same stream - predictions
    identical, final logits within 1e-3 (the north-star bar), in fp32 and against the fp64 oracle."""
    from uni_adapter_amd.Uni_Adapter import TtaEngine
    D, W, n, chunk = 1024, 8, 4, 2
    g = torch.Generator().manual_seed(100 + K)
    text = torch.nn.functional.normalize(torch.randn(K, D, generator=g), dim=1)
    lab = torch.randint(0, K, (n,), generator=g)
    scale = 1.0 + 4.0 * torch.rand(n, 1, generator=g)
    f = (text[lab] + 0.05 * torch.randn(n, D, generator=g)) * scale
    fa = (f / scale + 0.02 * torch.randn(n, D, generator=g)) * scale
    feats = torch.stack([f, fa], 1).cuda()
    tgt = lab.int().cuda()
    cfg = {'epsilon': 1e-4, 'sigma': 1e-4, 'eta': 0.1, 'rho': 0.02}
    one = TtaEngine(FeatureFeeder(feats), text, cfg, M, res_learning=False, chunk=n)
    want = one.process_chunk(torch.zeros(n, 8, 3), torch.ones(n, 8, 3), tgt.cpu())
    engs = [TtaEngine(None, text, cfg, M, res_learning=False, chunk=chunk, replay_mode='class', emulated_rank=(r, W)) for r in range(W)]
    final = torch.empty(W, n, K, device='cuda'); ranks = torch.empty(W, n, dtype=torch.int32, device='cuda')
    preds = torch.empty(W, n, dtype=torch.int32, device='cuda')
    for s0 in range(0, n, chunk):
        s1 = s0 + chunk
        gathered = torch.stack([e.class_local_chunk(feats[s0:s1]).clone() for e in engs])          # stands in for the one all_gather
        for r, e in enumerate(engs):
            e.class_fuse_chunk(gathered, tgt[s0:s1], final[r, s0:s1], ranks[r, s0:s1], preds[r, s0:s1])
    torch.cuda.synchronize()
    for r, e in enumerate(engs):
        assert torch.equal(final[r], want[0]) and torch.equal(ranks[r], want[1]) and torch.equal(preds[r], want[2]), r
        k0, k1 = e.class_k0, e.class_k1
        for a, b in ((e.mode_dota_model.mu, one.mode_dota_model.mu), (e.mode_dota_model.var, one.mode_dota_model.var),
                     (e.mode_dota_model.c, one.mode_dota_model.c)):
            assert torch.equal(a[k0:k1], b[k0:k1]), r
    del engs
    s32 = O.TtaStream(text, O.TtaConfig(num_modes=M, res_learning=False))
    s64 = O.TtaStream(text, O.TtaConfig(num_modes=M, res_learning=False), dtype=torch.float64)
    got = want[0].cpu()
    for i in range(n):
        _, _, r32, p32 = s32.step(f[i:i + 1], fa[i:i + 1])
        _, _, r64, p64 = s64.step(f[i:i + 1].double(), fa[i:i + 1].double())
        assert int(want[2][i]) == int(r32.argmax()) == int(r64.argmax()), i
        assert (got[i] - r32[0]).abs().max().item() <= 1e-3, (i, (got[i] - r32[0]).abs().max().item())
        assert (got[i].double() - r64[0]).abs().max().item() <= 1e-3, (i, (got[i].double() - r64[0]).abs().max().item())
    np.testing.assert_allclose(one.mode_dota_model.c.cpu().numpy(), s32.st.c.numpy(), rtol=2e-5, atol=1e-7)
    np.testing.assert_allclose(one.mode_dota_model.mu.cpu().numpy(), s32.st.mu.numpy(), rtol=1e-5, atol=2e-7)


def _class_mode_worker(rank, world, port, out):
    """One rank of a 2-process job sharing cuda:0 (gloo): TtaEngine.process_chunk in 'class' replay mode."""
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), UA_REPLAY_MODE='class')
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.cuda.set_device(0)
    import uni_adapter_amd  # noqa: F401
    finals, acc = _run_small_stream_no_res()
    torch.save({'finals': finals.cpu(), 'acc': acc}, out + f'.r{rank}')
    dist.destroy_process_group()


def _run_small_stream_no_res():
    from uni_adapter_amd import synthetic
    from uni_adapter_amd.point_encoder import create_uni3d_from_state_dict
    from uni_adapter_amd.Uni_Adapter import TtaEngine
    K, D, n, chunk = 11, 256, 10, 5
    sd = synthetic.uni3d_state_dict(1, trans_dim=128, embed_dim=D, depth=1, device='cuda')
    model = create_uni3d_from_state_dict(sd, num_group=128, group_size=64, num_heads=2, device='cuda')
    text = synthetic.text_bank(K, D, 2)
    xyz, rgb = synthetic.point_clouds(n, 1024, 3, device='cuda')
    noise = torch.randn(n, 1024, 3, device='cuda', generator=torch.Generator(device='cuda').manual_seed(4))
    tgt = torch.arange(n, device='cuda') % K
    eng = TtaEngine(model, text, {'epsilon': 1e-4, 'sigma': 1e-4, 'eta': 0.1, 'rho': 0.02}, 4, res_learning=False, chunk=chunk,
                    device=torch.device('cuda'), overlap=True)
    outs = [eng.process_chunk(xyz[c0:c0 + chunk], rgb[c0:c0 + chunk], tgt[c0:c0 + chunk], noise[c0:c0 + chunk])[0] for c0 in range(0, n, chunk)]
    eng.synchronize()
    return torch.cat(outs), eng.accuracy()


def test_class_sharded_replay_two_processes(tmp_path):
    """The real path (sample-sharded encoder, all_gather of features per chunk, all_gather of score slices per chunk,
    replay overlapped on the second stream) with 2 ranks sharing cuda:0 equals the single-process stream bit for bit."""
    import socket
    import torch.multiprocessing as mp
    ref, ref_acc = _run_small_stream_no_res()
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / 'class.pt')
    mp.spawn(_class_mode_worker, args=(2, port, out), nprocs=2, join=True)
    for r in range(2):
        got = torch.load(out + f'.r{r}', weights_only=True)
        assert torch.equal(got['finals'], ref.cpu()) and got['acc'] == ref_acc
