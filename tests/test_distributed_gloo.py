"""world_size-2/3 gloo (CPU) tests of the N>1 path: sample sharding, the one all_gather per chunk
and the in-order cache replay (replicated on every rank, or on the owner rank only with a reduced
encoder share for it) give results identical to the single-process stream."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close(); return p


def _fake_encode(x):
    """Deterministic per-sample 'encoder' (the real one needs a GPU): depends only on the sample."""
    return torch.stack([torch.cos(x.sum(-1, keepdim=True) * torch.arange(1, 65)), torch.sin(x.mean(-1, keepdim=True) * torch.arange(1, 65))], 1).squeeze(-2) \
        if x.dim() == 3 else None


def _worker(rank, world, port, n, chunk, out):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import uni_adapter_amd  # noqa: F401
    from uni_adapter_amd.stream import gather_features, shard_indices
    from oracle import ref_cpu as O
    g = torch.Generator().manual_seed(0)
    K, D = 6, 64
    text = torch.nn.functional.normalize(torch.randn(K, D, generator=g), dim=1)
    data = torch.randn(n, 5, 3, generator=g)
    lab = torch.randint(0, K, (n,), generator=g)
    s = O.TtaStream(text, O.TtaConfig(num_modes=2, res_learning=False))
    finals = []
    for c0 in range(0, n, chunk):
        nn_ = min(chunk, n - c0)
        mine = shard_indices(nn_, rank, world)
        x = data[c0:c0 + nn_][mine]
        f = torch.stack([text[lab[c0:c0 + nn_][mine]] + 0.05 * torch.cos(x.sum((1, 2)))[:, None] * torch.ones(D),
                         text[lab[c0:c0 + nn_][mine]] + 0.05 * torch.sin(x.sum((1, 2)))[:, None] * torch.ones(D)], 1)
        feats = gather_features(f, nn_, rank, world)
        assert feats.shape == (nn_, 2, D)
        for j in range(nn_):                       # replicated in-order replay
            finals.append(s.step(feats[j, 0:1], feats[j, 1:2])[2])
    res = torch.cat(finals)
    if world > 1:
        both = [torch.empty_like(res) for _ in range(world)]
        dist.all_gather(both, res)
        assert all(torch.equal(both[0], b) for b in both)      # identical on every rank
    if rank == 0:
        np.save(out, res.numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize('n,chunk', [(10, 4), (7, 3)])
def test_sharded_stream_equals_single_process(tmp_path, n, chunk):
    outs = []
    for world in (1, 2):
        out = str(tmp_path / f'w{world}.npy')
        mp.spawn(_worker, args=(world, _free_port(), n, chunk, out), nprocs=world, join=True)
        outs.append(np.load(out))
    assert np.array_equal(outs[0], outs[1])


def test_shard_indices_partition():
    import sys
    sys.path.insert(0, ROOT)
    import uni_adapter_amd  # noqa: F401
    from uni_adapter_amd.stream import shard_indices
    for n in (1, 7, 16):
        for w in (1, 2, 3, 8):
            allidx = torch.cat([shard_indices(n, r, w) for r in range(w)]).sort()[0]
            assert torch.equal(allidx, torch.arange(n))


def _worker_owner(rank, world, port, n, chunk, rho, out):
    """'owner' mode: contiguous, uneven shares; rank 0 alone replays; results broadcast from it."""
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import uni_adapter_amd  # noqa: F401
    from uni_adapter_amd.stream import broadcast_from_owner, gather_feature_blocks, shard_block, shard_counts
    from oracle import ref_cpu as O
    g = torch.Generator().manual_seed(0)
    K, D = 6, 64
    text = torch.nn.functional.normalize(torch.randn(K, D, generator=g), dim=1)
    data = torch.randn(n, 5, 3, generator=g)
    lab = torch.randint(0, K, (n,), generator=g)
    s = O.TtaStream(text, O.TtaConfig(num_modes=2, res_learning=False)) if rank == 0 else None
    finals = []
    for c0 in range(0, n, chunk):
        nn_ = min(chunk, n - c0)
        mine = shard_block(nn_, rank, world, rho)
        assert len(mine) == shard_counts(nn_, world, rho)[rank]
        x = data[c0:c0 + nn_][mine]
        f = torch.stack([text[lab[c0:c0 + nn_][mine]] + 0.05 * torch.cos(x.sum((1, 2)))[:, None] * torch.ones(D),
                         text[lab[c0:c0 + nn_][mine]] + 0.05 * torch.sin(x.sum((1, 2)))[:, None] * torch.ones(D)], 1) \
            if len(mine) else torch.zeros(0, 2, D)
        feats = gather_feature_blocks(f, nn_, rank, world, None, rho)
        assert feats.shape == (nn_, 2, D)
        if rank == 0:
            for j in range(nn_):
                finals.append(s.step(feats[j, 0:1], feats[j, 1:2])[2])
    res = torch.cat(finals) if rank == 0 else torch.zeros(n, K)
    if world > 1:
        broadcast_from_owner(res)
    if rank == world - 1:                       # a non-owner rank ends up with the owner's results
        np.save(out, res.numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize('world,rho', [(2, 0.14), (3, 0.14), (2, 2.0)])
def test_owner_replay_equals_single_process(tmp_path, world, rho):
    """rho = 2.0 drives the owner's encoder share to zero (the W = 8 situation): rank 0 only replays."""
    n, chunk = 11, 5
    outs = []
    for w in (1, world):
        out = str(tmp_path / f'o{w}.npy')
        mp.spawn(_worker_owner, args=(w, _free_port(), n, chunk, rho, out), nprocs=w, join=True)
        outs.append(np.load(out))
    assert np.array_equal(outs[0], outs[1])


def test_shard_counts_balance_rule():
    import sys
    sys.path.insert(0, ROOT)
    import uni_adapter_amd  # noqa: F401
    from uni_adapter_amd.stream import shard_block, shard_counts
    assert shard_counts(31, 1) == [31]
    for n in (1, 5, 62, 124, 248):
        for w in (2, 3, 4, 8):
            c = shard_counts(n, w)
            assert sum(c) == n and len(c) == w and min(c) >= 0
            assert max(c[1:]) - min(c[1:]) <= 1                       # the encoders are balanced among themselves
            assert c[0] <= min(c[1:]) + 1                             # the owner never encodes more than an encoder rank
            assert torch.equal(torch.cat([shard_block(n, r, w) for r in range(w)]), torch.arange(n))
    # the rule of stream.py: x0 = (n/W) max(0, 1 - rho (W-1))
    assert shard_counts(62, 2, rho=0.14)[0] == 27 and shard_counts(124, 4, rho=0.14)[0] == 18 and shard_counts(248, 8, rho=0.14)[0] == 1
    assert shard_counts(248, 8)[0] == 0 and shard_counts(248, 8, rho=0.2)[0] == 0


def _worker_delta(rank, world, port, n, step, out):
    """'delta' mode over gloo: every rank replays only its even block (oracle loop body), ONE gather_deltas per step,
    merge in rank order."""
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import uni_adapter_amd  # noqa: F401
    from uni_adapter_amd.stream import gather_deltas, shard_counts
    from oracle import ref_cpu as O
    text, feats = _delta_inputs(n)
    d = O.TtaDeltaStream(text, O.TtaConfig(num_modes=2), world)
    finals = []
    for s0 in range(0, n, step):
        nn_ = min(step, n - s0)
        counts = shard_counts(nn_, world, 0.0)
        start = sum(counts[:rank])
        packed, f, _, steps = d.local(feats[s0 + start:s0 + start + counts[rank]], start)
        res = torch.zeros(max(counts), text.shape[0])
        if f:
            res[:len(f)] = torch.stack(f)
        buf = torch.cat([packed, res.reshape(-1), torch.tensor([float(steps)])])
        g = gather_deltas(buf, world)                                 # (W, len) - the one collective of the step
        assert g.shape == (world, buf.numel())
        d.merge([g[r, :packed.numel()] for r in range(world)], nn_, int(g[:, -1].max()))
        rows = g[:, packed.numel():-1].view(world, max(counts), -1)
        finals.append(torch.cat([rows[r, :counts[r]] for r in range(world)]))
    res = torch.cat(finals)
    torch.save({'final': res, 'mu': d.s.st.mu, 'res': d.s.res, 'step': d.s.adam.step}, out + f'.r{rank}')
    dist.destroy_process_group()


def _delta_inputs(n):
    g = torch.Generator().manual_seed(5)
    K, D = 5, 64
    text = torch.nn.functional.normalize(torch.randn(K, D, generator=g), dim=1)
    lab = torch.randint(0, K, (n,), generator=g)
    feats = torch.nn.functional.normalize(text[lab][:, None, :] + 0.3 * torch.randn(n, 2, D, generator=g), dim=-1)
    return text, feats


@pytest.mark.parametrize('world', [2, 3])
def test_delta_mode_over_gloo_equals_single_process_emulation(tmp_path, world):
    """N>1 'delta' path: sharding (even blocks), the one all_gather of packed deltas and the rank-order merge give every
    rank exactly what the single-process emulation of the same rule gives."""
    from oracle import ref_cpu as O
    n, step = 11, 5
    text, feats = _delta_inputs(n)
    d = O.TtaDeltaStream(text, O.TtaConfig(num_modes=2), world)
    want = torch.cat([d.step_chunk(feats[s0:s0 + step])[0] for s0 in range(0, n, step)])
    out = str(tmp_path / 'delta')
    mp.spawn(_worker_delta, args=(world, _free_port(), n, step, out), nprocs=world, join=True)
    for r in range(world):
        got = torch.load(out + f'.r{r}', weights_only=True)
        assert torch.equal(got['final'], want)
        assert torch.equal(got['mu'], d.s.st.mu) and torch.equal(got['res'], d.s.res) and got['step'] == d.s.adam.step


def test_delta_mode_properties():
    """Size-independent properties of the 'delta' rule (oracle): W = 1 reproduces the sequential stream to fp32 rounding;
    for any W the count mass grows by exactly 2 per sample (each fit adds sum(prob) = 1) and pi stays normalised; with one
    sample per step and W = 1 nothing is stale, so predictions are the sequential ones."""
    from oracle import ref_cpu as O
    n = 9
    text, feats = _delta_inputs(n)
    seq = O.TtaStream(text, O.TtaConfig(num_modes=2))
    want = torch.stack([seq.step(feats[i, 0:1], feats[i, 1:2])[2][0] for i in range(n)])
    one = O.TtaDeltaStream(text, O.TtaConfig(num_modes=2), 1)
    got = torch.cat([one.step_chunk(feats[s0:s0 + 4])[0] for s0 in range(0, n, 4)])
    assert (got - want).abs().max().item() < 1e-3 and torch.equal(got.argmax(1), want.argmax(1))
    assert one.s.adam.step == seq.adam.step
    for world in (2, 4):
        d = O.TtaDeltaStream(text, O.TtaConfig(num_modes=2), world)
        for s0 in range(0, n, 4):
            d.step_chunk(feats[s0:s0 + 4])
        assert abs(d.s.st.c.sum().item() - (text.shape[0] + 2.0 * n)) < 1e-3
        assert (d.s.st.pi.sum(1) - 1).abs().max().item() < 1e-5
        assert abs(d.s.st.class_counts.sum().item() - 2.0 * n) < 1e-3
        assert (d.s.st.var >= 1e-8).all() and d.s.i == n


def _worker_class(rank, world, port, n, out, chunk=4):
    """'class' mode over gloo: each rank keeps classes [r*per,(r+1)*per) of the oracle state and steps it through a whole
    chunk; ONE gather per chunk of (score slice | partial sum of c after the sample) for all its samples, then the fusion."""
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import uni_adapter_amd  # noqa: F401
    from uni_adapter_amd.stream import gather_deltas
    from oracle import ref_cpu as O
    text, feats = _delta_inputs(n)
    K, M = text.shape[0], 2
    per = -(-K // world)
    k0 = min(rank * per, K); k1 = min(k0 + per, K)
    full = O.gmm_init(text, M, 1e-4, 1e-4)
    st = O.GmmState(full.mu[k0:k1].clone(), full.var[k0:k1].clone(), full.pi[k0:k1].clone(), full.c[k0:k1].clone(),
                    full.class_counts[k0:k1].clone(), full.epsilon)
    finals, collectives = [], 0
    for s0 in range(0, n, chunk):
        nb = min(chunk, n - s0)
        buf = torch.zeros(nb, per + 1, dtype=torch.float64)
        zs = []
        for j in range(nb):
            f, z, ent, prob, pred = O.clip_logits(feats[s0 + j, 0:1], text)          # all K, on every rank
            fa = O.clip_logits(feats[s0 + j, 1:2], text)[0]
            fa = fa / fa.norm(dim=-1, keepdim=True)
            zs.append(z)
            if k1 > k0:
                buf[j, :k1 - k0] = O.gmm_predict(st, f.mean(0)[None].half())[0].double()
                O.gmm_fit(st, f, prob[:, k0:k1]); O.gmm_fit(st, fa, prob[:, k0:k1])
                buf[j, per] = st.c.double().sum()
        g = gather_deltas(buf, world)                                              # the ONE collective of the chunk: (W, nb, per+1)
        collectives += 1
        for j in range(nb):
            dz = torch.cat([g[r, j, :per] for r in range(world)])[:K].float()[None]
            cmean = (g[:, j, per].sum() / (K * M)).float()
            finals.append(O.fuse_logits(zs[j], dz, cmean, 0.02, 0.1))
    assert collectives == -(-n // chunk)
    torch.save({'final': torch.cat(finals), 'mu': st.mu, 'k': (k0, k1)}, out + f'.r{rank}')
    dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 3])
def test_class_sharded_replay_over_gloo_equals_single_process(tmp_path, world):
    """N>1 'class' path: the cache step is independent per class and (without residual learning) the fusion feeds nothing
    back into the state, so rank-local slices stepped through a chunk + ONE gather per chunk reproduce the single-process
    stream (K = 5 is not a multiple of W; the last chunk is ragged)."""
    from oracle import ref_cpu as O
    n = 9
    text, feats = _delta_inputs(n)
    s = O.TtaStream(text, O.TtaConfig(num_modes=2, res_learning=False))
    want = torch.cat([s.step(feats[j, 0:1], feats[j, 1:2])[2] for j in range(n)])
    out = str(tmp_path / 'class')
    mp.spawn(_worker_class, args=(world, _free_port(), n, out), nprocs=world, join=True)
    for r in range(world):
        got = torch.load(out + f'.r{r}', weights_only=True)
        np.testing.assert_allclose(got['final'].numpy(), want.numpy(), rtol=0, atol=1e-5)
        assert torch.equal(got['final'].argmax(1), want.argmax(1))
        k0, k1 = got['k']
        assert torch.equal(got['mu'], s.st.mu[k0:k1])


def test_share_rules_partition_every_step():
    """Host-side sharding rules over a sweep of step sizes, world sizes and cost ratios: the per-rank blocks are a partition
    of the step in stream order; rank 0's share shrinks with rho and never exceeds the even share; class slices of the
    class-sharded replay tile [0, K) without overlap for any K and W."""
    import sys
    sys.path.insert(0, ROOT)
    import uni_adapter_amd  # noqa: F401
    from uni_adapter_amd.stream import shard_block, shard_counts, shard_indices
    from oracle import ref_cpu as O
    for world in (1, 2, 3, 4, 7, 8):
        for n in (0, 1, 2, 5, 31, 62, 124, 248, 255):
            for rho in (0.0, 0.1, 0.15, 0.3, 1.0):
                counts = shard_counts(n, world, rho)
                assert len(counts) == world and sum(counts) == n and min(counts) >= 0
                if world > 1:
                    assert counts[0] <= -(-n // world)
                    assert max(counts[1:]) - min(counts[1:]) <= 1
                got = torch.cat([shard_block(n, r, world, rho) for r in range(world)])
                assert torch.equal(got, torch.arange(n))
            assert shard_counts(n, world, 0.0) == O.even_counts(n, world)
            strided = torch.cat([shard_indices(n, r, world) for r in range(world)]).sort().values
            assert torch.equal(strided, torch.arange(n))
    for world in (2, 3, 8, 16):
        for K in (1, 5, 15, 40, 55, 1156, 4096):
            per = -(-K // world)
            edges = [(min(r * per, K), min(min(r * per, K) + per, K)) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == K
            assert all(a[1] == b[0] for a, b in zip(edges, edges[1:]))


def _worker_ratio(rank, world, port, out):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import uni_adapter_amd  # noqa: F401
    from uni_adapter_amd.stream import agree_on_ratio, shard_counts
    rho = agree_on_ratio(0.3 + 0.05 * rank, 2.0 + 0.1 * rank, world)            # every rank measures something slightly different
    torch.save({'rho': rho, 'counts': shard_counts(62, world, rho)}, out + f'.r{rank}')
    dist.destroy_process_group()


def test_measured_cost_ratio_is_rank_zeros_on_every_rank(tmp_path):
    """'owner' mode cuts every chunk with shard_counts(n, W, rho): rho is measured in the warm-up (TtaEngine.calibrate_replay_cost)
    and must be bit-identical on all ranks - rank 0's measurement is broadcast."""
    out = str(tmp_path / 'ratio')
    mp.spawn(_worker_ratio, args=(3, _free_port(), out), nprocs=3, join=True)
    got = [torch.load(out + f'.r{r}', weights_only=True) for r in range(3)]
    assert got[0]['rho'] == got[1]['rho'] == got[2]['rho'] == 0.3 / 2.0
    assert got[0]['counts'] == got[1]['counts'] == got[2]['counts'] and sum(got[0]['counts']) == 62
