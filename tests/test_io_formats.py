"""Section 8 row F4: the file formats either side of the hot path (corruption .npy streams, text bank,
driver flags) - CPU; the end-to-end driver and checkpoint/resume - GPU."""
import os
import types

import numpy as np
import pytest
import torch


def _args(root, **kw):
    a = types.SimpleNamespace(dataset_name='modelnet', myroot=str(root), corruption='gaussian', severity=5, debug=False)
    a.__dict__.update(kw)
    return a


def _write_stream(root, n=7, N=64, name='data_gaussian_5.npy', labels=None, K=40):
    g = np.random.default_rng(0)
    data = g.standard_normal((n, N, 3)).astype(np.float32)
    lab = g.integers(0, K, size=(n,)).astype(np.int64) if labels is None else labels
    np.save(os.path.join(root, name), data)
    np.save(os.path.join(root, 'label.npy'), lab)
    return data, lab


def test_file_naming_follows_the_reference(tmp_path):
    from uni_adapter_amd.tta_data import corruption_files
    r = str(tmp_path)
    assert corruption_files(r, 'gaussian', 5) == (os.path.join(r, 'data_gaussian_5.npy'), os.path.join(r, 'label.npy'))
    assert corruption_files(r, 'clean', 3) == (os.path.join(r, 'data_original.npy'), os.path.join(r, 'label.npy'))
    assert corruption_files(r, 'mixed_corruptions_2', 5) == (os.path.join(r, 'mixed_corruptions_2.npy'),
                                                             os.path.join(r, 'mixed_corruptions_labels.npy'))


@pytest.mark.parametrize('label_shape', ['flat', 'col', 'row'])
def test_corruption_stream_items(tmp_path, label_shape):
    from uni_adapter_amd.tta_data import CorruptionStream, CLASS_NAMES, load_tta_dataset
    data, lab = _write_stream(tmp_path)
    if label_shape != 'flat':                               # ScanObjectNN-C stores (n,1) or (1,n)
        np.save(os.path.join(tmp_path, 'label.npy'), lab.reshape(-1, 1) if label_shape == 'col' else lab.reshape(1, -1))
    ds = load_tta_dataset(_args(tmp_path))
    assert isinstance(ds, CorruptionStream) and len(ds) == 7
    pc, y, name, rgb = ds[3]
    assert pc.dtype == np.float32 and np.array_equal(pc, data[3]) and y == int(lab[3]) and name == CLASS_NAMES['modelnet'][y]
    assert rgb.shape == (64, 3) and bool((rgb == 1).all())
    # default collate of a DataLoader gives what test_zeroshot_3d_core consumes
    b = next(iter(torch.utils.data.DataLoader(ds, batch_size=2, shuffle=False)))
    assert b[0].shape == (2, 64, 3) and b[1].tolist() == lab[:2].tolist() and b[3].shape == (2, 64, 3)
    assert len(load_tta_dataset(_args(tmp_path, debug=True))) == 5
    assert len(load_tta_dataset(_args(tmp_path, mmap=True))) == 7


def test_dataset_selection_and_errors(tmp_path):
    from uni_adapter_amd.tta_data import load_tta_dataset, CLASS_NAMES
    _write_stream(tmp_path, K=15)
    assert load_tta_dataset(_args(tmp_path, dataset_name='ScanObjectNN_C')).class_name == CLASS_NAMES['scanobject']
    assert len(CLASS_NAMES['modelnet']) == 40 and len(CLASS_NAMES['scanobject']) == 15 and len(CLASS_NAMES['shapenet']) == 55
    with pytest.raises(NotImplementedError):
        load_tta_dataset(_args(tmp_path, dataset_name='objaverse'))
    with pytest.raises(FileNotFoundError):
        load_tta_dataset(_args(tmp_path, corruption='lidar'))
    np.save(os.path.join(tmp_path, 'label.npy'), np.zeros(3, dtype=np.int64))
    with pytest.raises(ValueError):
        load_tta_dataset(_args(tmp_path))
    # object arrays would need pickle: refused with an explanation, never unpickled
    np.save(os.path.join(tmp_path, 'label.npy'), np.array([{'a': 1}] * 7, dtype=object), allow_pickle=True)
    with pytest.raises(ValueError, match='pickle'):
        load_tta_dataset(_args(tmp_path))


def test_text_bank_and_flags(tmp_path):
    from uni_adapter_amd.tta_data import load_text_bank, CORRUPTIONS
    from uni_adapter_amd.main_test_time import parse_args
    t = torch.nn.functional.normalize(torch.randn(15, 32), dim=1)
    torch.save(t, os.path.join(tmp_path, 'text.pt'))
    assert torch.equal(load_text_bank(os.path.join(tmp_path, 'text.pt')), t)
    torch.save({'x': t}, os.path.join(tmp_path, 'bad.pt'))
    with pytest.raises(ValueError):
        load_text_bank(os.path.join(tmp_path, 'bad.pt'))
    a = parse_args(['--myroot', str(tmp_path)])                       # defaults of utils/params.py
    assert (a.mode_M, a.severity, a.corruption, a.batch_size, a.seed) == (4, 5, 'all', 1, 42)
    assert (a.dota_epsilon, a.dota_sigma, a.dota_eta, a.dota_rho) == (1e-4, 1e-4, 0.1, 0.02)
    assert a.use_mode_dota and a.res_learning and len(CORRUPTIONS) == 15


@pytest.mark.gpu
def test_driver_end_to_end_and_resume(tmp_path):
    """Two corruption streams through the driver with a small synthetic Uni3D; then checkpoint/resume:
    state_dict() after half a stream + load_state_dict() into a fresh engine reproduces the
    uninterrupted run bit for bit."""
    from uni_adapter_amd import synthetic
    from uni_adapter_amd.point_encoder import create_uni3d_from_state_dict
    from uni_adapter_amd.main_test_time import parse_args, run
    from uni_adapter_amd.Uni_Adapter import TtaEngine
    K, D, n = 15, 256, 12
    g = np.random.default_rng(1)
    for corr in ('gaussian', 'lidar'):
        pts = g.standard_normal((n, 1024, 3)).astype(np.float32)
        pts /= np.linalg.norm(pts, axis=-1).max(axis=1)[:, None, None]
        np.save(os.path.join(tmp_path, f'data_{corr}_5.npy'), pts)
    np.save(os.path.join(tmp_path, 'label.npy'), g.integers(0, K, size=(n, 1)))
    text = synthetic.text_bank(K, D, seed=4)
    torch.save(text, os.path.join(tmp_path, 'text.pt'))
    sd = synthetic.uni3d_state_dict(3, trans_dim=128, embed_dim=D, depth=1, device='cuda')
    model = create_uni3d_from_state_dict(sd, num_group=128, group_size=64, num_heads=2, device='cuda')
    res = {}
    for corr in ('gaussian', 'lidar'):
        a = parse_args(['--myroot', str(tmp_path), '--dataset_name', 'scanobject', '--corruption', corr, '--mode-M', '2',
                        '--precomputed-text-features', os.path.join(tmp_path, 'text.pt'), '--output-dir', str(tmp_path / 'out'),
                        '--name', 'run', '--chunk-size', '5'])
        res.update(run(a, model=model))
    assert set(res) == {'gaussian', 'lidar'} and all(0.0 <= v <= 100.0 for v in res.values())
    assert 'Average Top-1' in open(tmp_path / 'out' / 'run' / 'out.log').read()

    cfg = {'epsilon': 1e-4, 'sigma': 1e-4, 'eta': 0.1, 'rho': 0.02}
    pts = torch.from_numpy(np.load(os.path.join(tmp_path, 'data_gaussian_5.npy'))).cuda()
    rgb = torch.ones_like(pts); noise = torch.randn(pts.shape, device='cuda', generator=torch.Generator(device='cuda').manual_seed(0))
    tgt = torch.from_numpy(np.load(os.path.join(tmp_path, 'label.npy'))).reshape(-1).cuda()

    def fresh():
        return TtaEngine(model, text, cfg, 2, res_learning=True, chunk=6, device=torch.device('cuda'))
    full = fresh()
    f1, _, _ = full.process_chunk(pts[:6], rgb[:6], tgt[:6], noise[:6])
    f2, _, p2 = full.process_chunk(pts[6:], rgb[6:], tgt[6:], noise[6:])
    first = fresh()
    first.process_chunk(pts[:6], rgb[:6], tgt[:6], noise[:6])
    ck = os.path.join(tmp_path, 'state.pt')
    torch.save(first.state_dict(), ck)
    resumed = fresh()
    resumed.load_state_dict(torch.load(ck, weights_only=True))
    g2, _, q2 = resumed.process_chunk(pts[6:], rgb[6:], tgt[6:], noise[6:])
    assert torch.equal(f2, g2) and torch.equal(p2, q2)
    assert resumed.accuracy() == full.accuracy()
    with pytest.raises(ValueError):
        TtaEngine(model, text, cfg, 4, res_learning=True, chunk=6, device=torch.device('cuda')).load_state_dict(torch.load(ck, weights_only=True))


def _driver_worker(rank, world, port, root, mode):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), UA_REPLAY_MODE=mode)
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.cuda.set_device(0)
    import uni_adapter_amd  # noqa: F401
    res = _drive(root, f'run_{mode}_r{rank}', 'gaussian,lidar,impulse' if mode == 'streams' else 'gaussian')
    torch.save(res, os.path.join(root, f'res_{mode}_r{rank}.pt'))
    dist.destroy_process_group()


def _drive(root, name, corruption='gaussian'):
    from uni_adapter_amd import synthetic
    from uni_adapter_amd.point_encoder import create_uni3d_from_state_dict
    from uni_adapter_amd.main_test_time import parse_args, run
    sd = synthetic.uni3d_state_dict(3, trans_dim=128, embed_dim=256, depth=1, device='cuda')
    model = create_uni3d_from_state_dict(sd, num_group=128, group_size=64, num_heads=2, device='cuda')
    a = parse_args(['--myroot', root, '--dataset_name', 'scanobject', '--corruption', corruption, '--mode-M', '2',
                    '--precomputed-text-features', os.path.join(root, 'text.pt'), '--output-dir', os.path.join(root, 'out'),
                    '--name', name, '--chunk-size', '5'])
    return run(a, model=model)


@pytest.mark.gpu
@pytest.mark.parametrize('mode', ['owner', 'replicated'])
def test_driver_two_ranks_equals_single_process(tmp_path, mode):
    """The driver as one rank of a 2-process job (both on cuda:0, gloo): every rank reads the same corruption stream and
    draws the same augmentation noise, the engine shards the samples; both ranks report the single-process accuracy."""
    import socket
    import torch.multiprocessing as mp
    from uni_adapter_amd import synthetic
    K, n = 15, 12
    g = np.random.default_rng(2)
    pts = g.standard_normal((n, 1024, 3)).astype(np.float32)
    pts /= np.linalg.norm(pts, axis=-1).max(axis=1)[:, None, None]
    np.save(os.path.join(tmp_path, 'data_gaussian_5.npy'), pts)
    np.save(os.path.join(tmp_path, 'label.npy'), g.integers(0, K, size=(n, 1)))
    torch.save(synthetic.text_bank(K, 256, seed=4), os.path.join(tmp_path, 'text.pt'))
    want = _drive(str(tmp_path), 'single')
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_driver_worker, args=(2, port, str(tmp_path), mode), nprocs=2, join=True)
    for r in range(2):
        assert torch.load(os.path.join(tmp_path, f'res_{mode}_r{r}.pt'), weights_only=True) == want


@pytest.mark.gpu
def test_driver_deals_corruption_streams_to_ranks(tmp_path):
    """--replay-mode streams: the corruption streams are independent units (state rebuilt per corruption), rank r takes streams
    r, r+W, ...; no collective on the data path, results merged at the end - every rank reports all streams, with the
    single-process numbers (3 streams on 2 ranks: uneven deal)."""
    import socket
    import torch.multiprocessing as mp
    from uni_adapter_amd import synthetic
    K, n = 15, 10
    g = np.random.default_rng(3)
    for corr in ('gaussian', 'lidar', 'impulse'):
        pts = g.standard_normal((n, 1024, 3)).astype(np.float32)
        pts /= np.linalg.norm(pts, axis=-1).max(axis=1)[:, None, None]
        np.save(os.path.join(tmp_path, f'data_{corr}_5.npy'), pts)
    np.save(os.path.join(tmp_path, 'label.npy'), g.integers(0, K, size=(n, 1)))
    torch.save(synthetic.text_bank(K, 256, seed=4), os.path.join(tmp_path, 'text.pt'))
    want = _drive(str(tmp_path), 'single', 'gaussian,lidar,impulse')
    assert set(want) == {'gaussian', 'lidar', 'impulse'}
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_driver_worker, args=(2, port, str(tmp_path), 'streams'), nprocs=2, join=True)
    for r in range(2):
        assert torch.load(os.path.join(tmp_path, f'res_streams_r{r}.pt'), weights_only=True) == want
