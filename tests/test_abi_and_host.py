"""CPU-only checks: the C-ABI library loads and exports every symbol include/uniadapter.h
declares (no compute without a GPU), the ctypes signatures cover the header, host-side logic
(state init, sharding) matches the oracle / golden fixtures, and the product package never
imports the oracle."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    h = open(os.path.join(ROOT, 'include', 'uniadapter.h')).read()
    h = re.sub(r'/\*.*?\*/', '', h, flags=re.S)
    return sorted(set(re.findall(r'\b(ua_[a-z0-9_]+)\s*\(', h)))


def test_library_exports_every_declared_symbol():
    import uni_adapter_amd  # noqa: F401
    from uni_adapter_amd import lib
    so = ctypes.CDLL(lib.LIB_PATH)
    names = _declared_symbols()
    assert len(names) >= 17, names
    for n in names:
        assert hasattr(so, n), f'{n} declared in include/uniadapter.h but not exported'
    assert set(lib.SIGNATURES) | {'ua_version', 'ua_strerror', 'ua_gmm_loglik_mfma_workspace_bytes', 'ua_align_mfma_workspace_bytes', 'ua_align_mfma_supported'} == set(names)
    assert lib.version() == 103
    assert 'envelope' in lib.strerror(-3) and lib.strerror(0) == 'ok'


def test_bad_arguments_return_codes_without_gpu():
    """Argument validation happens on the host before any launch, so it is testable here."""
    import uni_adapter_amd  # noqa: F401
    from uni_adapter_amd import lib
    with pytest.raises(RuntimeError, match='invalid argument'):
        lib.call('ua_gmm_loglik', None, None, None, None, 1, 4, 4, 256, 1e-4, None, None, None, None)
    with pytest.raises(RuntimeError, match='invalid argument'):
        lib.call('ua_fps', None, 1, 1024, 512, 0, 1e-3, None, None, None)
    buf = (ctypes.c_float * 64)()
    p = ctypes.addressof(buf)
    p16 = (p + 15) // 16 * 16
    with pytest.raises(RuntimeError, match='envelope'):     # D not a multiple of 256
        lib.call('ua_gmm_loglik', p16, p16, p16, p16, 1, 4, 4, 100, 1e-4, None, None, None, None)
    with pytest.raises(RuntimeError, match='aligned'):
        lib.call('ua_gmm_loglik', p16 + 4, p16, p16, p16, 1, 4, 4, 256, 1e-4, None, None, None, None)
    with pytest.raises(RuntimeError, match='aligned'):      # K % 4 != 0
        lib.call('ua_gemm', p16, 8, p16, 8, p16, 8, 4, 4, 6, None, None)
    # fused residual step: K above the LDS envelope, D not a multiple of 256, step 0, missing tickets
    args = [p16] * 10
    with pytest.raises(RuntimeError, match='envelope'):
        lib.call('ua_align_step', *args, 129, 8, 1024, 1, 1e-3, 0.9, 0.999, 1e-8, None, None, p16, p16, None)
    with pytest.raises(RuntimeError, match='envelope'):
        lib.call('ua_align_step', *args, 40, 8, 1000, 1, 1e-3, 0.9, 0.999, 1e-8, None, None, p16, p16, None)
    with pytest.raises(RuntimeError, match='invalid argument'):
        lib.call('ua_align_step', *args, 40, 8, 1024, 0, 1e-3, 0.9, 0.999, 1e-8, None, None, p16, p16, None)
    with pytest.raises(RuntimeError, match='invalid argument'):
        lib.call('ua_align_step', *args, 40, 8, 1024, 1, 1e-3, 0.9, 0.999, 1e-8, None, None, p16, None, None)
    with pytest.raises(RuntimeError, match='invalid argument'):          # augmented copy missing
        lib.call('ua_clip_logits_pair', p16, None, p16, 40, 1024, 100.0, p16, p16, None, None, None)
    with pytest.raises(RuntimeError, match='envelope'):                  # D % 4
        lib.call('ua_clip_logits_pair', p16, p16, p16, 40, 1022, 100.0, p16, p16, None, None, None)
    # split GEMM: K must be a multiple of 32, scales positive; rowgroup_add groups are multiples of 32 rows
    with pytest.raises(RuntimeError, match='aligned'):
        lib.call('ua_gemm_split', p16, 48, 1.0, p16, 48, 1.0, p16, 64, 64, 64, 48, None, None)
    with pytest.raises(RuntimeError, match='invalid argument'):
        lib.call('ua_gemm_split', p16, 64, 0.0, p16, 64, 1.0, p16, 64, 64, 64, 64, None, None)
    # 'delta' multi-GPU mode: D % 4, buffer stride too short / not a multiple of 4, misaligned delta base, W < 1
    with pytest.raises(RuntimeError, match='invalid argument'):
        lib.call('ua_gmm_delta_pack', p16, p16, p16, p16, p16, p16, 4, 2, 6, p16, None)
    with pytest.raises(RuntimeError, match='invalid argument'):
        lib.call('ua_gmm_delta_merge', p16, p16, p16, p16, p16, p16, p16, p16, 2, 2 * 4 * 2 * 8, 4, 2, 8, None)
    with pytest.raises(RuntimeError, match='invalid argument'):
        lib.call('ua_gmm_delta_merge', p16, p16, p16, p16, p16, p16, p16, p16, 2, 2 * 4 * 2 * 8 + 4 * 2 + 2, 4, 2, 8, None)
    with pytest.raises(RuntimeError, match='aligned'):
        lib.call('ua_gmm_delta_merge', p16, p16, p16, p16, p16, p16, p16, p16 + 4, 2, 2 * 4 * 2 * 8 + 4 * 2, 4, 2, 8, None)
    with pytest.raises(RuntimeError, match='invalid argument'):
        lib.call('ua_vec_merge', p16, p16, 0, 16, 16, 1.0, p16, None)
    with pytest.raises(RuntimeError, match='invalid argument'):
        lib.call('ua_vec_merge', p16, p16, 2, 8, 16, 0.5, p16, None)
    with pytest.raises(RuntimeError, match='invalid argument'):
        lib.call('ua_vec_delta', p16, None, 16, p16, None)
    # class-sharded replay: more local classes than the per-rank share; W * per < K
    with pytest.raises(RuntimeError, match='invalid argument'):
        lib.call('ua_class_shard_pack', p16, p16, 5, 2, 4, p16, None)
    with pytest.raises(RuntimeError, match='invalid argument'):
        lib.call('ua_fuse_logits_sharded', p16, p16, 2, 1, 4, 9, 2, 0.02, 0.1, 1, None, p16, None, None, None, None)
    # folded LayerNorm: mean/rstd without column sums, statistics pitch shorter than M, finalize with parts = 0 / ld < rows
    e = lib.GemmEpilogue(); e.ln_mean_rstd = p16
    with pytest.raises(RuntimeError, match='invalid argument'):
        lib.call('ua_gemm_split', p16, 64, 1.0, p16, 64, 1.0, p16, 64, 64, 64, 64, ctypes.byref(e), None)
    e = lib.GemmEpilogue(); e.row_stats_out = p16; e.ld_stats = 32
    with pytest.raises(RuntimeError, match='invalid argument'):
        lib.call('ua_gemm_split', p16, 64, 1.0, p16, 64, 1.0, p16, 64, 64, 64, 64, ctypes.byref(e), None)
    e = lib.GemmEpilogue(); e.row_stats_out = p16; e.ld_stats = 64
    with pytest.raises(RuntimeError, match='envelope'):      # exact-f32 kernel has no folded LayerNorm
        lib.call('ua_gemm', p16, 64, p16, 64, p16, 64, 64, 64, 64, ctypes.byref(e), None)
    with pytest.raises(RuntimeError, match='invalid argument'):
        lib.call('ua_row_stats_finalize', p16, 0, 64, 64, 1024, 1e-6, p16, None)
    with pytest.raises(RuntimeError, match='invalid argument'):
        lib.call('ua_row_stats_finalize', p16, 4, 32, 64, 1024, 1e-6, p16, None)
    e = lib.GemmEpilogue(); e.rowgroup_add = p16; e.ldg = 64; e.group_rows = 48
    with pytest.raises(RuntimeError, match='envelope'):
        lib.call('ua_gemm', p16, 64, p16, 64, p16, 64, 96, 64, 64, ctypes.byref(e), None)


def test_cpu_tensors_are_rejected_loudly():
    import uni_adapter_amd  # noqa: F401
    from uni_adapter_amd import lib
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        lib.ptr(torch.zeros(4))
    if not torch.cuda.is_available():
        from uni_adapter_amd.dota_mixture import DOTA_mix
        with pytest.raises(RuntimeError, match='HIP device'):
            DOTA_mix({'epsilon': 1e-4, 'sigma': 1e-4}, 256, 4, torch.randn(256, 4), num_modes=2)


def test_init_state_cpu_matches_reference_golden(golden_dir):
    import uni_adapter_amd  # noqa: F401
    from uni_adapter_amd.dota_mixture import init_state_cpu
    g = np.load(os.path.join(golden_dir, 'g2_stream_modelnet40.npz'))
    text = torch.from_numpy(g['text'])
    mu, var, pi, c, cc, eps, sig = init_state_cpu({'epsilon': 1e-4, 'sigma': 1e-4}, 1024, 40, text.T, int(g['M']))
    assert np.array_equal(mu[:, :, :16].numpy(), g['init_mu_head'])
    assert np.array_equal(var[:, :, :16].numpy(), g['init_var_head'])
    assert np.array_equal(pi.numpy(), g['init_pi']) and np.array_equal(c.numpy(), g['init_c'])
    assert float(mu.double().sum()) == float(g['init_mu_sum']) and float(var.double().sum()) == float(g['init_var_sum'])
    g1 = np.load(os.path.join(golden_dir, 'g1_init_sigma_autocorrect.npz'))
    text = torch.from_numpy(g1['text'])
    mu, var, pi, c, cc, eps, sig = init_state_cpu({'epsilon': 1e-3, 'sigma': 1.0}, text.shape[1], text.shape[0], text.T, 3)
    assert np.array_equal(mu[:, :, :8].numpy(), g1['mu_head']) and np.array_equal(var[:, :, :8].numpy(), g1['var_head'])
    assert sig == 1.0 / text.shape[1]


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, 'uni-adapter_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h', '.cpp')):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle', src, flags=re.M), f
                assert 'ref_cpu' not in src, f
    # bench.py may use the oracle only inside cpu_baseline()
    b = open(os.path.join(ROOT, 'bench.py')).read()
    uses = [m.start() for m in re.finditer(r'from oracle', b)]
    assert len(uses) == 1 and b.rfind('def cpu_baseline', 0, uses[0]) > b.rfind('\ndef ', 0, b.rfind('def cpu_baseline', 0, uses[0]))


def test_synthetic_weights_have_reference_key_layout():
    import uni_adapter_amd  # noqa: F401
    from uni_adapter_amd import synthetic
    from oracle import ref_cpu as O
    a = synthetic.uni3d_state_dict(0, trans_dim=128, embed_dim=64, depth=2)
    b = O.make_uni3d_state_dict(0, trans_dim=128, embed_dim=64, enc_dim=512, depth=2)
    assert set(a) == set(b)
    for k in a:
        assert a[k].shape == b[k].shape, k
