"""Phase timing of gmm_step1_kernel (instrumented build of csrc/ua_gmm.hip, -DUA_GMM_TRACE): s_memtime stamps of one steady-state
class of workgroup 0, per wave.  Profiling aid.   python tools/gmm_trace.py [K M]"""
import ctypes as C, os, subprocess, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFS = [a for a in sys.argv[1:] if a.startswith('-D')]
TAG = ''.join(d[2:].lower().replace('ua_gmm_', '_').replace('=', '') for d in DEFS)
OUT = os.path.join(ROOT, 'tools', '_probe', f'libgmm_trace{TAG}.so')
def build():
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    subprocess.run(['/opt/rocm/bin/hipcc', '-O3', '--offload-arch=gfx950', '-fPIC', '-std=c++17', '-ffp-contract=off', '-DUA_GMM_TRACE'] + DEFS +
                   ['-I', os.path.join(ROOT, 'include'), '-shared', os.path.join(ROOT, 'uni-adapter_amd', 'csrc', 'ua_gmm.hip'), '-o', OUT], check=True)
if __name__ == '__main__':
    if '--build-only' in sys.argv:
        build(); raise SystemExit
    if not os.path.exists(OUT):
        build()
    lib = C.CDLL(OUT)
    print('variant:', ' '.join(DEFS) or '(default)')
    args = [a for a in sys.argv[1:] if not a.startswith('-')]
    K, M, D = (int(args[0]), int(args[1]), 1024) if len(args) >= 2 else (4096, 16, 1024)
    dev = 'cuda'
    g = torch.Generator(device=dev).manual_seed(0)
    text = torch.nn.functional.normalize(torch.randn(K, D, device=dev, generator=g), dim=1)
    mu = (text[:, None, :] + 1e-3 * torch.randn(K, M, D, device=dev, generator=g)).contiguous()
    var = (1e-4 * (1 + torch.rand(K, M, D, device=dev, generator=g))).contiguous()
    pi = torch.full((K, M), 1.0 / M, device=dev); c = torch.full((K, M), 1.0 / M, device=dev); cc = torch.zeros(K, device=dev)
    x = torch.nn.functional.normalize(text[5:6] + 0.05 * torch.randn(1, D, device=dev, generator=g), dim=1).contiguous()
    gam = torch.softmax(100 * x @ text.T, 1).contiguous(); sc = torch.empty(1, K, device=dev)
    p = lambda t: C.c_void_p(t.data_ptr())
    # scores of the first call (fresh state) against fp64 on the device: sum_d ln v + (x16 - mu)^2 / v, logsumexp over modes
    xq = x.half().double()
    vv = (var.double() + 1e-4).clamp_min(1e-8)
    ll = -0.5 * (vv.log().sum(-1) + ((xq[0] - mu.double()) ** 2 / vv).sum(-1))
    ref = torch.logsumexp((pi.double() + 1e-10).log() + ll, dim=1)
    mu0, var0, pi0, c0, cc0 = (t.clone() for t in (mu, var, pi, c, cc))
    lib.ua_gmm_step_fused(p(mu), p(var), p(pi), p(c), p(cc), p(x), 1, p(x), p(x), p(gam), p(sc), K, M, D, C.c_float(1e-4), None)
    torch.cuda.synchronize()
    print(f'first call: max |score - fp64| = {(sc[0].double() - ref).abs().max().item():.3e} on scores of magnitude {ref.abs().max().item():.1f}; '
          f'digest of scores {float(sc.double().sum()):.10e}, of mu {float(mu.double().sum()):.12e}, of var {float(var.double().sum()):.12e}')
    for _ in range(3):
        rc = lib.ua_gmm_step_fused(p(mu), p(var), p(pi), p(c), p(cc), p(x), 1, p(x), p(x), p(gam), p(sc), K, M, D, C.c_float(1e-4), None)
        assert rc == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        lib.ua_gmm_step_fused(p(mu), p(var), p(pi), p(c), p(cc), p(x), 1, p(x), p(x), p(gam), p(sc), K, M, D, C.c_float(1e-4), None)
    e1.record(); torch.cuda.synchronize()
    print(f'K={K} M={M}: {e0.elapsed_time(e1) * 100:.1f} us per launch (instrumented build)')
    buf = (C.c_ulonglong * 256)()
    assert lib.ua_debug_gmm_trace(buf) == 0
    names = ['top', 'prefetch issued', 'pass1 (logdet+2 maha)', 'joints->LDS', 'barrier A', 'lse + g (fit 1)', 'fit-1 update', 'pre pass2',
             'pass2 (logdet+maha)', 'barrier B', 'pi/joint/lse (fit 2)', 'fit-2 update', 'store issued', 'r = nx (waits prefetch)']
    print('cycles per phase of one steady-state class of workgroup 0 (per wave; s_memtime):')
    print('wave ' + ' '.join(f'{i:>7d}' for i in range(1, 14)) + '   total')
    for m in range(M):
        t = [buf[m * 16 + i] for i in range(14)]
        d = [t[i] - t[i - 1] for i in range(1, 14)]
        print(f'{m:4d} ' + ' '.join(f'{v:7d}' for v in d) + f' {t[13] - t[0]:7d}')
    for i, n in enumerate(names[1:], 1):
        print(f'  {i:2d}: {n}')
