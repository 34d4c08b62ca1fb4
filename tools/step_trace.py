"""Phase timing of align_step_kernel (instrumented build of csrc/ua_gmm.hip, -DUA_STEP_TRACE). Profiling aid."""
import ctypes as C, os, subprocess, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, 'tools', '_probe', 'libstep_trace.so')
def build():
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    subprocess.run(['/opt/rocm/bin/hipcc', '-O3', '--offload-arch=gfx950', '-fPIC', '-std=c++17', '-ffp-contract=off', '-DUA_STEP_TRACE',
                    '-I', os.path.join(ROOT, 'include'), '-shared', os.path.join(ROOT, 'uni-adapter_amd', 'csrc', 'ua_gmm.hip'), '-o', OUT], check=True)
if __name__ == '__main__':
    import sys
    build()
    if '--build-only' in sys.argv: raise SystemExit
    lib = C.CDLL(OUT)
    K, M, D = 40, 8, 1024
    dev = 'cuda'
    mu = torch.randn(K, M, D, device=dev) * 0.03; iv = 1e4 / (1 + torch.rand(K, M, D, device=dev))
    text0 = torch.nn.functional.normalize(torch.randn(K, D, device=dev), dim=1)
    res = torch.zeros(K, D, device=dev); am = torch.zeros(K, D, device=dev); av = torch.zeros(K, D, device=dev)
    e = text0.clone(); nrm = torch.ones(K, device=dev)
    L = -3000 + 10 * torch.randn(K, K, device=dev); J = L[:, :, None] - 2.0 + 0.1 * torch.randn(K, K, M, device=dev)
    gll = torch.zeros(K, K, M, device=dev); ge = torch.zeros(K, D, device=dev); loss = torch.zeros(1, device=dev)
    tk = torch.zeros(K, dtype=torch.int32, device=dev)
    p = lambda t: C.c_void_p(t.data_ptr())
    for it in range(5):
        rc = lib.ua_align_step(p(mu), p(iv), p(L), p(J), p(text0), p(res), p(am), p(av), p(e), p(nrm), K, M, D, it + 1,
                               C.c_double(1e-3), C.c_double(0.9), C.c_double(0.999), C.c_double(1e-8), p(loss), p(gll), p(ge), p(tk), None)
        assert rc == 0
        torch.cuda.synchronize()
    g = gll.flatten().cpu()
    names = ['start', 'prefetch', 'max', 'expexp', 'sums', 'loss', 'dLn+S', 'bwd+ge', 'handoff', 'adam']
    print('non-last block (cycles from start):', ' '.join(f'{n}={int(v)}' for n, v in zip(names[:9], g[:9])))
    print('last block     (cycles from start):', ' '.join(f'{n}={int(v)}' for n, v in zip(names, g[16:26])))
