"""Phase timing of align_bwd_adam_mfma_kernel (instrumented build of csrc/ua_align_mfma.hip, -DUA_ALIGN_TRACE). Profiling aid."""
import ctypes as C, os, subprocess, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, 'tools', '_probe', 'libalign_trace.so')
def build():
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    src = [os.path.join(ROOT, 'uni-adapter_amd', 'csrc', f) for f in ('ua_align_mfma.hip', 'ua_gemm_split.hip', 'ua_logits.hip')]
    subprocess.run(['/opt/rocm/bin/hipcc', '-O3', '--offload-arch=gfx950', '-fPIC', '-std=c++17', '-ffp-contract=off', '-DUA_ALIGN_TRACE',
                    '-I', os.path.join(ROOT, 'include'), '-shared'] + src + ['-o', OUT], check=True)
if __name__ == '__main__':
    build()
    if '--build-only' in sys.argv: raise SystemExit
    lib = C.CDLL(OUT)
    lib.ua_align_mfma_workspace_bytes.restype = C.c_size_t
    K, M, D = 40, 8, 1024
    dev = 'cuda'
    g = torch.Generator(device=dev).manual_seed(0)
    text0 = torch.nn.functional.normalize(torch.randn(K, D, device=dev, generator=g), dim=1)
    mu = text0[:, None, :].repeat(1, M, 1) + 0.002 * torch.randn(K, M, D, device=dev, generator=g)
    var = 1e-4 * (1 + torch.rand(K, M, D, device=dev, generator=g)); pi = torch.full((K, M), 1.0 / M, device=dev)
    res = torch.zeros(K, D, device=dev); am = torch.zeros(K, D, device=dev); av = torch.zeros(K, D, device=dev); t = text0.clone()
    L = torch.empty(K, K, device=dev); J = torch.empty(K, K, M, device=dev); U = torch.empty(K, K, M, device=dev); nrm = torch.empty(K, device=dev)
    loss = torch.zeros(16, device=dev)
    nb = lib.ua_align_mfma_workspace_bytes(K, M, D)
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    p = lambda x: C.c_void_p(x.data_ptr())
    assert lib.ua_align_mfma_prepare(p(mu), p(var), p(pi), K, M, D, C.c_float(1e-4), p(ws), C.c_size_t(nb), None) == 0
    for it in range(5):
        assert lib.ua_align_forward_mfma(p(t), K, M, D, p(ws), C.c_size_t(nb), p(L), p(J), p(U), p(nrm), None) == 0
        assert lib.ua_align_step_mfma(p(L), p(J), p(U), p(nrm), p(text0), p(res), p(am), p(av), p(t), K, M, D, it + 1, C.c_double(1e-3), C.c_double(0.9),
                                      C.c_double(0.999), C.c_double(1e-8), p(ws), C.c_size_t(nb), p(loss), None, None) == 0
        torch.cuda.synchronize()
    names = ['start', 'loads issued+landed', 'stats', 'S->LDS + weights', 'mfma', 'G to LDS', 'adam']
    print('cycles from start:', ' '.join(f'{n}={int(v)}' for n, v in zip(names, loss[1:8].cpu())))
