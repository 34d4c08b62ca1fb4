"""Prints one SHA-256 over the outputs of ua_gemm_split for the launch shapes that have a compile-time-specialised epilogue row
loop (transformer blocks, mini-PointNet convolutions).  Run under UA_EPI_LEAN=1 and =0 by
tests/test_gpu_kernels.py::test_specialised_epilogue_loops_write_the_same_bytes - the two digests must be equal."""
import ctypes as C
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import uni_adapter_amd  # noqa: E402,F401
from uni_adapter_amd import lib  # noqa: E402
from uni_adapter_amd.lib import GemmEpilogue  # noqa: E402


def split(x, scale):
    rows, cols = x.shape
    y = torch.empty(rows, cols, device='cuda')
    lib.call('ua_split_f16', x.data_ptr(), cols, y.data_ptr(), cols, rows, cols, cols, float(scale), lib.stream_ptr())
    return y


def main():
    g = torch.Generator(device='cuda').manual_seed(5)
    M, K = 768, 128                                             # 3 full row panels (the specialised loops serve full 64x64 sub-tiles)
    h = hashlib.sha256()
    for name, N, act, opt in [('qkv', 512, 0, dict(s=1)), ('proj', 256, 0, dict(c=1, s=1, res=1, st=1)), ('last fc2', 256, 0, dict(c=1, res=1)),
                              ('fc1', 512, 3, dict(s=1, st=1)), ('fc1 no sub-LN', 512, 3, dict(s=1)), ('fc1 fp32', 512, 3, dict(c=1)),
                              ('conv2', 256, 0, dict(s=1, pool=1)), ('conv3', 512, 1, dict(s=1, aff=1)), ('conv4', 512, 0, dict(pool=1))]:
        A = torch.randn(M, K, device='cuda', generator=g); W = torch.randn(N, K, device='cuda', generator=g) / 8
        As, Ws = split(A, 1.0), split(W, 256.0)
        n_out = N // 2 if act == 3 else N
        keep = [torch.randn(N, device='cuda', generator=g)]
        e = GemmEpilogue(); e.bias = keep[0].data_ptr(); e.act = act
        outs = []
        Cd = None
        if opt.get('c'):
            Cd = torch.zeros(M, n_out, device='cuda'); outs.append(Cd)
        if opt.get('s'):
            Cs = torch.zeros(M, n_out, device='cuda'); e.split_out = Cs.data_ptr(); e.ld_split = n_out; e.split_scale = 2.0; outs.append(Cs)
        if opt.get('res'):
            keep.append(torch.randn(M, n_out, device='cuda', generator=g)); e.residual = keep[-1].data_ptr(); e.ldr = n_out
        if opt.get('st'):
            st = torch.zeros((N + 255) // 256, M, 2, device='cuda'); e.row_stats_out = st.data_ptr(); e.ld_stats = M; outs.append(st)
        if opt.get('pool'):
            pool = torch.zeros(M // 64, n_out, device='cuda'); e.pool_out = pool.data_ptr(); e.ldp = n_out; e.pool_rows = 64; outs.append(pool)
        if opt.get('aff'):
            keep += [1 + 0.1 * torch.randn(N, device='cuda', generator=g), torch.randn(N, device='cuda', generator=g),
                     torch.randn(M // 64, N, device='cuda', generator=g)]
            e.scale, e.shift, e.rowgroup_add = keep[-3].data_ptr(), keep[-2].data_ptr(), keep[-1].data_ptr(); e.ldg = N; e.group_rows = 64
        lib.call('ua_gemm_split', As.data_ptr(), K, 1.0, Ws.data_ptr(), K, 256.0, Cd.data_ptr() if Cd is not None else None, n_out, M, N, K,
                 C.byref(e), lib.stream_ptr())
        torch.cuda.synchronize()
        for o in outs:
            h.update(o.cpu().numpy().tobytes())
    print('digest', h.hexdigest())


if __name__ == '__main__':
    main()
