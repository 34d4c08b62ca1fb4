"""Where does a gemm_split_kernel wave spend its cycles?  Builds an instrumented copy of
uni-adapter_amd/csrc/ua_gemm_split.hip (-DUA_SPLIT_TRACE: s_memtime stamps around the DMA issue,
the LDS fragment reads, the MFMA issue, the vmcnt(0) wait and the barrier) into tools/_probe/ and
prints the per-wave cycle split for the encoder's GEMM shapes.  Profiling aid, not product code.

    python tools/gemm_trace.py [--shapes qkv,fc1,...]
"""
import argparse
import ctypes as C
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'uni-adapter_amd', 'csrc')
OUT = os.path.join(ROOT, 'tools', '_probe', 'libgemm_trace.so')

SHAPES = {  # (M, N, K) at chunk 31 (rows = 31*513)
    'qkv': (15903, 3072, 1024), 'proj': (15903, 1024, 1024), 'fc1': (15903, 5504, 1024), 'fc2': (15903, 1024, 2752),
    'conv2': (31 * 512 * 64, 256, 128), 'conv3': (31 * 512 * 64, 512, 256), 'conv4': (31 * 512 * 64, 512, 512), 'small': (1024, 1024, 1024),
}


def build(extra=()):
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    src = os.path.join(CSRC, 'ua_gemm_split.hip')
    if not extra and os.path.exists(OUT) and os.path.getmtime(OUT) > max(os.path.getmtime(src), os.path.getmtime(os.path.join(CSRC, 'ua_gemm_common.h'))):
        return OUT
    cmd = ['/opt/rocm/bin/hipcc', '-O3', '--offload-arch=gfx950', '-fPIC', '-std=c++17', '-ffp-contract=off', '-DUA_SPLIT_TRACE', *extra,
           '-I', os.path.join(ROOT, 'include'), '-shared', src, '-o', OUT]
    subprocess.run(cmd, check=True)
    return OUT


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--shapes', default='qkv,proj,fc1,fc2,conv3,small')
    ap.add_argument('--build-only', action='store_true')
    ap.add_argument('--clouds', type=int, default=31, help='token rows = clouds * 513 (the bench step encodes 62 clouds)')
    ap.add_argument('--store-only', action='store_true', help='ablation: first epilogue half = 16 plain float4 row stores (no LDS, no math)')
    ap.add_argument('--pool-only', action='store_true', help='no C stores: epilogue writes only a 64-row max-pool (store-path ablation)')
    a = ap.parse_args()
    so = build(('-DUA_EPI_STORE_ONLY',) if a.store_only else ())
    if a.build_only:
        return
    lib = C.CDLL(so)
    lib.ua_gemm_split.restype = C.c_int
    dev = torch.device('cuda')
    for name in a.shapes.split(','):
        M, N, K = SHAPES[name]
        if a.clouds != 31 and M == 15903:
            M = a.clouds * 513
        A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) / K ** 0.5
        As = torch.empty(M, K, device=dev); Ws = torch.empty(N, K, device=dev)        # H32L32 images (4 B / column)
        for x, y, r, sc in ((A, As, M, 1.0), (W, Ws, N, 256.0)):
            rc = lib.ua_split_f16(C.c_void_p(x.data_ptr()), C.c_int64(K), C.c_void_p(y.data_ptr()), C.c_int64(K), r, K, K, C.c_float(sc), None)
            assert rc == 0
        Cm = torch.empty(M, N, device=dev)
        nt = ((M + 255) // 256) * ((N + 255) // 256)
        tr = torch.zeros(nt * 8 * 6, dtype=torch.int64, device=dev)

        sys.path.insert(0, ROOT)
        import uni_adapter_amd  # noqa: F401
        from uni_adapter_amd.lib import GemmEpilogue
        epi = GemmEpilogue()
        pool = torch.empty((M + 63) // 64, N, device=dev)
        if a.pool_only:
            epi.pool_out = pool.data_ptr(); epi.ldp = N; epi.pool_rows = 64

        def run():
            return lib.ua_gemm_split(C.c_void_p(As.data_ptr()), C.c_int64(K), C.c_float(1.0), C.c_void_p(Ws.data_ptr()), C.c_int64(K),
                                     C.c_float(256.0), None if a.pool_only else C.c_void_p(Cm.data_ptr()), C.c_int64(N), M, N, K,
                                     C.byref(epi), None)
        lib.ua_gemm_split_set_trace(C.c_void_p(0), C.c_void_p(0))
        ref = None
        res = {0: [], 1: []}
        for rep in range(8):                                 # tile-order A/B on the default schedule, interleaved
            for order in (0, 1):
                lib.ua_gemm_split_set_schedule(16); lib.ua_gemm_split_set_tile_order(order)
                for _ in range(2):
                    assert run() == 0
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    run()
                e1.record(); torch.cuda.synchronize()
                res[order].append(e0.elapsed_time(e1) / 10 * 1e3)
        med = {o: sorted(v)[len(v) // 2] for o, v in res.items()}
        print(f'{name}: tile order row-major band {med[0]:.0f} us (min {min(res[0]):.0f}) | blocked 4x8 {med[1]:.0f} us (min {min(res[1]):.0f})', flush=True)
        lib.ua_gemm_split_set_tile_order(1)
        for sched in (2, 8, 16, 2, 8, 16):                 # A/B in one process, interleaved
            lib.ua_gemm_split_set_schedule(sched)
            for _ in range(3):
                assert run() == 0
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                run()
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            if ref is None:
                ref = Cm.clone()
            same = torch.equal(ref, Cm)
            err = (ref - Cm).abs().max().item() / ref.abs().max().item()
            print(f'{name}: schedule {sched}: {ms*1e3:.0f} us, {6.0*M*N*K/ms/1e9:.0f} TF executed, bit-identical to first run: {same} (max rel diff {err:.1e})', flush=True)
        for sched in (2, 8, 16):
          lib.ua_gemm_split_set_schedule(sched)
          tl = torch.zeros(nt * 10, dtype=torch.int64, device=dev)
          lib.ua_gemm_split_set_trace(C.c_void_p(0), C.c_void_p(tl.data_ptr()))
          run(); torch.cuda.synchronize()
          T = tl.cpu().reshape(nt, 10).double()
          fullm = T[:, 3] == 1
          ml, ep = (T[:, 1] - T[:, 0]) / 100, (T[:, 2] - T[:, 1]) / 100
          clk = (T[fullm, 9] / (T[fullm, 1] - T[fullm, 0])).median().item() * 0.1       # cycles per 10 ns -> GHz
          print(f'    schedule {sched}: full tiles main loop {ml[fullm].mean():.1f} us, epilogue {ep[fullm].mean():.1f} us, in-kernel clock {clk:.2f} GHz, '
                f'{T[fullm, 9].mean() / (K // 32):.0f} cycles per K-tile', flush=True)
          span_s = (T[:, 2].max() - T[:, 0].min()) / 100
          busy_s = ((T[:, 2] - T[:, 0]) / 100).sum() / 256
          print(f'      kernel span {span_s:.1f} us; sum(block time)/256 CUs = {busy_s:.1f} us -> {span_s - busy_s:.1f} us per CU not inside a block '
                f'(dispatch gaps, round quantisation, tail); {nt} tiles = {nt / 256:.2f} rounds', flush=True)
          F = T[fullm]
          seg = [(F[:, 4] - F[:, 1]), (F[:, 5] - F[:, 4]), (F[:, 6] - F[:, 5]), (F[:, 7] - F[:, 6]), (F[:, 8] - F[:, 7]), (F[:, 2] - F[:, 8])]
          print('      wave 0 epilogue (us): stage0 %.2f | rows0 %.2f | stage1 %.2f | rows1 %.2f | store drain %.2f | barrier %.2f' % tuple(float(x.mean()) / 100 for x in seg), flush=True)
        lib.ua_gemm_split_set_schedule(2)
        tl = torch.zeros(nt * 10, dtype=torch.int64, device=dev)
        lib.ua_gemm_split_set_trace(C.c_void_p(0), C.c_void_p(tl.data_ptr()))
        run(); torch.cuda.synchronize()
        T = tl.cpu().reshape(nt, 10).double()
        t0 = T[:, 0].min()
        fullm = T[:, 3] == 1
        st, ml, ep = (T[:, 0] - t0) / 100, (T[:, 1] - T[:, 0]) / 100, (T[:, 2] - T[:, 1]) / 100
        span = (T[:, 2].max() - t0) / 100
        busy = ((T[:, 2] - T[:, 0]) / 100).sum() / 256
        print(f'{name}: timeline (us): kernel span {span:.1f}; full tiles: main loop {ml[fullm].mean():.1f} (min {ml[fullm].min():.1f} max {ml[fullm].max():.1f}), '
              f'epilogue {ep[fullm].mean():.1f} (max {ep[fullm].max():.1f}); thin tiles: {(ml[~fullm] + ep[~fullm]).mean() if (~fullm).any() else 0:.1f}; '
              f'sum(block time)/256 CUs = {busy:.1f}; last block start {st.max():.1f}', flush=True)
        import numpy as np
        hist = np.histogram(st.numpy(), bins=8)[0]
        print(f'    block start histogram over the span: {hist.tolist()}', flush=True)
        lib.ua_gemm_split_set_trace(C.c_void_p(tr.data_ptr()), C.c_void_p(0))
        run(); torch.cuda.synchronize()
        t = tr.cpu().reshape(nt, 8, 6).double()
        nk = K // 32
        full = t[:, :, :5].sum(-1)[:, 0] > 0
        names = ['dma_issue', 'lds_read_wait', 'mfma_issue', 'dma_wait', 'barrier']
        out = [f'{name}: M={M} N={N} K={K} {ms*1e3:.0f} us, {6.0*M*N*K/ms/1e9:.0f} TF executed; cycles per slab per wave:']
        for late in (0, 1):
            sel = t[:, :, 5] == late
            v = t[sel][:, :5].mean(0) / nk
            out.append(f"  {'late ' if late else 'early'} waves: " + ' '.join(f'{n}={x:.0f}' for n, x in zip(names, v)) + f' total={v.sum():.0f}')
        print('\n'.join(out), flush=True)


if __name__ == '__main__':
    main()
