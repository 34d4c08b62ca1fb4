"""A/B of the persistent tile walk of ua_gemm_split (one process, interleaved rounds): the Uni3D-L encoder forward and the four
block GEMM shapes at the bench's 62 clouds.   python tools/gemm_persist_ab.py"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import uni_adapter_amd  # noqa
from uni_adapter_amd import lib, synthetic
from uni_adapter_amd.lib import GemmEpilogue
from uni_adapter_amd.point_encoder import create_uni3d_from_state_dict

def ev(fn, reps):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps

def shapes():
    M = 62 * 513
    for N, K, name in ((3072, 1024, 'qkv'), (1024, 1024, 'proj'), (5504, 1024, 'fc1'), (1024, 2752, 'fc2'), (512, 512, 'conv4'), (512, 256, 'conv3'),
                       (5504, 1024, 'fc1x4'), (3072, 1024, 'qkvx4')):
        M = 8 * 512 * 64 if name.startswith('conv') else (248 * 513 if name.endswith('x4') else 62 * 513)
        A = torch.randn(M, K, device='cuda'); W = torch.randn(N, K, device='cuda') / K ** 0.5
        As = torch.empty(M, K, device='cuda'); Ws = torch.empty(N, K, device='cuda'); st = lib.stream_ptr()
        lib.call('ua_split_f16', A.data_ptr(), K, As.data_ptr(), K, M, K, K, 1.0, st)
        lib.call('ua_split_f16', W.data_ptr(), K, Ws.data_ptr(), K, N, K, K, 1024.0, st)
        Cc = torch.empty(M, N, device='cuda'); e = GemmEpilogue()
        fn = lambda: lib.call('ua_gemm_split', As.data_ptr(), K, 1.0, Ws.data_ptr(), K, 1024.0, Cc.data_ptr(), N, M, N, K, C.byref(e), st)
        outs, t = {}, {0: [], 1: []}
        for rnd in range(5):
            for mode in (0, 1):
                lib.call('ua_gemm_split_set_persistent', mode)
                fn(); t[mode].append(ev(fn, 10))
                if rnd == 0: outs[mode] = Cc.clone()
        fl = 6.0 * M * N * K
        print(f'{name:5s} M={M} N={N} K={K}: one-tile {min(t[0]) * 1e3:7.1f} us ({fl / min(t[0]) / 1e9:6.0f} TF exec) | persistent {min(t[1]) * 1e3:7.1f} us '
              f'({fl / min(t[1]) / 1e9:6.0f} TF exec) | ratio {min(t[1]) / min(t[0]):.3f} | bit-identical {torch.equal(outs[0], outs[1])}', flush=True)

def encoder(NC):
    sd = synthetic.uni3d_state_dict(0, depth=24, device='cuda')
    m = create_uni3d_from_state_dict(sd, num_group=512, group_size=64, num_heads=16, precision='split')
    xyz, rgb = synthetic.point_clouds(NC, 1024, 3, device='cuda')
    pc = torch.cat([xyz, rgb], -1)
    outs, t = {}, {0: [], 1: [], 2: []}
    for rnd in range(3):
        for mode in (0, 1, 2):
            lib.call('ua_gemm_split_set_persistent', mode)
            o = m.encode_pc(pc); t[mode].append(ev(lambda: m.encode_pc(pc), 3))
            if rnd == 0: outs[mode] = o.clone()
    print(f'Uni3D-L forward, {NC} clouds: one-tile {min(t[0]):.2f} ms | persistent always {min(t[1]):.2f} ms | persistent >= 8 rounds {min(t[2]):.2f} ms '
          f'| ratios {min(t[1]) / min(t[0]):.3f} {min(t[2]) / min(t[0]):.3f} | bit-identical {torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])}', flush=True)

def comparator():
    """What the vendor library (hipBLASLt through torch.matmul, plain fp16 operands, fp32 accumulate) does with the SAME number of
    matrix-core products: (M, 3K) x (3K, N) in fp16 executes as many MFMAs as the 3-term split GEMM (M, K) x (K, N).  Comparator only."""
    M = 62 * 513
    for N, K, name in ((3072, 1024, 'qkv'), (1024, 1024, 'proj'), (5504, 1024, 'fc1'), (1024, 2752, 'fc2'), (8192, 8192, 'cube8k')):
        if name == 'cube8k': Mx = 8192
        else: Mx = M
        A = torch.randn(Mx, 3 * K, device='cuda', dtype=torch.float16); W = torch.randn(N, 3 * K, device='cuda', dtype=torch.float16)
        fn = lambda: torch.matmul(A, W.t())
        fn(); t = min(ev(fn, 10) for _ in range(3))
        print(f'comparator hipBLASLt fp16 {name}: ({Mx} x {3 * K}) x ({3 * K} x {N}): {t * 1e3:7.1f} us = {2.0 * Mx * N * 3 * K / t / 1e9:6.0f} TF', flush=True)

if __name__ == '__main__':
    shapes(); encoder(62); encoder(248); comparator()
    lib.call('ua_gemm_split_set_persistent', -1)
