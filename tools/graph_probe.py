"""Does a HIP graph of the cache replay's launch sequence run faster than the same launches enqueued one by one?
Captures ua_tta_chunk for n samples into a graph (torch.cuda.graph -> hipStreamBeginCapture) and times replays of it
against direct calls.  Timing experiment only: the Adam bias-correction scalars are kernel arguments, so a replayed graph
repeats the captured step numbers.    python tools/graph_probe.py [n]"""
import ctypes, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import uni_adapter_amd  # noqa: E402,F401
from uni_adapter_amd import lib, synthetic  # noqa: E402
from uni_adapter_amd.Uni_Adapter import TtaEngine  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 31
    gpath = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'g2_stream_modelnet40.npz')
    text = torch.from_numpy(np.load(gpath)['text']) if os.path.exists(gpath) else synthetic.text_bank(40, 1024)
    eng = TtaEngine(None, text, {'epsilon': 1e-4, 'sigma': 1e-4, 'eta': 0.1, 'rho': 0.02}, 8, res_learning=True, chunk=n, device=torch.device('cuda'))
    D, K = 1024, text.shape[0]
    g = torch.Generator(device='cuda').manual_seed(3)
    feats = torch.randn(n, 2, D, device='cuda', generator=g) * 0.3 + text.cuda()[torch.randint(0, K, (n,), device='cuda', generator=g)][:, None, :]
    final = torch.empty(n, K, device='cuda'); preds = torch.empty(n, dtype=torch.int32, device='cuda')

    def call(stream):
        lib.call('ua_tta_chunk', ctypes.byref(eng._state), feats.data_ptr(), None, n, 1, 0, final.data_ptr(), None, preds.data_ptr(), stream.cuda_stream)

    def timed(fn, reps=8):
        ts = []
        for _ in range(reps):
            torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        return sorted(ts)[len(ts) // 2]
    cur = torch.cuda.current_stream()
    for _ in range(2):
        call(cur)
    direct = timed(lambda: call(cur))
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        call(torch.cuda.current_stream())
    graph.replay(); torch.cuda.synchronize()
    replayed = timed(graph.replay)
    print(f'direct launches: {direct / n * 1e6:.1f} us/sample; graph replay: {replayed / n * 1e6:.1f} us/sample ({n} samples per graph)')


if __name__ == '__main__':
    main()
