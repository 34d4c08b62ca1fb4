import os, sys, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import uni_adapter_amd
from uni_adapter_amd import lib, synthetic
from uni_adapter_amd.dota_mixture import DOTA_mix
def timeit(fn, reps=10, warm=3):
    for _ in range(warm): fn()
    evs = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); evs.append((a, b))
    torch.cuda.synchronize()
    t = sorted(x.elapsed_time(y) for x, y in evs); return t[len(t)//2]
for K, M in ((4096, 16), (1156, 8), (40, 8)):
    D = 1024
    text = synthetic.text_bank(K, D, 1)
    mdl = DOTA_mix({'epsilon': 1e-4, 'sigma': 1e-4}, D, K, text.t(), num_modes=M)
    x = torch.nn.functional.normalize(text[:1].cuda() + 0.05 * torch.randn(1, D, device='cuda'), dim=1)
    g = torch.softmax(100 * x @ text.cuda().t(), 1)
    sc = torch.empty(1, K, device='cuda')
    ms = timeit(lambda: mdl.step_fused(x, x, x, g, scores=sc))
    print(f"variant={os.environ.get('UA_STEP_VARIANT','1')} grid2={os.environ.get('UA_STEP_GRID2','1')} K={K} M={M}: {ms*1e3:.1f} us  {16.0*K*M*D/ms/1e6:.0f} GB/s ({16.0*K*M*D/ms/1e6/8000:.3f} of 8 TB/s)  checksum {float(mdl.mu.double().sum()):.10f} {float(sc.double().sum()):.6f}")
