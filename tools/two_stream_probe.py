"""Probe: the two encoder forwards of a chunk (original and augmented clouds) as ONE batch on one stream vs as two
half-batches enqueued on two HIP streams (separate workspaces), Uni3D-L, 31 samples.  Profiling aid."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import uni_adapter_amd  # noqa: E402,F401
from uni_adapter_amd import synthetic, point_encoder as PE  # noqa: E402

dev = torch.device('cuda')
sd = synthetic.uni3d_state_dict(0, device=dev)
model = PE.create_uni3d_from_state_dict(sd, num_group=512, group_size=64, num_heads=16, device=dev, precision='split')
enc = model.point_encoder
n = int(sys.argv[1]) if len(sys.argv) > 1 else 31
xyz, rgb = synthetic.point_clouds(2 * n, 1024, seed=1, device=dev)
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()


def one_batch():
    enc.ws_slot = 0
    return enc(xyz, rgb)


def two_streams():
    cur = torch.cuda.current_stream()
    sA.wait_stream(cur); sB.wait_stream(cur)
    with torch.cuda.stream(sA):
        enc.ws_slot = 0
        a = enc(xyz[:n], rgb[:n])
    with torch.cuda.stream(sB):
        enc.ws_slot = 1
        b = enc(xyz[n:], rgb[n:])
    cur.wait_stream(sA); cur.wait_stream(sB)
    enc.ws_slot = 0
    return torch.cat([a, b])


for name, fn in (('one batch of %d clouds' % (2 * n), one_batch), ('two streams x %d clouds' % n, two_streams), ('one batch again', one_batch), ('two streams again', two_streams)):
    for _ in range(2):
        out = fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(6):
        out = fn()
    torch.cuda.synchronize()
    print(f'{name:28s} {(time.perf_counter() - t0) / 6 * 1e3:7.2f} ms per {2 * n} clouds', flush=True)
ref = one_batch(); got = two_streams(); torch.cuda.synchronize()
print('max |difference| between the two ways:', (ref - got).abs().max().item())
