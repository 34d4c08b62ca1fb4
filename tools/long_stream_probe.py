"""HIP engine on the golden residual-learning streams, printed against the reference runs of g12 (what the calibrated
assertions of tests/test_gpu_harness.py::test_long_stream_against_the_references_own_spread are built from).  Run on the GPU box."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import uni_adapter_amd  # noqa
from uni_adapter_amd.Uni_Adapter import TtaEngine

class Feeder(torch.nn.Module):
    def __init__(self, feats): super().__init__(); self.feats = feats; self.pos = 0
    def encode_pc(self, clouds):
        n2 = clouds.shape[0]; out = self.feats.view(-1, self.feats.shape[-1])[self.pos:self.pos + n2]; self.pos += n2; return out.contiguous()

def run(g, n, chunk, res=True):
    text = torch.from_numpy(g['text'])
    feats = torch.stack([torch.from_numpy(g['feat'][:n]), torch.from_numpy(g['feat_aug'][:n])], 1).cuda()
    eng = TtaEngine(Feeder(feats), text, {'epsilon': 1e-4, 'sigma': 1e-4, 'eta': 0.1, 'rho': 0.02}, int(g['M']), res_learning=res, chunk=chunk)
    fs, ps = [], []
    for s in range(0, n, chunk):
        m = min(chunk, n - s)
        f, r, p = eng.process_chunk(torch.zeros(m, 8, 3), torch.ones(m, 8, 3))
        fs.append(f.cpu()); ps.append(p.cpu())
    return torch.cat(fs).double().numpy(), torch.cat(ps).numpy()

gd = os.path.join(ROOT, 'tests', 'golden')
for name in ('modelnet40', 'scanobjectnn', 'shapenet55'):
    g = dict(np.load(os.path.join(gd, f'g2_stream_{name}.npz')))
    n = g['feat'].shape[0]
    f, p = run(g, n, 16)
    d = np.abs(f - g['final']).max(1)
    print(name, 'n', n, 'pred mismatch at', np.nonzero(p != g['pred'])[0].tolist(), 'diffs', np.round(d, 4).tolist())
g = dict(np.load(os.path.join(gd, 'g12_long_stream_modelnet40.npz')))
n = g['feat'].shape[0]
f, p = run(g, n, 32)
np.savez(os.path.join(ROOT, 'gpurun_out', 'r02_hip_g12.npz'), final=f, pred=p)
refs = {v: g['final_' + v].astype(np.float64) for v in 'abcde'}
env = np.zeros(n)
for u in 'abcde':
    for v in 'abcde':
        if u < v: env = np.maximum(env, np.abs(refs[u] - refs[v]).max(1))
for v in 'abcde':
    d = np.abs(f - refs[v]).max(1)
    print('g12 HIP vs', v, 'first>1e-3', int(np.argmax(d > 1e-3)) if (d > 1e-3).any() else None, 'max', d.max(), 'agree', (p == g['pred_' + v]).mean(),
          'violations of env', int((d > np.maximum(env, 1e-3)).sum()))
d = np.abs(f - refs['a']).max(1)
print('HIP vs a first 40', np.round(d[:40], 4).tolist())
print('env first 40', np.round(env[:40], 4).tolist())
allagree = np.all(np.stack([g['pred_' + v] for v in 'abcde']) == g['pred_a'][None], 0)
print('steps where all refs agree:', int(allagree.sum()), 'HIP mismatches there:', np.nonzero((p != g['pred_a']) & allagree)[0].tolist())
print('HIP acc', (p == g['labels']).mean(), 'ref accs', [(g['pred_' + v] == g['labels']).mean() for v in 'abcde'])
raise SystemExit
# no residual learning
from oracle import ref_cpu as O
f2, p2 = run(g, n, 32, res=False)
s64 = O.TtaStream(torch.from_numpy(g['text']), O.TtaConfig(num_modes=int(g['M']), res_learning=False), dtype=torch.float64)
s32 = O.TtaStream(torch.from_numpy(g['text']), O.TtaConfig(num_modes=int(g['M']), res_learning=False))
e64, e32, r3264 = [], [], []
for i in range(n):
    a = s64.step(torch.from_numpy(g['feat'][i:i+1]).double(), torch.from_numpy(g['feat_aug'][i:i+1]).double())[2][0].numpy()
    b = s32.step(torch.from_numpy(g['feat'][i:i+1]), torch.from_numpy(g['feat_aug'][i:i+1]))[2][0].double().numpy()
    e64.append(np.abs(f2[i] - a).max()); e32.append(np.abs(f2[i] - b).max()); r3264.append(np.abs(a - b).max())
print('no-res: HIP vs fp64 max', max(e64), 'HIP vs fp32 oracle max', max(e32), 'oracle32 vs 64 max', max(r3264))
print('no-res per 32:', [round(max(e64[i:i+32]), 5) for i in range(0, n, 32)], [round(max(r3264[i:i+32]), 5) for i in range(0, n, 32)])
