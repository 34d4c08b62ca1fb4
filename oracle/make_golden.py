#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING the reference (read-only at /root/reference).

TEST INFRASTRUCTURE ONLY - run in the build container (the GPU box has no /root/reference):

    python oracle/make_golden.py [--skip-long | --only-long | --only-extras]     # writes tests/golden/
                                        (the g12 long streams take ~20 minutes; --only-long regenerates just those)

What executes reference code here (nothing is copied into the repo; only inputs/outputs are
stored):
  * `dota_mixture.py` (it needs only torch) and `utils/math_utils.py` are executed from their SOURCE TEXT
    (`compile(open(path).read())` into a fresh module object): Python's bytecode cache is bypassed, so a stale
    `.pyc` shipped in the reference is never executed and nothing is written under /root/reference.
  * `Uni_Adapter.py` cannot be imported (needs `clip`, `open_clip`), so the pure functions
    `softmax_entropy`, `get_logits_wrapper`, `compute_text_alignment_loss` and the
    `--use-mode-dota` statements of the per-sample loop body (Uni_Adapter.py:388-521) are located
    with `ast`, compiled from the reference's own syntax tree and executed in a namespace that
    supplies the loop's free variables.  The encoder is replaced by a feeder that returns
    pre-generated features (the encoder itself is third-party timm + pointnet2_ops).
  * `models/point_encoder.py` needs `pointnet2_ops`; only the definitions of
    `square_distance`, `knn_point`, `Group`, `Encoder` are compiled from its syntax tree, with
    `fps` supplied explicitly (FPS centres are an input of the fixture).
  * `models/ulip/pointbert/misc.py:fps` (pure torch) is compiled the same way for the FPS fixture.
"""
import ast
import os
import sys
import types

sys.dont_write_bytecode = True

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

REF = os.environ.get('UA_REFERENCE', '/root/reference')
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), 'tests', 'golden')
sys.path.insert(0, os.path.dirname(HERE))

from oracle import ref_cpu as O  # noqa: E402  (weights helper only)


def _load_source_module(name, path):
    """Execute a reference file from its source text into a fresh module (no import machinery, no bytecode cache)."""
    mod = types.ModuleType(name)
    mod.__file__ = path
    with open(path) as f:
        exec(compile(f.read(), path, 'exec'), mod.__dict__)
    return mod


def _tree(path):
    with open(path) as f:
        return ast.parse(f.read(), filename=path)


def _exec_defs(path, names, ns):
    """Compile selected top-level def/class nodes of a reference file into `ns`."""
    tree = _tree(path)
    body = [n for n in tree.body
            if isinstance(n, (ast.FunctionDef, ast.ClassDef)) and n.name in names]
    assert {n.name for n in body} == set(names), (names, [n.name for n in body])
    mod = ast.Module(body=body, type_ignores=[])
    exec(compile(mod, path, 'exec'), ns)
    return ns


def _loop_body_statements(path):
    """Statements of the per-sample loop in test_zeroshot_3d_core that form the
    --use-mode-dota path: text/clip_weights block, get_logits_wrapper call, big if/elif."""
    tree = _tree(path)
    fn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == 'test_zeroshot_3d_core')
    loops = [n for n in ast.walk(fn) if isinstance(n, ast.For)
             and isinstance(n.iter, ast.Call) and getattr(n.iter.func, 'id', '') == 'enumerate'
             and 'test_loader' in ast.unparse(n.iter)]
    loop = loops[0]
    keep = []
    for st in loop.body:
        src = ast.unparse(st)
        if isinstance(st, ast.If) and 'text_current' in src and 'clip_weights' in src:
            keep.append(st)                                    # Uni_Adapter.py:388-401
        elif isinstance(st, ast.Assign) and 'get_logits_wrapper' in src:
            keep.append(st)                                    # :406
        elif isinstance(st, ast.If) and 'mode_dota_model.predict' in src:
            keep.append(st)                                    # :409-574
    assert len(keep) == 3, [ast.unparse(s)[:60] for s in keep]
    return compile(ast.Module(body=keep, type_ignores=[]), path, 'exec')


def _fusion_statements(path):
    """Just the fusion assignments (Uni_Adapter.py:491-521) for the known-answer fixture."""
    tree = _tree(path)
    fn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == 'test_zeroshot_3d_core')
    want = ('dota_weights_val', 'dota_logits', 'entropy_clip', 'entropy_dota', 'weight_clip',
            'weight_dota', 'final_logits')
    assigns = sorted((n for n in ast.walk(fn) if isinstance(n, ast.Assign)
                      and isinstance(n.targets[0], ast.Name)), key=lambda n: n.lineno)
    out = []
    started = False
    for st in assigns:
        nm = st.targets[0].id
        if nm == 'dota_weights_val' and 'mode_dota_model.c.mean' in ast.unparse(st):
            started = True
        if started and nm in want:
            out.append(st)
            if nm == 'final_logits':
                break
    if len(out) < 8:
        raise RuntimeError('fusion statements not found')
    return compile(ast.Module(body=out, type_ignores=[]), path, 'exec')


class Feeder:
    """Stands in for the encoder: returns queued features from encode_pc()."""
    def __init__(self):
        self.q = []
    def encode_pc(self, feature):
        return self.q.pop(0)
    def float(self):
        return self
    def eval(self):
        return self


def make_features(text, n, seed, noise=0.05, aug_noise=0.02):
    """Synthetic encoder outputs: un-normalised vectors near a random class anchor."""
    g = torch.Generator().manual_seed(seed)
    K, D = text.shape
    labels = torch.randint(0, K, (n,), generator=g)
    scale = 1.0 + 4.0 * torch.rand(n, 1, generator=g)
    f = (text[labels] + noise * torch.randn(n, D, generator=g)) * scale
    fa = (f / scale + aug_noise * torch.randn(n, D, generator=g)) * scale
    return f.contiguous(), fa.contiguous(), labels


def run_stream(text, M, n_steps, seed, ua_ns, loop_code, dota_mixture, dtype=torch.float32):
    """dtype=torch.float64 runs the SAME reference statements in double precision: the state is initialised in fp32 (so
    that every arithmetic starts from identical numbers) and cast to double, and `Tensor.float()` - which the reference
    calls on its inputs (dota_mixture.py:170-171,250; Uni_Adapter.py:394) - is mapped to `Tensor.double()` for the run."""
    K, D = text.shape
    cfg = {'epsilon': 1e-4, 'sigma': 1e-4, 'eta': 0.1, 'rho': 0.02}
    f, fa, labels = make_features(text, n_steps, seed)
    if dtype == torch.float64:
        orig_float = torch.Tensor.float
        torch.Tensor.float = lambda self, *a, **k: self.double()
        try:
            return _run_stream(text, M, n_steps, f, fa, labels, cfg, ua_ns, loop_code, dota_mixture, dtype)
        finally:
            torch.Tensor.float = orig_float
    return _run_stream(text, M, n_steps, f, fa, labels, cfg, ua_ns, loop_code, dota_mixture, dtype)


def _run_stream(text, M, n_steps, f, fa, labels, cfg, ua_ns, loop_code, dota_mixture, dtype, progress=None):
    K, D = text.shape
    args = types.SimpleNamespace(vlm3d='uni3d', use_dota=False, use_mode_dota=True, res_learning=True,
                                 print_freq=10 ** 9, device='cpu', use_gmm_dota=False)
    model = Feeder()
    text_features = text.clone()
    text_features_initial = text_features.clone().detach()
    mdl = dota_mixture.DOTA_mix(cfg, D, K, text_features_initial.T, num_modes=M)
    mdl.eval()
    if dtype != torch.float32:
        for nm in ('mu', 'var', 'pi', 'c', 'class_counts'):
            setattr(mdl, nm, getattr(mdl, nm).to(dtype))
        text_features, text_features_initial = text_features.to(dtype), text_features_initial.to(dtype)
        f, fa = f.to(dtype), fa.to(dtype)
    raw = {}
    orig_predict = mdl.predict
    def predict_spy(x, source_priors=None):
        o = orig_predict(x, source_priors)
        raw['z'] = o.clone()
        return o
    mdl.predict = predict_spy
    text_residuals = torch.zeros_like(text_features_initial, requires_grad=True)
    opt = torch.optim.Adam([text_residuals], lr=0.001)
    import logging
    ns = dict(ua_ns)
    ns.update(dict(torch=torch, F=F, args=args, model=model, mode_dota_model=mdl, dota_model=None,
                   dota_cfg=cfg, text_features=text_features, text_features_initial=text_features_initial,
                   text_residuals=text_residuals, residual_optimizer=opt, residual_lr_start=0.001,
                   residual_lr_end=0.001, num_batches=n_steps, logging=logging))
    rec = {k: [] for k in ('clip', 'dota_raw', 'final', 'pred', 'cmean', 'loss', 'w')}
    snaps = {}
    init = dict(mu=mdl.mu.clone(), var=mdl.var.clone(), pi=mdl.pi.clone(), c=mdl.c.clone())
    for i in range(n_steps):
        model.q = [f[i:i + 1].clone(), fa[i:i + 1].clone()]
        ns['i'] = i
        ns['pc'] = torch.zeros(1, 8, 3)
        ns['rgb'] = torch.ones(1, 8, 3)
        ns['feature'] = torch.cat((ns['pc'], ns['rgb']), dim=-1)     # Uni_Adapter.py:385
        with torch.no_grad():
            exec(loop_code, ns)
        rec['clip'].append(ns['clip_logits'][0].detach().clone())
        rec['dota_raw'].append(raw['z'][0].clone())
        rec['final'].append(ns['final_logits'][0].detach().clone())
        rec['pred'].append(ns['pred'])
        rec['cmean'].append(float(mdl.c.mean()))
        rec['w'].append(float(ns['dota_weights_val']))
        rec['loss'].append(float(ns['alignment_loss']) if i > 0 else float('nan'))
        if progress and i % 16 == 0:
            print(f'[long stream {progress}] step {i}/{n_steps}', flush=True)
        if i in (0, 1, n_steps // 2, n_steps - 1):
            snaps[i] = dict(c=mdl.c.clone(), pi=mdl.pi.clone(), res=text_residuals.detach().clone())
    out = dict(
        text=text.numpy(), M=np.int64(M), feat=f.numpy().astype(np.float32), feat_aug=fa.numpy().astype(np.float32),
        labels=labels.numpy(),
        clip=torch.stack(rec['clip']).numpy(), dota_raw=torch.stack(rec['dota_raw']).numpy(),
        final=torch.stack(rec['final']).numpy(), pred=np.array(rec['pred'], dtype=np.int64),
        cmean=np.array(rec['cmean']), w=np.array(rec['w']), loss=np.array(rec['loss']),
        init_pi=init['pi'].numpy(), init_c=init['c'].numpy(),
        init_mu_head=init['mu'][:, :, :16].numpy(), init_var_head=init['var'][:, :, :16].numpy(),
        init_mu_sum=np.float64(init['mu'].double().sum()), init_var_sum=np.float64(init['var'].double().sum()),
        end_c=mdl.c.numpy(), end_pi=mdl.pi.numpy(), end_class_counts=mdl.class_counts.numpy(),
        end_mu_head=mdl.mu[:, :, :16].numpy(), end_var_head=mdl.var[:, :, :16].numpy(),
        end_mu_sum=np.float64(mdl.mu.double().sum()), end_var_sum=np.float64(mdl.var.double().sum()),
        end_res=text_residuals.detach().numpy(),
        end_adam_m=opt.state[text_residuals]['exp_avg'].numpy(),
        end_adam_v=opt.state[text_residuals]['exp_avg_sq'].numpy(),
        snap_steps=np.array(sorted(snaps)),
    )
    for s, d in snaps.items():
        out[f'snap{s}_c'] = d['c'].numpy()
        out[f'snap{s}_pi'] = d['pi'].numpy()
        out[f'snap{s}_res_head'] = d['res'][:, :32].numpy()
    return out, mdl, text_residuals.detach().clone()


def _reference_pieces():
    dota_mixture = _load_source_module('ref_dota_mixture', os.path.join(REF, 'dota_mixture.py'))   # reference module
    ua_path = os.path.join(REF, 'Uni_Adapter.py')
    ua_ns = {'torch': torch, 'F': F, 'np': np}
    _exec_defs(ua_path, ['softmax_entropy', 'get_logits_wrapper', 'compute_text_alignment_loss'], ua_ns)
    loop_code = _loop_body_statements(ua_path)
    fusion_code = _fusion_statements(ua_path)
    tf = os.path.join(REF, 'precomputed_text_features', 'Uni3D')
    banks = {
        'modelnet40': torch.load(os.path.join(tf, 'text_features_large.pt'), map_location='cpu', weights_only=True),
        'scanobjectnn': torch.load(os.path.join(tf, 'text_features_large_scanobjectnn.pt'), map_location='cpu', weights_only=True),
        'shapenet55': torch.load(os.path.join(tf, 'text_features_large_shapenetcorev2.pt'), map_location='cpu', weights_only=True),
    }
    banks = {k: v.float().contiguous() for k, v in banks.items()}
    return dota_mixture, ua_ns, loop_code, fusion_code, banks


# ---- G12: long stream, the reference's loop against ITSELF under other fp32 summation orders and in fp64 -------------
# Residual learning makes the reference's fp32 trajectory sensitive to rounding (the loss routes a gradient through
# likelihood_matrix.max(), Uni_Adapter.py:247-248, and Adam's first steps are sign-like).  To pin how far two correct fp32
# evaluations of the SAME reference statements drift apart over a stream, the loop body is run on one 256-sample stream
#   a: as is, 1 thread (the arithmetic of the g2 fixtures)      b: the same with 8 threads
#   c: with the feature dimension permuted (text bank and features; the first 16 coordinates - the ones the
#      initialisation offsets touch, dota_mixture.py:70-79 - stay in place): mathematically the same stream,
#      every sum over D associates differently
#   d: with the class order permuted (rows of the text bank; logits mapped back): every sum over K associates differently
#   e: in fp64 (see run_stream) - the "truth" the fp32 runs scatter around
# and per-step final logits / predictions of each are stored.  Tests assert the HIP path against the envelope of a-d.
LONG = dict(bank='modelnet40', M=8, n=256, seed=14)
LONG_STREAMS = {'modelnet40': dict(bank='modelnet40', M=8, n=256, seed=14),          # g12_long_stream_modelnet40.npz
                'scanobjectnn': dict(bank='scanobjectnn', M=8, n=256, seed=15)}      # g12_long_stream_scanobjectnn.npz
LONG_VARIANTS = ('a', 'b', 'c', 'd', 'e')


def long_stream_variant(variant, out_path, stream='modelnet40'):
    LONG.update(LONG_STREAMS[stream])
    dota_mixture, ua_ns, loop_code, _, banks = _reference_pieces()
    text = banks[LONG['bank']]
    K, D = text.shape
    M, n = LONG['M'], LONG['n']
    cfg = {'epsilon': 1e-4, 'sigma': 1e-4, 'eta': 0.1, 'rho': 0.02}
    f, fa, labels = make_features(text, n, LONG['seed'])
    torch.set_num_threads(8 if variant == 'b' else 1)
    g = torch.Generator().manual_seed(1000 + ord(variant))
    back = None
    if variant == 'c':
        perm = torch.cat([torch.arange(16), 16 + torch.randperm(D - 16, generator=g)])
        text, f, fa = text[:, perm].contiguous(), f[:, perm].contiguous(), fa[:, perm].contiguous()
    if variant == 'd':
        q = torch.randperm(K, generator=g)
        text = text[q].contiguous()
        back = torch.argsort(q)                                   # class k of the original order sits at row back[k]
    dtype = torch.float64 if variant == 'e' else torch.float32
    if dtype == torch.float64:
        orig_float = torch.Tensor.float
        torch.Tensor.float = lambda self, *a, **k: self.double()
    try:
        out, _, _ = _run_stream(text, M, n, f, fa, labels, cfg, ua_ns, loop_code, dota_mixture, dtype, progress=variant)
    finally:
        if dtype == torch.float64:
            torch.Tensor.float = orig_float
    final, pred, dz, clip = out['final'], out['pred'], out['dota_raw'], out['clip']
    if back is not None:
        final, dz, clip = final[:, back.numpy()], dz[:, back.numpy()], clip[:, back.numpy()]
        pred = q.numpy()[pred]
    np.savez_compressed(out_path, final=final, pred=pred.astype(np.int64), dota_raw=dz, clip=clip, w=out['w'], loss=out['loss'])


def long_streams(stream='modelnet40'):
    """Runs the five variants as parallel child processes (several minutes each) and merges them into one fixture."""
    import subprocess
    import tempfile
    LONG.update(LONG_STREAMS[stream])
    tmp = tempfile.mkdtemp(prefix='ua_g12_')
    procs = {v: subprocess.Popen([sys.executable, os.path.abspath(__file__), '--long-variant', v, os.path.join(tmp, v + '.npz'), stream])
             for v in LONG_VARIANTS}
    for v, pr in procs.items():
        if pr.wait() != 0:
            raise RuntimeError(f'long-stream variant {v} failed')
    dota_mixture, ua_ns, loop_code, _, banks = _reference_pieces()
    text = banks[LONG['bank']]
    f, fa, labels = make_features(text, LONG['n'], LONG['seed'])
    out = dict(text=text.numpy(), M=np.int64(LONG['M']), feat=f.numpy(), feat_aug=fa.numpy(), labels=labels.numpy(),
               variants=np.array(LONG_VARIANTS))
    for v in LONG_VARIANTS:
        d = np.load(os.path.join(tmp, v + '.npz'))
        out['final_' + v] = d['final'] if v == 'e' else d['final'].astype(np.float32)
        out['pred_' + v] = d['pred']
        out['dota_raw_' + v] = d['dota_raw'] if v == 'e' else d['dota_raw'].astype(np.float32)
        if v in ('a', 'e'):
            out['clip_' + v] = d['clip']
            out['loss_' + v] = d['loss']
        if v == 'a':
            out['w'] = d['w']
    np.savez_compressed(os.path.join(OUT, f'g12_long_stream_{stream}.npz'), **out)
    fa_, fe = out['final_a'].astype(np.float64), out['final_e']
    for v in LONG_VARIANTS[1:]:
        dv = np.abs(out['final_' + v].astype(np.float64) - fa_).max(1)
        print('g12', v, 'vs a: max', dv.max(), 'first step over 1e-3:', int(np.argmax(dv > 1e-3)) if (dv > 1e-3).any() else None,
              'pred agreement', float((out['pred_' + v] == out['pred_a']).mean()))


# ---- extras (round 2): branches of the reference that its CLI does not reach but its code defines -----------------------
def extras():
    torch.manual_seed(0)
    torch.set_num_threads(1)
    # G10b: dota.DOTA(streaming_update_Sigma=False) - the covariances stay at sigma*I, only means and counts move (dota.py:54-58)
    ns10 = {'torch': torch, 'nn': nn}
    _exec_defs(os.path.join(REF, 'dota.py'), ['DOTA'], ns10)
    K10, D10, n10 = 8, 256, 10
    g10 = torch.Generator().manual_seed(33)
    text10 = torch.nn.functional.normalize(torch.randn(K10, D10, generator=g10), dim=1)
    ref10 = ns10['DOTA']({'epsilon': 1e-4, 'sigma': 1e-4}, D10, K10, text10.T.clone(), streaming_update_Sigma=False)
    lab10 = torch.randint(0, K10, (n10,), generator=g10)
    X10 = torch.nn.functional.normalize(text10[lab10] + 0.35 * torch.randn(n10, D10, generator=g10), dim=1)
    Y10 = torch.softmax(100.0 * X10 @ text10.T, dim=1)
    sc10 = []
    for i in range(n10):
        sc10.append(ref10.predict(X10[i:i + 1].half()).float())
        ref10.fit(X10[i:i + 1], Y10[i:i + 1]); ref10.update()
    np.savez_compressed(os.path.join(OUT, 'g10b_dota_mean_only.npz'), text=text10.numpy(), X=X10.numpy(), Y=Y10.numpy(), eps=np.float64(1e-4),
                        sigma=np.float64(1e-4), scores=torch.cat(sc10).numpy(), mu=ref10.mu.numpy(), c=ref10.c.numpy(),
                        Sigma0=ref10.Sigma[0].numpy(), Lambda=ref10.Lambda.float().numpy())
    print('g10b dota mean-only', torch.cat(sc10).shape)
    # G11b: the exact-inverse refinement of the cluster cache (compute_cache_logits_old, Uni_Adapter.py:159-189 with
    #       math_utils.online_value_refinement_old) on the g11 stream
    mu11 = _load_source_module('ref_math_utils', os.path.join(REF, 'utils', 'math_utils.py'))
    ns11 = {'torch': torch, 'F': F, 'np': np, 'math_utils': mu11}
    _exec_defs(os.path.join(REF, 'Uni_Adapter.py'), ['softmax_entropy', 'get_entropy', 'update_cache', 'compute_cache_logits_old'], ns11)
    K11, D11, n11, cap11 = 6, 256, 40, 3
    g11 = torch.Generator().manual_seed(77)
    text11 = torch.nn.functional.normalize(torch.randn(K11, D11, generator=g11), dim=1)
    lab11 = torch.randint(0, K11, (n11,), generator=g11)
    feats11 = text11[lab11] + 0.9 * torch.randn(n11, D11, generator=g11) / 16
    hp11 = {'shot_capacity': cap11, 'beta': 150, 'threshold': 0.5, 'lambda_reg': 0.11}
    cw11 = text11.T.contiguous()
    cache11 = {}
    fin11, z211 = [], []
    for i in range(n11):                                                       # loop body :406, :541-574 with use_new_approximation False
        pcf = feats11[i:i + 1] / feats11[i:i + 1].norm(dim=-1, keepdim=True)
        cl = 100. * pcf @ cw11
        loss = ns11['softmax_entropy'](cl); pm = cl.softmax(1)
        pred = int(cl.topk(1, dim=1)[1].t()[0])
        pe = torch.tensor(ns11['get_entropy'](loss, cw11))
        ns11['update_cache'](cache11, pred, [pcf, loss, pe, pm], shot_capacity=cap11, clip_weights=cw11, beta=150)
        fl = cl.clone() / 100.0
        p1 = torch.softmax(fl, dim=1); e1 = ns11['softmax_entropy'](p1, False)
        fl2 = ns11['compute_cache_logits_old'](pcf, cache11, cw11, hp11)
        p2 = torch.softmax(fl2, dim=1); e2 = ns11['softmax_entropy'](p2, False)
        fin11.append((1 / e1).reshape(-1, 1) * p1 + (1 / e2).reshape(-1, 1) * p2); z211.append(fl2)
    np.savez_compressed(os.path.join(OUT, 'g11b_cluster_cache_exact_inverse.npz'), text=text11.numpy(), feats=feats11.numpy(), cap=np.int64(cap11),
                        beta=np.float64(150), threshold=np.float64(0.5), lambda_reg=np.float64(0.11), finals=torch.cat(fin11).numpy(),
                        cache_logits=torch.cat(z211).numpy())
    print('g11b cluster cache, exact inverse', torch.cat(z211).shape)
    # G13: attention maps by the reference's own hook arithmetic for EVA-style blocks (extract_attention.py:115-160: separate
    #      q/k/v projections, heads, scale, softmax) - the in-tree description of the third-party timm EvaAttention
    import logging
    import typing
    ns13 = {'torch': torch, 'nn': nn, 'F': F, 'np': np, 'logging': logging, 'Dict': typing.Dict, 'List': typing.List, 'Tuple': typing.Tuple,
            'Optional': typing.Optional, 'logger': logging.getLogger('g13')}
    _exec_defs(os.path.join(REF, 'extract_attention.py'), ['AttentionExtractor'], ns13)
    ext = object.__new__(ns13['AttentionExtractor'])
    ext.attention_maps = {}
    hook = ext._attention_hook_with_save(0)
    C13, H13, G13, B13 = 128, 2, 128, 2
    sd13 = O.make_uni3d_state_dict(5, trans_dim=C13, embed_dim=64, enc_dim=512, depth=1)
    pre = 'point_encoder.visual.blocks.0.attn.'
    lin = lambda name, bias: _linear_from(sd13[pre + name + '.weight'], sd13.get(pre + name + '.bias') if bias else None)
    module = types.SimpleNamespace(q_proj=lin('q_proj', True), k_proj=lin('k_proj', False), v_proj=lin('v_proj', True), num_heads=H13)
    g13 = torch.Generator().manual_seed(41)
    x13 = torch.randn(B13, G13 + 1, C13, generator=g13)
    with torch.no_grad():
        hook(module, (x13,), None)
    amap = ext.attention_maps['layer_0']
    np.savez_compressed(os.path.join(OUT, 'g13_eva_attention_map.npz'), x=x13.numpy(), weights_seed=np.int64(5), heads=np.int64(H13),
                        attn=amap.numpy())
    print('g13 attention map', tuple(amap.shape), 'row sums', float(amap.sum(-1).min()), float(amap.sum(-1).max()))


def _linear_from(w, b):
    m = nn.Linear(w.shape[1], w.shape[0], bias=b is not None)
    with torch.no_grad():
        m.weight.copy_(w)
        if b is not None:
            m.bias.copy_(b)
    return m


def main():
    os.makedirs(OUT, exist_ok=True)
    if len(sys.argv) >= 4 and sys.argv[1] == '--long-variant':
        return long_stream_variant(sys.argv[2], sys.argv[3], sys.argv[4] if len(sys.argv) > 4 else 'modelnet40')
    if '--only-long' in sys.argv:
        for st in LONG_STREAMS:
            if '--long-stream' not in sys.argv or sys.argv[sys.argv.index('--long-stream') + 1] == st:
                long_streams(st)
        return
    if '--only-extras' in sys.argv:
        return extras()
    torch.manual_seed(0)
    torch.set_num_threads(1)            # fixes CPU reduction order for reproducible fixtures
    dota_mixture, ua_ns, loop_code, fusion_code, banks = _reference_pieces()
    ua_path = os.path.join(REF, 'Uni_Adapter.py')

    # ---- G1/G2: init + full loop-body stream (reference DOTA_mix + loop statements) ----
    for name, M, n, seed in (('modelnet40', 8, 48, 11), ('scanobjectnn', 8, 24, 12), ('shapenet55', 4, 16, 13)):
        out, mdl, res = run_stream(banks[name], M, n, seed, ua_ns, loop_code, dota_mixture)
        np.savez_compressed(os.path.join(OUT, f'g2_stream_{name}.npz'), **out)
        print('g2', name, 'preds', out['pred'][:8], 'final[0,:3]', out['final'][0, :3])
        if name == 'scanobjectnn':
            # ---- G3: alignment loss / matrix / gradient on the end state (K=15) ----
            text0 = banks[name]
            r = res.clone().requires_grad_(True)
            e = text0 + r
            e = e / e.norm(dim=1, keepdim=True)
            loss, Lm = ua_ns['compute_text_alignment_loss'](e, mdl)
            (g,) = torch.autograd.grad(loss, r)
            row3 = mdl.predict(e[3:4].detach())
            np.savez_compressed(os.path.join(OUT, 'g3_alignment_scanobjectnn.npz'),
                                text=text0.numpy(), res=res.numpy(), mu=mdl.mu.numpy(), var=mdl.var.numpy(),
                                pi=mdl.pi.numpy(), c=mdl.c.numpy(), eps=np.float64(mdl.epsilon),
                                loss=np.float64(loss.item()), L=Lm.detach().numpy(), grad_res=g.numpy(),
                                predict_row3=row3.numpy())
            print('g3 loss', loss.item())

    # ---- cache-only stream without residual learning, B>1 fit (DOTA_mix API surface) ----
    text = banks['scanobjectnn']
    K, D = text.shape
    cfg = {'epsilon': 1e-4, 'sigma': 1e-4}
    mdl = dota_mixture.DOTA_mix(cfg, D, K, text.T, num_modes=4)
    g = torch.Generator().manual_seed(5)
    xs, gs, zs = [], [], []
    for it in range(6):
        B = (1, 3, 2, 4, 1, 5)[it]
        lab = torch.randint(0, K, (B,), generator=g)
        x = F.normalize(text[lab] + 0.05 * torch.randn(B, D, generator=g), dim=1)
        gam = torch.softmax(100 * x @ text.T, 1)
        zs.append(mdl.predict(x.half()).clone())
        mdl.fit(x, gam)
        xs.append(x); gs.append(gam)
    np.savez_compressed(os.path.join(OUT, 'g2b_batched_fit.npz'), text=text.numpy(), M=np.int64(4),
                        **{f'x{i}': xs[i].numpy() for i in range(6)}, **{f'g{i}': gs[i].numpy() for i in range(6)},
                        **{f'z{i}': zs[i].numpy() for i in range(6)},
                        end_mu=mdl.mu.numpy(), end_var=mdl.var.numpy(), end_c=mdl.c.numpy(), end_pi=mdl.pi.numpy(),
                        end_class_counts=mdl.class_counts.numpy(), end_t=np.int64(mdl.t))

    # ---- sigma auto-correction branch (dota_mixture.py:46-54) ----
    m2 = dota_mixture.DOTA_mix({'epsilon': 1e-3, 'sigma': 1.0}, D, K, text.T, num_modes=3)
    np.savez_compressed(os.path.join(OUT, 'g1_init_sigma_autocorrect.npz'), text=text.numpy(),
                        mu_head=m2.mu[:, :, :8].numpy(), var_head=m2.var[:, :, :8].numpy(),
                        pi=m2.pi.numpy(), c=m2.c.numpy())

    # ---- G4: fusion known answers (reference statements :491-521) ----
    g = torch.Generator().manual_seed(7)
    cases = []
    for t in range(8):
        Kc = (40, 15, 55, 40, 7, 40, 40, 3)[t]
        clip_z = 100 * (0.1 + 0.2 * torch.rand(1, Kc, generator=g))
        dota_z = 1500 + 1500 * torch.rand(1, Kc, generator=g)
        c = (1.0 / 8) + (10 ** (t - 3)) * torch.rand(Kc, 8, generator=g)
        ns = dict(ua_ns)
        ns.update(dict(torch=torch, dota_cfg={'rho': 0.02, 'eta': 0.1}, pc_features=torch.zeros(1, 4),
                       mode_dota_model=types.SimpleNamespace(c=c), clip_logits=clip_z.clone(),
                       dota_logits=dota_z.clone()))
        exec(fusion_code, ns)
        cases.append((clip_z, dota_z, c, ns['final_logits'].clone(), float(ns['dota_weights_val'])))
    np.savez_compressed(os.path.join(OUT, 'g4_fusion.npz'), n=np.int64(len(cases)),
                        **{f'clip{i}': c[0].numpy() for i, c in enumerate(cases)},
                        **{f'dota{i}': c[1].numpy() for i, c in enumerate(cases)},
                        **{f'c{i}': c[2].numpy() for i, c in enumerate(cases)},
                        **{f'final{i}': c[3].numpy() for i, c in enumerate(cases)},
                        w=np.array([c[4] for c in cases]))

    # ---- G5/G6: grouping + mini-PointNet from models/point_encoder.py ----
    pe_path = os.path.join(REF, 'models', 'point_encoder.py')
    box = {}
    def given_fps(xyz, number):                        # FPS centres are a fixture INPUT
        return box['centers']
    pe_ns = {'torch': torch, 'nn': nn, 'fps': given_fps, 'logging': __import__('logging')}
    _exec_defs(pe_path, ['square_distance', 'knn_point', 'Group', 'Encoder'], pe_ns)
    g = torch.Generator().manual_seed(21)
    xyz = torch.randn(2, 1024, 3, generator=g)
    xyz = xyz / xyz.norm(dim=-1).max()
    xyz[1, 100:110] = xyz[1, 90:100]                   # duplicate points -> exact distance ties
    rgb = torch.rand(2, 1024, 3, generator=g)
    cidx = O.fps(xyz, 128, start=0, skip_origin_eps=-1.0)
    centers = xyz[torch.arange(2)[:, None], cidx]
    box['centers'] = centers
    sq = pe_ns['square_distance'](centers, xyz)
    grp = pe_ns['Group'](num_group=128, group_size=32)
    nb, cen, feat = grp(xyz, rgb)
    idx = pe_ns['knn_point'](32, xyz, centers)
    np.savez_compressed(os.path.join(OUT, 'g5_group.npz'), xyz=xyz.numpy(), rgb=rgb.numpy(),
                        center_idx=cidx.numpy(), sqdist_rows=sq[:, :4].numpy(),
                        idx_sorted=idx.sort(-1)[0].numpy(),
                        feat_sorted_sum=feat.double().sum((2,)).numpy(), feat_g0=feat[:, :2].numpy(),
                        idx_g0=idx[:, :2].numpy())
    sd = O.make_uni3d_state_dict(3, trans_dim=64, embed_dim=32, enc_dim=512, depth=0)
    enc = pe_ns['Encoder'](encoder_channel=512)
    enc.load_state_dict({k[len('point_encoder.encoder.'):]: v for k, v in sd.items()
                         if k.startswith('point_encoder.encoder.')}, strict=False)
    enc.eval()
    with torch.no_grad():
        tok = enc(feat[:, :16])
    np.savez_compressed(os.path.join(OUT, 'g6_patch_encoder.npz'), feat=feat[:, :16].numpy(),
                        weights_seed=np.int64(3), tokens=tok.numpy(),
                        **{'w_' + k[len('point_encoder.encoder.'):]: v.numpy() for k, v in sd.items()
                           if k.startswith('point_encoder.encoder.first_conv')})
    print('g6 tokens', tok.shape, tok[0, 0, :3])

    # ---- G7: pure-torch FPS (models/ulip/pointbert/misc.py:40-60) with recorded start ----
    misc_ns = {'torch': torch, 'np': np}
    _exec_defs(os.path.join(REF, 'models', 'ulip', 'pointbert', 'misc.py'), ['fps', 'index_points'], misc_ns)
    torch.manual_seed(1234)
    start = int(torch.randint(0, 1024, (1,), dtype=torch.long))   # same draw the reference makes
    torch.manual_seed(1234)
    pts = misc_ns['fps'](xyz[:1], 64)
    np.savez_compressed(os.path.join(OUT, 'g7_fps.npz'), xyz=xyz[:1].numpy(), start=np.int64(start),
                        centers=pts.numpy())
    print('g7 start', start)
    # ---- G8: ULIP-2 PointBERT (models/ulip/pointbert/point_encoder.py + dvae.py + ulip_model.py) ----
    dv_ns = {'torch': torch, 'nn': nn, 'F': F, 'fps': given_fps}
    _exec_defs(os.path.join(REF, 'models', 'ulip', 'pointbert', 'dvae.py'), ['knn_point', 'square_distance', 'Group', 'Encoder'], dv_ns)
    pt_ns = {'torch': torch, 'nn': nn, 'F': F, 'DropPath': nn.Identity, 'Group': dv_ns['Group'], 'Encoder': dv_ns['Encoder']}
    _exec_defs(os.path.join(REF, 'models', 'ulip', 'pointbert', 'point_encoder.py'),
               ['Mlp', 'Attention', 'Block', 'TransformerEncoder', 'PointTransformer'], pt_ns)
    uargs = types.SimpleNamespace(pc_feat_dim=768, pc_depth=2, drop_path_rate=0.0, num_head=6, group_size=32, num_group=128, encoder_dim=256)
    usd = O.make_ulip_state_dict(9, trans_dim=384, depth=2)
    ptm = pt_ns['PointTransformer'](uargs)
    missing = ptm.load_state_dict({k[len('point_encoder.'):]: v for k, v in usd.items() if k.startswith('point_encoder.')}, strict=True)
    ptm.eval()
    uxyz = xyz[:, :, :]
    ucidx = O.fps(uxyz, 128, start=3, skip_origin_eps=-1.0)
    box['centers'] = uxyz[torch.arange(2)[:, None], ucidx]
    with torch.no_grad():
        ucat = ptm(uxyz)
        uemb = ucat @ usd['pc_projection']                                   # ulip_model.py:15-18
    np.savez_compressed(os.path.join(OUT, 'g8_ulip.npz'), xyz=uxyz.numpy(), center_idx=ucidx.numpy(), weights_seed=np.int64(9),
                        concat=ucat.numpy(), embed=uemb.numpy())
    print('g8 ulip', uemb.shape, uemb[0, :3])
    # ---- G9: OpenShape PPAT parts that run without torch_redstone (models/openshape/pointnet_util.py, ppta.py:9-81) ----
    from einops import rearrange
    feats9 = torch.cat([xyz, rgb], -1)                                  # in_channel 6 = xyz + rgb
    G9, S9, R9, H9 = 128, 64, 0.2, 8
    cidx9 = O.fps(xyz, G9, start=1, skip_origin_eps=-1.0)
    pu = {'torch': torch, 'nn': nn, 'F': F, 'farthest_point_sample': lambda xyz_, npoint: cidx9}    # FPS start is random in the reference (:75)
    _exec_defs(os.path.join(REF, 'models', 'openshape', 'pointnet_util.py'),
               ['square_distance', 'index_points', 'query_ball_point', 'sample_and_group', 'PointNetSetAbstraction'], pu)
    sd9 = O.make_ppat_state_dict(21, depth=2)
    sa9 = pu['PointNetSetAbstraction'](npoint=G9, radius=R9, nsample=S9, in_channel=9, mlp=[64, 64, 256], group_all=False)
    sa9.load_state_dict({k[len('ppat.sa.'):]: v for k, v in sd9.items() if k.startswith('ppat.sa.')}, strict=False)
    sa9.eval()
    with torch.no_grad():
        nx9, nf9 = sa9(xyz.transpose(1, 2).contiguous(), feats9.transpose(1, 2).contiguous())
        ball9 = pu['query_ball_point'](R9, S9, xyz, xyz[torch.arange(2)[:, None], cidx9])
    pt9 = {'torch': torch, 'nn': nn, 'rearrange': rearrange}
    _exec_defs(os.path.join(REF, 'models', 'openshape', 'ppta.py'), ['PreNorm', 'FeedForward', 'Attention', 'Transformer'], pt9)
    tr9 = pt9['Transformer'](512, 2, H9, 64, 1536, 0.0, False)
    tr9.load_state_dict({k[len('ppat.transformer.'):]: v for k, v in sd9.items() if k.startswith('ppat.transformer.')}, strict=True)
    tr9.eval()
    tok9 = O.ppat_encode(xyz, feats9, sd9, G9, R9, S9, H9, center_idx=cidx9, return_parts=True)[3]
    with torch.no_grad():
        y9 = tr9(tok9, None)
    np.savez_compressed(os.path.join(OUT, 'g9_ppat.npz'), xyz=xyz.numpy(), feats=feats9.numpy(), center_idx=cidx9.numpy(),
                        weights_seed=np.int64(21), ball_idx=ball9.numpy().astype(np.int32),
                        sa_centers=nx9.transpose(1, 2).contiguous().numpy(), sa_features=nf9.transpose(1, 2).contiguous().numpy(),
                        tokens_in=tok9.numpy(), tokens_out=y9.numpy())
    print('g9 ppat', y9.shape)
    # ---- G10: full-covariance DOTA (dota.py:19-87; `import clip` at :8 is unused, so only the class is compiled) ----
    ns10 = {'torch': torch, 'nn': nn}
    _exec_defs(os.path.join(REF, 'dota.py'), ['DOTA'], ns10)
    K10, D10, n10 = 8, 256, 12
    g10 = torch.Generator().manual_seed(31)
    text10 = torch.nn.functional.normalize(torch.randn(K10, D10, generator=g10), dim=1)
    cfg10 = {'epsilon': 1e-4, 'sigma': 1e-4}
    ref10 = ns10['DOTA'](cfg10, D10, K10, text10.T.clone())
    lab10 = torch.randint(0, K10, (n10,), generator=g10)
    X10 = torch.nn.functional.normalize(text10[lab10] + 0.35 * torch.randn(n10, D10, generator=g10), dim=1)
    Y10 = torch.softmax(100.0 * X10 @ text10.T, dim=1)
    sc10 = []
    for i in range(n10):
        sc10.append(ref10.predict(X10[i:i + 1].half()).float())
        ref10.fit(X10[i:i + 1], Y10[i:i + 1]); ref10.update()
    ref10.fit(X10[:3], Y10[:3]); ref10.update()                              # one batched fit (B = 3)
    sb10 = ref10.predict(X10[3:5].half()).float()
    np.savez_compressed(os.path.join(OUT, 'g10_dota_fullcov.npz'), text=text10.numpy(), X=X10.numpy(), Y=Y10.numpy(), eps=np.float64(1e-4),
                        sigma=np.float64(1e-4), scores=torch.cat(sc10).numpy(), scores_after_batch=sb10.numpy(), mu=ref10.mu.numpy(),
                        c=ref10.c.numpy(), Sigma0=ref10.Sigma[0].numpy(), overall_Sigma=ref10.overall_Sigma.numpy(),
                        Lambda=ref10.Lambda.float().numpy())
    print('g10 dota full covariance', torch.cat(sc10).shape)
    # ---- G11: original cluster cache + graph refinement (Uni_Adapter.py:78-156,541-574; utils/math_utils.py imported as is) ----
    mu11 = _load_source_module('ref_math_utils', os.path.join(REF, 'utils', 'math_utils.py'))
    ns11 = {'torch': torch, 'F': F, 'np': np, 'math_utils': mu11}
    _exec_defs(os.path.join(REF, 'Uni_Adapter.py'), ['softmax_entropy', 'get_entropy', 'update_cache', 'compute_cache_logits'], ns11)
    K11, D11, n11, cap11 = 6, 256, 40, 3
    g11 = torch.Generator().manual_seed(77)
    text11 = torch.nn.functional.normalize(torch.randn(K11, D11, generator=g11), dim=1)
    lab11 = torch.randint(0, K11, (n11,), generator=g11)
    feats11 = text11[lab11] + 0.9 * torch.randn(n11, D11, generator=g11) / 16
    hp11 = {'shot_capacity': cap11, 'beta': 150, 'threshold': 0.5, 'lambda_reg': 0.11}
    cw11 = text11.T.contiguous()
    cache11, Lo11, Li11 = {}, 0, 0
    fin11, z211 = [], []
    for i in range(n11):                                                       # the statements of the loop body :406, :541-574
        pcf = feats11[i:i + 1] / feats11[i:i + 1].norm(dim=-1, keepdim=True)
        cl = 100. * pcf @ cw11
        loss = ns11['softmax_entropy'](cl); pm = cl.softmax(1)
        pred = int(cl.topk(1, dim=1)[1].t()[0])
        pe = torch.tensor(ns11['get_entropy'](loss, cw11))
        add = ns11['update_cache'](cache11, pred, [pcf, loss, pe, pm], shot_capacity=cap11, clip_weights=cw11, beta=150)
        fl = cl.clone() / 100.0
        p1 = torch.softmax(fl, dim=1); e1 = ns11['softmax_entropy'](p1, False)
        fl2, info = ns11['compute_cache_logits'](pcf, cache11, cw11, [add, Lo11, Li11], i, hp11)
        Lo11, Li11 = info[0], info[1]
        p2 = torch.softmax(fl2, dim=1); e2 = ns11['softmax_entropy'](p2, False)
        fin11.append((1 / e1).reshape(-1, 1) * p1 + (1 / e2).reshape(-1, 1) * p2); z211.append(fl2)
    keys11 = torch.cat([it[0] for c in sorted(cache11) for it in cache11[c]], 0)
    np.savez_compressed(os.path.join(OUT, 'g11_cluster_cache.npz'), text=text11.numpy(), feats=feats11.numpy(), cap=np.int64(cap11), beta=np.float64(150),
                        threshold=np.float64(0.5), lambda_reg=np.float64(0.11), finals=torch.cat(fin11).numpy(), cache_logits=torch.cat(z211).numpy(),
                        final_keys=keys11.numpy(), final_counts=np.array([it[3] for c in sorted(cache11) for it in cache11[c]], dtype=np.int64))
    print('g11 cluster cache', keys11.shape)
    extras()
    if '--skip-long' not in sys.argv:
        for st in LONG_STREAMS:
            long_streams(st)
    tot = sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT))
    print('golden bytes', tot)


if __name__ == '__main__':
    main()
