"""What one rank of a W-GPU run looks like, measured on ONE GPU: per step the encoder handles this rank's
31 samples while the in-order cache replay covers all W*31 samples of the step (features of the other
ranks are stand-ins: this rank's features repeated).  Reports ms/step and the implied whole-job
point-clouds/s = W*31 / step time.   python tools/emulate_scaling.py [--world 1,2,4,8] [--priority 0|-1]
Profiling aid for the replicated-replay serial fraction."""
import argparse, ctypes, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import uni_adapter_amd  # noqa: E402,F401
from uni_adapter_amd import lib, synthetic, point_encoder as PE  # noqa: E402
from uni_adapter_amd.Uni_Adapter import TtaEngine  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--world', default='1,2,4,8')
    ap.add_argument('--priority', type=int, default=0)
    ap.add_argument('--steps', type=int, default=6)
    ap.add_argument('--chunk', type=int, default=31)
    ap.add_argument('--mode', default='replicated', choices=['replicated', 'owner', 'delta', 'class'])
    ap.add_argument('--rho', type=float, default=None)
    ap.add_argument('--classes', type=int, default=0, help='synthetic text bank of this many classes instead of the ModelNet40 bank')
    ap.add_argument('--modes', type=int, default=8)
    ap.add_argument('--no-res', action='store_true', help='no residual learning (required for K > 160 and for --mode class)')
    a = ap.parse_args()
    dev = torch.device('cuda')
    gpath = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'g2_stream_modelnet40.npz')
    text = torch.from_numpy(np.load(gpath)['text']) if not a.classes else synthetic.text_bank(a.classes, 1024, 1)
    res = not a.no_res
    sd = synthetic.uni3d_state_dict(0, trans_dim=1024, embed_dim=1024, enc_dim=512, depth=24, device=dev)
    model = PE.create_uni3d_from_state_dict(sd, num_group=512, group_size=64, num_heads=16, device=dev, precision='split')
    n = a.chunk
    from uni_adapter_amd.stream import REPLAY_COST_RATIO, shard_counts
    rho = a.rho if a.rho is not None else REPLAY_COST_RATIO
    if a.mode == 'owner':
        # rank 0: x0 samples through the encoder + the replay of all W*n samples; an encoder rank: its share, no replay
        for W in [int(w) for w in a.world.split(',')]:
            counts = shard_counts(n * W, W, rho)
            times = []
            for role, n_enc, n_rep in (('rank 0', counts[0], n * W), ('encoder rank', max(counts[1:]) if W > 1 else 0, 0)):
                if W == 1 and role != 'rank 0':
                    continue
                eng = TtaEngine(model, text, {'epsilon': 1e-4, 'sigma': 1e-4, 'eta': 0.1, 'rho': 0.02}, 8, res_learning=True, chunk=n * W, device=dev)
                rs = torch.cuda.Stream(device=dev)
                total = a.steps + 2
                xyz, rgb = synthetic.point_clouds(max(1, n_enc) * total, 1024, seed=42, device=dev)
                noise = torch.randn(xyz.shape, device=dev, generator=torch.Generator(device=dev).manual_seed(7))
                final = torch.empty(max(1, n_rep), eng.K, device=dev); preds = torch.empty(max(1, n_rep), dtype=torch.int32, device=dev)
                stand_in = torch.nn.functional.normalize(torch.randn(max(1, n_rep), 2, eng.D, device=dev), dim=-1)
                main_s = torch.cuda.current_stream(); i_glob, adam = 1, 0
                for it in range(total):
                    if it == 2:
                        torch.cuda.synchronize(); t0 = time.perf_counter()
                    if n_enc:
                        s_ = slice(it * n_enc, (it + 1) * n_enc)
                        feats = eng.encode_chunk(xyz[s_], rgb[s_], noise[s_])
                    if n_rep:
                        rs.wait_stream(main_s)
                        with torch.cuda.stream(rs):
                            lib.call('ua_tta_chunk', ctypes.byref(eng._state), lib.ptr(stand_in), None, n_rep, i_glob, adam, lib.ptr(final), None,
                                     lib.ptr(preds, torch.int32), rs.cuda_stream)
                        i_glob += n_rep; adam += 10 * n_rep
                torch.cuda.synchronize()
                times.append((role, n_enc, n_rep, (time.perf_counter() - t0) / a.steps))
                del eng
            step = max(t[3] for t in times)
            print(f'W={W} owner mode (rho {rho}): ' + '; '.join(f'{r}: {ne} encoded + {nr} replayed = {t * 1e3:.1f} ms' for r, ne, nr, t in times) +
                  f' -> {n * W / step:8.1f} point-clouds/s whole job ({n * W / step / W:6.1f} per GPU)', flush=True)
        return
    if a.mode == 'delta':
        # one rank of a W-rank 'delta' job: encode + replay its own n samples, pack, merge W buffers (stand-ins for the
        # all_gather result: this rank's buffer W times - the copy costs what receiving them would write; wire time excluded)
        for W in [int(w) for w in a.world.split(',')]:
            eng = TtaEngine(model, text, {'epsilon': 1e-4, 'sigma': 1e-4, 'eta': 0.1, 'rho': 0.02}, 8, res_learning=True, chunk=n * W, device=dev,
                            replay_mode='delta', emulated_rank=(0, max(W, 2)), overlap=True)
            Wm = eng.world
            total = a.steps + 2
            xyz, rgb = synthetic.point_clouds(n * total, 1024, seed=42, device=dev)
            noise = torch.randn(xyz.shape, device=dev, generator=torch.Generator(device=dev).manual_seed(7))
            for it in range(total):
                if it == 2:
                    torch.cuda.synchronize(); t0 = time.perf_counter()
                s_ = slice(it * n, (it + 1) * n)
                rep = lambda t: t[s_].repeat(Wm, 1, 1)                     # rank 0 takes the first n of the n*W-sample step
                feats = eng.encode_chunk(rep(xyz), rep(rgb), rep(noise))
                eng.delta_step(feats, None, n * Wm, exchange=lambda b: b[None].repeat(Wm, 1))
            buf = eng._delta['buf']
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / a.steps
            print(f'W={W} delta mode: {dt * 1e3:7.1f} ms/step (merge of {Wm} buffers of {buf.numel() * 4 / 1e6:.2f} MB) -> '
                  f'{n * W / dt:8.1f} point-clouds/s whole job ({n / dt:6.1f} per GPU)', flush=True)
            del eng
        return
    if a.mode == 'class':
        # one rank of a W-rank 'class' job: encode its n samples, then for the n*W-sample step its class slice of
        # the cache step + the chunk's fusion from W stand-in buffers (its own, repeated; wire time of the one all_gather excluded)
        for W in [int(w) for w in a.world.split(',')]:
            eng = TtaEngine(model, text, {'epsilon': 1e-4, 'sigma': 1e-4, 'eta': 0.1, 'rho': 0.02}, a.modes, res_learning=False, chunk=n * W,
                            device=dev, replay_mode='class', emulated_rank=(0, max(W, 2)), overlap=True)
            Wm = eng.world
            total = a.steps + 2
            xyz, rgb = synthetic.point_clouds(n * total, 1024, seed=42, device=dev)
            noise = torch.randn(xyz.shape, device=dev, generator=torch.Generator(device=dev).manual_seed(7))
            final = torch.empty(n * W, eng.K, device=dev); preds = torch.empty(n * W, dtype=torch.int32, device=dev)
            main_s, rs = torch.cuda.current_stream(), eng.replay_stream
            eng.world = 1                                    # encode_chunk: all n samples are this rank's share
            for it in range(total):
                if it == 2:
                    torch.cuda.synchronize(); t0 = time.perf_counter()
                s_ = slice(it * n, (it + 1) * n)
                feats = eng.encode_chunk(xyz[s_], rgb[s_], noise[s_]).repeat(W, 1, 1).contiguous()
                rs.wait_stream(main_s); feats.record_stream(rs)
                eng.world = Wm
                with torch.cuda.stream(rs):
                    buf = eng.class_local_chunk(feats)
                    eng.class_fuse_chunk(buf[None].repeat(Wm, 1, 1), None, final, None, preds)
                eng.world = 1
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / a.steps
            print(f'W={W} class mode (K={eng.K}, M={eng.M}, {eng.class_k1 - eng.class_k0} classes on this rank): {dt * 1e3:7.1f} ms/step -> '
                  f'{n * W / dt:8.1f} point-clouds/s whole job ({n / dt:6.1f} per GPU)', flush=True)
            del eng
        return
    for W in [int(w) for w in a.world.split(',')]:
        eng = TtaEngine(model, text, {'epsilon': 1e-4, 'sigma': 1e-4, 'eta': 0.1, 'rho': 0.02}, a.modes, res_learning=res, chunk=n * W, device=dev)
        rs = torch.cuda.Stream(device=dev, priority=a.priority)
        total = a.steps + 2
        xyz, rgb = synthetic.point_clouds(n * total, 1024, seed=42, device=dev)
        noise = torch.randn(xyz.shape, device=dev, generator=torch.Generator(device=dev).manual_seed(7))
        final = torch.empty(n * W, eng.K, device=dev); preds = torch.empty(n * W, dtype=torch.int32, device=dev)
        main_s = torch.cuda.current_stream()
        i_glob, adam = 1, 0
        t0 = None
        for it in range(total):
            if it == 2:
                torch.cuda.synchronize(); t0 = time.perf_counter()
            s = slice(it * n, (it + 1) * n)
            feats = eng.encode_chunk(xyz[s], rgb[s], noise[s])                        # this rank's share (world = 1 engine)
            feats = feats.repeat(W, 1, 1).contiguous()                                 # stand-in for the all_gather result
            rs.wait_stream(main_s)
            feats.record_stream(rs)
            with torch.cuda.stream(rs):
                lib.call('ua_tta_chunk', ctypes.byref(eng._state), lib.ptr(feats), None, n * W, i_glob, adam, lib.ptr(final), None,
                         lib.ptr(preds, torch.int32), rs.cuda_stream)
            i_glob += n * W; adam += 10 * n * W
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.steps
        print(f'W={W}: {dt * 1e3:7.1f} ms/step -> {n * W / dt:8.1f} point-clouds/s whole job ({n * W / dt / W:6.1f} per GPU), replay priority {a.priority}', flush=True)
        del eng


if __name__ == '__main__':
    main()
