#!/usr/bin/env python
"""scenes/sec forward+backward of the FCAF3D sparse-voxel hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run)

One step = forward_train + loss.backward() + gradient all-reduce (N>1) + grad-clip + AdamW step over one
batch of `--batch` synthetic ScanNet-shaped scenes per GPU (100 000 points, 2 cm voxels, 18 classes,
fcaf3d_scannet-3d-18class topology: MEResNet3D-34, 4 levels).  Scenes are resident in HBM before the
timed region.  Prints ONE JSON line on rank 0 (contract in the task statement), including
  roofline     — the dominant kernel (MFMA gather-GEMM sparse conv), algorithmic FLOPs / HIP-event time
  cpu_baseline — the CPU oracle (ME-CPU-algorithm restatement, oracle/model_oracle.py) on this host.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=8, help='scenes per GPU per step (reference samples_per_gpu=8)')
    ap.add_argument('--workload', default='scannet-100k', choices=['plumbing-20k', 'scannet-100k', 'sunrgbd-100k', 's3dis-500k'])
    ap.add_argument('--voxel-size', type=float, default=0.02)
    ap.add_argument('--levels', type=int, default=4)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-points', type=int, default=0, help='points of the cpu_baseline sample scene (0 = same as workload)')
    ap.add_argument('--no-instrument', action='store_true', help='skip per-kernel HIP events (roofline = null)')
    ap.add_argument('--probe-every', type=int, default=4, help='HIP events bracket the conv launches of every n-th timed step '
                    '(each event pair is a pipeline bubble: sampling keeps the probe from slowing the thing it measures)')
    ap.add_argument('--spatial-sort', action='store_true', help='Z-order sort of the collated points (measured: no gain, r1)')
    ap.add_argument('--wgrad-overlap', action='store_true', help='weight-gradient kernels on a second stream (measured r1: no gain, the GPU is already full)')
    ap.add_argument('--breakdown', action='store_true', help='diagnostic: HIP-event time per C-ABI entry point and per conv shape (stderr)')
    return ap.parse_args()


CONFIG_OF = {'plumbing-20k': 'fcaf3d_scannet-3d-18class', 'scannet-100k': 'fcaf3d_scannet-3d-18class',
             'sunrgbd-100k': 'fcaf3d_sunrgbd-3d-10class', 's3dis-500k': 'fcaf3d_s3dis-3d-5class'}


def build_model(args):
    import fcaf3d_amd as fa
    cfg = fa.get_config(CONFIG_OF[args.workload], voxel_size=args.voxel_size)
    m = cfg.model
    if args.levels != 4:
        m.backbone['n_outs'] = args.levels
        m.neck_with_head['in_channels'] = (64, 128, 256, 512)[:args.levels]
        m.neck_with_head.assigner['n_scales'] = args.levels
    torch.manual_seed(0)
    model = fa.build_detector(m, train_cfg=m.get('train_cfg'), test_cfg=m.get('test_cfg'))
    return model, cfg


def make_batches(args, rank, dev, n_batches=2):
    import fcaf3d_amd as fa
    from fcaf3d_amd.synthetic import WORKLOADS, make_scene
    kw = WORKLOADS[args.workload]['scene']
    batches = []
    for j in range(n_batches):
        pts, gts, labs = [], [], []
        for i in range(args.batch):
            p, g, l = make_scene(1000 * rank + j * args.batch + i, **kw)
            pts.append(torch.from_numpy(p).to(dev))
            gts.append(fa.DepthInstance3DBoxes(torch.from_numpy(g), origin=(.5, .5, .5)).to(dev))
            labs.append(torch.from_numpy(l).to(dev))
        batches.append(dict(points=pts, gt_bboxes_3d=gts, gt_labels_3d=labs,
                            img_metas=[dict(box_type_3d=fa.DepthInstance3DBoxes)] * args.batch))
    return batches


class Breakdown:
    """Diagnostic only (perturbs the timing): events around EVERY C-ABI call, grouped by name / conv shape."""

    def __init__(self):
        self.rec = []

    def install(self):
        import fcaf3d_amd._lib as L
        orig = L.call
        rec = self.rec

        def call(name, *a):
            s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
            s.record(); orig(name, *a); e.record()
            key = name
            if name == 'fc_conv_fwd':
                key = f'conv_fwd n={a[6]} K={a[7]} {a[8]}->{a[9]}'
            elif name == 'fc_conv_wgrad':
                key = f'wgrad n={a[6]} K={a[7]} {a[8]}->{a[9]}'
            rec.append((key, name, s, e))
        L.call = call

    def report(self, steps, t_total_ms):
        torch.cuda.synchronize()
        by_key, by_name = {}, {}
        for key, name, s, e in self.rec:
            t = s.elapsed_time(e)
            by_key.setdefault(key, [0, 0.0]); by_key[key][0] += 1; by_key[key][1] += t
            by_name.setdefault(name, [0, 0.0]); by_name[name][0] += 1; by_name[name][1] += t
        print(f'--- breakdown per step ({steps} steps, wall {t_total_ms / steps:.2f} ms/step) ---', file=sys.stderr)
        tot = 0.0
        for name, (c, t) in sorted(by_name.items(), key=lambda kv: -kv[1][1]):
            print(f'{name:28s} calls/step {c / steps:7.1f}  ms/step {t / steps:8.3f}', file=sys.stderr)
            tot += t
        print(f'{"sum of C-ABI kernels":28s} {"":18s} ms/step {tot / steps:8.3f}', file=sys.stderr)
        print('--- top conv shapes ---', file=sys.stderr)
        for key, (c, t) in sorted(by_key.items(), key=lambda kv: -kv[1][1])[:40]:
            if key.startswith(('conv_fwd', 'wgrad')):
                print(f'{key:44s} calls/step {c / steps:6.1f}  ms/step {t / steps:8.3f}  us/call {t / c * 1e3:9.1f}', file=sys.stderr)


class ConvProbe:
    """HIP-event bracket around every MFMA sparse-conv launch (forward + backward-data) of the timed steps.
    The launches' algorithmic FLOPs (2 * valid pairs * Cin * Cout, pairs counted on the device from the map
    actually used) are collected in one extra, untimed step per distinct batch — the launch sequence of a
    step is deterministic — so that nothing but the two event records sits inside the timed region."""

    def __init__(self):
        self.timed = {}        # batch index -> list of per-step lists of (start, end)
        self.counted = {}      # batch index -> list of (pairs_dev | None, flops_per_pair, n_out, bytes)
        self.mode = None
        self._cur = None

    def begin_step(self, batch_index, mode):
        self.mode = mode
        if mode == 'time':
            self._cur = []
            self.timed.setdefault(batch_index, []).append(self._cur)
        else:
            self._cur = self.counted[batch_index] = []

    def install(self):
        import fcaf3d_amd._lib as L
        probe = self
        orig = L.call

        def call(name, *a):
            if name not in ('fc_conv_fwd', 'fc_conv_fwd_pairs') or probe.mode is None:
                return orig(name, *a)
            if name == 'fc_conv_fwd':
                # (in, W, nbr, out_index, out, n_in, n_out, K, Cin, Cout, flags, ws, ws_bytes, stream)
                n_in, n_out, K, Cin, Cout, has_map = a[5], a[6], a[7], a[8], a[9], bool(a[2])
            else:
                # (in, W, pair_in, pair_cnt, pair_pos, out, n_in, n_out, K, Cin, Cout, flags, ws, ws_bytes, stream)
                n_in, n_out, K, Cin, Cout, has_map = a[6], a[7], a[8], a[9], a[10], True
            if Cin % 32 or Cout % 64:
                return orig(name, *a)          # generic FMA / stem path: not the kernel under the probe
            if probe.mode == 'time':
                s = torch.cuda.Event(enable_timing=True)
                e = torch.cuda.Event(enable_timing=True)
                s.record()
                orig(name, *a)
                e.record()
                probe._cur.append((s, e))
            else:
                orig(name, *a)
                nbytes = 4.0 * (n_in * Cin + n_out * Cout + K * Cin * Cout) + (4.0 * K * n_out if has_map else 0.0)
                probe._cur.append((probe._pairs, 2.0 * Cin * Cout, n_out, nbytes))
        L.call = call
        self._pairs = None
        import fcaf3d_amd.functional as Fn
        fwd0, bwd0 = Fn._SparseConv.forward, Fn._SparseConv.backward

        def pairs_of(kmap):
            if kmap is None or probe.mode != 'count':
                return None
            if getattr(kmap, '_pairs_dev', None) is None:
                kmap._pairs_dev = (kmap.nbr >= 0).sum()
            return kmap._pairs_dev

        def fwd(ctx, feats, weight, kmap, n_out):
            probe._pairs = pairs_of(kmap)
            return fwd0(ctx, feats, weight, kmap, n_out)

        def bwd(ctx, gout):
            probe._pairs = pairs_of(ctx.kmap)
            return bwd0(ctx, gout)
        Fn._SparseConv.forward = staticmethod(fwd)
        Fn._SparseConv.backward = staticmethod(bwd)

    def summary(self):
        if not self.timed:
            return None
        torch.cuda.synchronize()
        flops = ms = alg_bytes = 0.0
        n = 0
        for bi, steps in self.timed.items():
            counted = self.counted[bi]
            per_launch = [((float(p.item()) if p is not None else float(n_out)) * fpp, nb) for p, fpp, n_out, nb in counted]
            for ev in steps:
                assert len(ev) == len(per_launch), 'launch sequence of a step is expected to be deterministic'
                for (s, e), (f, nb) in zip(ev, per_launch):
                    ms += s.elapsed_time(e)
                    flops += f
                    alg_bytes += nb
                    n += 1
        achieved = flops / (ms * 1e-3) / 1e12
        traffic, traffic_src = None, None
        tj = os.path.join(ROOT, 'profiles', 'r1_traffic.json')
        if os.path.exists(tj):          # PMC passes cannot run inside the timed region: committed rocprofv3 result
            t = json.load(open(tj))
            traffic, traffic_src = t['hbm_bytes_per_launch'], 'profiles/r1_traffic.json (' + t['method'] + ')'
        return dict(bound='mfma', kernel='k_conv_mfma (sparse conv fwd + dgrad, dense GEMMs of convT/heads)',
                    achieved=round(achieved, 3), peak=PEAK_F32_MFMA_TFLOPS, unit='TFLOP/s',
                    frac=round(achieved / PEAK_F32_MFMA_TFLOPS, 4), traffic=traffic, traffic_unit='B/launch',
                    traffic_source=traffic_src, algorithmic_bytes_per_launch=round(alg_bytes / n), launches=n,
                    avg_launch_us=round(ms * 1e3 / n, 2), flops_per_launch=round(flops / n / 1e9, 4),
                    time_share_ms_per_step=None)


def cpu_baseline(args, model, cfg):
    """The oracle (ME-CPU-algorithm restatement) forward+backward on ONE scene of the same workload."""
    from fcaf3d_amd.synthetic import WORKLOADS, make_scene
    from oracle import model_oracle as MO
    kw = dict(WORKLOADS[args.workload]['scene'])
    if args.cpu_points:
        kw['n_points'] = args.cpu_points
    cores = min(os.cpu_count() or 1, 64)
    torch.set_num_threads(cores)
    P = {k: v.detach().cpu().clone().requires_grad_(v.dtype.is_floating_point) for k, v in model.state_dict().items()}
    p, g, l = make_scene(999, **kw)
    t0 = time.time()
    losses = MO.forward_train(P, cfg.model, [p], [g], [l])
    sum(losses.values()).backward()
    dt = time.time() - t0
    return dict(value=round(1.0 / dt, 5), unit='scenes/s', cores=cores, kind='port',
                sample=f'1 scene of {kw["n_points"]} pts, fwd+bwd once ({dt:.1f} s), oracle/model_oracle.py '
                       f'(ME-CPU-algorithm restatement: hash kernel maps + per-offset gather-GEMM-scatter, torch CPU fp32)')


def count_steps(args, n_batches):
    """number of untimed FLOP-count steps after the timed region — a function of the flags only, never of the rank"""
    return 0 if (args.no_instrument or args.breakdown) else n_batches      # every distinct batch a probed step may use


def main():
    args = parse()
    if os.environ.get('FC_FAULT_DUMP'):           # debugging aid: dump every thread's stack after n seconds and exit
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ['FC_FAULT_DUMP']), exit=True)
    from fcaf3d_amd import dist as D
    D.init_dist()
    rank = int(os.environ.get('RANK', '0'))
    world = D.world_size()
    local = int(os.environ.get('LOCAL_RANK', '0'))
    assert torch.cuda.is_available(), 'bench.py needs the MI355X (there is no CPU fallback for the product path)'
    if os.environ.get('FC_DIST_BACKEND') == 'gloo':
        local %= torch.cuda.device_count()       # smoke mode only: several ranks may share one GPU
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)

    model, cfg = build_model(args)
    model = model.to(dev).train()
    model.async_maps = True           # scenes are resident in HBM: coordinate work may run on its side stream
    model.spatial_sort = args.spatial_sort
    import fcaf3d_amd.functional as Fn
    Fn.WGRAD_ASYNC = args.wgrad_overlap
    if world > 1:
        for p in model.parameters():
            torch.distributed.broadcast(p.data, 0)
    averager = D.GradientAverager(model.parameters())
    opt = torch.optim.AdamW(model.parameters(), lr=cfg.optimizer.lr, weight_decay=cfg.optimizer.weight_decay, fused=True)
    max_norm = cfg.optimizer_config.grad_clip.max_norm
    batches = make_batches(args, rank, dev)

    probe = None
    bd = None
    if args.breakdown:
        args.no_instrument = True                # on every rank: the untimed count steps below hold collectives
    if args.breakdown and rank == 0:
        bd = Breakdown()
        bd.install()
    if not args.no_instrument and rank == 0:
        probe = ConvProbe()
        probe.install()

    def step(i, probe_mode=None):
        batch = batches[i % len(batches)]
        if probe:
            if probe_mode:
                probe.begin_step(i % len(batches), probe_mode)
            else:
                probe.mode = None
        opt.zero_grad(set_to_none=True)
        losses = model(return_loss=True, **batch)
        loss = losses['loss_centerness'] + losses['loss_bbox'] + losses['loss_cls']
        loss.backward()
        averager.finish()
        torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm)
        opt.step()
        return loss

    for i in range(args.warmup):
        step(i)
    if bd:
        torch.cuda.synchronize()
        bd.rec.clear()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = step(args.warmup + i, 'time' if i % max(args.probe_every, 1) == 0 else None)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    final_loss = float(loss.item())
    # untimed: FLOPs of every launch, one step per distinct batch.  EVERY rank steps (a step holds collectives: a
    # rank-0-only extra step deadlocks the job); only rank 0 carries the probe
    for b in range(count_steps(args, len(batches))):
        step(b, 'count' if probe else None)
    if bd:
        bd.report(args.steps, dt * 1e3)
    assert np.isfinite(final_loss), 'loss diverged'

    if rank == 0:
        scenes = args.batch * world * args.steps
        out = {
            'metric': 'scenes/sec fwd+bwd, ScanNet 100k-pt 2cm voxels, 1/2/4/8 MI355X',
            'value': round(scenes / dt, 3), 'unit': 'scenes/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 3), 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': f'{args.workload} synthetic, {CONFIG_OF[args.workload]}.py topology '
                                   f'(MEResNet3D-34, {args.levels} levels), voxel {args.voxel_size} m',
                       'scenes_per_gpu_per_step': args.batch, 'global_batch': args.batch * world,
                       'step': 'forward_train + backward + grad all-reduce + grad-clip + AdamW',
                       'parallelism': f'dp{world}', 'final_loss': round(final_loss, 4)},
        }
        rl = probe.summary() if probe else None
        if rl:
            probed_steps = len(range(0, args.steps, max(args.probe_every, 1)))
            conv_ms = rl['avg_launch_us'] * rl['launches'] / 1e3 / probed_steps
            rl['probed_steps'] = probed_steps
            rl['time_share_ms_per_step'] = round(conv_ms, 3)
        out['roofline'] = rl
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(args, model, cfg)
        else:
            out['cpu_baseline'] = None
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
