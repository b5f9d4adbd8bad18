#!/usr/bin/env python
"""scenes/sec forward+backward of the FCAF3D sparse-voxel hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run)

One step = forward_train + loss.backward() + gradient all-reduce (N>1) + grad-clip + AdamW step over one
batch of `--batch` synthetic ScanNet-shaped scenes per GPU (100 000 points, 2 cm voxels, 18 classes,
fcaf3d_scannet-3d-18class topology: MEResNet3D-34, 4 levels).  Scenes are resident in HBM before the
timed region.  Prints ONE JSON line on rank 0 (contract in the task statement), including
  roofline     — the dominant kernel (MFMA gather-GEMM sparse conv), algorithmic FLOPs / HIP-event time, and
                 `hbm_kernels`: GB/s of every bandwidth-bound kernel (compulsory bytes / HIP-event time) vs the 8 TB/s peak
  cpu_baseline — MinkowskiEngine's CPU algorithm restated in C/OpenMP (oracle/conv_oracle.c) on this host's cores,
                 with the Python/torch oracle's whole-step number beside it
  config.{fwd_bwd_only, config4_global_batch_16, inference, data_parallel} — measured outside the timed region.
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
PEAK_X6_TFLOPS = round(2500.0 / 6, 1)   # fp32 products as 6 bf16 MFMA products each (csrc/conv_x6.h): bf16 dense peak / 6
PEAK_H3_TFLOPS = round(2500.0 / 3, 1)   # r6: as 3 fp16 MFMA products each (csrc/conv_x6.h "h3"): fp16 dense peak (= bf16's) / 3


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--settle-seconds', type=float, default=45.0,
                    help='after the warm-up steps: further UNTIMED steps (windows of 4) for at most this long while the host, not the GPU, '
                         'sets the step time (a guard: since the allocator reserve and the warmed probed step — profiles/r6_notes.md section 14 — no run needed more than the one window that always runs); 0 = off.  Reported in config.host.settle')
    ap.add_argument('--batch', type=int, default=8, help='scenes per GPU per step (reference samples_per_gpu=8)')
    ap.add_argument('--workload', default='scannet-100k', choices=['plumbing-20k', 'scannet-100k', 'sunrgbd-100k', 's3dis-500k'])
    ap.add_argument('--voxel-size', type=float, default=0.02)
    ap.add_argument('--levels', type=int, default=4)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-points', type=int, default=0, help='points of the cpu_baseline sample scene (0 = same as workload)')
    ap.add_argument('--reserve-gb', type=float, default=64.0, help='device memory taken into the caching allocator before the first step (runner.reserve_device_memory); 0: none')
    ap.add_argument('--sample-main-thread', action='store_true', help='tools/wchan.py over the timed region: what the main thread did when it was not running (config.host.steps.main_thread)')
    ap.add_argument('--no-instrument', action='store_true', help='skip per-kernel HIP events (roofline = null)')
    ap.add_argument('--probe-every', type=int, default=20, help='HIP events bracket the conv launches of every n-th timed step '
                    '(each event pair is a pipeline bubble: sampling keeps the probe from slowing the thing it measures)')
    ap.add_argument('--spatial-sort', action='store_true', help='Z-order sort of the collated points (measured: no gain, r1)')
    ap.add_argument('--no-wgrad-overlap', action='store_true', help='keep the weight-gradient kernels on the main stream (default: a second '
                    'stream, so that they overlap the backward-data chain: +3 % measured r2)')
    ap.add_argument('--no-priority-stream', dest='priority_stream', action='store_false',
                    help='default: the step\'s dependent chain runs on a high-priority HIP stream (r3, beside the split kernels: 23.41 -> 23.25 ms)')
    ap.add_argument('--infer-steps', type=int, default=8, help='untimed-region extra: simple_test batches for the inference scenes/s line (0 = skip)')
    ap.add_argument('--cpu-reps', type=int, default=3, help='repetitions of the conv-only part of the cpu_baseline (median)')
    ap.add_argument('--no-fp32-route', action='store_true', help='skip the untimed FC_X6=0 extra')
    ap.add_argument('--no-force-dp', action='store_true', help='skip the untimed N=1-through-the-averager extra')
    ap.add_argument('--no-executor', action='store_true', help='per-operator module path for every step (default: the network body through the '
                    'native launch-list executor, fcaf3d_amd/executor.py; FC_EXEC=0 does the same)')
    ap.add_argument('--breakdown', action='store_true', help='diagnostic: HIP-event time per C-ABI entry point and per conv shape (stderr)')
    ap.add_argument('--no-extras', action='store_true', help='skip the untimed BASELINE.md section 2 "also report" configurations '
                    '(literal 1 cm, two scales, SUN RGB-D, S3DIS: 5 steps each, N = 1 only) and the bf16 fast-mode number')
    ap.add_argument('--extra-steps', type=int, default=5)
    ap.add_argument('--no-lookahead', action='store_true', help='plan every step in line (default: the next batch\'s coordinate phase runs on a '
                    'worker thread beside the current step, fcaf3d_amd/plan.py Lookahead)')
    return ap.parse_args()


# BASELINE.md section 2 "also report": the reference's other FCAF3D configurations, as flagged extras of the default line
# (configs/fcaf3d/fcaf3d.py:1 literal 1 cm voxels, fcaf3d_2scales_scannet-3d-18class.py:2, fcaf3d_sunrgbd-3d-10class.py,
# fcaf3d_s3dis-3d-5class.py).  key -> (workload, voxel size, levels, scenes per step)
EXTRAS = {'literal_1cm': ('scannet-100k', 0.01, 4, 4), 'two_scales': ('scannet-100k', 0.02, 2, 8),
          'sunrgbd': ('sunrgbd-100k', 0.02, 4, 8), 's3dis': ('s3dis-500k', 0.02, 4, 2)}


def run_extra(args, key, dev, steps):
    """one BASELINE.md 'also report' configuration: its own model, scenes and TrainStep; `steps` timed steps after 2 warm-ups,
    with the stream configuration of the main line (nothing of it is probed)"""
    import copy
    import fcaf3d_amd.functional as Fn
    from fcaf3d_amd.runner import TrainStep
    wl, vs, lv, bs = EXTRAS[key]
    a = copy.copy(args)
    a.workload, a.voxel_size, a.levels, a.batch = wl, vs, lv, bs
    model, cfg = build_model(a)
    model = model.to(dev).train()
    model.async_maps = True
    model.inputs_resident = True
    Fn.WGRAD_ASYNC = not args.no_wgrad_overlap
    tr = TrainStep.from_config(model, cfg)
    batches = make_batches(a, 0, dev)
    for i in range(4):                  # warm-up with the timed region's own pattern (lookahead: two plans' arenas alive): the allocator's pools fill here
        tr(batches[i % 2], batches[(i + 1) % 2] if (i < 3 and not args.no_lookahead) else None)
    dt, loss = timed_region(lambda i: tr(batches[i % 2], batches[(i + 1) % 2] if (i < steps - 1 and not args.no_lookahead) else None)[0], steps, 1, dev)
    out = dict(workload=f'{wl}, voxel {vs} m, {lv} levels', scenes_per_step=bs, steps=steps, ms_per_step=round(dt / steps * 1e3, 3),
               value=round(bs * steps / dt, 3), unit='scenes/s', final_loss=round(float(loss), 4),
               host_enqueue_ms_per_step=round(LAST_HOST_S / steps * 1e3, 3), host_busy_ms_per_step=round((LAST_HOST_S - LAST_WAIT_S) / steps * 1e3, 3),
               calls=LAST_STEPS)
    del tr, model, batches             # (no empty_cache(): the allocator keeps what it holds — see --reserve-gb)
    return out


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


# Compulsory HBM bytes (every operand tensor touched once) of the bandwidth-bound entry points, from their C-ABI arguments
# (include/fcaf3d_hip.h).  `a` = positional arguments, `pr` = the probe (carries sizes the ABI call does not).
def _stem(a):                   # fc_conv_fwd / fc_conv_wgrad with Cin == 3: (.., n_in, n_out, K, Cin, Cout, ..) at 5..9
    return a[8] == 3


HBM_BYTES = {
    'fc_conv_fwd': lambda a, pr: (4.0 * (a[5] * 3 + a[6] * a[9] + a[7] * 3 * a[9]) + 4.0 * a[7] * a[6]) if _stem(a) else None,
    'fc_conv_wgrad': lambda a, pr: (4.0 * (a[5] * 3 + a[6] * a[9]) + 4.0 * a[7] * a[6]) if _stem(a) else None,
    # (in, W, nbr, out, col, n_in, n_out, K): inputs + table read, out + col (84 floats per row) written
    'fc_stem_conv_fwd': lambda a, pr: 4.0 * (a[5] * 3 + a[6] * 64 + a[7] * 3 * 64 + (a[6] * 84 if a[4] else 0)) + 4.0 * a[7] * a[6],
    # (col, gout, gW, n_out, K, ..): col and gout streamed once
    'fc_stem_conv_wgrad': lambda a, pr: 4.0 * a[3] * (84 + 64),
    # (in, nbr, n_out, K, C, out, argrow): every input row read once (k2s2: each voxel has one parent), out + argmax written
    'fc_maxpool_fwd': lambda a, pr: 4.0 * a[4] * (pr.pool_n_in + 2 * a[2]) + 4.0 * a[3] * a[2],
    'fc_maxpool_bwd': lambda a, pr: 4.0 * a[3] * 3 * a[2],
    # (x, seg, seg_stride, n, C, mean, var, eps, gamma, beta, residual, act, y)
    'fc_norm_act_fwd': lambda a, pr: 4.0 * a[3] * a[4] * (2 + (1 if a[10] else 0)),
    # (x, y, gy, seg, seg_stride, n, C, nseg, mean, var, cnt, eps, gamma, beta, act, gx, gres, ..): x, gy (and y when the forward
    # had a residual) read; gx (gres) written
    'fc_norm_act_bwd': lambda a, pr: 4.0 * a[5] * a[6] * (3 + (1 if a[1] else 0) + (1 if a[16] else 0)),
    'fc_col_stats': lambda a, pr: 4.0 * a[3] * a[4],
    'fc_bn_stats_train': lambda a, pr: 4.0 * a[1] * a[2],
    # (x, n, C, eps, gamma, beta, residual, act, ..): x read, y written (+ residual)
    'fc_bn_act_train_fwd': lambda a, pr: 4.0 * a[1] * a[2] * (2 + (1 if a[6] else 0)),
    # (x, y, gy, n, C, mean, var, eps, gamma, beta, act, gx, gres, ..)
    'fc_bn_act_train_bwd': lambda a, pr: 4.0 * a[3] * a[4] * (3 + (1 if a[1] else 0) + (1 if a[12] else 0)),
    # r5 (x, n, C, eps, gamma, beta, residual, act, momentum, y, mean, var, cnt, rmean, rvar, nbt, part, ..): with a statistics table
    # from the producing convolution x is read once and y written; without one x is read twice
    'fc_bn_train_fwd': lambda a, pr: 4.0 * a[1] * a[2] * ((2 if a[16] else 3) + (1 if a[6] else 0)),
    # (x, y, gy, gy2, n, C, mean, var, cnt, eps, gamma, beta, act, gx, gres, ..)
    'fc_bn_train_bwd': lambda a, pr: 4.0 * a[4] * a[5] * (3 + (1 if a[1] else 0) + (1 if a[3] else 0) + (1 if a[14] else 0)),
    # (coords, n, q, keys, vals, cap, out_coords, ..): 16 B coordinate read + 16 B key/value insert per row
    'fc_hash_unique': lambda a, pr: 32.0 * a[1],
    # (out_coords, n_out, keys, vals, cap, offsets, K, nbr): 16 B coordinate + K x (12 B probe + 4 B write) per row
    'fc_kernel_map': lambda a, pr: a[1] * (16.0 + 16.0 * a[6]),
    'fc_gather_rows': lambda a, pr: 4.0 * a[2] * (2 * a[3] + 1),
    'fc_scatter_rows_add': lambda a, pr: 4.0 * a[2] * (3 * a[3] + 1),
}
PEAK_HBM_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable on a float4 copy)


class ConvProbe:
    """HIP-event bracket around every MFMA sparse-conv launch (forward + backward-data) of the timed steps.
    The launches' algorithmic FLOPs (2 * valid pairs * Cin * Cout, pairs counted on the device from the map
    actually used) are collected in one extra, untimed step per distinct batch — the launch sequence of a
    step is deterministic — so that nothing but the two event records sits inside the timed region."""

    def __init__(self):
        self.hbm = []          # (kernel, compulsory bytes, start, end) of the bandwidth-bound launches of the probed steps
        self.hbm_native = []   # (kernel, compulsory bytes, ms): brackets taken inside the native plan calls
        self.pool_n_in = 0
        self.timed = {}        # batch index -> list of per-step lists of (start, end)
        self.counted = {}      # batch index -> list of (pairs_dev | None, flops_per_pair, n_out, bytes)
        self.shapes = {}       # batch index -> list of (entry point, n_in, n_out, K, Cin, Cout, has map)
        self.mode = None
        self._cur = None
        self.exec_steps = []   # executor.PROBE entries of the probed steps that went through the native executor

    def begin_step(self, batch_index, mode):
        self.mode = mode
        self._bi = batch_index
        if mode == 'hbm':
            self._cur = None
            return
        if mode != 'time':
            self.shapes[batch_index] = []
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
            full_name = name
            if name in ('fc_conv_fwd_stats', 'fc_conv_fwd_pairs_tiles_stats'):
                name = name[:-6]          # (r5: the same launch + the statistics epilogue for the BatchNorm behind it; same argument positions)
            return call_(name, full_name, *a)

        def call_(name, full_name, *a):
            orig_ = orig
            if probe.mode in ('time', 'hbm') and name in HBM_BYTES:
                try:
                    nbytes = HBM_BYTES[name](a, probe)
                except Exception:
                    nbytes = None
                if nbytes:
                    s = torch.cuda.Event(enable_timing=True)
                    e = torch.cuda.Event(enable_timing=True)
                    s.record()
                    orig_(full_name, *a)
                    e.record()
                    probe.hbm.append((name if not (name in ('fc_conv_fwd', 'fc_conv_wgrad') and a[8] == 3) else name + '(stem)', nbytes, s, e))
                    return
            if name not in ('fc_conv_fwd', 'fc_conv_fwd_pairs', 'fc_conv_fwd_pairs_tiles') or probe.mode in (None, 'hbm'):
                return orig_(full_name, *a)
            if name == 'fc_conv_fwd':
                # (in, W, nbr, out_index, out, n_in, n_out, K, Cin, Cout, flags, ws, ws_bytes, stream)
                n_in, n_out, K, Cin, Cout, has_map = a[5], a[6], a[7], a[8], a[9], bool(a[2])
            else:
                # (in, W, pair_in, pair_cnt, pair_pos, out, n_in, n_out, K, Cin, Cout, [live_tiles,] flags, ws, ws_bytes, stream)
                n_in, n_out, K, Cin, Cout, has_map = a[6], a[7], a[8], a[9], a[10], True
            if Cin % 32 or Cout % 64:
                return orig_(full_name, *a)          # generic FMA / stem path: not the kernel under the probe
            if probe.mode == 'time':
                s = torch.cuda.Event(enable_timing=True)
                e = torch.cuda.Event(enable_timing=True)
                s.record()
                orig_(full_name, *a)
                e.record()
                probe._cur.append((s, e))
            else:
                orig_(full_name, *a)
                nbytes = 4.0 * (n_in * Cin + n_out * Cout + K * Cin * Cout) + (4.0 * K * n_out if has_map else 0.0)
                probe._cur.append((probe._pairs, 2.0 * Cin * Cout, n_out, nbytes))
                probe.shapes.setdefault(probe._bi, []).append((name, n_in, n_out, K, Cin, Cout, int(has_map)))
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

        def fwd(ctx, feats, weight, kmap, n_out, *rest):
            probe._pairs = pairs_of(kmap)
            return fwd0(ctx, feats, weight, kmap, n_out, *rest)

        def bwd(ctx, gout, *rest):
            probe._pairs = pairs_of(ctx.kmap)
            return bwd0(ctx, gout, *rest)
        Fn._SparseConv.forward = staticmethod(fwd)
        Fn._SparseConv.backward = staticmethod(bwd)
        pool0 = Fn._MaxPool.forward

        def pool(ctx, feats, kmap):
            probe.pool_n_in = feats.shape[0]
            return pool0(ctx, feats, kmap)
        Fn._MaxPool.forward = staticmethod(pool)

    def hbm_summary(self):
        """per bandwidth-bound kernel: compulsory bytes / HIP-event time over the probed steps"""
        torch.cuda.synchronize()
        agg = {}
        for name, nbytes, s, e in self.hbm:
            d = agg.setdefault(name, [0, 0.0, 0.0])
            d[0] += 1
            d[1] += nbytes
            d[2] += s.elapsed_time(e)
        for name, nbytes, ms in self.hbm_native:
            d = agg.setdefault(name, [0, 0.0, 0.0])
            d[0] += 1
            d[1] += nbytes
            d[2] += ms
        rows = []
        for name, (n, b, ms) in sorted(agg.items(), key=lambda kv: -kv[1][2]):
            gbs = b / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
            rows.append(dict(kernel=name, launches=n, algorithmic_MB_per_launch=round(b / n / 1e6, 3),
                             avg_us=round(ms * 1e3 / n, 2), GBps=round(gbs, 1), frac_of_hbm_peak=round(gbs / PEAK_HBM_GBS, 4)))
        return rows

    def exec_records(self):
        """probed steps that went through the native executor: (FLOPs, bytes, ms) per convolution launch from the event brackets
        inside fc_exec (csrc/exec.hip) and the pair counts bound with each step"""
        import ctypes
        import fcaf3d_amd._lib as L
        torch.cuda.synchronize()
        cap = sum(e['nf'] + e['nb'] + e.get('na', 0) for e in self.exec_steps) + 16
        ms = (ctypes.c_float * cap)()
        meta = (ctypes.c_int64 * (8 * cap))()
        n = L.lib().fc_exec_probe_read(ctypes.cast(ms, ctypes.c_void_p), ctypes.cast(meta, ctypes.c_void_p), cap)
        assert n == cap - 16, (n, cap)
        out, i = [], 0
        for e in self.exec_steps:
            pairs = {k: float(v.item()) for k, v in e['pairs'].items()}
            for _ in range(e['nf'] + e['nb'] + e.get('na', 0)):
                mp, _dir, n_in, n_out, K, Cin, Cout, _pm = (int(meta[8 * i + j]) for j in range(8))
                if mp == -2:         # the amax pass of a convolution operand (h3 split): time of the operator family, no FLOPs, no launch of its own
                    out.append((0.0, 0.0, float(ms[i]), True))
                    i += 1
                    continue
                P = pairs[mp] if mp >= 0 else float(n_out)
                nbytes = 4.0 * (n_in * Cin + n_out * Cout + K * Cin * Cout) + (4.0 * K * n_out if mp >= 0 else 0.0)
                out.append((2.0 * P * Cin * Cout, nbytes, float(ms[i]), False))
                i += 1
        return out

    def summary(self):
        if not any(ev for st_ in self.timed.values() for ev in st_) and not self.exec_steps:
            return None
        if self.exec_steps and not any(e['nf'] for e in self.exec_steps):
            self.exec_steps = []
        torch.cuda.synchronize()
        flops = ms = alg_bytes = amax_ms = 0.0
        n = 0
        if self.exec_steps:
            for f, nb, t, is_amax in self.exec_records():
                flops += f
                alg_bytes += nb
                ms += t
                n += 0 if is_amax else 1
                amax_ms += t if is_amax else 0.0
        for bi, steps in self.timed.items():
            steps = [ev for ev in steps if ev]             # (steps that went through the executor left no per-call brackets)
            if not steps or bi not in self.counted:
                continue
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
        if os.environ.get('FC_PROBE_DUMP') and self.timed:          # diagnostic: every probed launch (shape, pairs, FLOPs, microseconds) as JSON lines
            with open(os.environ['FC_PROBE_DUMP'], 'w') as fh:
                for bi, steps in self.timed.items():
                    for ev in steps:
                        if not ev:
                            continue
                        for (s_, e_), (p_, fpp, n_out, nb), shp in zip(ev, self.counted[bi], self.shapes[bi]):
                            pr = float(p_.item()) if p_ is not None else float(n_out)
                            fh.write(json.dumps(dict(shape=shp, pairs=pr, gflop=pr * fpp / 1e9, us=s_.elapsed_time(e_) * 1e3)) + '\n')
        traffic, traffic_src = None, None
        for tname in ('r6_traffic.json', 'r5_traffic.json', 'r4_traffic.json'):          # PMC passes cannot run inside the timed region: committed rocprofv3 result
            tj = os.path.join(ROOT, 'profiles', tname)
            if os.path.exists(tj):
                t = json.load(open(tj))
                traffic, traffic_src = t['hbm_bytes_per_launch'], f'profiles/{tname} (' + t['method'] + ')'
                break
        import fcaf3d_amd.functional as Fn
        x6 = bool(Fn.X6)
        h3 = x6 and Fn.split_mode() == 2
        peak = (PEAK_H3_TFLOPS if h3 else PEAK_X6_TFLOPS) if x6 else PEAK_F32_MFMA_TFLOPS
        return dict(bound='mfma',
                    kernel=('k_conv_x6 MODE 2 "h3" (sparse conv fwd + dgrad, dense GEMMs of convT/heads: fp32 in / fp32 accumulate, every fp32 '
                            'product as 3 exact fp16 x fp16 products of two-piece operands on v_mfma_f32_32x32x16_f16; the operands\' amax '
                            'passes are inside the measured time)') if h3 else
                    ('k_conv_x6 (sparse conv fwd + dgrad, dense GEMMs of convT/heads: fp32 in / fp32 accumulate, every fp32 '
                     'product as 6 exact bf16 x bf16 products on v_mfma_f32_32x32x16_bf16)') if x6 else
                    'k_conv_mfma (sparse conv fwd + dgrad, dense GEMMs of convT/heads)',
                    achieved=round(achieved, 3), peak=peak, unit='TFLOP/s',
                    peak_source=('fp16 dense MFMA peak 2500 TFLOP/s / 3 matrix products per fp32 product (MI355X_MICROARCH.md); '
                                 'achieved counts the ALGORITHMIC fp32 FLOPs 2 P Cin Cout') if h3 else
                    ('bf16 dense MFMA peak 2500 TFLOP/s / 6 matrix products per fp32 product (MI355X_MICROARCH.md); '
                     'achieved counts the ALGORITHMIC fp32 FLOPs 2 P Cin Cout') if x6 else
                    'v_mfma_f32_32x32x2_f32 dense peak (MI355X_MICROARCH.md)',
                    frac=round(achieved / peak, 4), vs_f32_mfma_peak=round(achieved / PEAK_F32_MFMA_TFLOPS, 4),
                    vs_six_product_peak=round(achieved / PEAK_X6_TFLOPS, 4) if x6 else None,
                    products_per_fp32_product=(3 if h3 else 6) if x6 else 1,
                    amax_ms_in_measured_time=round(amax_ms, 4),
                    traffic=traffic, traffic_unit='B/launch',
                    traffic_source=traffic_src, algorithmic_bytes_per_launch=round(alg_bytes / n), launches=n,
                    avg_launch_us=round(ms * 1e3 / n, 2), flops_per_launch=round(flops / n / 1e9, 4),
                    time_share_ms_per_step=None)


def physical_cores():
    """(physical cores, logical CPUs, model name) of this host, from lscpu / /proc/cpuinfo"""
    import subprocess
    info = {}
    try:
        for line in subprocess.run(['lscpu'], capture_output=True, text=True, timeout=10).stdout.splitlines():
            if ':' in line:
                k, v = line.split(':', 1)
                info[k.strip()] = v.strip()
        cores = int(info.get('Core(s) per socket', '0')) * int(info.get('Socket(s)', '1'))
    except Exception:
        cores = 0
    return cores or (os.cpu_count() or 1), os.cpu_count() or 1, info.get('Model name', 'unknown')


def cpu_baseline(args):
    """MinkowskiEngine's CPU algorithm restated (the reference's CPU backend itself is not installable here) on ONE scene of
    the same workload, on this host's physical cores — oracle/cpu_bench.py, run as a SUBPROCESS so that OMP_NUM_THREADS is in
    the environment before any OpenMP runtime starts (r2 set it too late and ran on half of the cores): the whole
    forward_train + backward (kernel maps, normalisation, assignment, losses included) with the sparse convolutions in
    C / OpenMP SIMD kernels, and the convolutions alone with their GFLOP/s and fraction of the host's fp32 peak."""
    import subprocess
    phys, logical, cpu_name = physical_cores()
    # lscpu reports the HOST's cores; the container may be allowed fewer (cgroup / affinity mask): 128 OpenMP threads pinned onto
    # a smaller mask ran the r3 kernels at 24 GFLOP/s.  Use the CPUs this process may actually run on, at most one per core.
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = logical
    try:                                                   # ... and a cgroup CPU quota caps it further
        q, per = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if q != 'max':
            avail = min(avail, max(1, int(int(q) / int(per))))
    except Exception:
        pass
    threads = max(1, min(phys, avail))
    env = dict(os.environ, OMP_NUM_THREADS=str(threads), HIP_VISIBLE_DEVICES='')
    env.pop('OMP_PROC_BIND', None); env.pop('OMP_PLACES', None)
    cmd = [sys.executable, '-m', 'oracle.cpu_bench', '--workload', args.workload, '--config', CONFIG_OF[args.workload],
           '--voxel-size', str(args.voxel_size), '--levels', str(args.levels), '--points', str(args.cpu_points),
           '--reps', str(max(args.cpu_reps, 1))]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    if r.returncode != 0:
        return dict(error=(r.stderr or r.stdout)[-400:])
    out = json.loads(r.stdout.strip().splitlines()[-1])
    out['affinity_cpus'] = avail
    return out


def count_steps(args, n_batches):
    """number of untimed FLOP-count steps after the timed region — a function of the flags only, never of the rank.  (A probed
    step that went through the native executor brings its FLOPs with it; one that fell back to the per-operator path — pruning
    that bites, as on the S3DIS workload — needs these counts.)"""
    return 0 if (args.no_instrument or args.breakdown) else n_batches      # every distinct batch a probed step may use


def hbm_steps(args, exec_on):
    """untimed per-operator steps for `roofline.hbm_kernels` when the timed region went through the executor — again a function
    of the flags only"""
    return 1 if (exec_on and not (args.no_instrument or args.breakdown)) else 0


LAST_HOST_S = 0.0
LAST_WAIT_S = 0.0
LAST_CPU_S = 0.0


def gpu_local_cpus(dev_index):
    """CPUs of the NUMA node the GPU hangs off (sysfs, by PCI address), or None"""
    try:
        p = torch.cuda.get_device_properties(dev_index)
        bdf = f'{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0'
        txt = open(f'/sys/bus/pci/devices/{bdf}/local_cpulist').read().strip()
        cpus = set()
        for part in txt.split(','):
            a, _, b = part.partition('-')
            cpus.update(range(int(a), int(b or a) + 1))
        return cpus or None
    except Exception:
        return None


def host_micro(dev):
    """three micro-measurements of the host, taken right after the timed region (diagnostic for the slow mode of profiles/r5_notes.md
    section 16 / r6_notes.md section 3: is the CPU slow, is a launch slow, or is a device round trip slow?)"""
    t0 = time.perf_counter()
    acc = 0
    for i in range(200000):
        acc += i & 3
    py_ms = (time.perf_counter() - t0) * 1e3
    x = torch.zeros(64, device=dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(300):
        x.add_(1.0)
    launch_us = (time.perf_counter() - t0) / 300 * 1e6
    torch.cuda.synchronize()
    ev = torch.cuda.Event()
    t0 = time.perf_counter()
    for _ in range(50):
        x.add_(1.0)
        ev.record()
        ev.synchronize()
    sync_us = (time.perf_counter() - t0) / 50 * 1e6
    return dict(python_200k_loop_ms=round(py_ms, 2), torch_launch_us=round(launch_us, 2), launch_plus_event_sync_us=round(sync_us, 2),
                cpu=os.sched_getcpu() if hasattr(os, 'sched_getcpu') else None)


def host_state():
    """load average and usable CPUs of the host (diagnostic beside `host_enqueue_ms_per_step`: the slow mode of profiles/r5_notes.md
    section 16 leaves the kernels at their speed — whether the HOST was slow or busy is what these fields record)"""
    try:
        la = [round(x, 2) for x in os.getloadavg()]
    except OSError:
        la = None
    return dict(loadavg=la, cpus=len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else os.cpu_count())


def thread_cpu_snapshot():
    """{tid: (comm, cpu seconds)} of every thread of this process (/proc/self/task): who burns the CPU time `process_cpu_ms_per_step` reports"""
    out = {}
    try:
        tck = os.sysconf('SC_CLK_TCK')
        for tid in os.listdir('/proc/self/task'):
            try:
                st = open(f'/proc/self/task/{tid}/stat').read()
                comm = st[st.index('(') + 1:st.rindex(')')]
                f = st[st.rindex(')') + 2:].split()
                out[tid] = (comm, (int(f[11]) + int(f[12])) / tck)
            except (OSError, ValueError, IndexError):
                pass
    except (OSError, ValueError):
        pass
    return out


LAST_THREADS = None
SAMPLE_MAIN_THREAD = False
LAST_STEPS = None           # per-call enqueue times of the last timed region and what the kernel did to the process meanwhile


GC_LOG = []                 # (generation, seconds) of every garbage collection of the interpreter since install_gc_log()


def install_gc_log():
    """time every collection of CPython's cyclic collector (gc.callbacks): a FULL collection walks every tracked object of the
    process — tens of milliseconds with torch imported — on whichever thread allocated last, holding the GIL"""
    import gc
    t = [0.0]

    def cb(phase, info):
        if phase == 'start':
            t[0] = time.perf_counter()
        else:
            GC_LOG.append((info['generation'], time.perf_counter() - t[0]))
    gc.callbacks.append(cb)


def sched_snapshot():
    """what the kernel did to this process: context switches and page faults (getrusage) and, where the files are readable, the CPU
    quota of the box's cgroup (cpu.stat: periods in which the group was THROTTLED) and the CPU pressure of the host (PSI)"""
    import resource
    ru = resource.getrusage(resource.RUSAGE_SELF)
    out = dict(nvcsw=ru.ru_nvcsw, nivcsw=ru.ru_nivcsw, majflt=ru.ru_majflt, minflt=ru.ru_minflt)
    try:
        for ln in open('/sys/fs/cgroup/cpu.stat'):
            k, v = ln.split()
            if k in ('nr_periods', 'nr_throttled', 'throttled_usec'):
                out[k] = int(v)
    except (OSError, ValueError):
        pass
    try:                                           # the caching allocator: device allocations (hipMalloc) and frees, bytes it holds
        ms = torch.cuda.memory_stats()
        out['device_mallocs'] = int(ms.get('num_device_alloc', ms.get('segment.all.allocated', 0)))
        out['device_frees'] = int(ms.get('num_device_free', ms.get('segment.all.freed', 0)))
        out['reserved_MB'] = int(ms.get('reserved_bytes.all.current', 0)) >> 20
        out['alloc_retries'] = int(ms.get('num_alloc_retries', 0))
    except Exception:                              # noqa: no device
        pass
    try:
        some = open('/proc/pressure/cpu').readline().split()
        out['psi_cpu_some_total_us'] = int(some[-1].split('=')[1])
    except (OSError, ValueError, IndexError):
        pass
    return out


def timed_region(fn, n, world, dev):
    """barrier + synchronize | n calls | synchronize + barrier; returns the MAX over ranks of the elapsed seconds"""
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    import fcaf3d_amd._lib as L
    w0 = L.HOST_WAIT[0]
    th0 = thread_cpu_snapshot()
    sampler = None
    if SAMPLE_MAIN_THREAD:
        import subprocess
        import tempfile
        sfile = os.path.join(tempfile.gettempdir(), f'fc_wchan_{os.getpid()}.json')
        sampler = subprocess.Popen([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tools', 'wchan.py'),
                                    str(os.getpid()), str(os.getpid()), sfile])
        time.sleep(0.3)
        sampler.send_signal(__import__('signal').SIGUSR1)
    s0 = sched_snapshot()
    g0 = len(GC_LOG)
    c0 = time.process_time()
    t0 = time.perf_counter()
    last = None
    marks = [t0]
    for i in range(n):
        last = fn(i)
        marks.append(time.perf_counter())
    global LAST_HOST_S, LAST_WAIT_S, LAST_STEPS
    LAST_HOST_S = time.perf_counter() - t0       # the host has ENQUEUED all n calls (diagnostic: close to the region's time = host-bound)
    LAST_WAIT_S = L.HOST_WAIT[0] - w0            # ... of which it was blocked on the device by design (run-ahead bound, staging ring, lookahead plan)
    torch.cuda.synchronize()
    global LAST_CPU_S
    LAST_CPU_S = time.process_time() - c0        # CPU time of ALL threads of the process over the region (spinning waits show here)
    s1 = sched_snapshot()
    per = sorted((b - a) * 1e3 for a, b in zip(marks, marks[1:]))
    LAST_STEPS = dict(call_ms_min=round(per[0], 2), call_ms_median=round(per[len(per) // 2], 2), call_ms_max=round(per[-1], 2),
                      calls_over_twice_the_median=sum(1 for x in per if x > 2 * per[len(per) // 2]),
                      kernel={k: s1[k] - s0[k] for k in s1 if k in s0},
                      gc={f'gen{g}': [sum(1 for x in GC_LOG[g0:] if x[0] == g), round(sum(x[1] for x in GC_LOG[g0:] if x[0] == g) * 1e3, 2)]
                          for g in (0, 1, 2)})
    if sampler is not None:
        sampler.terminate()
        try:
            sampler.wait(timeout=5)
            LAST_STEPS['main_thread'] = json.load(open(sfile))
        except Exception as e:                   # noqa: a diagnostic
            LAST_STEPS['main_thread'] = dict(error=str(e))
    global LAST_THREADS
    th1 = thread_cpu_snapshot()
    agg = {}
    plan_tids = {str(t.native_id) for t in __import__('threading').enumerate() if t.name.startswith('fc-plan')}
    for tid, (comm, c1) in th1.items():
        d = c1 - th0.get(tid, (comm, 0.0))[1]
        if d > 0:
            name = 'main' if tid == str(os.getpid()) else ('fc-plan' if tid in plan_tids else comm + ' (other)')
            agg[name] = agg.get(name, 0.0) + d
    LAST_THREADS = {k: round(v / n * 1e3, 2) for k, v in sorted(agg.items(), key=lambda kv: -kv[1])[:6]}
    if world > 1:
        torch.distributed.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    return dt, last


def main():
    args = parse()
    install_gc_log()
    host0 = host_state()
    # stdout carries ONE line, the JSON result: everything else that writes to file descriptor 1 during the run (RCCL's
    # version banner, library warnings) goes to stderr
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    if os.environ.get('FC_FAULT_DUMP'):           # debugging aid: dump every thread's stack after n seconds and exit
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ['FC_FAULT_DUMP']), exit=True)
    from fcaf3d_amd import dist as D
    from fcaf3d_amd.runner import TrainStep
    D.init_dist()
    rank = int(os.environ.get('RANK', '0'))
    world = D.world_size()
    local = int(os.environ.get('LOCAL_RANK', '0'))
    assert torch.cuda.is_available(), 'bench.py needs the MI355X (there is no CPU fallback for the product path)'
    if os.environ.get('FC_DIST_BACKEND') == 'gloo':
        local %= torch.cuda.device_count()       # smoke mode only: several ranks may share one GPU
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    # keep the process (main thread, the lookahead plan thread, the HIP runtime's threads created from here on) on the CPUs of the NUMA
    # node the GPU hangs off: doorbells, signal polling and pinned staging buffers then stay on one socket (FC_NUMA_PIN=0: leave the
    # scheduler alone)
    pinned = None
    if os.environ.get('FC_NUMA_PIN', '1') != '0' and hasattr(os, 'sched_setaffinity'):
        loc = gpu_local_cpus(local)
        if loc:
            allowed = os.sched_getaffinity(0) & loc
            if len(allowed) >= 4:
                os.sched_setaffinity(0, allowed)
                pinned = len(allowed)

    _dummies = [torch.cuda.Stream(device=dev) for _ in range(int(os.environ.get('FC_DUMMY_STREAMS', '0')))]     # diagnostic: shifts the stream -> hardware-queue mapping
    if args.priority_stream:
        # the step's dependent chain runs on a HIGH-priority HIP stream; the overlapped weight-gradient stream and the
        # coordinate stream keep the default (lower) priority, so they fill idle CUs instead of competing for them
        torch.cuda.set_stream(torch.cuda.Stream(device=dev, priority=-1))
    model, cfg = build_model(args)
    model = model.to(dev).train()
    model.async_maps = True           # scenes are resident in HBM: coordinate work may run on its side stream
    model.inputs_resident = True      # ... without waiting for the main stream (nothing enqueued there produces them)
    model.spatial_sort = args.spatial_sort
    import fcaf3d_amd.functional as Fn
    import fcaf3d_amd.executor as EX
    exec_on = EX.ENABLED and not args.no_executor
    lookahead = not args.no_lookahead
    Fn.WGRAD_ASYNC = not args.no_wgrad_overlap
    head_overlap = model.neck_with_head.head_overlap and not args.no_wgrad_overlap
    if world > 1:
        for p in model.parameters():
            torch.distributed.broadcast(p.data, 0)
    # the reference's recipe (configs/fcaf3d/fcaf3d.py:30-33): AdamW 1e-3 / 1e-4, grad-clip 10, step LR — fcaf3d_amd/runner.py
    trainer = TrainStep.from_config(model, cfg)
    batches = make_batches(args, rank, dev)
    # r6: the caching allocator's pools (one per stream) are filled before the first step — no device allocation inside a timed step
    reserved = 0
    if args.reserve_gb > 0:
        import fcaf3d_amd.sparse as SPm
        from fcaf3d_amd.runner import reserve_device_memory
        sharers = -(-world // torch.cuda.device_count()) if os.environ.get('FC_DIST_BACKEND') == 'gloo' else 1     # smoke mode: ranks per GPU
        reserved = reserve_device_memory(args.reserve_gb / sharers, dev, streams=[
            (torch.cuda.current_stream(dev), 0.5), (SPm.map_stream(dev), 0.25), (Fn.wgrad_stream(dev), 0.125),
            (model.neck_with_head._head_stream(dev), 0.125)], chunk_gb=8)

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

    def step(i, probe_mode=None, which=None, prefetch=False):
        src = which if which is not None else batches
        batch = src[i % len(src)]
        # prefetch: the NEXT batch's coordinate phase starts beside this step, on a worker thread (TrainStep next_batch ->
        # SingleStageSparse3DDetector.prefetch).  Every timed region asks for it on all steps but its last and plans its first
        # step in line, so that exactly n coordinate phases run inside a region of n steps; probed steps never prefetch
        # (their brackets must hold the kernels' own time).
        nxt = src[(i + 1) % len(src)] if (prefetch and lookahead and probe_mode is None) else None
        if probe:
            if probe_mode:
                probe.begin_step(i % len(src), probe_mode)
            else:
                probe.mode = None
        # probed steps keep every kernel on the main stream: a HIP-event bracket is only a kernel's own duration when
        # nothing else shares the GPU (the overlapped weight-gradient stream would inflate every bracket it touches)
        one_stream = probe_mode in ('time', 'hbm')
        Fn.WGRAD_ASYNC = (not args.no_wgrad_overlap) and not one_stream
        # probed steps of the timed region go through the native executor like every other step (its one-stream program), the
        # brackets sit inside fc_exec around every convolution operator; the untimed extras (FLOP counts of the per-operator
        # path, `hbm_kernels`) bracket individual C-ABI calls of the per-operator module path — the same calls with the same
        # arguments (tests/test_gpu_exec.py: bit for bit)
        EX.ENABLED = exec_on and probe_mode in (None, 'time') and bd is None
        EX.PROBE = probe.exec_steps if (probe and probe_mode == 'time' and EX.ENABLED) else None
        model.neck_with_head.head_overlap = head_overlap and not one_stream             # ... nor the head branch's stream
        n_exec = len(probe.exec_steps) if probe else 0
        loss, _ = trainer(batch, next_batch=nxt)
        EX.PROBE = None
        if probe and probe_mode == 'time' and len(probe.exec_steps) > n_exec and probe._cur:
            probe._cur.clear()       # the step went through the executor (its brackets are inside fc_exec); a pruned finest level's
        return loss                  # per-operator tail (2 convolutions) is not double-booked

    for i in range(args.warmup):
        step(i)
    # r6: settling.  Some first processes ran with 20-23 ms of host time per step instead of 4-5 at unchanged kernel times; read at first
    # as the box waking up, it was device allocations and first-use costs inside the steps (r6_notes.md section 14: removed at the root
    # by --reserve-gb, the warmed probed step below and the cycle-free coordinate objects).  The guard stays: while the host's enqueue
    # time of forward + backward exceeds half of the step's wall time, keep stepping UNTIMED (windows of 4 steps, at most
    # --settle-seconds), every rank in lockstep; what was done is reported in config.host.settle.
    settle = dict(windows=0, extra_untimed_steps=0, seconds=0.0, host_share_first=None, host_share_last=None)
    if args.settle_seconds > 0 and not args.breakdown:
        t_settle = time.perf_counter()
        k = args.warmup
        while True:
            torch.cuda.synchronize()
            p0, w0 = list(trainer.phase_s), time.perf_counter()
            for j in range(4):
                step(k + j, prefetch=j < 3)
            torch.cuda.synchronize()
            wall = time.perf_counter() - w0
            share = ((trainer.phase_s[1] - p0[1]) + (trainer.phase_s[2] - p0[2])) / max(wall, 1e-9)
            k += 4
            settle['windows'] += 1
            settle['extra_untimed_steps'] += 4
            if settle['host_share_first'] is None:
                settle['host_share_first'] = round(share, 3)
            settle['host_share_last'] = round(share, 3)
            slow = 1.0 if share > 0.5 else 0.0
            if world > 1:
                t = torch.tensor([slow], device=dev)
                torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
                slow = float(t.item())
            if slow == 0.0 or time.perf_counter() - t_settle > args.settle_seconds:
                break
        settle['seconds'] = round(time.perf_counter() - t_settle, 2)
    if exec_on and probe:
        # the probed step of the timed region runs the executor's ONE-STREAM program: build its operator list now, not inside
        # the timed region (r4: ~20 ms of host work that the first 20-step line carried as +1 ms per step)
        wa0, ho0 = Fn.WGRAD_ASYNC, model.neck_with_head.head_overlap
        Fn.WGRAD_ASYNC, model.neck_with_head.head_overlap = False, False
        EX.program_for(model, True)
        Fn.WGRAD_ASYNC, model.neck_with_head.head_overlap = wa0, ho0
    if exec_on and args.warmup > 0 and not args.no_instrument:
        # ... and run it once, untimed, brackets and all (r6: the first probed step of a process creates a few hundred HIP events and
        # whatever pools the runtime keeps behind them — 36 MB of fresh host memory, 20-120 ms — and it was step 8 of the timed region).
        # EVERY rank steps (the step holds collectives; the condition is a function of the flags only), rank 0 with its probe
        step(args.warmup, 'time' if probe else None)
        if probe:
            probe.exec_records()
            probe.timed, probe.exec_steps = {}, []
    if bd:
        torch.cuda.synchronize()
        bd.rec.clear()
    if world > 1:
        trainer.averager.log = []                # host-side launch times of the gradient buckets of the timed steps
    # probed steps: the LAST step of the timed region and every `probe_every`-th before it.  r1-r4 probed the first step of each window
    # — with the next step's coordinate phase (its own stream) running beside the probed step's kernels and inside their brackets: 10 %
    # above the kernels' own durations in the rocprofv3 trace.  Nothing is enqueued behind the last step before the region's closing
    # synchronise, so its brackets are the kernels' own time (r5: 135 -> ~125 us per operator against 122 us in the trace)
    pe = max(args.probe_every, 1)
    ph0 = list(trainer.phase_s)
    global SAMPLE_MAIN_THREAD
    SAMPLE_MAIN_THREAD = bool(args.sample_main_thread) and rank == 0          # (stays on for the extras' regions: a diagnostic run)
    dt, loss = timed_region(lambda i: step(args.warmup + i, 'time' if (args.steps - 1 - i) % pe == 0 else None, prefetch=i < args.steps - 1),
                            args.steps, world, dev)
    host_main = dict(host_enqueue_ms_per_step=round(LAST_HOST_S / args.steps * 1e3, 3),
                     host_blocked_ms_per_step=round(LAST_WAIT_S / args.steps * 1e3, 3),
                     host_busy_ms_per_step=round((LAST_HOST_S - LAST_WAIT_S) / args.steps * 1e3, 3),
                     process_cpu_ms_per_step=round(LAST_CPU_S / args.steps * 1e3, 3), thread_cpu_ms_per_step=LAST_THREADS, steps=LAST_STEPS, settle=settle,
                     phases_ms_per_step=[round((b - a) / args.steps * 1e3, 3) for a, b in zip(ph0, trainer.phase_s)],
                     micro=host_micro(dev), pinned_to_gpu_numa_node=pinned, after=host_state(),
                     allocator_reserve_GB=round(reserved / 2 ** 30, 1))
    final_loss = float(loss.item())
    dp_log = getattr(trainer.averager, 'log', None)
    trainer.averager.log = None
    # untimed: FLOPs of every launch, one step per distinct batch.  EVERY rank steps (a step holds collectives: a
    # rank-0-only extra step deadlocks the job); only rank 0 carries the probe
    for b in range(count_steps(args, len(batches))):
        step(b, 'count' if probe else None)
    for b in range(hbm_steps(args, exec_on)):      # untimed, per-operator path: the bandwidth-bound entry points for `hbm_kernels`
        # ... with NOTHING else on the GPU: the previous step's weight-image build (weight-gradient stream) has drained and this step's
        # coordinate phase / target assignment stays on the main stream (r6: with the host no longer the bottleneck the image build of
        # step i - 1 ran beside the stem convolution of the probed step and doubled its bracket)
        torch.cuda.synchronize()
        model.async_maps = False
        import fcaf3d_amd.plan as PL
        PL.PROBE = bool(probe)
        try:
            step(b, 'hbm' if probe else None)
        finally:
            model.async_maps = True
            PL.PROBE = False
        if probe:                                  # the coordinate phase's own launches (brackets inside fc_plan_levels / fc_plan_maps)
            torch.cuda.synchronize()
            for kind, nbytes, ms in PL.probe_read():
                probe.hbm_native.append(('fc_plan:' + kind, nbytes, ms))
    if probe:
        probe.mode = None
    EX.ENABLED = exec_on
    if bd:
        bd.report(args.steps, dt * 1e3)
    assert np.isfinite(final_loss), 'loss diverged'
    # the roofline of the timed region's probed steps, read out now (the extra below probes once more)
    rl = probe.summary() if probe else None
    if rl:
        probed_steps = len(range(0, args.steps, max(args.probe_every, 1)))
        rl['probed_steps'] = probed_steps
        rl['time_share_ms_per_step'] = round(rl['avg_launch_us'] * rl['launches'] / 1e3 / probed_steps, 3)
        rl['hbm_kernels'] = probe.hbm_summary()
    # r5: the convolution launches now ALSO compute the BatchNorm statistics / backward reductions of the layer next to them in their
    # epilogues (csrc/conv_x6.h X6Epi) — work that used to be launches of its own sits inside the brackets of `roofline`.  Two untimed
    # probed steps, back to back under identical conditions (nothing else enqueued: no next step's coordinate phase beside them, which
    # is why both read higher than the timed region's figure): the default route, and the r4 arrangement (FC_BN_FUSE=0: the same
    # convolution kernels without those epilogues) — the second is the per-launch figure comparable with r1-r4's
    if rl and exec_on and Fn.BN_FUSE and world == 1 and not args.no_extras:
        ab = {}
        try:
            for fused in (True, False):
                probe.timed, probe.exec_steps = {}, []
                Fn.BN_FUSE = fused
                model.__dict__.pop('_programs', None)
                for i in range(2):
                    step(i)
                torch.cuda.synchronize()
                step(0, 'time')
                u = probe.summary()
                ab['with_bn_epilogues' if fused else 'without_bn_epilogues'] = dict(achieved=u['achieved'], frac=u['frac'],
                                                                                    avg_launch_us=u['avg_launch_us'], launches=u['launches'])
            ab['what'] = ('two untimed probed steps on an otherwise idle GPU: the default route (BatchNorm statistics / backward reductions in '
                          'the convolution epilogues, inside the brackets) and FC_BN_FUSE=0 (those reductions as passes of their own, as in r1-r4)')
        except Exception as e:
            ab['error'] = repr(e)[:200]
        finally:
            Fn.BN_FUSE = True
            model.__dict__.pop('_programs', None)
            probe.mode = None
            probe.timed, probe.exec_steps = {}, []
        rl['bn_epilogue_ab'] = ab

    # ---- extras, all OUTSIDE the timed region above (every rank runs them: they hold collectives) ------------------
    # (a) SURVEY 8(d) protocol: forward_train + backward only, synchronised around every iteration, median
    fb = []
    for i in range(5):
        batch = batches[i % len(batches)]
        trainer.optimizer.zero_grad(set_to_none=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        losses = model(return_loss=True, **batch)
        (losses['loss_centerness'] + losses['loss_bbox'] + losses['loss_cls']).backward()
        trainer.averager.finish()
        torch.cuda.synchronize()
        fb.append(time.perf_counter() - t0)
    fb_med = float(np.median(fb))
    # (b) BASELINE config 4: global batch 16 split over the ranks (2 scenes per GPU on 8), same step
    cfg4 = None
    if world > 1 and 16 % world == 0 and args.workload == 'scannet-100k':
        per = 16 // world
        small = [{k: v[:per] for k, v in b.items()} for b in batches] if per <= args.batch else None
        if small is not None:
            for i in range(2):
                step(i, which=small)
            dt4, _ = timed_region(lambda i: step(i, which=small, prefetch=i < 5), 6, world, dev)
            cfg4 = dict(global_batch=16, scenes_per_gpu_per_step=per, steps=6, ms_per_step=round(dt4 / 6 * 1e3, 3),
                        value=round(16 * 6 / dt4, 3), unit='scenes/s')
    # (b2) N = 1 through the data-parallel machinery (1-rank RCCL group: autograd hooks, bucket launches on the weight-gradient
    #      stream, in-place all-reduce of the flat gradient buffer) — what the DP path itself costs, without a second GPU
    forced = cfg4_1 = None
    if world == 1 and not args.no_force_dp and trainer.flat is not None:
        try:
            import socket
            sk = socket.socket(); sk.bind(('127.0.0.1', 0)); port = sk.getsockname()[1]; sk.close()
            torch.distributed.init_process_group('nccl', init_method=f'tcp://127.0.0.1:{port}', rank=0, world_size=1)
            D.FORCE_AVERAGER = True
            plain = trainer.averager
            trainer.averager = D.GradientAverager(trainer.params, bucket_mb=64, flat=trainer.flat)
            for i in range(3):
                step(i)
            dtf, _ = timed_region(lambda i: step(i, prefetch=i < 9), 10, 1, dev)
            forced = dict(steps=10, ms_per_step=round(dtf / 10 * 1e3, 3), value=round(args.batch * 10 / dtf, 3), unit='scenes/s', calls=LAST_STEPS,
                          host_busy_ms_per_step=round((LAST_HOST_S - LAST_WAIT_S) / 10 * 1e3, 3),
                          buckets=len(trainer.averager.buckets), what='the same step with the gradient averager active in a 1-rank RCCL group')
            if args.workload == 'scannet-100k' and args.batch >= 2:
                # BASELINE config 4 as ONE of its 8 GPUs sees it: 2 scenes per step (global batch 16 / 8), through the data-parallel
                # machinery (1-rank RCCL group: bucket launches, in-place all-reduce of the 282 MB flat gradient buffer)
                small = [{k: v[:2] for k, v in b.items()} for b in batches]
                for i in range(3):
                    step(i, which=small)
                dt2, _ = timed_region(lambda i: step(i, which=small, prefetch=i < 11), 12, 1, dev)
                ms2 = dt2 / 12 * 1e3
                gbytes = trainer.flat.n * 4
                # prediction, not a measurement: ring all-reduce moves 2 (N-1)/N of the buffer per GPU; one xGMI link carries
                # ~153 GB/s (MI355X_MICROARCH.md / task statement: 7 links x ~153 GB/s, ring collectives are per-link bound).
                # Worst case = nothing of it hidden behind backward, one ring; best case = fully hidden.
                t_ar = 2 * 7 / 8 * gbytes / 153e9 * 1e3
                cfg4_1 = dict(scenes_per_gpu_per_step=2, steps=12, ms_per_step=round(ms2, 3), value=round(2 * 12 / dt2, 3), unit='scenes/s per GPU',
                              what='N = 1, 2 scenes per step (what each of the 8 GPUs of BASELINE config 4 runs), gradient averager in a 1-rank RCCL group',
                              gradient_MB=round(gbytes / 1e6, 1),
                              predicted_8gpu=dict(note='PREDICTION from this N = 1 measurement, not measured on 8 GPUs',
                                                  allreduce_ms_one_ring_153GBps=round(t_ar, 2),
                                                  scenes_per_s_allreduce_hidden=round(16 / ms2 * 1e3, 1),
                                                  scenes_per_s_allreduce_exposed=round(16 / (ms2 + t_ar) * 1e3, 1)))
            trainer.averager.close()
            trainer.averager = plain
        except Exception as e:                                   # an extra: never fails the bench line
            forced = dict(error=repr(e)[:200])
        finally:
            D.FORCE_AVERAGER = False
            if torch.distributed.is_initialized():
                torch.distributed.destroy_process_group()
    # (b3) the same step on the fp32 MFMA kernels (FC_X6=0: v_mfma_f32_32x32x2_f32, IEEE fp32 products) — the number to fall back on
    #      if the split-bf16 route were not accepted as fp32-within-tolerance
    fp32_route = None
    if world == 1 and Fn.X6 and not args.no_fp32_route:
        x6_0, Fn.X6 = Fn.X6, False
        try:
            for i in range(3):
                step(i)
            dt32, _ = timed_region(lambda i: step(i, prefetch=i < 7), 8, 1, dev)
            fp32_route = dict(steps=8, ms_per_step=round(dt32 / 8 * 1e3, 3), value=round(args.batch * 8 / dt32, 3), unit='scenes/s',
                              what='FC_X6=0: every convolution on v_mfma_f32_32x32x2_f32 (per-operator module path)')
        finally:
            Fn.X6 = x6_0
    # (b4) SURVEY.md 8(f) rank 4: bf16 fast mode — the split-bf16 launches multiply the operands rounded to bf16 only (one MFMA
    #      product instead of six).  NOT a parity route (tests/test_gpu_ops.py bounds it at 2e-2); a flagged extra, never `value`
    bf16_fast = None
    if world == 1 and Fn.X6 and not args.no_extras:
        import fcaf3d_amd._lib as L
        try:
            torch.cuda.synchronize()
            L.lib().fc_set_bf16_fast(1)
            trainer.invalidate_images()          # the fast mode reads six-product (bf16) images
            for i in range(3):
                step(i)
            dtb, lb = timed_region(lambda i: step(i, prefetch=i < 7), 8, 1, dev)
            bf16_fast = dict(parity=False, steps=8, ms_per_step=round(dtb / 8 * 1e3, 3), value=round(args.batch * 8 / dtb, 3), unit='scenes/s',
                             loss_after=round(float(lb), 4),
                             what='fc_set_bf16_fast(1): forward / backward-data / weight-gradient MFMA launches on bf16-rounded operands '
                                  '(1 of the 6 products), fp32 accumulate; everything else unchanged')
        except Exception as e:
            bf16_fast = dict(error=repr(e)[:200])
        finally:
            torch.cuda.synchronize()
            L.lib().fc_set_bf16_fast(0)
            trainer.invalidate_images()
    # (b5) BASELINE.md section 2 "also report" configurations (their own models; 5 steps each)
    extras = {}
    if world == 1 and not args.no_extras and args.workload == 'scannet-100k' and args.levels == 4 and args.voxel_size == 0.02:
        for key in EXTRAS:
            try:
                extras[key] = run_extra(args, key, dev, args.extra_steps)
            except Exception as e:                               # an extra never fails the bench line
                extras[key] = dict(error=repr(e)[:200])
        Fn.WGRAD_ASYNC = not args.no_wgrad_overlap
    # (c) inference: simple_test (eval-mode BatchNorm, decode, multi-class BEV NMS on the device) — the only quantity the
    #     reference publishes a speed for (README.md:91-93, scenes/s on one GPU)
    infer = infer_pipe = None
    if args.infer_steps > 0:
        model.eval()
        model.static_weights = True                  # inference serving: the weights do not change between calls
        test_batches = [dict(points=b['points'], img_metas=b['img_metas']) for b in batches]
        # a NORMAL-priority stream for serving: with two batches in flight the coordinate stream (high priority) must outrank
        # the stream that carries the previous batch's forward pass, or the next batch's count read-backs queue behind it
        # (tools/pipeprof.py: 12.7 -> 9.5 ms per batch here, 12.7 -> 12.8 on the training loop's high-priority stream)
        torch.cuda.synchronize()
        with torch.cuda.stream(torch.cuda.Stream(device=dev)), torch.no_grad():
            model(return_loss=False, **test_batches[0])
            dti, _ = timed_region(lambda i: model(return_loss=False, **test_batches[i % len(test_batches)]), args.infer_steps,
                                  world, dev)
            # ... and with two batches in flight (simple_test_async): the next batch's coordinate phase on the host while this
            # one's forward pass and decode run on the GPU
            n_pipe = 2 * args.infer_steps

            pending = [None]                     # the batch enqueued last, its read-backs still to be done

            def pipelined(i):
                h = model.simple_test_async(**test_batches[i % len(test_batches)])
                if pending[0] is not None:
                    pending[0]()                     # results of batch i - 1 while batch i runs on the GPU
                pending[0] = h
                if i == n_pipe - 1:
                    pending[0]()
                    pending[0] = None
            dtp, _ = timed_region(pipelined, n_pipe, world, dev)
        torch.cuda.synchronize()
        model.train()
        infer_pipe = dict(value=round(args.batch * world * n_pipe / dtp, 3), unit='scenes/s', steps=n_pipe,
                          ms_per_batch=round(dtp / n_pipe * 1e3, 3), what='simple_test_async: two batches in flight')
        infer = dict(value=round(args.batch * world * args.infer_steps / dti, 3), unit='scenes/s', steps=args.infer_steps,
                     ms_per_batch=round(dti / args.infer_steps * 1e3, 3), what='simple_test: extract_feat + get_bboxes + multi-class BEV NMS')

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
                       'step': 'forward_train + backward + grad all-reduce + grad-clip + AdamW (fcaf3d_amd/runner.py TrainStep)',
                       'parallelism': f'dp{world}', 'final_loss': round(final_loss, 4),
                       'host': dict(host_main, at_start=host0,
                                    what='host_enqueue_ms_per_step: host time until the timed steps were all ENQUEUED / steps (the last, '
                                         'probed step synchronises inside); host_blocked: the part of it the host waited for the device BY DESIGN '
                                         '(run-ahead bound of 2 steps in runner.TrainStep, pinned staging ring, a lookahead plan not ready yet); '
                                         'host_busy = the difference = the host\'s own work per step (r5: 17.5 of 20.5 ms, the step was host-bound '
                                         'whenever that grew); phases_ms_per_step: [run-ahead bound + prefetch, forward_train enqueue, backward enqueue, '
                                         'all-reduce finish + clip + AdamW + weight images]; micro: a pure-Python loop, a torch launch, a launch + event '
                                         'round trip, taken right after the region; loadavg / usable CPUs when the process started and after the timed region; '
                                         'process_cpu / thread_cpu_ms_per_step: CPU time of the process and of its busiest threads over the region (spinning '
                                         'waits included); settle: UNTIMED steps run after the --warmup steps, in windows of 4, while the host\'s forward + '
                                         'backward enqueue time exceeded half of the step\'s wall time (host_share) — at most '
                                         '--settle-seconds; one window is always run; steps: host time of the timed region\'s calls (min / median / max; calls above twice the '
                                         'median) and what the kernel did to the process over the region — voluntary / involuntary context switches, page '
                                         'faults, periods in which the box\'s cgroup was CPU-throttled (cpu.stat), CPU pressure of the host (PSI, us)'),
                       'coordinate_phase': ('native plan (csrc/plan.hip: fc_plan_levels + fc_plan_maps, 2 read-backs per step)' +
                                            (', the NEXT batch planned on a worker thread beside the current step (plan.Lookahead); exactly '
                                             f'{args.steps} plans inside the timed region' if lookahead else ', in line')),
                       'fwd_bwd_only': {'protocol': 'SURVEY 8(d): forward_train + backward (+ all-reduce), synchronised per iteration, median of 5',
                                        'ms': round(fb_med * 1e3, 3), 'scenes_per_s': round(args.batch * world / fb_med, 3)},
                       'config4_global_batch_16': cfg4, 'forced_dp_n1': forced, 'config4_per_gpu': cfg4_1,
                       'fp32_mfma_route': fp32_route, 'bf16_fast_mode': bf16_fast, 'inference': infer, 'inference_pipelined': infer_pipe,
                       **extras,
                       'kernel_source_sha16': __import__('fcaf3d_amd.build', fromlist=['source_hash']).source_hash(),
                       'executor': bool(exec_on) and 'network body through the native launch-list executor (fcaf3d_amd/executor.py, '
                                   'csrc/exec.hip), the probed step included (event brackets inside fc_exec)'},
        }
        out['roofline'] = rl
        if world > 1 and dp_log:
            out['config']['data_parallel'] = D.summarize_bucket_log(trainer.averager, dp_log)
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(args)
        else:
            out['cpu_baseline'] = None
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + '\n').encode())
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
