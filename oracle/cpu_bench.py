"""ORACLE — test infrastructure only: the `cpu_baseline` leg of bench.py, run as a SUBPROCESS
(`python -m oracle.cpu_bench ...`) so that OMP_NUM_THREADS / OMP_PROC_BIND are in the environment before any OpenMP
runtime initialises (r2 set them after torch had started its pool and used half of the cores).

What is timed, on ONE scene of the benchmark's workload, on this host's physical cores:
  whole step  — forward_train + backward of oracle/model_oracle.py (MinkowskiEngine's CPU algorithm restated: hash-map
                kernel maps, per-offset gather -> GEMM -> scatter-add, torch norms / losses), with every sparse
                convolution (forward, backward-data, backward-weights) running through the C / OpenMP SIMD kernels of
                oracle/conv_oracle.c — map construction, normalisation, target assignment and losses INCLUDED;
  conv only   — the same C kernels over the recorded layers, median of `reps`, with GFLOP/s and the fraction of the
                host's fp32 FMA peak (cores x SIMD width x 2 FMA pipes x 2 x clock).
Prints one JSON object on stdout."""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def host():
    info = {}
    try:
        for line in subprocess.run(['lscpu'], capture_output=True, text=True, timeout=10).stdout.splitlines():
            if ':' in line:
                k, v = line.split(':', 1)
                info[k.strip()] = v.strip()
        cores = int(info.get('Core(s) per socket', '0')) * int(info.get('Socket(s)', '1'))
    except Exception:
        cores = 0
    mhz = 0.0
    for key in ('CPU max MHz', 'CPU MHz'):
        try:
            mhz = float(info.get(key, '0').replace(',', '.'))
        except ValueError:
            mhz = 0.0
        if mhz:
            break
    if not mhz:
        import re
        m = re.search(r'([\d.]+)\s*GHz', info.get('Model name', ''))
        mhz = float(m.group(1)) * 1e3 if m else 0.0
    return cores or (os.cpu_count() or 1), os.cpu_count() or 1, info.get('Model name', 'unknown'), mhz


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--workload', default='scannet-100k')
    ap.add_argument('--config', default='fcaf3d_scannet-3d-18class')
    ap.add_argument('--voxel-size', type=float, default=0.02)
    ap.add_argument('--levels', type=int, default=4)
    ap.add_argument('--points', type=int, default=0)
    ap.add_argument('--reps', type=int, default=3)
    args = ap.parse_args()
    import torch
    import fcaf3d_amd as fa
    from fcaf3d_amd.synthetic import WORKLOADS, make_scene
    from oracle import bev, conv_c, me_oracle as mo, model_oracle as MO
    phys, logical, cpu_name, mhz = host()
    torch.set_num_threads(int(os.environ.get('OMP_NUM_THREADS', phys)))
    cfg = fa.get_config(args.config, voxel_size=args.voxel_size)
    m = cfg.model
    if args.levels != 4:
        m.backbone['n_outs'] = args.levels
        m.neck_with_head['in_channels'] = (64, 128, 256, 512)[:args.levels]
        m.neck_with_head.assigner['n_scales'] = args.levels
    torch.manual_seed(0)
    model = fa.build_detector(m, train_cfg=m.get('train_cfg'), test_cfg=m.get('test_cfg'))
    P = {k: v.detach().clone().requires_grad_(v.dtype.is_floating_point) for k, v in model.state_dict().items()}
    kw = dict(WORKLOADS[args.workload]['scene'])
    if args.points:
        kw['n_points'] = args.points
    p, g, l = make_scene(999, **kw)

    layers = []

    class CConv(torch.autograd.Function):
        """mo.conv with the three passes in C / OpenMP (oracle/conv_oracle.c)"""

        @staticmethod
        def forward(ctx, feats, weight, nbr):
            ctx.save_for_backward(feats, weight)
            ctx.nbr = nbr
            layers.append((nbr, feats.shape[0], weight.shape[1], weight.shape[2]))
            return torch.from_numpy(conv_c.conv_fwd(feats.detach().numpy(), weight.detach().numpy(), nbr))

        @staticmethod
        def backward(ctx, go):
            feats, weight = ctx.saved_tensors
            go = go.contiguous().numpy()
            gin = torch.from_numpy(conv_c.conv_dgrad(go, weight.detach().numpy(), ctx.nbr, feats.shape[0]))
            gw = torch.from_numpy(conv_c.conv_wgrad(feats.detach().numpy(), go, ctx.nbr, weight.shape[1], weight.shape[2]))
            return gin, gw, None

    conv0 = mo.conv
    mo.conv = lambda feats, weight, nbr: (CConv.apply(feats, weight, nbr) if weight.shape[1] >= 8 else conv0(feats, weight, nbr))
    try:
        MO.forward_train(P, m, [p], [g], [l])                  # warm-up: thread pools, page faults
        layers.clear()
        for v in P.values():
            v.grad = None
        t0 = time.time()
        losses = MO.forward_train(P, m, [p], [g], [l])
        sum(losses.values()).backward()
        dt_step = time.time() - t0
    finally:
        mo.conv = conv0
    rng = np.random.default_rng(0)
    data, flops = [], 0.0
    for nbr, n_in, Cin, Cout in layers:
        K, n_out = nbr.shape
        data.append((nbr, n_in, Cin, Cout, rng.standard_normal((n_in, Cin), dtype=np.float32),
                     rng.standard_normal((K, Cin, Cout), dtype=np.float32) * 0.05, rng.standard_normal((n_out, Cout), dtype=np.float32)))
        flops += 3 * 2.0 * float((nbr >= 0).sum()) * Cin * Cout
    times = []
    for _ in range(max(args.reps, 1)):
        t0 = time.time()
        for nbr, n_in, Cin, Cout, x, w, go in data:
            conv_c.conv_fwd(x, w, nbr)
            conv_c.conv_dgrad(go, w, nbr, n_in)
            conv_c.conv_wgrad(x, go, nbr, Cin, Cout)
        times.append(time.time() - t0)
    dt_c = float(np.median(times))
    lib = bev.lib()
    lib.oc_simd_width.restype = __import__('ctypes').c_int
    simd = int(lib.oc_simd_width())
    threads = conv_c.num_threads()
    peak = min(threads, phys) * simd * 2 * 2 * mhz / 1e3 if mhz else None         # GFLOP/s: 2 FMA pipes x 2 flops
    print(json.dumps(dict(
        value=round(1.0 / dt_step, 5), unit='scenes/s', cores=threads, physical_cores=phys, logical_cpus=logical, cpu=cpu_name,
        kind='port', step_s=round(dt_step, 2),
        sample=f'1 scene of {kw["n_points"]} pts, whole forward_train + backward once after one warm-up pass ({dt_step:.2f} s): '
               f'oracle/model_oracle.py (MinkowskiEngine CPU algorithm restated: hash-map kernel maps, per offset gather-GEMM-scatter, '
               f'torch norms / assigner / losses) with the {len(data)} sparse convolutions x (fwd, dgrad, wgrad) in C / OpenMP SIMD '
               f'kernels (oracle/conv_oracle.c, {simd * 32}-bit vectors), {threads} threads',
        conv_only=dict(seconds=round(dt_c, 3), gflops=round(flops / dt_c / 1e9, 1), gflop=round(flops / 1e9, 1),
                       host_peak_gflops=round(peak, 0) if peak else None,
                       frac_of_host_peak=round(flops / dt_c / 1e9 / peak, 4) if peak else None,
                       what=f'the convolutions alone, median of {len(times)} runs'))))


if __name__ == '__main__':
    main()
