"""Diagnostic: sweep tile / split configurations of the MFMA sparse-conv kernels on the REAL kernel maps of
the benchmark workload.   python tools/convbench.py [--batch 8] [--wgrad]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import fcaf3d_amd.functional as Fn  # noqa: E402
from fcaf3d_amd import _lib as L  # noqa: E402


def timeit(fn, reps=8):
    fn(); fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3     # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--wgrad', action='store_true')
    ap.add_argument('--only', default='')
    ap.add_argument('--bm256', action='store_true')
    ap.add_argument('--unsorted', action='store_true', help='use the plain neighbour table (no occupancy-mask row order)')
    ap.add_argument('--flags', type=int, default=0, help='extra flags for the default run (bit16: BK=32, bit17: BK=64)')
    ap.add_argument('--default-only', action='store_true')
    ap.add_argument('--pairconv', action='store_true', help='forward: per-offset gather-GEMM over the pair lists')
    ap.add_argument('--pairs', action='store_true', help='with --wgrad: reduce over the exact pair lists')
    a = ap.parse_args()
    sys.argv = [sys.argv[0], '--batch', str(a.batch)]
    args = bench.parse()
    dev = torch.device('cuda:0')
    model, cfg = bench.build_model(args)
    model = model.to(dev)
    batch = bench.make_batches(args, 0, dev, n_batches=1)[0]
    coords, feats = model.voxelize(batch['points'])
    from fcaf3d_amd.sparse import SparseTensor
    x = SparseTensor(feats, coordinates=coords, batch_size=args.batch)
    model.plan_maps(x.cmap)
    cm0 = x.cmap
    m1 = cm0.strided(2); m2 = m1.strided(2)
    levels, prev = [], m2
    for _ in range(4):
        mi = prev.strided(2); levels.append((prev, mi)); prev = mi
    lv = [m for _, m in levels]
    necks = []
    xm = lv[-1]
    for i in (2, 1, 0):
        g = xm.generate(); u, _, _ = lv[i].union(g); necks.append((g, u)); xm = u
    cases = [
        ('L1 k3s1 64->64', lv[0].kernel_map(lv[0], 3), 64, 64),
        ('L2 k3s1 128->128', lv[1].kernel_map(lv[1], 3), 128, 128),
        ('L3 k3s1 256->256', lv[2].kernel_map(lv[2], 3), 256, 256),
        ('L4 k3s1 512->512', lv[3].kernel_map(lv[3], 3), 512, 512),
        ('L4 out 512->128', lv[3].kernel_map(lv[3], 3), 512, 128),
        ('N2 k3s1 256->256', necks[0][1].kernel_map(necks[0][1], 3), 256, 256),
        ('N2 out 256->128', necks[0][1].kernel_map(necks[0][1], 3), 256, 128),
        ('N1 k3s1 128->128', necks[1][1].kernel_map(necks[1][1], 3), 128, 128),
        ('N0 k3s1 64->64', necks[2][1].kernel_map(necks[2][1], 3), 64, 64),
        ('N0 out 64->128', necks[2][1].kernel_map(necks[2][1], 3), 64, 128),
        ('N0 dgrad 128->64', necks[2][1].kernel_map(necks[2][1], 3), 128, 64),
        ('L1 k3s2 64->64', levels[0][0].kernel_map(levels[0][1], 3), 64, 64),
        ('L2 k3s2 64->128', levels[1][0].kernel_map(levels[1][1], 3), 64, 128),
    ]
    for name, km, Cin, Cout in cases:
        if a.only and a.only not in name:
            continue
        K = km.K
        km.sort_rows = not a.unsorted
        nbr_f, oidx = km.sorted_fwd()
        if a.wgrad:
            nbr_f, oidx = km.nbr, None
        pairs = km.n_pairs()
        xin = torch.randn(km.n_in, Cin, device=dev)
        w = torch.randn(K, Cin, Cout, device=dev)
        out = torch.empty(km.n_out, Cout, device=dev)
        gout = torch.randn(km.n_out, Cout, device=dev)
        gw = torch.empty_like(w)
        gflop = 2.0 * pairs * Cin * Cout / 1e9
        res = []
        for bm in (() if a.default_only else ((3,) if a.bm256 else (1, 2))):
            for bn in ((1, 2) if Cout % 128 == 0 else (1,)):
                for S in ((0,) if a.wgrad else (0, 1, 2, 3, 4, 6, 9, 14, 27)):
                    if a.wgrad and bm == 2 and (Cin % 128 or a.pairs):
                        continue
                    for Sw in ((0, 2, 4, 8, 16, 32, 64, 128) if a.wgrad else (0,)):
                        fl = (bm << 4) | (bn << 6) | ((S or Sw) << 8)
                        Fn.FLAGS = fl
                        try:
                            if a.wgrad and a.pairs:
                                wsb = L.query('fc_conv_wgrad_ws_bytes', km.n_out, K, Cin, Cout, fl)
                                ws = L.workspace(wsb, dev)
                                pi, po, _, cnt = km.pairs()
                                t = timeit(lambda: L.call('fc_conv_wgrad_pairs', L.ptr(xin), L.ptr(gout), L.ptr(pi), L.ptr(po), L.ptr(cnt),
                                                          L.ptr(gw), km.n_in, km.n_out, K, Cin, Cout, fl, L.ptr(ws), ws.numel(), L.stream()))
                            elif a.wgrad:
                                wsb = L.query('fc_conv_wgrad_ws_bytes', km.n_out, K, Cin, Cout, fl)
                                ws = L.workspace(wsb, dev)
                                t = timeit(lambda: L.call('fc_conv_wgrad', L.ptr(xin), L.ptr(gout), L.ptr(nbr_f), L.ptr(oidx), L.ptr(gw),
                                                          km.n_in, km.n_out, K, Cin, Cout, fl, L.ptr(ws), ws.numel(), L.stream()))
                            else:
                                t = timeit(lambda: Fn._conv_fwd(xin, w, nbr_f, out, km.n_in, km.n_out, K, Cin, Cout, oidx))
                        finally:
                            Fn.FLAGS = 0
                        res.append((t, {1: 64, 2: 128, 3: 256}[bm], bn * 64, S or Sw))
        Fn.FLAGS = a.flags
        if a.wgrad and a.pairs:
            wsb = L.query('fc_conv_wgrad_ws_bytes', km.n_out, K, Cin, Cout, 0)
            ws = L.workspace(wsb, dev)
            pi, po, _, cnt = km.pairs()
            t0 = timeit(lambda: L.call('fc_conv_wgrad_pairs', L.ptr(xin), L.ptr(gout), L.ptr(pi), L.ptr(po), L.ptr(cnt), L.ptr(gw),
                                       km.n_in, km.n_out, K, Cin, Cout, 0, L.ptr(ws), ws.numel(), L.stream()))
        elif a.wgrad:
            wsb = L.query('fc_conv_wgrad_ws_bytes', km.n_out, K, Cin, Cout, 0)
            ws = L.workspace(wsb, dev)
            t0 = timeit(lambda: L.call('fc_conv_wgrad', L.ptr(xin), L.ptr(gout), L.ptr(nbr_f), L.ptr(oidx), L.ptr(gw), km.n_in, km.n_out,
                                       K, Cin, Cout, 0, L.ptr(ws), ws.numel(), L.stream()))
        elif a.pairconv:
            lists = km.pairs()
            t0 = timeit(lambda: Fn._conv_pairs(xin, w, lists, out, km.n_in, km.n_out, K, Cin, Cout))
        else:
            t0 = timeit(lambda: Fn._conv_fwd(xin, w, nbr_f, out, km.n_in, km.n_out, K, Cin, Cout, oidx))
        if not res:
            res = [(t0, 0, 0, 0)]
        res.sort()
        best = ', '.join(f'{t:.0f}us({bm}x{bn},S{S})' for t, bm, bn, S in res[:4])
        print(f'{name:20s} n={km.n_out:7d} P={pairs:9d} {gflop:7.1f} GF  default {t0:7.0f}us = {gflop / t0 * 1e3:5.1f} TF | best: {best} '
              f'= {gflop / res[0][0] * 1e3:5.1f} TF', flush=True)


if __name__ == '__main__':
    main()
