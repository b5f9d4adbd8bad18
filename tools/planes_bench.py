"""Diagnostic (tools/): the split-bf16 convolution with the gathered operand split IN the kernel (k_conv_x6) against the same
convolution on PRE-SPLIT PLANES (fc_x6_planes + k_conv_x6p, flags bit27), on the kernel maps of a benchmark batch: bitwise
equality of the results and microseconds per launch (HIP events, 10 repetitions after 2 warm-ups), plus the cost of building
the planes.

    python tools/planes_bench.py [--batch 8]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import fcaf3d_amd._lib as L  # noqa: E402
import fcaf3d_amd.functional as Fn  # noqa: E402
from fcaf3d_amd.sparse import SparseTensor  # noqa: E402

APL = 1 << 27


def timed(fn, reps=10):
    for _ in range(2):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / reps


def main():
    args = bench.parse()
    dev = torch.device('cuda:0')
    model, _ = bench.build_model(args)
    model = model.to(dev).train()
    batch = bench.make_batches(args, 0, dev, n_batches=1)[0]
    coords, feats = model.voxelize(batch['points'])
    x = SparseTensor(feats, coordinates=coords, batch_size=len(batch['points']))
    cm0 = x.cmap
    m1 = cm0.strided(2); m2 = m1.strided(2)
    lv, prev = [], m2
    for _ in range(4):
        mi = prev.strided(2)
        lv.append((prev, mi))
        prev = mi
    g2 = lv[3][1].generate(); g1 = g2.generate(); g0 = g1.generate()
    cases = []
    for name, (pm, mi), C in zip(('L1', 'L2', 'L3', 'L4'), lv, (64, 128, 256, 512)):
        cases.append((f'{name} same {C}->{C}', mi.kernel_map(mi, 3), C, C))
    cases.append((f'L2 down 64->128', lv[1][0].kernel_map(lv[1][1], 3), 64, 128))
    cases += [('g2 256->256', g2.kernel_map(g2, 3), 256, 256), ('g1 128->128', g1.kernel_map(g1, 3), 128, 128),
              ('g0 64->64', g0.kernel_map(g0, 3), 64, 64), ('g0 64->128', g0.kernel_map(g0, 3), 64, 128),
              ('g0 128->64 (dgrad shape)', g0.kernel_map(g0, 3), 128, 64)]
    torch.manual_seed(0)
    print(f'{"case":28s} {"rows":>8s} {"route":>6s} {"in-kernel us":>13s} {"planes us":>10s} {"dma128 us":>10s} {"dma256 us":>10s} {"to_planes us":>12s} {"GFLOP":>8s} {"TF in-k":>8s} {"TF dma":>9s}  equal')
    for name, km, Cin, Cout in cases:
        n_in, n_out = km.n_in, km.n_out
        f = torch.relu(torch.randn((n_in, Cin), device=dev))
        w = torch.randn((km.K, Cin, Cout), device=dev) * 0.05
        img = Fn._x6_image(w, False)
        fl = Fn.FLAGS | Fn.CONV_X6
        planes = torch.empty(L.query('fc_x6_planes_bytes', n_in, Cin), dtype=torch.uint8, device=dev)

        def to_planes():
            L.call('fc_x6_planes', L.ptr(f), L.ptr(planes), n_in, Cin, L.stream())
        to_planes()
        out_a = torch.empty((n_out, Cout), device=dev)
        out_b = torch.empty((n_out, Cout), device=dev)
        pairs = Fn._pair_conv(km, n_out, Cin, Cout)
        if pairs:
            lists, tiles = km.pairs(), km.pair_tiles()
            run_a = lambda: Fn._conv_pairs(f, img, lists, out_a, n_in, n_out, km.K, Cin, Cout, tiles, flags=fl)
            run_b = lambda: Fn._conv_pairs(planes, img, lists, out_b, n_in, n_out, km.K, Cin, Cout, tiles, flags=fl | APL)
            P = float(lists[3].sum().item())
        else:
            nbr, oidx = km.sorted_fwd()
            run_a = lambda: Fn._conv_fwd(f, img, nbr, out_a, n_in, n_out, km.K, Cin, Cout, oidx, flags=fl)
            run_b = lambda: Fn._conv_fwd(planes, img, nbr, out_b, n_in, n_out, km.K, Cin, Cout, oidx, flags=fl | APL)
            P = float((km.nbr >= 0).sum().item())
        ta, tb, tp = timed(run_a), timed(run_b), timed(to_planes)
        gf = 2 * P * Cin * Cout / 1e9
        td, eq = [float('nan')] * 2, [None] * 2
        if not pairs:
            for q, bm in enumerate((128, 256)):
                out_c = torch.empty((n_out, Cout), device=dev)
                run_c = lambda: L.call('fc_conv_x6d', L.ptr(planes), L.ptr(img), L.ptr(nbr), L.ptr(oidx) if oidx is not None else None,
                                       L.ptr(out_c), n_out, km.K, Cin, Cout, bm, L.stream())
                td[q] = timed(run_c)
                eq[q] = bool(torch.equal(out_a, out_c))
        best = min(t for t in td + [float('inf')] if t == t)
        print(f'{name:28s} {n_out:8d} {"pairs" if pairs else "table":>6s} {ta:13.1f} {tb:10.1f} {td[0]:10.1f} {td[1]:10.1f} {tp:12.1f} {gf:8.2f} {gf / ta * 1e3:8.1f} {gf / best * 1e3:9.1f}  '
              f'{bool(torch.equal(out_a, out_b))} {eq}')


if __name__ == '__main__':
    main()
