"""Summarise a per-wave timeline written by `tools/nbench_trace --trace` (FC_TRACE build of the kernels).
record = 8 x u64: [block x|y|z|wave, xcc|units, t0 start, t1 prologue done, t2 first operands landed, t3 main loop done, -, t5 end]
times are wall_clock64() ticks (100 MHz)."""
import sys
import numpy as np


def main(path):
    r = np.fromfile(path, dtype=np.uint64).reshape(-1, 8)
    units = (r[:, 1] & 0xffffffff).astype(np.int64)
    xcc = (r[:, 1] >> 32).astype(np.int64)
    t = r[:, 2:8].astype(np.float64) * 0.01           # us
    t0 = t[:, 0].min()
    t -= t0
    live = units > 0
    span = t[:, 5].max()
    print(f'{path}: {len(r)} waves ({live.sum()} with work), kernel span {span:.1f} us, XCDs {sorted(set(xcc.tolist()))}')
    d = lambda a, b: (t[live, b] - t[live, a])
    for name, a, b in (('prologue', 0, 1), ('first operands', 1, 2), ('main loop', 2, 3), ('epilogue', 3, 5), ('whole wave', 0, 5)):
        x = d(a, b)
        print(f'  {name:15s} mean {x.mean():7.2f}  p50 {np.median(x):7.2f}  p95 {np.percentile(x, 95):7.2f}  max {x.max():7.2f} us')
    ml = d(2, 3)
    per_unit = ml / np.maximum(units[live], 1)
    print(f'  main loop per unit: mean {per_unit.mean() * 1e3:.0f} ns (16 MFMAs of a 64x64 unit = 1024 cycles = 427 ns @2.4 GHz alone on its SIMD)')
    print(f'  wave start times: p5 {np.percentile(t[:, 0], 5):.1f} p50 {np.median(t[:, 0]):.1f} p95 {np.percentile(t[:, 0], 95):.1f} max {t[:, 0].max():.1f} us')
    # how many waves are inside their main loop over time
    edges = np.linspace(0, span, 41)
    busy = [( (t[live, 2] <= e) & (t[live, 3] > e)).sum() for e in edges]
    print('  waves in main loop over time:', ' '.join(str(b) for b in busy))
    tot_units = units[live].sum()
    print(f'  total units {tot_units}; sum(main loop) / span / 1024 SIMDs = {ml.sum() / span / 1024:.2f} waves per SIMD in their main loop on average')


if __name__ == '__main__':
    for p in sys.argv[1:]:
        main(p)
