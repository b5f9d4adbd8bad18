"""Diagnostic (CPU, oracle maps): how much MFMA work the dense neighbour table issues for absent neighbours, per backbone
level, for natural / mask-sorted row order and 128/64/32-row skip granularity.   python tests/diag_density.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fcaf3d_amd.synthetic import make_scene  # noqa: E402
from oracle import me_oracle as mo  # noqa: E402


def stats(nbr, name):
    K, n = nbr.shape
    present = nbr >= 0
    dens = present.sum() / (K * n)
    masks = (present.astype(np.int64) << np.arange(K)[:, None]).sum(0)
    out = [f'{name:10s} n={n:7d} density {dens:.3f}']
    for label, order in (('nat', np.arange(n)), ('sort', np.argsort(masks, kind='stable'))):
        p = present[:, order]
        for g in (128, 64, 32):
            pad = (-n) % g
            pp = np.pad(p, ((0, 0), (0, pad)))
            act = pp.reshape(K, -1, g).any(2).sum() * g / (K * n)
            out.append(f'{label}{g}: {act:.3f}')
    print('  '.join(out), flush=True)


def main():
    B = 4
    cs = []
    for b in range(B):
        pts, _, _ = make_scene(b)
        q = np.floor(pts[:, :3] / np.float32(0.02)).astype(np.int32)
        cs.append(np.concatenate([np.full((len(q), 1), b, np.int32), q], 1))
    c0, _, _ = mo.unique_first(np.concatenate(cs))
    c = mo.stride_coords(c0, 1, 2)
    c = mo.stride_coords(c, 2, 2)
    stride = 4
    for lv in range(4):
        nxt = mo.stride_coords(c, stride, 2)
        stats(mo.kernel_map(c, nxt, mo.kernel_offsets(3, stride)), f'L{lv + 1} s2')
        stride *= 2
        c = nxt
        stats(mo.kernel_map(c, c, mo.kernel_offsets(3, stride)), f'L{lv + 1} s1')


if __name__ == '__main__':
    main()
