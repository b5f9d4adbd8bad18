"""The batched weight-image pass alone on the chip (tools/, not product): fc_x6_weight_images over every convolution kernel of the
benchmark's detector (what TrainStep rebuilds after each optimizer step), HIP-event timed.
    python tools/imgbench.py [--reps 30]        (FC_LIB=<other build> for an A/B)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                                    # noqa: E402
import fcaf3d_amd.functional as Fn                              # noqa: E402
from fcaf3d_amd.nn import MinkowskiConvolution                  # noqa: E402


def main():
    reps = 30
    if '--reps' in sys.argv:
        i = sys.argv.index('--reps')
        reps = int(sys.argv[i + 1])
        del sys.argv[i:i + 2]
    args = bench.parse()
    dev = torch.device('cuda:0')
    model, _ = bench.build_model(args)
    model = model.to(dev).train()
    ws = [m.kernel for m in model.modules() if isinstance(m, MinkowskiConvolution) and m.kernel.requires_grad]
    im = Fn.WeightImages(ws)
    for _ in range(3):
        im.build()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        im.build()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    n_w = sum(w.numel() for w in ws)
    print(f'{im.n} images of {len(ws)} kernels ({n_w / 1e6:.1f} M weights, {im.blocks} units): median {ts[len(ts) // 2]:.1f} us, min {ts[0]:.1f} us per build '
          f'(zero + amax + image pass); lib {os.environ.get("FC_LIB", "in-tree")}')


if __name__ == '__main__':
    main()
