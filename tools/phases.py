"""GPU time per phase of the training step in PIPELINED mode (tools/, not product): HIP events recorded on the main stream at
the phase boundaries of bench.py's step, no synchronisation inside the measured steps — the time between two events is how
long the main stream (the step's dependent chain) took to get from one boundary to the next, with the coordinate /
weight-gradient / head streams running beside it as in the benchmark.

  python tools/phases.py [--steps 12] [bench.py flags...]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                                    # noqa: E402
import fcaf3d_amd.functional as Fn                              # noqa: E402
from fcaf3d_amd.runner import TrainStep, parse_losses           # noqa: E402


def main():
    steps = 12
    if '--steps' in sys.argv:
        i = sys.argv.index('--steps')
        steps = int(sys.argv[i + 1])
        del sys.argv[i:i + 2]
    args = bench.parse()
    dev = torch.device('cuda:0')
    model, cfg = bench.build_model(args)
    model = model.to(dev).train()
    model.async_maps = True
    model.inputs_resident = True
    Fn.WGRAD_ASYNC = not args.no_wgrad_overlap
    model.spatial_sort = args.spatial_sort
    tr = TrainStep.from_config(model, cfg)
    batches = bench.make_batches(args, 0, dev)
    names = ['forward + loss', 'backward', 'optimizer (clip + AdamW)', 'gap to next step']
    marks, cur = [], [None]

    def ev(stream=None):
        e = torch.cuda.Event(enable_timing=True)
        e.record(stream) if stream is not None else e.record()
        return e

    # boundaries of TrainStep.__call__ (runner.py:223): after the run-ahead bound, after forward_train + parse_losses, after
    # loss.backward(), at return — recorded on the main stream; the step itself is TrainStep's, lookahead included
    import fcaf3d_amd.runner as R
    bound0, parse0, fin0 = tr._bound_run_ahead, R.parse_losses, tr.averager.finish

    def bound(*a, **k):
        bound0(*a, **k)
        if cur[0] is not None:
            cur[0].append(ev())

    def parse(losses):
        r = parse0(losses)
        if cur[0] is not None:
            cur[0].append(ev())
        return r

    def fin():
        if cur[0] is not None:
            cur[0].append(ev())
        return fin0()
    tr._bound_run_ahead, R.parse_losses, tr.averager.finish = bound, parse, fin
    side = Fn.wgrad_stream(dev)
    img = []

    def step(i, rec):
        cur[0] = [] if rec else None
        tr(batches[i % 2], next_batch=batches[(i + 1) % 2] if not args.no_lookahead else None)
        if rec:
            cur[0].append(ev())
            img.append(ev(side))                 # the weight images of the next step are ready (weight-gradient stream)
            marks.append(cur[0])

    for i in range(6):
        step(i, False)
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for i in range(steps):
        step(i, True)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / steps * 1e3
    tot = [0.0] * 4
    for j, m in enumerate(marks):
        assert len(m) == 4, len(m)
        for k in range(3):
            tot[k] += m[k].elapsed_time(m[k + 1])
        if j + 1 < len(marks):
            tot[3] += m[3].elapsed_time(marks[j + 1][0])
    n = len(marks)
    for k in range(4):
        print(f'{names[k]:28s} {tot[k] / (n if k < 3 else n - 1):8.3f} ms')
    print(f'{"sum":28s} {sum(tot[:3]) / n + tot[3] / (n - 1):8.3f} ms   (wall {wall:.3f} ms per step)')
    print(f'{"images ready after step end":28s} {sum(m[3].elapsed_time(e) for m, e in zip(marks, img)) / n:8.3f} ms')


if __name__ == '__main__':
    main()
