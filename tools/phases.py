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
    tr = TrainStep.from_config(model, cfg)
    batches = bench.make_batches(args, 0, dev)
    names = ['forward (extract_feat)', 'loss', 'backward', 'optimizer', 'gap to next step']
    marks = []

    def ev():
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def step(i, rec):
        b = batches[i % 2]
        m = [ev()] if rec else None
        tr.optimizer.zero_grad(set_to_none=True)
        x = model.extract_feat(b['points'], b['img_metas'], (b['gt_bboxes_3d'], b['gt_labels_3d']))
        x = [list(v) for v in x]
        if rec: m.append(ev())
        losses = model.neck_with_head.loss(*x, b['gt_bboxes_3d'], b['gt_labels_3d'], b['img_metas'])
        loss = parse_losses(losses)
        if rec: m.append(ev())
        loss.backward()
        tr.averager.finish()
        if rec: m.append(ev())
        tr.optimizer.step(tr.max_norm)
        if rec:
            m.append(ev())
            marks.append(m)

    for i in range(4):
        step(i, False)
    torch.cuda.synchronize()
    for i in range(steps):
        step(i, True)
    torch.cuda.synchronize()
    tot = [0.0] * 5
    for j, m in enumerate(marks):
        for k in range(4):
            tot[k] += m[k].elapsed_time(m[k + 1])
        if j + 1 < len(marks):
            tot[4] += m[4].elapsed_time(marks[j + 1][0])
    n = len(marks)
    for k in range(5):
        print(f'{names[k]:28s} {tot[k] / (n if k < 4 else n - 1):8.3f} ms')
    print(f'{"sum":28s} {sum(tot[:4]) / n + tot[4] / (n - 1):8.3f} ms')


if __name__ == '__main__':
    main()
