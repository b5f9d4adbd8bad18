"""Diagnostic: where does the HOST spend its time in one training step? (cProfile + phase timers with syncs)"""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    sys.argv = [sys.argv[0]] + sys.argv[1:]
    args = bench.parse()
    dev = torch.device('cuda:0')
    model, cfg = bench.build_model(args)
    model = model.to(dev).train()
    model.async_maps = True
    model.inputs_resident = True
    batches = bench.make_batches(args, 0, dev)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, fused=True)

    def step(i, timers=None):
        b = batches[i % 2]
        t = [time.perf_counter()]

        def mark():
            if timers is not None:
                torch.cuda.synchronize()
            t.append(time.perf_counter())
        opt.zero_grad(set_to_none=True)
        x = model.extract_feat(b['points'], b['img_metas'])
        x = [list(v) for v in x]
        mark()
        losses = model.neck_with_head.loss(*x, b['gt_bboxes_3d'], b['gt_labels_3d'], b['img_metas'])
        loss = sum(losses.values())
        mark()
        loss.backward()
        mark()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 10.0)
        opt.step()
        mark()
        if timers is not None:
            for j in range(4):
                timers[j] += t[j + 1] - t[j]

    for i in range(3):
        step(i)
    torch.cuda.synchronize()
    timers = [0.0] * 4
    for i in range(4):
        step(i, timers)
    print('synced phase ms/step: extract_feat %.1f | loss %.1f | backward %.1f | clip+adamw %.1f' % tuple(1e3 * v / 4 for v in timers))
    # pure host time (no syncs inside): how long until all work is ENQUEUED
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(4):
        step(i)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print('enqueue-only ms/step %.1f ; drained ms/step %.1f' % (1e3 * (t1 - t0) / 4, 1e3 * (t2 - t0) / 4))
    pr = cProfile.Profile()
    pr.enable()
    for i in range(3):
        step(i)
    torch.cuda.synchronize()
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats('cumulative').print_stats(45)
    st.sort_stats('tottime').print_stats(25)


if __name__ == '__main__':
    main()
