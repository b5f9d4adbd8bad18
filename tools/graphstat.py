"""Diagnostic: autograd node census of one training step + aten-op census (torch.profiler) — where the small
torch kernels (fill / add / copy) in the kernel trace come from."""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    args = bench.parse()
    dev = torch.device('cuda:0')
    model, cfg = bench.build_model(args)
    model = model.to(dev).train()
    model.async_maps = True
    model.inputs_resident = True
    b = bench.make_batches(args, 0, dev, n_batches=1)[0]

    def fwd():
        x = model.extract_feat(b['points'], b['img_metas'])
        x = [list(v) for v in x]
        losses = model.neck_with_head.loss(*x, b['gt_bboxes_3d'], b['gt_labels_3d'], b['img_metas'])
        return sum(losses.values())

    loss = fwd()
    seen, stack, cnt = set(), [loss.grad_fn], collections.Counter()
    while stack:
        f = stack.pop()
        if f is None or f in seen:
            continue
        seen.add(f)
        cnt[type(f).__name__] += 1
        stack.extend(n for n, _ in f.next_functions)
    print('autograd nodes:', sum(cnt.values()))
    for k, v in cnt.most_common(40):
        print(f'  {v:5d} {k}')
    loss.backward()
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        loss = fwd()
        loss.backward()
        torch.cuda.synchronize()
    rows = [(e.key, e.count, e.self_device_time_total) for e in prof.key_averages() if e.key.startswith('aten::')]
    rows.sort(key=lambda r: -r[2])
    print('aten ops by self device time (us):')
    for k, c, t in rows[:30]:
        print(f'  {t:9.0f} us {c:5d} x {k}')


if __name__ == '__main__':
    main()
