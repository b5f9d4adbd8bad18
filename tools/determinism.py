"""Run-to-run determinism of forward_train + backward on the GPU (tools/, not product): the same batch twice through the
same model must give bitwise the same losses and gradients (no float atomics anywhere in the path).
    python tools/determinism.py [--points 30000] [--scenes 2] [--levels 4] [--reps 3] [--bench-streams]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fcaf3d_amd as fa                                   # noqa: E402
from fcaf3d_amd.synthetic import make_scene               # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--points', type=int, default=30000)
    ap.add_argument('--scenes', type=int, default=2)
    ap.add_argument('--levels', type=int, default=4)
    ap.add_argument('--reps', type=int, default=3)
    ap.add_argument('--bench-streams', action='store_true', help="bench.py's stream layout: weight gradients and coordinates on their own streams")
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    cfg = fa.get_config('fcaf3d_scannet-3d-18class', voxel_size=0.02)
    m = cfg.model
    m.backbone['n_outs'] = a.levels
    m.neck_with_head['in_channels'] = (64, 128, 256, 512)[:a.levels]
    m.neck_with_head.assigner['n_scales'] = a.levels
    model = fa.build_detector(m, train_cfg=m.get('train_cfg'), test_cfg=m.get('test_cfg')).to(dev).train()
    if a.bench_streams:
        import fcaf3d_amd.functional as Fn
        Fn.WGRAD_ASYNC = True
        model.async_maps = True
        model.inputs_resident = True
    sc = [make_scene(100 + i, n_points=a.points) for i in range(a.scenes)]
    batch = dict(points=[torch.from_numpy(s[0]).to(dev) for s in sc],
                 gt_bboxes_3d=[fa.DepthInstance3DBoxes(torch.from_numpy(s[1]), origin=(.5, .5, .5)) for s in sc],
                 gt_labels_3d=[torch.from_numpy(s[2]).to(dev) for s in sc],
                 img_metas=[dict(box_type_3d=fa.DepthInstance3DBoxes) for _ in sc])
    runs = []
    for r in range(a.reps):
        model.zero_grad(set_to_none=True)
        losses = model(return_loss=True, **batch)
        sum(losses.values()).backward()
        torch.cuda.synchronize()
        runs.append(({k: float(v) for k, v in losses.items()}, {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}))
    ok = True
    for r in range(1, a.reps):
        if runs[r][0] != runs[0][0]:
            ok = False
            print(f'run {r}: losses differ', runs[0][0], runs[r][0])
        bad = [(k, float((runs[r][1][k] - g).abs().max()), float(g.abs().max())) for k, g in runs[0][1].items()
               if not torch.equal(runs[r][1][k], g)]
        if bad:
            ok = False
            print(f'run {r}: {len(bad)} of {len(runs[0][1])} gradients differ; first:', bad[:6])
    import hashlib
    h = hashlib.sha1()
    for k in sorted(runs[0][1]):
        h.update(runs[0][1][k].cpu().numpy().tobytes())
    print('deterministic' if ok else 'NOT deterministic', '| digest', h.hexdigest()[:16], '| losses', runs[0][0])


if __name__ == '__main__':
    main()
