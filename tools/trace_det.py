"""Where does a bench-mode training run first differ between two processes?  (tools/, not product)
Runs TrainStep in the bench's stream configuration for a few steps and records, WITHOUT host syncs, per step: checksums of
the assigned targets, the three losses, the gradient buffer and the parameter buffer.  Run it twice and diff the output.
    python tools/trace_det.py [--steps 6]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fcaf3d_amd as fa                                   # noqa: E402
import fcaf3d_amd.functional as Fn                        # noqa: E402
from fcaf3d_amd.runner import TrainStep                   # noqa: E402
from fcaf3d_amd.synthetic import make_scene               # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=6)
    ap.add_argument('--scenes', type=int, default=8)
    ap.add_argument('--points', type=int, default=100000)
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    cfg = fa.get_config('fcaf3d_scannet-3d-18class', voxel_size=0.02)
    m = cfg.model
    model = fa.build_detector(m, train_cfg=m.get('train_cfg'), test_cfg=m.get('test_cfg')).to(dev).train()
    model.async_maps = True
    model.inputs_resident = True
    Fn.WGRAD_ASYNC = os.environ.get('TD_WGRAD_ASYNC', '0') == '1'
    tr = TrainStep.from_config(model, cfg)
    sc = [make_scene(100 + i, n_points=a.points) for i in range(a.scenes)]
    batch = dict(points=[torch.from_numpy(s[0]).to(dev) for s in sc],
                 gt_bboxes_3d=[fa.DepthInstance3DBoxes(torch.from_numpy(s[1]).to(dev), origin=(.5, .5, .5)) for s in sc],
                 gt_labels_3d=[torch.from_numpy(s[2]).to(dev) for s in sc],
                 img_metas=[dict(box_type_3d=fa.DepthInstance3DBoxes) for _ in sc])
    head = model.neck_with_head
    rec = []
    diffs = []
    keep = {}
    t0 = head._targets

    def targets(*args, **kw):
        out = t0(*args, **kw)
        rec.append(('targets', torch.stack([out['ct'].double().sum(), out['bt'].double().sum(), out['labels'].double().sum(),
                                            out['inv_pos'].double().sum(), out['inv_den'].double().sum()])))
        return out
    head._targets = targets
    pm0 = model.plan_maps

    def plan_maps(cm0):
        out = pm0(cm0)
        # every coordinate set and kernel map reachable from the input set: checksums (the batch is the same every step)
        seen, stack, sums = set(), [cm0], []
        while stack:
            cm = stack.pop()
            if id(cm) in seen:
                continue
            seen.add(id(cm))
            sums.append(cm.coords.double().sum() + cm.n)
            for km in (k for per in cm._kmaps.values() for k in per.values()):
                sums.append(km.nbr.double().sum())
                for attr in ('nbr_t',):
                    v = getattr(km, attr, None)
                    if torch.is_tensor(v):
                        sums.append(v.double().sum())
                    elif isinstance(v, (tuple, list)):
                        sums += [t.double().sum() for t in v if torch.is_tensor(t)]
            stack += list(cm._strided.values())
            for a in ('_gen', '_generated', '_union'):
                v = getattr(cm, a, None)
                if v is not None and hasattr(v, 'coords'):
                    stack.append(v)
        if out:
            stack = list(out)
            for cm in out:
                if id(cm) not in seen:
                    seen.add(id(cm)); sums.append(cm.coords.double().sum() + cm.n)
                    for km in (k for per in cm._kmaps.values() for k in per.values()):
                        sums.append(km.nbr.double().sum())
        rec.append((f'maps ({len(sums)} tensors)', torch.stack([torch.stack(sums).sum(), torch.stack(sums).abs().max()])))
        return out
    model.plan_maps = plan_maps
    a0 = head.assigner.assign_batched

    def assign(pts, scene, level, cmaps, gtb, gtl):
        ins = torch.stack([pts.double().sum(), scene.double().sum(), level.double().sum(),
                           torch.cat([g.tensor for g in gtb]).double().sum(), torch.cat(list(gtl)).double().sum()])
        import fcaf3d_amd._lib as L
        B, Lv = len(gtb), len(cmaps)
        M = max(1, max(len(g) for g in gtb))
        key = (pts.device, L.stream())
        if os.environ.get('TD_PERSIST') == '1':          # the assigner's big inputs in buffers that are never freed
            if 'pts' not in keep:
                keep['pts'], keep['scene'], keep['level'] = torch.empty_like(pts), torch.empty_like(scene), torch.empty_like(level)
            pts, scene, level = keep['pts'].copy_(pts), keep['scene'].copy_(scene), keep['level'].copy_(level)
        r1 = a0(pts, scene, level, cmaps, gtb, gtl)
        w1 = L._ws_cache[key][:B * M * (Lv + 4) * 4].clone().view(torch.int32)
        r2 = a0(pts, scene, level, cmaps, gtb, gtl)          # the same call again, right behind the first on the same stream
        w2 = L._ws_cache[key][:B * M * (Lv + 4) * 4].clone().view(torch.int32)
        n = B * M
        segs = dict(counts=(0, n * Lv), best=(n * Lv, n * Lv + n), kth=(n * Lv + n, n * Lv + 2 * n), trig=(n * Lv + 2 * n, n * Lv + 4 * n))
        rec.append(('  ws mismatches counts/best/kth/trig', torch.stack([(w1[a:b] != w2[a:b]).sum().double() for a, b in segs.values()])))
        rec.append(('  ws sums call1 counts/best', torch.stack([w1[0:n * Lv].double().sum(), w1[n * Lv:n * Lv + n].double().sum()])))
        rec.append(('  ws sums call2 counts/best', torch.stack([w2[0:n * Lv].double().sum(), w2[n * Lv:n * Lv + n].double().sum()])))
        d = (r1[2] != r2[2]) | (r1[0] != r2[0])
        idx = torch.nonzero(d).flatten()
        n_d = d.sum()
        first = torch.where(n_d > 0, idx.min() if idx.numel() else torch.zeros((), dtype=torch.int64, device=d.device), torch.zeros((), dtype=torch.int64, device=d.device)) if False else None
        rec.append(('  rows differing between the two calls', torch.stack([n_d.double()])))
        diffs.append((idx, r1[2][idx], r2[2][idx], r1[0][idx], r2[0][idx], scene[idx], level[idx]))
        rec.append(('  assign inputs', ins))
        rec.append(('  assign first / second', torch.stack([r1[0].double().sum(), r1[2].double().sum(), r2[0].double().sum(),
                                                              r2[2].double().sum()])))
        return r1
    head.assigner.assign_batched = assign
    torch.cuda.synchronize()
    for step in range(a.steps):
        loss, losses = tr(batch)
        rec.append((f'step {step} losses', torch.stack([losses[k].detach().double() for k in sorted(losses)])))
        rec.append((f'step {step} grad/param', torch.stack([tr.flat.grad.double().sum(), tr.flat.grad.double().abs().sum(),
                                                            tr.flat.data.double().sum()])))
    torch.cuda.synchronize()
    for name, v in rec:
        print(name, ' '.join(f'{x:.17g}' for x in v.tolist()))
    for k, (idx, l1, l2, c1, c2, sc_, lv) in enumerate(diffs):
        if idx.numel():
            i = idx.tolist()
            print(f'call pair {k}: {len(i)} rows differ, index range {i[0]}..{i[-1]}, distinct 256-row blocks {len(set(x // 256 for x in i))}, '
                  f'scenes {sorted(set(sc_.tolist()))}, levels {sorted(set(lv.tolist()))}')
            for j in range(min(6, len(i))):
                print('   row', i[j], 'scene', int(sc_[j]), 'level', int(lv[j]), 'label', int(l1[j]), int(l2[j]), 'ct', float(c1[j]), float(c2[j]))


if __name__ == '__main__':
    main()
