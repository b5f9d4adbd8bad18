"""Generates tests/golden/*.npz by IMPORTING the reference's own pure-torch pieces
from /root/reference (build container only — the reference never travels).

Run:  python tests/golden/make_golden.py

What is imported from the reference (by file path, under stub modules for the
third-party packages that are absent here — MinkowskiEngine, mmdet, mmcv):
  * mmdet3d/models/dense_heads/fcaf3d_neck_with_head.py
      Fcaf3DAssigner.assign, compute_centerness, Fcaf3DNeckWithHead._bbox_pred_to_bbox
  * mmdet3d/core/bbox/structures/utils.py      rotation_3d_in_axis
  * mmdet3d/core/bbox/iou_calculators/iou3d_calculator.py   axis_aligned_bbox_overlaps_3d
  * mmdet3d/ops/rotated_iou/{oriented_iou_loss,box_intersection_2d}.py   cal_iou_3d
      (its un-vendored CUDA `sort_v` is replaced by an angular argsort; the polygon
       area is invariant to the start vertex)
  * oracle/_ref/pcdet_iou3d_cpu (compiled from mmdet3d/ops/pcdet_nms/src/iou3d_cpu.cpp
    by oracle/Makefile) -> boxes_iou_bev_cpu
Only inputs + outputs are stored.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


class _Reg:
    def register_module(self, *a, **k):
        return lambda cls: cls


def load_reference():
    np.int = int  # min_enclosing_box.py:53 uses the removed alias
    _stub('MinkowskiEngine')
    _stub('mmdet')
    _stub('mmdet.core', BaseAssigner=object, reduce_mean=lambda x: x, build_assigner=lambda c: None)
    _stub('mmdet.models')
    _stub('mmdet.models.builder', HEADS=_Reg(), build_loss=lambda c: None)
    _stub('mmdet.core.bbox')
    _stub('mmdet.core.bbox.builder', BBOX_ASSIGNERS=_Reg())
    _stub('mmdet.core.bbox.iou_calculators')
    _stub('mmdet.core.bbox.iou_calculators.builder', IOU_CALCULATORS=_Reg())
    _stub('mmcv')
    _stub('mmcv.cnn', Scale=object, bias_init_with_prob=lambda p: float(-np.log((1 - p) / p)))
    utils = _load('ref_box_utils', f'{REF}/mmdet3d/core/bbox/structures/utils.py')
    _stub('mmdet3d')
    _stub('mmdet3d.core')
    _stub('mmdet3d.core.bbox')
    _stub('mmdet3d.core.bbox.structures', rotation_3d_in_axis=utils.rotation_3d_in_axis,
          get_box_type=None)
    _stub('mmdet3d.ops')
    _stub('mmdet3d.ops.pcdet_nms', pcdet_nms_gpu=None, pcdet_nms_normal_gpu=None)
    head = _load('ref_head', f'{REF}/mmdet3d/models/dense_heads/fcaf3d_neck_with_head.py')
    # aligned IoU
    _stub('mmdet3d.core.bbox.iou_calculators')
    src = open(f'{REF}/mmdet3d/core/bbox/iou_calculators/iou3d_calculator.py').read()
    ns = {'torch': torch}
    start = src.index('def axis_aligned_bbox_overlaps_3d')
    exec(compile(src[start:], 'iou3d_calculator.py', 'exec'), ns)
    # rotated IoU package with a python sort_v
    pkg = types.ModuleType('rotated_iou'); pkg.__path__ = [f'{REF}/mmdet3d/ops/rotated_iou']
    sys.modules['rotated_iou'] = pkg
    cu = types.ModuleType('rotated_iou.cuda_op'); cu.__path__ = []
    sys.modules['rotated_iou.cuda_op'] = cu
    _stub('rotated_iou.cuda_op.cuda_ext', sort_v=_sort_v_py)
    _load('rotated_iou.box_intersection_2d', f'{REF}/mmdet3d/ops/rotated_iou/box_intersection_2d.py')
    _load('rotated_iou.min_enclosing_box', f'{REF}/mmdet3d/ops/rotated_iou/min_enclosing_box.py')
    riou = _load('rotated_iou.oriented_iou_loss', f'{REF}/mmdet3d/ops/rotated_iou/oriented_iou_loss.py')
    return head, utils, ns['axis_aligned_bbox_overlaps_3d'], riou


def _sort_v_py(vertices, mask, num_valid):
    """Angular sort standing in for the un-vendored CUDA sort_v (Appendix D)."""
    v = vertices.detach().numpy(); m = mask.numpy(); nv = num_valid.numpy()
    B, N = nv.shape
    out = np.zeros((B, N, 9), np.int64)
    for b in range(B):
        for n in range(N):
            pad = 8 + int(np.argmin(m[b, n, 8:]))         # first invalid intersection slot
            ids = np.nonzero(m[b, n])[0]
            if len(ids) < 3:
                out[b, n] = pad
                continue
            ang = np.arctan2(v[b, n, ids, 1], v[b, n, ids, 0])
            order = ids[np.argsort(ang, kind='stable')]
            # drop near-duplicates (identical boxes put each corner in twice)
            keep = [order[0]]
            for i in order[1:]:
                if np.abs(v[b, n, i] - v[b, n, keep[-1]]).max() > 1e-6:
                    keep.append(i)
            if len(keep) > 1 and np.abs(v[b, n, keep[-1]] - v[b, n, keep[0]]).max() <= 1e-6:
                keep.pop()
            keep = keep[:8]
            row = keep + [keep[0]] + [pad] * (8 - len(keep))
            out[b, n] = row
    return torch.from_numpy(out)


class _GT:
    """minimal GT container: bottom-centre tensor like DepthInstance3DBoxes"""
    def __init__(self, gravity_boxes):
        t = torch.as_tensor(gravity_boxes, dtype=torch.float32).clone()
        self.gravity_center = t[:, :3].clone()
        t[:, 2] -= t[:, 5] / 2
        self.tensor = t
        self.volume = t[:, 3] * t[:, 4] * t[:, 5]

    def __len__(self):
        return len(self.tensor)


def gen_assigner(head, out):
    from fcaf3d_amd.synthetic import make_scene
    cases = {}
    for ci, (seed, n_scales, rotated) in enumerate([(0, 4, False), (1, 2, False), (2, 1, False),
                                                    (3, 4, True), (4, 2, True)]):
        pts, gt, labels = make_scene(seed, n_points=6000, n_boxes=7, rotated=rotated)
        rng = np.random.default_rng(100 + seed)
        levels = []
        for l in range(n_scales):
            step = 0.16 * 2 ** l
            q = np.unique(np.floor(pts[:, :3] / step), axis=0) * step
            levels.append(torch.from_numpy(q[rng.permutation(len(q))].astype(np.float32)))
        a = head.Fcaf3DAssigner(limit=27, topk=18, n_scales=n_scales)
        ct, bt, lb = a.assign(levels, _GT(gt), torch.from_numpy(labels))
        cases[f'c{ci}_n_scales'] = np.int64(n_scales)
        cases[f'c{ci}_gt'] = gt; cases[f'c{ci}_labels'] = labels
        for l, p in enumerate(levels):
            cases[f'c{ci}_points{l}'] = p.numpy()
        cases[f'c{ci}_centerness'] = ct.numpy(); cases[f'c{ci}_bbox_targets'] = bt.numpy()
        cases[f'c{ci}_assigned'] = lb.numpy()
        print('assigner case', ci, 'positives', int((lb >= 0).sum()))
    cases['n_cases'] = np.int64(5)
    np.savez_compressed(out, **cases)


def gen_decode(head, out):
    rng = np.random.default_rng(7)
    pts = torch.from_numpy(rng.uniform(0, 6, (64, 3)).astype(np.float32))
    d = {'points': pts.numpy()}
    p6 = torch.from_numpy(np.exp(rng.normal(0, 0.5, (64, 6))).astype(np.float32))
    p8 = torch.cat([p6, torch.from_numpy(rng.normal(0, 1, (64, 2)).astype(np.float32))], 1)
    d['pred6'] = p6.numpy(); d['pred8'] = p8.numpy()
    f = head.Fcaf3DNeckWithHead._bbox_pred_to_bbox
    d['out6'] = f(types.SimpleNamespace(yaw_parametrization='fcaf3d'), pts, p6).numpy()
    for mode in ('fcaf3d', 'sin-cos'):
        d[f'out8_{mode}'] = f(types.SimpleNamespace(yaw_parametrization=mode), pts, p8).numpy()
    d['out7_naive'] = f(types.SimpleNamespace(yaw_parametrization='naive'), pts, p8[:, :7]).numpy()
    bt = torch.from_numpy(np.exp(rng.normal(0, 1, (50, 7))).astype(np.float32))
    d['cent_in'] = bt.numpy(); d['cent_out'] = head.compute_centerness(bt).numpy()
    np.savez_compressed(out, **d)


def _rand_boxes(rng, n, rotated):
    c = rng.uniform(0, 3, (n, 3)); s = rng.uniform(0.3, 2.0, (n, 3))
    yaw = rng.uniform(-np.pi, np.pi, (n, 1)) if rotated else np.zeros((n, 1))
    return np.concatenate([c, s, yaw], 1).astype(np.float32)


def gen_iou(aiou, riou, out):
    rng = np.random.default_rng(11)
    d = {}
    # aligned: (cx,cy,cz,w,l,h) both, the loss transforms to corners (iou3d_loss.py:21-35)
    a = _rand_boxes(rng, 200, False)[:, :6]; b = a + rng.normal(0, 0.3, a.shape).astype(np.float32)
    b[:, 3:] = np.abs(b[:, 3:]) + 0.1
    b[:5] = a[:5]                        # identical
    b[5:10, :3] += 10                    # disjoint
    pa = torch.from_numpy(a).requires_grad_(True); pb = torch.from_numpy(b)

    def tr(x):
        return torch.cat([x[:, :3] - x[:, 3:6] / 2, x[:, :3] + x[:, 3:6] / 2], 1)
    iou = aiou(tr(pa), tr(pb), is_aligned=True)
    w = torch.from_numpy(rng.random(200).astype(np.float32))
    ((1 - iou) * w).sum().backward()
    d.update(al_pred=a, al_target=b, al_w=w.numpy(), al_iou=iou.detach().numpy(), al_grad=pa.grad.numpy())
    # rotated
    a = _rand_boxes(rng, 300, True); b = a + rng.normal(0, 0.25, a.shape).astype(np.float32)
    b[:, 3:6] = np.abs(b[:, 3:6]) + 0.1
    b[:4] = a[:4]                                             # identical
    b[4:8, :3] += 10                                          # disjoint
    b[8:12] = a[8:12]; b[8:12, 3:6] *= 0.5                    # contained
    a[12:16, 6] = 0; b[12:16] = a[12:16]; b[12:16, 6] = np.pi / 4   # 45 degrees
    pa = torch.from_numpy(a).requires_grad_(True); pb = torch.from_numpy(b)
    iou = riou.cal_iou_3d(pa[None], pb[None])[0]
    w = torch.from_numpy(rng.random(300).astype(np.float32))
    ((1 - iou) * w).sum().backward()
    d.update(ro_pred=a, ro_target=b, ro_w=w.numpy(), ro_iou=iou.detach().numpy(), ro_grad=pa.grad.numpy())
    np.savez_compressed(out, **d)


def gen_bev(out):
    sys.path.insert(0, os.path.join(ROOT, 'oracle', '_ref'))
    import pcdet_iou3d_cpu as ref
    rng = np.random.default_rng(5)
    d = {}
    for n in (1, 63, 64, 65, 300):
        b = _rand_boxes(rng, n, True)
        b[:, :2] = rng.uniform(0, 4, (n, 2))
        ans = torch.zeros(n, n)
        ref.boxes_iou_bev_cpu(torch.from_numpy(b), torch.from_numpy(b), ans)
        d[f'boxes{n}'] = b; d[f'iou{n}'] = ans.numpy()
    hand = np.array([[0, 0, 0, 2, 2, 1, 0], [1, 0, 0, 2, 2, 1, 0], [0, 0, 0, 2, 2, 1, np.pi / 4],
                     [0, 0, 5, 2, 2, 1, 0], [5, 5, 0, 1, 1, 1, 0.3]], np.float32)
    ans = torch.zeros(5, 5)
    ref.boxes_iou_bev_cpu(torch.from_numpy(hand), torch.from_numpy(hand), ans)
    d['boxes_hand'] = hand; d['iou_hand'] = ans.numpy()
    np.savez_compressed(out, **d)


if __name__ == '__main__':
    head, utils, aiou, riou = load_reference()
    gen_assigner(head, os.path.join(HERE, 'assigner.npz'))
    gen_decode(head, os.path.join(HERE, 'decode.npz'))
    gen_iou(aiou, riou, os.path.join(HERE, 'iou3d.npz'))
    rng = np.random.default_rng(3)
    pts = torch.from_numpy(rng.normal(0, 1, (4, 5, 3)).astype(np.float32))
    ang = torch.from_numpy(rng.uniform(-3, 3, 4).astype(np.float32))
    np.savez_compressed(os.path.join(HERE, 'rotation.npz'), points=pts.numpy(), angles=ang.numpy(),
                        out=utils.rotation_3d_in_axis(pts, ang, axis=2).numpy())
    if os.path.exists(os.path.join(ROOT, 'oracle', '_ref')):
        gen_bev(os.path.join(HERE, 'bev_iou.npz'))
    print('done')
