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
  * mmdet3d/core/evaluation/indoor_eval.py      indoor_eval / eval_map_recall / eval_det_cls / average_precision
      (mmcv.print_log and terminaltables stubbed; per-box `overlaps` served by the oracle's 3D IoU)
  * mmdet3d/core/bbox/structures/{base_box3d,depth_box3d}.py, mmdet3d/core/points/{base_points,depth_points}.py
      rotate / flip / scale / translate of boxes + points (what RandomFlip3D, GlobalRotScaleTrans, GlobalAlignment apply)
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
    # >= 5 seeds x {1, 2, 4} levels, rotated GT included (cases 0-4 are r1's; 5-14 added in r3: every level
    # count with axis-aligned AND rotated boxes on new seeds, crowded scenes (15 boxes) and a single box)
    spec = [(0, 4, False, 7), (1, 2, False, 7), (2, 1, False, 7), (3, 4, True, 7), (4, 2, True, 7),
            (5, 1, True, 7), (6, 4, False, 15), (7, 2, True, 15), (8, 1, False, 1), (9, 4, True, 1),
            (10, 2, False, 3), (11, 4, True, 15), (12, 1, True, 15), (13, 2, False, 15), (14, 4, False, 3)]
    for ci, (seed, n_scales, rotated, n_boxes) in enumerate(spec):
        pts, gt, labels = make_scene(seed, n_points=6000, n_boxes=n_boxes, rotated=rotated)
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
    cases['n_cases'] = np.int64(len(spec))
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


def gen_iou64(riou, src, out):
    """r5: the rotated-IoU gradients of iou3d.npz's inputs once more with the reference's own code (oriented_iou_loss.py:86-109)
    run in FLOAT64 — the yardstick for the 1e-4 gradient bound of tests/test_gpu_model.py (the fp32 reference gradient itself
    sits up to 1e-3 away from it: clipped-polygon vertices differenced at fp32)"""
    d = np.load(src)
    pa = torch.from_numpy(d['ro_pred']).double().requires_grad_(True)
    pb = torch.from_numpy(d['ro_target']).double()
    iou = riou.cal_iou_3d(pa[None], pb[None])[0]
    ((1 - iou) * torch.from_numpy(d['ro_w']).double()).sum().backward()
    g32 = d['ro_grad'].astype(np.float64)
    print('rotated IoU, fp32 reference vs its fp64 run: iou', float(np.abs(iou.detach().numpy() - d['ro_iou']).max()),
          'grad', float(np.abs(pa.grad.numpy() - g32).max()), 'of scale', float(np.abs(pa.grad.numpy()).max()))
    np.savez_compressed(out, ro_iou64=iou.detach().numpy(), ro_grad64=pa.grad.numpy())


def gen_indoor_eval(out):
    """mmdet3d/core/evaluation/indoor_eval.py run as it is (mmcv.print_log / terminaltables stubbed) on random
    detections; its per-box `overlaps` is served by the oracle's 3D IoU, so the golden pins the matching / AP logic."""
    from oracle import bev as obev

    class Boxes:                                   # the slice of BaseInstance3DBoxes indoor_eval touches
        def __init__(self, tensor, box_dim=7, with_yaw=True, origin=(0.5, 0.5, 0)):
            t = torch.as_tensor(np.asarray(tensor), dtype=torch.float32).reshape(-1, 7).clone()
            if tuple(origin) != (0.5, 0.5, 0):
                t[:, 2] += t[:, 5] * (0 - origin[2])
            self.tensor = t

        def __len__(self):
            return len(self.tensor)

        def __getitem__(self, i):
            b = Boxes.__new__(Boxes); b.tensor = self.tensor[i:i + 1]
            return b

        def new_box(self, t):
            b = Boxes.__new__(Boxes); b.tensor = torch.as_tensor(t).reshape(-1, 7).clone()
            return b

        def convert_to(self, mode):
            return self

        @classmethod
        def overlaps(cls, a, b):
            return torch.from_numpy(iou3d(_grav(a.tensor), _grav(b.tensor)))

    def _grav(t):
        t = t.clone().numpy(); t[:, 2] += t[:, 5] / 2
        return t

    def iou3d(a, b):                               # pcdet_nms_utils.py:44-78 on the oracle's BEV IoU
        iou = obev.iou_matrix(a, b, True).astype(np.float64)
        sa, sb = (a[:, 3] * a[:, 4])[:, None], (b[:, 3] * b[:, 4])[None]
        ov = iou * (sa + sb) / (1 + iou)
        oh = np.clip(np.minimum(a[:, 2:3] + a[:, 5:6] / 2, (b[:, 2] + b[:, 5] / 2)[None])
                     - np.maximum(a[:, 2:3] - a[:, 5:6] / 2, (b[:, 2] - b[:, 5] / 2)[None]), 0, None)
        o3 = ov * oh
        return (o3 / np.clip((a[:, 3] * a[:, 4] * a[:, 5])[:, None] + (b[:, 3] * b[:, 4] * b[:, 5])[None] - o3, 1e-6, None)).astype(np.float32)

    _stub('mmcv.utils', print_log=lambda *a, **k: None)

    class AsciiTable:
        def __init__(self, data):
            self.table = ''
    _stub('terminaltables', AsciiTable=AsciiTable)
    ev = _load('ref_indoor_eval', f'{REF}/mmdet3d/core/evaluation/indoor_eval.py')
    rng = np.random.default_rng(7)
    d = {}
    for case, (n_scenes, n_cls, rotated) in enumerate([(6, 4, False), (5, 3, True)]):
        gt_annos, dt_annos = [], []
        for s in range(n_scenes):
            m = int(rng.integers(0, 7)) if s else 5
            gb = np.concatenate([rng.uniform(0, 6, (m, 3)), rng.uniform(0.4, 1.6, (m, 3)),
                                 rng.uniform(-3, 3, (m, 1)) if rotated else np.zeros((m, 1))], 1).astype(np.float32)
            gl = rng.integers(0, n_cls, m)
            if s == 0:
                gl[:n_cls] = np.arange(n_cls)[:m]                       # every class has a GT box somewhere
            # detections: jittered copies of GT (some duplicated) + random false positives
            k = int(rng.integers(0, 2 * m + 3))
            src = rng.integers(0, max(m, 1), k)
            db = (gb[src] if m else np.zeros((k, 7), np.float32)).copy()
            db[:, :3] += rng.normal(0, 0.15, (k, 3)); db[:, 3:6] *= rng.uniform(0.8, 1.25, (k, 3))
            if rotated:
                db[:, 6] += rng.normal(0, 0.2, k)
            dl = np.where(rng.random(k) < 0.8, gl[src] if m else 0, rng.integers(0, n_cls, k)).astype(np.int64)
            fp = np.concatenate([rng.uniform(0, 6, (3, 3)), rng.uniform(0.4, 1.6, (3, 3)), np.zeros((3, 1))], 1).astype(np.float32)
            db = np.concatenate([db, fp]).astype(np.float32); dl = np.concatenate([dl, rng.integers(0, n_cls, 3)])
            ds = rng.random(len(db)).astype(np.float32)
            gt_annos.append(dict(gt_num=m, gt_boxes_upright_depth=gb, **{'class': gl}))
            dt_annos.append(dict(boxes_3d=Boxes(db, origin=(0.5, 0.5, 0.5)), scores_3d=torch.from_numpy(ds),
                                 labels_3d=torch.from_numpy(dl)))
            d[f'c{case}_gt_boxes{s}'] = gb; d[f'c{case}_gt_class{s}'] = gl
            d[f'c{case}_dt_boxes{s}'] = db; d[f'c{case}_dt_scores{s}'] = ds; d[f'c{case}_dt_labels{s}'] = dl
        label2cat = {i: f'cat{i}' for i in range(n_cls)}
        ret = ev.indoor_eval(gt_annos, dt_annos, (0.25, 0.5), label2cat, box_type_3d=Boxes, box_mode_3d=None)
        d[f'c{case}_n_scenes'] = n_scenes; d[f'c{case}_n_cls'] = n_cls
        keys = sorted(ret)
        d[f'c{case}_keys'] = np.array(keys); d[f'c{case}_vals'] = np.array([ret[k] for k in keys], np.float64)
        print('indoor_eval golden case', case, {k: round(ret[k], 4) for k in keys if k.startswith('m')})
    np.savez_compressed(out, **d)


def gen_pipeline(out):
    """the reference's own DepthInstance3DBoxes / DepthPoints (depth_box3d.py, base_box3d.py, base_points.py,
    depth_points.py; the compiled iou3d / roiaware ops they import but do not use here are stubbed) driven through
    rotate / flip / scale / translate with fixed parameters, and GlobalAlignment's rotate+translate."""
    _stub('mmdet3d')
    _stub('mmdet3d.ops', points_in_boxes_batch=None)
    _stub('mmdet3d.ops.iou3d', iou3d_cuda=None)
    _stub('mmdet3d.core')
    pts_pkg = types.ModuleType('mmdet3d.core.points'); pts_pkg.__path__ = [f'{REF}/mmdet3d/core/points']
    sys.modules['mmdet3d.core.points'] = pts_pkg
    bp = _load('mmdet3d.core.points.base_points', f'{REF}/mmdet3d/core/points/base_points.py')
    pts_pkg.BasePoints = bp.BasePoints
    dp = _load('mmdet3d.core.points.depth_points', f'{REF}/mmdet3d/core/points/depth_points.py')
    pkg = types.ModuleType('refstruct'); pkg.__path__ = [f'{REF}/mmdet3d/core/bbox/structures']
    sys.modules['refstruct'] = pkg
    _load('refstruct.utils', f'{REF}/mmdet3d/core/bbox/structures/utils.py')
    _load('refstruct.base_box3d', f'{REF}/mmdet3d/core/bbox/structures/base_box3d.py')
    db = _load('refstruct.depth_box3d', f'{REF}/mmdet3d/core/bbox/structures/depth_box3d.py')
    rng = np.random.default_rng(11)
    d = {}
    for case, with_yaw in enumerate([True, False]):
        pts = np.concatenate([rng.uniform(-3, 3, (500, 3)), rng.uniform(0, 255, (500, 3))], 1).astype(np.float32)
        m = 9
        bx = np.concatenate([rng.uniform(-3, 3, (m, 3)), rng.uniform(0.3, 2.0, (m, 3))], 1).astype(np.float32)
        if with_yaw:
            bx = np.concatenate([bx, rng.uniform(-3, 3, (m, 1)).astype(np.float32)], 1)
        angle, scale, trans = 0.0613, 1.0731, np.array([0.12, -0.07, 0.031], np.float32)
        d[f'c{case}_points'] = pts; d[f'c{case}_boxes'] = bx
        d[f'c{case}_params'] = np.array([angle, scale, *trans], np.float64)

        def fresh():
            b = db.DepthInstance3DBoxes(torch.from_numpy(bx.copy()), box_dim=bx.shape[1], with_yaw=with_yaw, origin=(0.5, 0.5, 0))
            p = dp.DepthPoints(torch.from_numpy(pts.copy()), points_dim=6, attribute_dims=dict(color=[3, 4, 5]))
            return b, p
        for direction in ('horizontal', 'vertical'):
            b, p = fresh()
            p = b.flip(direction, points=p)
            d[f'c{case}_flip_{direction}_points'] = p.tensor.numpy(); d[f'c{case}_flip_{direction}_boxes'] = b.tensor.numpy()
        b, p = fresh()
        p, _ = b.rotate(angle, p)                      # GlobalRotScaleTrans._rot_bbox_points
        p.scale(scale); b.scale(scale)                 # ._scale_bbox_points
        p.translate(trans); b.translate(trans)         # ._trans_bbox_points
        d[f'c{case}_rst_points'] = p.tensor.numpy(); d[f'c{case}_rst_boxes'] = b.tensor.numpy()
    # GlobalAlignment.__call__: points.rotate(rot.T) then translate
    th = 0.4
    A = np.eye(4, dtype=np.float32); A[:2, :2] = [[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]]; A[:3, 3] = [0.5, -1.25, 0.1]
    p = dp.DepthPoints(torch.from_numpy(pts.copy()), points_dim=6, attribute_dims=dict(color=[3, 4, 5]))
    p.rotate(A[:3, :3].T); p.translate(A[:3, 3])
    d['align_matrix'] = A; d['align_points_in'] = pts; d['align_points_out'] = p.tensor.numpy()
    np.savez_compressed(out, **d)
    print('pipeline golden written')


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
    if sys.argv[1:] == ['indoor_eval']:            # only this fixture (the others are unchanged)
        gen_indoor_eval(os.path.join(HERE, 'indoor_eval.npz'))
        sys.exit(0)
    if sys.argv[1:] == ['iou64']:
        head, utils, aiou, riou = load_reference()
        gen_iou64(riou, os.path.join(HERE, 'iou3d.npz'), os.path.join(HERE, 'iou3d_f64.npz'))
        sys.exit(0)
    if sys.argv[1:] == ['pipeline']:
        gen_pipeline(os.path.join(HERE, 'pipeline.npz'))
        sys.exit(0)
    head, utils, aiou, riou = load_reference()
    gen_indoor_eval(os.path.join(HERE, 'indoor_eval.npz'))
    gen_pipeline(os.path.join(HERE, 'pipeline.npz'))
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
