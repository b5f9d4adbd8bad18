"""CPU: pins the oracle (oracle/loss_oracle.py, oracle/bev_nms_oracle.c) against the golden vectors
that tests/golden/make_golden.py produced by running the REFERENCE's own code."""
import os

import numpy as np
import torch

from oracle import bev, loss_oracle as lo

G = os.path.join(os.path.dirname(__file__), 'golden')


def test_assigner_oracle():
    d = np.load(os.path.join(G, 'assigner.npz'))
    for ci in range(int(d['n_cases'])):
        L = int(d[f'c{ci}_n_scales'])
        pts = [torch.from_numpy(d[f'c{ci}_points{l}']) for l in range(L)]
        ct, bt, lb = lo.assign(pts, torch.from_numpy(d[f'c{ci}_gt']), torch.from_numpy(d[f'c{ci}_labels']), n_scales=L)
        ref = d[f'c{ci}_assigned']
        assert np.array_equal(lb.numpy(), ref)
        pos = ref >= 0
        assert np.allclose(ct.numpy()[pos], d[f'c{ci}_centerness'][pos], atol=1e-6)
        assert np.allclose(bt.numpy()[pos], d[f'c{ci}_bbox_targets'][pos], atol=1e-6)


def test_decode_oracle():
    d = np.load(os.path.join(G, 'decode.npz'))
    p = torch.from_numpy(d['points'])
    assert np.allclose(lo.bbox_pred_to_bbox(p, torch.from_numpy(d['pred6'])).numpy(), d['out6'], atol=1e-6)
    for mode in ('fcaf3d', 'sin-cos'):
        assert np.allclose(lo.bbox_pred_to_bbox(p, torch.from_numpy(d['pred8']), mode).numpy(), d[f'out8_{mode}'], atol=1e-6)
    assert np.allclose(lo.bbox_pred_to_bbox(p, torch.from_numpy(d['pred8'][:, :7]), 'naive').numpy(), d['out7_naive'], atol=1e-6)
    assert np.allclose(lo.compute_centerness(torch.from_numpy(d['cent_in'])).numpy(), d['cent_out'], atol=1e-6)


def test_aligned_and_rotated_iou_oracle():
    d = np.load(os.path.join(G, 'iou3d.npz'))
    for key, fn in (('al', lo.axis_aligned_iou), ('ro', lo.rotated_iou_3d)):
        pred = torch.from_numpy(d[f'{key}_pred']).requires_grad_(True)
        tgt = torch.from_numpy(d[f'{key}_target'])
        iou = fn(pred, tgt)
        ((1 - iou) * torch.from_numpy(d[f'{key}_w'])).sum().backward()
        assert np.allclose(iou.detach().numpy(), d[f'{key}_iou'], atol=2e-6), key
        assert np.allclose(pred.grad.numpy(), d[f'{key}_grad'], atol=1e-4, rtol=1e-4), key


def test_bev_iou_oracle_matches_compiled_reference():
    d = np.load(os.path.join(G, 'bev_iou.npz'))
    for n in ('1', '63', '64', '65', '300', '_hand'):
        b = d[f'boxes{n}']
        assert np.allclose(bev.iou_matrix(b, b, True), d[f'iou{n}'], atol=1e-6), n
    hand = d['iou_hand']
    assert abs(hand[0, 1] - 1 / 3) < 1e-6 and abs(hand[0, 2] - 0.70710677) < 1e-5 and hand[0, 3] == 1.0
    # greedy keep lists == greedy scan over the reference's IoU matrix (iou3d_nms.cpp:119-132)
    for n in (63, 64, 65, 300):
        b = d[f'boxes{n}']
        scores = np.linspace(1, 0, len(b)).astype(np.float32)
        keep = bev.nms(b, scores, 0.3, True)
        ref = lo.greedy_nms_from_iou(d[f'iou{n}'], 0.3)
        assert np.array_equal(keep, ref)


def test_focal_closed_form():
    x = torch.tensor([[0.3, -1.2, 2.0], [0.1, 0.0, -0.5]], requires_grad=True)
    lb = torch.tensor([1, -1])
    v = lo.sigmoid_focal_loss_sum(x, lb)
    p = torch.sigmoid(x.detach())
    exp = 0.0
    for n in range(2):
        for c in range(3):
            pc = float(p[n, c])
            exp += (-0.25 * (1 - pc) ** 2 * np.log(pc)) if int(lb[n]) == c else (-0.75 * pc ** 2 * np.log(1 - pc))
    assert abs(float(v) - exp) < 1e-6


def test_rotation_golden_matches_assigner_convention():
    d = np.load(os.path.join(G, 'rotation.npz'))
    p = torch.from_numpy(d['points']); a = torch.from_numpy(d['angles'])
    c, s = torch.cos(a)[:, None], torch.sin(a)[:, None]
    out = torch.stack([p[..., 0] * c + p[..., 1] * s, -p[..., 0] * s + p[..., 1] * c, p[..., 2]], -1)
    assert np.allclose(out.numpy(), d['out'], atol=1e-6)


def test_sort_v_restatement_gives_the_pinned_iou():
    """oracle.sort_v (Appendix D, standalone op) selects the same polygon as the pinned rotated-IoU path"""
    import numpy as np
    import torch
    from oracle import loss_oracle as lo
    g = torch.Generator().manual_seed(3)
    n = 300
    b1 = torch.cat([torch.rand(n, 2, generator=g) * 2, torch.rand(n, 2, generator=g) * 2 + 0.3,
                    (torch.rand(n, 1, generator=g) - 0.5) * 6.28], 1)
    b2 = torch.cat([b1[:, :2] + (torch.rand(n, 2, generator=g) - 0.5) * 1.5, torch.rand(n, 2, generator=g) * 2 + 0.3,
                    (torch.rand(n, 1, generator=g) - 0.5) * 6.28], 1)
    b2[:20] = b1[:20]
    b2[20:40, :2] += 10.0
    verts, mask = lo.intersection_vertices(b1, b2)
    nv = mask.sum(1).int()
    ctr = (verts * mask[..., None]).sum(1, keepdim=True) / nv.clamp(min=1)[:, None, None]
    vc = (verts - ctr) * mask[..., None]
    idx = lo.sort_v(vc[None].numpy(), mask[None].numpy(), nv[None].numpy())[0]
    assert idx.shape == (n, 9) and idx.dtype == np.int32
    sel = np.take_along_axis(vc.numpy(), idx[:, :, None].astype(np.int64), 1)
    area = np.abs((sel[:, :-1, 0] * sel[:, 1:, 1] - sel[:, :-1, 1] * sel[:, 1:, 0]).sum(1)) / 2
    z, h = torch.zeros(n, 1), torch.ones(n, 1)
    B1 = torch.cat([b1[:, :2], z, b1[:, 2:4], h, b1[:, 4:5]], 1)
    B2 = torch.cat([b2[:, :2], z, b2[:, 2:4], h, b2[:, 4:5]], 1)
    iou = lo.rotated_iou_3d(B1, B2).numpy()
    a1, a2 = (b1[:, 2] * b1[:, 3]).numpy(), (b2[:, 2] * b2[:, 3]).numpy()
    assert np.abs(area / (a1 + a2 - area) - iou).max() < 1e-5
    assert (idx[20:40] >= 8).all()                       # disjoint pairs: every slot is the pad index


def _oracle_iou3d(a, b):
    """3D IoU of gravity-centre boxes on the oracle's BEV IoU (the IoU make_golden.py served to the reference)"""
    import numpy as np
    from oracle import bev as obev
    a = np.asarray(a, np.float32); b = np.asarray(b, np.float32)
    iou = obev.iou_matrix(a, b, True).astype(np.float64)
    sa, sb = (a[:, 3] * a[:, 4])[:, None], (b[:, 3] * b[:, 4])[None]
    ov = iou * (sa + sb) / (1 + iou)
    oh = np.clip(np.minimum(a[:, 2:3] + a[:, 5:6] / 2, (b[:, 2] + b[:, 5] / 2)[None])
                 - np.maximum(a[:, 2:3] - a[:, 5:6] / 2, (b[:, 2] - b[:, 5] / 2)[None]), 0, None)
    o3 = ov * oh
    return (o3 / np.clip((a[:, 3] * a[:, 4] * a[:, 5])[:, None] + (b[:, 3] * b[:, 4] * b[:, 5])[None] - o3, 1e-6, None)).astype(np.float32)


def _indoor_eval_case(d, case):
    import numpy as np
    import torch
    import fcaf3d_amd as fa
    gt_annos, dt_annos = [], []
    for s in range(int(d[f'c{case}_n_scenes'])):
        gb, gl = d[f'c{case}_gt_boxes{s}'], d[f'c{case}_gt_class{s}']
        gt_annos.append({'gt_num': len(gb), 'gt_boxes_upright_depth': gb, 'class': gl})
        dt_annos.append(dict(boxes_3d=fa.DepthInstance3DBoxes(torch.from_numpy(d[f'c{case}_dt_boxes{s}']), origin=(.5, .5, .5)),
                             scores_3d=torch.from_numpy(d[f'c{case}_dt_scores{s}']),
                             labels_3d=torch.from_numpy(d[f'c{case}_dt_labels{s}'])))
    label2cat = {i: f'cat{i}' for i in range(int(d[f'c{case}_n_cls']))}
    want = dict(zip([str(k) for k in d[f'c{case}_keys']], d[f'c{case}_vals']))
    return gt_annos, dt_annos, label2cat, want


def test_indoor_eval_matches_reference_golden():
    """mAP / mAR / per-class AP + recall == the reference's indoor_eval run on the same annotations (tests/golden/
    make_golden.py::gen_indoor_eval), axis-aligned and rotated cases, thresholds 0.25 / 0.5"""
    import os
    import numpy as np
    from fcaf3d_amd.evaluation import average_precision, indoor_eval
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'indoor_eval.npz'))
    for case in (0, 1):
        gt_annos, dt_annos, label2cat, want = _indoor_eval_case(d, case)
        got = indoor_eval(gt_annos, dt_annos, (0.25, 0.5), label2cat, iou_fn=_oracle_iou3d)
        assert sorted(got) == sorted(want)
        for k in want:
            assert abs(got[k] - want[k]) < 1e-6, (case, k, got[k], want[k])
    # VOC area AP on a hand-checkable curve: recall steps 0.5, 1.0 with precisions 1.0, 2/3
    assert abs(float(average_precision(np.array([0.5, 0.5, 1.0]), np.array([1.0, 0.5, 2 / 3]))[0]) - (0.5 + 0.5 * 2 / 3)) < 1e-6


def test_pipeline_transforms_match_reference_classes():
    """fcaf3d_amd/pipelines.py == the reference's DepthInstance3DBoxes / DepthPoints rotate / flip / scale / translate
    and GlobalAlignment on the same inputs (tests/golden/pipeline.npz, generated by importing those classes)"""
    import os
    import numpy as np
    import torch
    from fcaf3d_amd import pipelines as pl
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'pipeline.npz'))
    for case, with_yaw in enumerate([True, False]):
        pts = torch.from_numpy(d[f'c{case}_points'])
        bx = torch.from_numpy(d[f'c{case}_boxes'])
        if bx.shape[1] == 6:
            bx = torch.cat((bx, bx.new_zeros(len(bx), 1)), 1)
        angle, scale, *trans = d[f'c{case}_params'].tolist()
        for direction in ('horizontal', 'vertical'):
            p, b = pl.flip_bev(pts, bx, direction, with_yaw)
            assert np.allclose(p.numpy(), d[f'c{case}_flip_{direction}_points'], atol=1e-6)
            assert np.allclose(b.numpy(), d[f'c{case}_flip_{direction}_boxes'], atol=1e-6)
        p, b = pl.rot_scale_trans(pts, bx, angle, scale, trans, with_yaw)
        assert np.allclose(p.numpy(), d[f'c{case}_rst_points'], atol=2e-6), np.abs(p.numpy() - d[f'c{case}_rst_points']).max()
        assert np.allclose(b.numpy(), d[f'c{case}_rst_boxes'], atol=2e-6), np.abs(b.numpy() - d[f'c{case}_rst_boxes']).max()
    out = pl.global_alignment(torch.from_numpy(d['align_points_in']), d['align_matrix'])
    assert np.allclose(out.numpy(), d['align_points_out'], atol=2e-6)
    # sampling: exact row count, without replacement when possible, colours ride along
    g = torch.Generator().manual_seed(0)
    s, idx = pl.indoor_point_sample(torch.from_numpy(d['c0_points']), 300, g)
    assert s.shape == (300, 6) and len(set(idx.tolist())) == 300
    s, idx = pl.indoor_point_sample(torch.from_numpy(d['c0_points']), 800, g)
    assert s.shape == (800, 6) and int(idx.max()) < 500
    aug = pl.TrainAugment(num_points=400, with_yaw=True)
    p, b, params = aug(torch.from_numpy(d['c0_points']), torch.from_numpy(d['c0_boxes']), g)
    assert p.shape == (400, 6) and b.shape == (9, 7) and 0.9 <= params['scale'] <= 1.1
    assert abs(params['angle']) <= 0.087266 + 1e-9
    assert torch.allclose(b[:, 3:6], torch.from_numpy(d['c0_boxes'])[:, 3:6] * params['scale'], atol=1e-5)


def test_pipeline_on_the_reference_scannet_fixture():
    """the reference's own dataset fixture (tests/data/scannet: scannet_infos.pkl + scene0000_00.bin, copied as data)
    through load -> align -> sample -> flip H,V -> rotate, against the numbers its test holds
    (tests/test_data/test_datasets/test_scannet_dataset.py:65-86, np.random.seed(0))"""
    import math
    import os
    import numpy as np
    import torch
    from fcaf3d_amd import pipelines as pl
    G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
    ds = pl.IndoorInfoDataset(G, os.path.join(G, 'scannet_infos.pkl'), with_yaw=False)
    assert len(ds) == 1
    pts, boxes, labels, meta = ds.load(0, points_file=os.path.join(G, 'scannet_scene0000_00.bin'))
    assert meta['sample_idx'] == 'scene0000_00' and pts.shape == (100, 6) and boxes.shape == (27, 7)
    expected_labels = [6, 6, 4, 9, 11, 11, 10, 0, 15, 17, 17, 17, 3, 12, 4, 4, 14, 1, 0, 0, 0, 0, 0, 0, 5, 5, 5]
    assert labels.tolist() == expected_labels
    # the reference's draws under np.random.seed(0): the sample indices, both flips (ratio 1.0), and the rotation whose
    # matrix its test pins as pcd_rotation = [[0.99654, 0.08311407, 0], [-0.08311407, 0.99654, 0], [0, 0, 1]]
    idx = np.random.RandomState(0).choice(100, 5, replace=False)
    angle = math.asin(0.08311407)
    p = pts[torch.from_numpy(idx)]
    p, b = pl.flip_bev(p, boxes, 'horizontal', False)
    p, b = pl.flip_bev(p, b, 'vertical', False)
    p, b = pl.rot_scale_trans(p, b, angle, 1.0, [0.0, 0.0, 0.0], False)
    expected_points = torch.tensor([[1.8339e+00, 2.1093e+00, 2.2900e+00], [3.6079e+00, 1.4592e-01, 2.0687e+00],
                                    [4.1886e+00, 5.0614e+00, -1.0841e-01], [6.8790e+00, 1.5086e+00, -9.3154e-02],
                                    [4.8253e+00, 2.6668e-01, 1.4917e+00]])
    expected_boxes = torch.tensor([[-1.1835, -3.6317, 1.5704, 1.7577, 0.3761, 0.5724, 0.0000],
                                   [-3.1832, 3.2269, 1.1911, 0.6727, 0.2251, 0.6715, 0.0000],
                                   [-0.9598, -2.2864, 0.0093, 0.7506, 2.5709, 1.2145, 0.0000],
                                   [-2.6988, -2.7354, 0.8288, 0.7680, 1.8877, 0.2870, 0.0000],
                                   [3.2989, 0.2885, -0.0090, 0.7600, 3.8814, 2.1603, 0.0000]])
    assert torch.allclose(b[:5], expected_boxes, rtol=1e-2, atol=2e-4), (b[:5] - expected_boxes).abs().max()
    assert torch.allclose(p[:, :3], expected_points, rtol=1e-2, atol=2e-4), (p[:, :3] - expected_points).abs().max()


def test_compose_of_the_config_pipeline_reproduces_the_reference_test_numbers():
    """the train_pipeline of configs/fcaf3d/fcaf3d_scannet-3d-18class.py:16-40, built BY NAME (LoadPointsFromFile,
    LoadAnnotations3D, GlobalAlignment, IndoorPointSample, RandomFlip3D, GlobalRotScaleTrans, DefaultFormatBundle3D,
    Collect3D) through fcaf3d_amd.pipelines.Compose, on the reference's ScanNet fixture under np.random.seed(0): the
    classes take numpy's draws in the reference's order, so the points and boxes are the numbers the reference's own test
    pins (tests/test_data/test_datasets/test_scannet_dataset.py:65-86) — without replaying a single draw by hand."""
    import os
    import numpy as np
    import torch
    from fcaf3d_amd import pipelines as pl
    G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
    ds = pl.IndoorInfoDataset(G, os.path.join(G, 'scannet_infos.pkl'), with_yaw=False)
    cfg = [dict(type='LoadPointsFromFile', coord_type='DEPTH', shift_height=False, load_dim=6, use_dim=[0, 1, 2, 3, 4, 5]),
           dict(type='LoadAnnotations3D'), dict(type='GlobalAlignment', rotation_axis=2),
           dict(type='IndoorPointSample', num_points=5),
           dict(type='RandomFlip3D', sync_2d=False, flip_ratio_bev_horizontal=1.0, flip_ratio_bev_vertical=1.0),
           dict(type='GlobalRotScaleTrans', rot_range=[-0.087266, 0.087266], scale_ratio_range=[1.0, 1.0], shift_height=False),
           dict(type='DefaultFormatBundle3D', class_names=('cabinet',)),
           dict(type='Collect3D', keys=['points', 'gt_bboxes_3d', 'gt_labels_3d'])]
    np.random.seed(0)
    out = pl.Compose(cfg)(ds.pre_pipeline(0, points_file=os.path.join(G, 'scannet_scene0000_00.bin')))
    expected_points = torch.tensor([[1.8339e+00, 2.1093e+00, 2.2900e+00], [3.6079e+00, 1.4592e-01, 2.0687e+00],
                                    [4.1886e+00, 5.0614e+00, -1.0841e-01], [6.8790e+00, 1.5086e+00, -9.3154e-02],
                                    [4.8253e+00, 2.6668e-01, 1.4917e+00]])
    expected_boxes = torch.tensor([[-1.1835, -3.6317, 1.5704, 1.7577, 0.3761, 0.5724, 0.0000],
                                   [-3.1832, 3.2269, 1.1911, 0.6727, 0.2251, 0.6715, 0.0000],
                                   [-0.9598, -2.2864, 0.0093, 0.7506, 2.5709, 1.2145, 0.0000],
                                   [-2.6988, -2.7354, 0.8288, 0.7680, 1.8877, 0.2870, 0.0000],
                                   [3.2989, 0.2885, -0.0090, 0.7600, 3.8814, 2.1603, 0.0000]])
    assert torch.allclose(out['points'][:, :3], expected_points, rtol=1e-2, atol=2e-4)
    assert torch.allclose(out['gt_bboxes_3d'].tensor[:5], expected_boxes, rtol=1e-2, atol=2e-4)
    meta = out['img_metas']
    assert meta['pcd_horizontal_flip'] and meta['pcd_vertical_flip'] and meta['sample_idx'] == 'scene0000_00'
    expected_rot = torch.tensor([[0.99654, 0.08311407, 0.0], [-0.08311407, 0.99654, 0.0], [0.0, 0.0, 1.0]])
    assert torch.allclose(meta['pcd_rotation'], expected_rot, atol=1e-5) and meta['pcd_scale_factor'] == 1.0
    assert out['gt_labels_3d'].dtype == torch.int64 and len(out['gt_labels_3d']) == 27


def test_s3dis_fixture_loads_through_the_s3dis_pipeline_head():
    """the reference's S3DIS fixture (tests/data/s3dis: s3dis_infos.pkl + points/Area_1_office_2.bin, copied as data) through
    the head of configs/fcaf3d/fcaf3d_s3dis-3d-5class.py:16-24 — LoadPointsFromFile + LoadAnnotations3D, no GlobalAlignment
    (S3DIS rooms carry no axis_align_matrix): 100 x 6 points as stored, a scene without an `annos` entry yields empty
    Depth-mode boxes (S3DISDataset.get_ann_info, s3dis_dataset.py:66-100)."""
    import os
    import numpy as np
    from fcaf3d_amd import pipelines as pl
    G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
    ds = pl.IndoorInfoDataset(G, os.path.join(G, 's3dis_infos.pkl'), with_yaw=False)
    assert len(ds) == 1
    res = ds.pre_pipeline(0, points_file=os.path.join(G, 's3dis_Area_1_office_2.bin'))
    assert res['sample_idx'] == 'Area_1_office_2' and len(res['ann_info']['gt_bboxes_3d']) == 0
    out = pl.Compose([dict(type='LoadPointsFromFile', coord_type='DEPTH', shift_height=False, load_dim=6, use_dim=[0, 1, 2, 3, 4, 5]),
                      dict(type='LoadAnnotations3D'), dict(type='IndoorPointSample', num_points=64),
                      dict(type='RandomFlip3D', sync_2d=False, flip_ratio_bev_horizontal=0.5, flip_ratio_bev_vertical=0.5),
                      dict(type='GlobalRotScaleTrans', rot_range=[-0.087266, 0.087266], scale_ratio_range=[.9, 1.1],
                           translation_std=[.1, .1, .1], shift_height=False),
                      dict(type='DefaultFormatBundle3D', class_names=('table', 'chair', 'sofa', 'bookcase', 'board')),
                      dict(type='Collect3D', keys=['points', 'gt_bboxes_3d', 'gt_labels_3d'])])(res)
    raw = np.fromfile(os.path.join(G, 's3dis_Area_1_office_2.bin'), np.float32).reshape(-1, 6)
    assert raw.shape == (100, 6) and out['points'].shape == (64, 6) and out['gt_bboxes_3d'].tensor.shape[0] == 0
    assert set(np.round(out['points'][:, 3:].numpy().ravel(), 3)) <= set(np.round(raw[:, 3:].ravel(), 3))    # colours untouched


def test_pipeline_on_the_reference_sunrgbd_fixture():
    """rotated boxes (with_yaw): the reference's SUN RGB-D fixture through flip(no) -> rotate -> scale -> sample against
    the numbers of tests/test_data/test_datasets/test_sunrgbd_dataset.py:96-126 (np.random.seed(0): the draw sequence —
    one choice + two rand of RandomFlip3D, uniform rot, uniform scale, normal x3, then the sample — is replayed)"""
    import os
    import numpy as np
    import torch
    from fcaf3d_amd import pipelines as pl
    G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
    ds = pl.IndoorInfoDataset(G, os.path.join(G, 'sunrgbd_infos.pkl'), with_yaw=True)
    pts, boxes, labels, meta = ds.load(0, points_file=os.path.join(G, 'sunrgbd_000001.bin'))
    assert labels.tolist() == [0, 7, 6] and boxes.shape == (3, 7)
    rs = np.random.RandomState(0)
    rs.rand(); flip_h = rs.rand() < 0.5; rs.rand()
    angle = rs.uniform(-0.523599, 0.523599)
    scale = rs.uniform(0.85, 1.15)
    rs.normal(scale=[0.0, 0.0, 0.0], size=3)
    assert not flip_h and abs(scale - 0.9770964398016714) < 1e-9 and abs(np.sin(angle) - 0.04698427) < 1e-6
    p, b = pl.rot_scale_trans(pts, boxes, angle, scale, [0.0, 0.0, 0.0], True)
    idx = rs.choice(p.shape[0], 5, replace=False)
    expected_boxes = torch.tensor([[0.8308, 4.1168, -1.2035, 2.2493, 1.8444, 1.9245, 1.6486],
                                   [2.3002, 4.8149, -1.2442, 0.5718, 0.8629, 0.9510, 1.6030],
                                   [-1.1477, 1.8090, -1.1725, 0.6965, 1.5273, 2.0563, 0.0552]])
    expected_points = torch.tensor([[-0.9904, 1.2596, 0.1105], [-0.9948, 1.2758, 0.0437], [-0.9866, 1.2641, 0.0504],
                                    [-0.9915, 1.2586, 0.1265], [-0.9890, 1.2561, 0.1216]])
    assert torch.allclose(b, expected_boxes, rtol=1e-3, atol=1e-4), (b - expected_boxes).abs().max()
    assert torch.allclose(p[torch.from_numpy(idx), :3], expected_points, rtol=1e-2, atol=2e-4)


def _ref_indoor_eval_vectors():
    """the inputs of the reference's own tests/test_metrics/test_indoor_eval.py (data)"""
    import numpy as np
    import torch
    import fcaf3d_amd as fa
    b = np.array([[-2.4089e-03, -3.3174e+00, 4.9438e-01, 2.1668e+00, 2.8431e-01, 1.6506e+00, 0.0],
                  [-3.4269e-01, -2.7565e+00, 2.8144e-02, 6.8554e-01, 9.6854e-01, 6.1755e-01, 0.0],
                  [-3.8320e+00, -1.0646e+00, 1.7074e-01, 2.4981e-01, 4.4708e-01, 6.2538e-01, 0.0],
                  [4.1073e-01, 3.3757e+00, 3.4311e-01, 8.0617e-01, 2.8679e-01, 1.6060e+00, 0.0],
                  [6.1199e-01, -3.1041e+00, 4.1873e-01, 1.2310e+00, 4.0162e-01, 1.7303e+00, 0.0],
                  [-5.9877e-01, -2.6011e+00, 1.1148e+00, 1.5704e-01, 7.5957e-01, 9.6930e-01, 0.0],
                  [2.7462e-01, -3.0088e+00, 6.5231e-02, 8.1208e-01, 4.1861e-01, 3.7339e-01, 0.0],
                  [-1.4704e+00, -2.0024e+00, 2.7479e-01, 1.7888e+00, 1.0566e+00, 1.3704e+00, 0.0],
                  [8.2727e-02, -3.1160e+00, 2.5690e-01, 1.4054e+00, 2.0772e-01, 9.6792e-01, 0.0],
                  [2.6896e+00, 1.9881e+00, 1.1566e+00, 9.9885e-02, 3.5713e-01, 4.5638e-01, 0.0]], np.float32)
    det = [dict(labels_3d=torch.tensor([0, 1, 2, 2, 0, 3, 1, 2, 3, 2]),
                boxes_3d=fa.DepthInstance3DBoxes(torch.from_numpy(b), origin=(0.5, 0.5, 0)),
                scores_3d=torch.tensor([1.7516e-05, 1.0167e-06, 8.4486e-07, 7.1048e-02, 6.4274e-05, 1.5003e-07,
                                        5.8102e-06, 1.9399e-08, 5.3126e-07, 1.8630e-09]))]
    gt = [{'gt_num': 10, 'gt_boxes_upright_depth': b.copy(), 'class': np.array([0, 1, 2, 0, 0, 3, 1, 3, 3, 2])}]
    one = np.array([[1., 1., 1., 1., 1., 1., 1.]], np.float32)
    det2 = [dict(labels_3d=torch.tensor([0]), boxes_3d=fa.DepthInstance3DBoxes(torch.from_numpy(one)), scores_3d=torch.tensor([.5])),
            dict(labels_3d=torch.tensor([1]), boxes_3d=fa.DepthInstance3DBoxes(torch.from_numpy(one)), scores_3d=torch.tensor([.5]))]
    gt2 = [{'gt_num': 2, 'gt_boxes_upright_depth': np.array([[0., 0., 0., 1., 1., 1., 1.], [1., 1., 1., 1., 1., 1., 1.]], np.float32),
            'class': np.array([2, 0])},
           {'gt_num': 1, 'gt_boxes_upright_depth': one.copy(), 'class': np.array([1])}]
    return (gt, det, {0: 'cabinet', 1: 'bed', 2: 'chair', 3: 'sofa'}), (gt2, det2, {0: 'cabinet', 1: 'bed', 2: 'chair'})


def _check_ref_indoor_eval(iou_fn):
    import numpy as np
    from fcaf3d_amd.evaluation import indoor_eval
    (gt, det, l2c), (gt2, det2, l2c2) = _ref_indoor_eval_vectors()
    r = indoor_eval(gt, det, [0.25, 0.5], l2c, iou_fn=iou_fn)
    assert np.isclose(r['cabinet_AP_0.25'], 0.666667) and np.isclose(r['bed_AP_0.25'], 1.0)
    assert np.isclose(r['chair_AP_0.25'], 0.5) and np.isclose(r['mAP_0.25'], 0.708333) and np.isclose(r['mAR_0.25'], 0.833333)
    r = indoor_eval(gt2, det2, [0.25, 0.5], l2c2, iou_fn=iou_fn)
    assert np.isclose(r['mAP_0.25'], 0.666667) and np.isclose(r['mAR_0.25'], 0.666667)


def test_indoor_eval_reference_test_vectors():
    """the known-answer vectors of the reference's tests/test_metrics/test_indoor_eval.py (its test needs CUDA; here the
    IoU comes from the oracle) incl. its 11-point AP value"""
    import numpy as np
    from fcaf3d_amd.evaluation import average_precision
    _check_ref_indoor_eval(_oracle_iou3d)
    ap = average_precision(np.array([[0.25, 0.5, 0.75], [0.25, 0.5, 0.75]]), np.array([[1., 1., 1.], [1., 1., 1.]]), '11points')
    assert abs(ap[0] - 0.06611571) < 0.001


def test_box_transforms_reference_test_vectors():
    """known-answer vectors of the reference's tests/test_utils/test_box3d.py::test_depth_boxes3d (gravity centre, flip
    H / V with points, rotate with yaw, rotate of yaw-less boxes) through fcaf3d_amd.boxes / fcaf3d_amd.pipelines"""
    import torch
    import fcaf3d_amd as fa
    from fcaf3d_amd import pipelines as pl
    b1 = torch.tensor([[1.4856, 2.5299, -0.5570, 0.9385, 2.1404, 0.8954, 3.0601],
                       [2.3262, 3.3065, 0.44255, 0.8234, 0.5325, 1.0099, 2.9971]])
    assert torch.allclose(fa.DepthInstance3DBoxes(b1).gravity_center,
                          torch.tensor([[1.4856, 2.5299, -0.1093], [2.3262, 3.3065, 0.9475]]), atol=1e-4)
    boxes = torch.cat([b1, torch.tensor([[2.4593, 2.5870, -0.4321, 0.8597, 0.6193, 1.0204, 3.0693],
                                         [1.4856, 2.5299, -0.5570, 0.9385, 2.1404, 0.8954, 3.0601]])])
    points = torch.tensor([[0.6762, 1.2559, -1.4658, 2.5359], [0.8784, 4.7814, -1.3857, 0.7167],
                           [-0.2517, 6.7053, -0.9697, 0.5599], [0.5520, 0.6533, -0.5265, 1.0032],
                           [-0.5358, 4.5870, -1.4741, 0.0556]])
    p, b = pl.flip_bev(points, boxes, 'horizontal', True)
    assert torch.allclose(b, torch.tensor([[-1.4856, 2.5299, -0.5570, 0.9385, 2.1404, 0.8954, 0.0815],
                                           [-2.3262, 3.3065, 0.4426, 0.8234, 0.5325, 1.0099, 0.1445],
                                           [-2.4593, 2.5870, -0.4321, 0.8597, 0.6193, 1.0204, 0.0723],
                                           [-1.4856, 2.5299, -0.5570, 0.9385, 2.1404, 0.8954, 0.0815]]), rtol=1e-3, atol=1e-4)
    assert torch.allclose(p[:, 0], -points[:, 0]) and torch.equal(p[:, 1:], points[:, 1:])
    p, b = pl.flip_bev(p, b, 'vertical', True)
    assert torch.allclose(b[:, [0, 1, 6]], torch.tensor([[-1.4856, -2.5299, -0.0815], [-2.3262, -3.3065, -0.1445],
                                                          [-2.4593, -2.5870, -0.0723], [-1.4856, -2.5299, -0.0815]]),
                          rtol=1e-3, atol=1e-4)
    p, b = pl.rotate(p, b, -0.022998953275003075, True)
    assert torch.allclose(b, torch.tensor([[-1.5434, -2.4951, -0.5570, 0.9385, 2.1404, 0.8954, -0.0585],
                                           [-2.4016, -3.2521, 0.4426, 0.8234, 0.5325, 1.0099, -0.1215],
                                           [-2.5181, -2.5298, -0.4321, 0.8597, 0.6193, 1.0204, -0.0493],
                                           [-1.5434, -2.4951, -0.5570, 0.9385, 2.1404, 0.8954, -0.0585]]), rtol=1e-3, atol=1e-4)
    assert torch.allclose(p, torch.tensor([[-0.7049, -1.2400, -1.4658, 2.5359], [-0.9881, -4.7599, -1.3857, 0.7167],
                                           [0.0974, -6.7093, -0.9697, 0.5599], [-0.5669, -0.6404, -0.5265, 1.0032],
                                           [0.4302, -4.5981, -1.4741, 0.0556]]), rtol=1e-3, atol=1e-4)
    # yaw-less boxes: the rotated box is replaced by its axis-aligned extent
    th = torch.tensor([[0.61211395, 0.8129094, 0.10563634, 1.497534, 0.16927195, 0.27956772],
                       [1.430009, 0.49797538, 0.9382923, 0.07694054, 0.9312509, 1.8919173]])
    b6 = fa.DepthInstance3DBoxes(th, box_dim=6, with_yaw=False).tensor
    _, br = pl.rotate(torch.zeros((1, 3)), b6, -0.04599790655000615, False)
    assert torch.allclose(br, torch.tensor([[0.64884546, 0.78390356, 0.10563634, 1.50373348, 0.23795205, 0.27956772, 0],
                                            [1.45139421, 0.43169443, 0.93829232, 0.11967964, 0.93380373, 1.89191735, 0]]),
                          atol=1e-6)


def test_split_bf16_arithmetic_properties():
    """oracle/x6_oracle.py — what fcaf3d_amd/csrc/conv_x6.h relies on: the three-way bf16 split of an fp32 value is exact, every
    piece is a bf16, every piece product is exact in fp32, the dropped products sum to at most 2^-24 of the product and have
    no common sign (round-to-nearest split, the kernels' since r4; the truncating split of r3: 2^-21, all towards zero), and a
    K = 1728 reduction (27 offsets x 64 channels) done the kernels' way is as close to fp64 as a plain fp32 accumulation."""
    from oracle import x6_oracle as X
    rng = np.random.default_rng(0)
    # values over the whole fp32 range, incl. denormal-adjacent magnitudes and exact powers of two, zeros, negatives
    x = np.concatenate([rng.standard_normal(20000).astype(np.float32) * np.float32(10.0) ** rng.integers(-30, 30, 20000).astype(np.float32),
                        np.float32([0.0, -0.0, 1.0, -1.0, 2.0 ** -120, 3.0, 16777215.0, 1.0 + 2.0 ** -23, -(1.0 - 2.0 ** -24),
                                    1.0 + 2.0 ** -8, 1.0 + 2.0 ** -8 + 2.0 ** -23, 255.5, 1.998046875, 1.99999988])])
    y = rng.permutation(x)
    for split, b2, b3, worst, mean_signed in ((X.split3, 2.0 ** -8, 2.0 ** -17, 2.0 ** -24, 2.0 ** -29),
                                              (X.split3_trunc, 2.0 ** -7, 2.0 ** -15, 2.0 ** -21, None)):
        x1, x2, x3 = split(x)
        assert np.array_equal((x1.astype(np.float64) + x2 + x3).astype(np.float32), x)          # exact (the fp64 sum is exact too)
        assert np.array_equal(x1.astype(np.float64) + x2.astype(np.float64) + x3.astype(np.float64), x.astype(np.float64))
        for p in (x1, x2, x3):
            assert not np.any(p.view(np.uint32) & np.uint32(0xffff))                            # representable in bf16
        nz = x != 0
        assert np.all(np.abs(x2[nz]) <= np.abs(x[nz]) * b2) and np.all(np.abs(x3[nz]) <= np.abs(x[nz]) * b3)
        py = split(y)
        px = (x1, x2, x3)
        keep = np.zeros(len(x), np.float64)
        for i, j in X.TERMS:
            prod64 = px[i].astype(np.float64) * py[j].astype(np.float64)
            with np.errstate(over='ignore', under='ignore'):
                prod32 = px[i] * py[j]
            fin = np.isfinite(prod32) & (np.abs(prod64) > 1e-30)                                 # away from fp32 overflow / underflow
            assert np.array_equal(prod32[fin].astype(np.float64), prod64[fin])                   # piece products are exact in fp32
            keep += prod64
        exact = x.astype(np.float64) * y.astype(np.float64)
        ok = np.abs(exact) > 1e-300
        signed = (keep[ok] - exact[ok]) / exact[ok]
        assert np.abs(signed).max() <= worst and np.abs(signed).mean() <= worst / 8, (np.abs(signed).max(), np.abs(signed).mean())
        if mean_signed is not None:
            assert abs(signed.mean()) <= mean_signed, signed.mean()                               # unbiased: no common sign
        else:
            assert np.all(signed <= 0)                                                           # truncation: every product shrunk
    # a conv-shaped reduction: 27 x 64 terms, ReLU-like activations
    a = np.maximum(rng.standard_normal((64, 1728)), 0).astype(np.float32)
    w = (rng.standard_normal((1728, 32)) * 0.05).astype(np.float32)
    ref = a.astype(np.float64) @ w.astype(np.float64)
    got = X.matmul_x6(a, w)
    plain = np.zeros((64, 32), np.float32)
    for k in range(1728):
        plain = plain + np.outer(a[:, k], w[k]).astype(np.float32)
    scale = np.abs(ref).max()
    e6, e32 = np.sqrt(((got - ref) ** 2).mean()) / scale, np.sqrt(((plain - ref) ** 2).mean()) / scale
    assert e6 < 3e-7 and e6 <= 1.2 * e32, (e6, e32)


def test_two_piece_fp16_split_properties():
    """oracle/x6_oracle.py h3 — what csrc/conv_x6.h MODE 2 relies on (r6): with the tensor-wide power-of-two scale no piece leaves
    fp16's range, the pair (h, l) holds the scaled value to 2^-23 relative — ONE fp32 ulp, and exactly for three values in four: the residual
    s x - h has at most 12 significant bits — where the low piece is normal and to 2^-25 absolute (= 2^-39 of the tensor's maximum)
    below, the three kept piece products are exact in fp32, the dropped part of a product is at most 2^-21 of it (2^-22 for the
    low x low product, 2^-23 per residual) and 2^-25 of it on average, with no common sign, and a conv-shaped K = 1728 reduction lands as close to fp64 as plain fp32 accumulation does."""
    from oracle import x6_oracle as X
    rng = np.random.default_rng(1)
    for mag in (1e-30, 3e-7, 1.0, 77.0, 4e3, 1e20, 3e38):
        x = (rng.standard_normal(20000) * np.exp(2.0 * rng.standard_normal(20000))).astype(np.float32)
        x = (x / np.abs(x).max() * np.float32(mag)).astype(np.float32)
        x[:6] = np.float32([0.0, -0.0, mag, -mag, mag * 2.0 ** -20, mag * (1 + 2.0 ** -11)])
        am = X.amax_finite(x)
        s = X.h3_scale(am)
        assert 2.0 ** 14 <= float(am) * float(s) < 2.0 ** 15, (mag, float(am) * float(s))           # scale rule: top element in [2^14, 2^15)
        assert float(s) == 2.0 ** round(np.log2(float(s)))                                           # a power of two: scaling is exact
        h, lo = X.split2_h(x, s)
        assert np.all(np.isfinite(h.astype(np.float32))) and np.all(np.isfinite(lo.astype(np.float32)))
        xs = x.astype(np.float64) * float(s)
        err = np.abs(h.astype(np.float64) + lo.astype(np.float64) - xs)
        normal_lo = np.abs(xs) >= 0.25                  # low piece >= 2^-14: a normal fp16
        assert np.all(err[normal_lo] <= np.abs(xs[normal_lo]) * 2.0 ** -23)
        assert np.all(err <= np.maximum(np.abs(xs) * 2.0 ** -23, 2.0 ** -25))
        assert (err[normal_lo] == 0).mean() >= 0.7                                                   # exact for about three values in four
        assert np.all(np.abs(lo.astype(np.float64)) <= np.abs(xs) * 2.0 ** -11 + 2.0 ** -25)
    # piece products exact in fp32; dropped part of a product
    x = (rng.standard_normal(50000) * np.exp(rng.standard_normal(50000))).astype(np.float32)
    y = (rng.standard_normal(50000) * 0.05).astype(np.float32)
    sx, sy = X.h3_scale(X.amax_finite(x)), X.h3_scale(X.amax_finite(y))
    px, py = X.split2_h(x, sx), X.split2_h(y, sy)
    keep = np.zeros(len(x), np.float64)
    for i, j in X.TERMS_H3:
        p64 = px[i].astype(np.float64) * py[j].astype(np.float64)
        p32 = px[i].astype(np.float32) * py[j].astype(np.float32)
        assert np.array_equal(p32.astype(np.float64), p64)                                           # 11 x 11 bits: exact in fp32
        keep += p64
    exact = x.astype(np.float64) * float(sx) * y.astype(np.float64) * float(sy)
    big = (np.abs(x.astype(np.float64) * float(sx)) >= 0.25) & (np.abs(y.astype(np.float64) * float(sy)) >= 0.25)
    signed = (keep[big] - exact[big]) / exact[big]
    assert np.abs(signed).max() <= 2.0 ** -21 and np.abs(signed).mean() <= 2.0 ** -24 and abs(signed.mean()) <= 2.0 ** -28, \
        (np.abs(signed).max(), np.abs(signed).mean(), signed.mean())
    # unscale
    assert float(X.h3_unscale(sx, sy)) * float(sx) * float(sy) == 1.0
    # a conv-shaped reduction: 27 x 64 terms, ReLU-like activations; heavy-tailed gradients
    for a in (np.maximum(rng.standard_normal((64, 1728)), 0).astype(np.float32),
              (rng.standard_normal((64, 1728)) * np.exp(2.0 * rng.standard_normal((64, 1728))) * 1e-6).astype(np.float32)):
        w = (rng.standard_normal((1728, 32)) * 0.05).astype(np.float32)
        ref = a.astype(np.float64) @ w.astype(np.float64)
        got = X.matmul_h3(a, w)
        plain = np.zeros((64, 32), np.float32)
        for k in range(1728):
            plain = plain + np.outer(a[:, k], w[k]).astype(np.float32)
        scale = np.abs(ref).max()
        e3, e32 = np.sqrt(((got - ref) ** 2).mean()) / scale, np.sqrt(((plain - ref) ** 2).mean()) / scale
        assert e3 < 3e-7 and e3 <= 1.2 * e32, (e3, e32)
