"""CPU: host-side logic of the product (registry/config shim, module graph & parameter names, target
assignment, box decoding, centerness) against the golden vectors generated from the reference's own
pure-torch code (tests/golden/make_golden.py)."""
import os

import numpy as np
import pytest
import torch

import fcaf3d_amd as fa

G = os.path.join(os.path.dirname(__file__), 'golden')


def _gt(boxes):
    return fa.DepthInstance3DBoxes(torch.from_numpy(boxes), origin=(.5, .5, .5))


def test_assigner_matches_reference_goldens():
    d = np.load(os.path.join(G, 'assigner.npz'))
    for ci in range(int(d['n_cases'])):
        L = int(d[f'c{ci}_n_scales'])
        pts = [torch.from_numpy(d[f'c{ci}_points{l}']) for l in range(L)]
        a = fa.Fcaf3DAssigner(limit=27, topk=18, n_scales=L)
        ct, bt, lb = a.assign(pts, _gt(d[f'c{ci}_gt']), torch.from_numpy(d[f'c{ci}_labels']))
        ref_lb = d[f'c{ci}_assigned']
        assert np.array_equal(lb.numpy(), ref_lb), f'case {ci}'
        pos = ref_lb >= 0
        assert pos.sum() >= 12 * len(d[f'c{ci}_gt'])               # the case is not vacuous (topk = 18 locations per box, minus overlaps)
        assert np.allclose(ct.numpy()[pos], d[f'c{ci}_centerness'][pos], atol=1e-6)
        assert np.allclose(bt.numpy()[pos], d[f'c{ci}_bbox_targets'][pos], atol=1e-6)


def test_assigner_no_boxes():
    a = fa.Fcaf3DAssigner(limit=27, topk=18, n_scales=2)
    ct, bt, lb = a.assign([torch.rand(10, 3), torch.rand(4, 3)], _gt(np.zeros((0, 7), np.float32)),
                          torch.zeros(0, dtype=torch.long))
    assert (lb == -1).all() and len(lb) == 14 and bt.shape == (14, 7)


def test_decode_and_centerness_goldens():
    d = np.load(os.path.join(G, 'decode.npz'))
    head = fa.Fcaf3DNeckWithHead.__new__(fa.Fcaf3DNeckWithHead)
    pts = torch.from_numpy(d['points'])
    head.yaw_parametrization = 'fcaf3d'
    assert np.allclose(head._bbox_pred_to_bbox(pts, torch.from_numpy(d['pred6'])).numpy(), d['out6'], atol=1e-6)
    assert np.allclose(head._bbox_pred_to_bbox(pts, torch.from_numpy(d['pred8'])).numpy(), d['out8_fcaf3d'], atol=1e-6)
    head.yaw_parametrization = 'sin-cos'
    assert np.allclose(head._bbox_pred_to_bbox(pts, torch.from_numpy(d['pred8'])).numpy(), d['out8_sin-cos'], atol=1e-6)
    head.yaw_parametrization = 'naive'
    assert np.allclose(head._bbox_pred_to_bbox(pts, torch.from_numpy(d['pred8'][:, :7])).numpy(), d['out7_naive'], atol=1e-6)
    assert np.allclose(fa.compute_centerness(torch.from_numpy(d['cent_in'])).numpy(), d['cent_out'], atol=1e-6)
    assert head._bbox_pred_to_bbox(pts[:0], torch.zeros(0, 6)).shape == (0, 6)


def test_configs_build_with_reference_names_and_shapes():
    cfg = fa.get_config('fcaf3d_scannet-3d-18class')
    assert cfg.model.voxel_size == 0.01 and cfg.model.neck_with_head.loss_bbox.with_yaw is False
    assert cfg.model.neck_with_head.assigner.n_scales == 4 and cfg.optimizer.type == 'AdamW'
    model = fa.build_detector(cfg.model, train_cfg=cfg.model.get('train_cfg'), test_cfg=cfg.model.get('test_cfg'))
    sd = model.state_dict()
    shapes = {
        'backbone.conv1.0.kernel': (27, 3, 64), 'backbone.conv1.1.weight': (1, 64),
        'backbone.layer1.0.conv1.kernel': (27, 64, 64), 'backbone.layer1.0.downsample.0.kernel': (1, 64, 64),
        'backbone.layer1.0.downsample.1.bn.running_var': (64,), 'backbone.layer2.0.conv1.kernel': (27, 64, 128),
        'backbone.layer3.5.conv2.kernel': (27, 256, 256), 'backbone.layer4.2.norm2.bn.weight': (512,),
        'neck_with_head.up_block_1.0.kernel': (8, 128, 64), 'neck_with_head.up_block_3.3.kernel': (27, 256, 256),
        'neck_with_head.out_block_0.0.kernel': (27, 64, 128), 'neck_with_head.out_block_3.1.bn.bias': (128,),
        'neck_with_head.centerness_conv.kernel': (128, 1), 'neck_with_head.reg_conv.kernel': (128, 6),
        'neck_with_head.cls_conv.kernel': (128, 18), 'neck_with_head.cls_conv.bias': (1, 18),
        'neck_with_head.scales.3.scale': (),
    }
    for k, s in shapes.items():
        assert tuple(sd[k].shape) == s, (k, sd[k].shape)
    n_params = sum(p.numel() for p in model.parameters())
    assert 69e6 < n_params < 72e6, n_params          # ~70.4 M
    assert abs(float(sd['neck_with_head.cls_conv.bias'][0, 0]) + np.log(99)) < 1e-5
    for name in ('fcaf3d_sunrgbd-3d-10class', 'fcaf3d_s3dis-3d-5class', 'fcaf3d_2scales_scannet-3d-18class',
                 'fcaf3d_3scales_scannet-3d-18class'):
        c = fa.get_config(name)
        fa.build_detector(c.model, train_cfg=c.model.get('train_cfg'), test_cfg=c.model.get('test_cfg'))
    c2 = fa.get_config('fcaf3d_2scales_scannet-3d-18class')
    assert c2.model.voxel_size == 0.02 and c2.model.backbone.n_outs == 2 and c2.model.neck_with_head.n_classes == 18
    c3 = fa.get_config('fcaf3d_scannet-3d-18class', voxel_size=0.02)
    assert c3.model.neck_with_head.voxel_size == 0.02


def test_registry_errors():
    import pytest
    with pytest.raises(KeyError):
        fa.build_backbone(dict(type='NoSuchBackbone'))
    with pytest.raises(ValueError):
        fa.build_backbone(dict(type='MEResNet3D', in_channels=3, depth=7))
    with pytest.raises(TypeError):
        fa.build_loss('IoU3DLoss')


def test_boxes_container():
    b = fa.DepthInstance3DBoxes(torch.tensor([[1., 2., 3., 2., 4., 6., 0.5]]), origin=(.5, .5, .5))
    assert torch.allclose(b.tensor[0, :3], torch.tensor([1., 2., 0.]))
    assert torch.allclose(b.gravity_center[0], torch.tensor([1., 2., 3.]))
    assert float(b.volume[0]) == 48.0 and len(b) == 1
    b6 = fa.DepthInstance3DBoxes(torch.zeros(0, 6), box_dim=6, with_yaw=False, origin=(.5, .5, .5))
    assert b6.tensor.shape == (0, 7) and b6.with_yaw is False


def test_segmented_topk_equals_the_per_segment_loop():
    """the batched candidate selection of get_bboxes == the reference's loop (`if len(scores) > nms_pre: topk(nms_pre)` per
    scene and level, fcaf3d_neck_with_head.py:238-243): same rows, same order; segments interleaved in memory, empty and
    exactly-k segments, tied scores"""
    from fcaf3d_amd.fcaf3d_neck_with_head import segmented_topk
    g = torch.Generator().manual_seed(1)
    n_seg, k = 7, 50
    sizes = [0, 1, 49, 50, 51, 400, 1300]
    seg = torch.cat([torch.full((n,), i, dtype=torch.int64) for i, n in enumerate(sizes)])
    seg = seg[torch.randperm(seg.numel(), generator=g)]                     # rows of a segment are not contiguous
    for trial, scale in enumerate((1.0, 1e-9, 1e-3, 1e-30)):                 # tiny scores too: no precision floor
        score = torch.rand(seg.numel(), generator=g) * scale
        score[::7] = score[3]                                                # ties
        sel = segmented_topk(seg, score, n_seg, k)
        want = []
        for s_ in range(n_seg):
            rows = torch.nonzero(seg == s_).flatten()
            if len(rows) > k:
                sc = score[rows]
                idx = torch.sort(-sc.double(), stable=True).indices[:k]      # descending score, ties in row order
                rows = rows[idx]
            want.append(rows)
        assert torch.equal(sel, torch.cat(want)), (trial, scale)


def test_bbox3d2result_batch_equals_per_scene_conversion():
    """simple_test's batched result conversion (three copies for the whole batch) == bbox3d2result per scene
    (mmdet3d/core/bbox/transforms.py bbox3d2result: boxes / scores / labels on the CPU), empty scenes included"""
    from fcaf3d_amd.boxes import bbox3d2result, bbox3d2result_batch
    g = torch.Generator().manual_seed(0)
    lst = []
    for n, dim in ((5, 7), (0, 7), (3, 7)):
        b = fa.DepthInstance3DBoxes(torch.rand(n, dim, generator=g), box_dim=dim, with_yaw=True, origin=(.5, .5, .5))
        lst.append((b, torch.rand(n, generator=g), torch.randint(0, 18, (n,), generator=g)))
    got = bbox3d2result_batch(lst)
    assert len(got) == 3 and bbox3d2result_batch([]) == []
    for (b, s, l), r in zip(lst, got):
        ref = bbox3d2result(b, s, l)
        assert torch.equal(r['boxes_3d'].tensor, ref['boxes_3d'].tensor) and type(r['boxes_3d']) is type(ref['boxes_3d'])
        assert r['boxes_3d'].box_dim == ref['boxes_3d'].box_dim and r['boxes_3d'].with_yaw == ref['boxes_3d'].with_yaw
        assert torch.equal(r['scores_3d'], ref['scores_3d']) and torch.equal(r['labels_3d'], ref['labels_3d'])
        assert torch.equal(r['boxes_3d'].gravity_center, ref['boxes_3d'].gravity_center)
    # 6-dim (yaw-less) boxes keep their flags
    b6 = fa.DepthInstance3DBoxes(torch.rand(4, 6, generator=g), box_dim=6, with_yaw=False, origin=(.5, .5, .5))
    r6 = bbox3d2result_batch([(b6, torch.rand(4, generator=g), torch.zeros(4, dtype=torch.long))])[0]
    assert r6['boxes_3d'].with_yaw is False and r6['boxes_3d'].box_dim == b6.box_dim


def test_checkpoint_roundtrip_mmcv_layout(tmp_path):
    """mmcv-layout checkpoints ({'meta','state_dict'}, optional DDP 'module.' prefix) load into the detector;
    mismatches are reported, not silently dropped (tools/test.py:172 flow of the reference)"""
    import torch
    import fcaf3d_amd as fa
    from fcaf3d_amd.checkpoint import load_checkpoint, load_state_dict, save_checkpoint
    cfg = fa.get_config('fcaf3d_scannet-3d-18class', voxel_size=0.02)
    a = fa.build_detector(cfg.model, train_cfg=cfg.get('train_cfg'), test_cfg=cfg.get('test_cfg'))
    b = fa.build_detector(cfg.model, train_cfg=cfg.get('train_cfg'), test_cfg=cfg.get('test_cfg'))
    with torch.no_grad():
        for p in a.parameters():
            p.add_(torch.randn_like(p) * 0.01)
    path = str(tmp_path / 'ck.pth')
    ck = save_checkpoint(a, path, meta=dict(epoch=12))
    assert set(ck) == {'meta', 'state_dict'} and ck['meta']['epoch'] == 12
    out = load_checkpoint(b, path, map_location='cpu', strict=True)
    assert out['meta']['epoch'] == 12
    for (k, x), (_, y) in zip(a.state_dict().items(), b.state_dict().items()):
        assert torch.equal(x, y), k
    # DDP-prefixed, bare state_dict, one wrong shape, one unknown key
    sd = {'module.' + k: v.clone() for k, v in a.state_dict().items()}
    sd['module.backbone.conv1.0.kernel'] = torch.zeros(27, 3, 32)
    sd['module.not_a_param'] = torch.zeros(1)
    torch.save(sd, path)
    c = fa.build_detector(cfg.model, train_cfg=cfg.get('train_cfg'), test_cfg=cfg.get('test_cfg'))
    msgs = []

    class Log:
        def warning(self, m):
            msgs.append(m)
    load_checkpoint(c, path, map_location='cpu', strict=False, logger=Log())
    assert 'size mismatch for backbone.conv1.0.kernel' in msgs[0] and 'not_a_param' in msgs[0]
    assert torch.equal(c.state_dict()['backbone.layer1.0.conv1.kernel'], a.state_dict()['backbone.layer1.0.conv1.kernel'])
    import pytest
    with pytest.raises(RuntimeError):
        load_state_dict(c, {k[len('module.'):]: v for k, v in sd.items()}, strict=True)


def test_weight_image_table_layout():
    """functional.WeightImages: one buffer, one descriptor table for every convolution kernel and both directions — entries
    only for shapes the split kernels take (reduction % 32, columns % 64), 256-byte aligned images of 6 K R C bytes, block
    ranges that tile [0, total) in descriptor order, lookup by (data pointer, K, Cin, Cout) incl. the 2-D 1x1 kernels."""
    import fcaf3d_amd.functional as Fn
    ws = [torch.zeros(27, 64, 128), torch.zeros(27, 3, 64), torch.zeros(128, 64), torch.zeros(8, 32, 64), torch.zeros(27, 64, 32)]
    wi = Fn.WeightImages(ws)
    d = wi.desc.view(-1, 8).tolist()
    # (27,64,128): both directions; (27,3,64): none; (128,64) as (1,128,64): both; (8,32,64): forward only (backward: reduction 64, columns 32);
    # (27,64,32): backward only (forward columns 32)
    assert wi.n == len(d) == 6
    blocks = 0
    for w_ptr, img_ptr, K, R, C, tr, first, _ in d:
        assert R % 32 == 0 and C % 64 == 0 and tr in (0, 1)
        assert first == blocks
        blocks += K * (R // 32) * (C // 64)
        assert (img_ptr - wi.buf.data_ptr()) % 256 == 0
        assert img_ptr + 6 * K * R * C <= wi.buf.data_ptr() + wi.buf.numel()
    assert blocks == wi.blocks
    f, b = wi.table[(ws[0].data_ptr(), 27, 64, 128)]
    assert f.numel() == 6 * 27 * 64 * 128 == b.numel() and f.data_ptr() != b.data_ptr()
    assert (ws[1].data_ptr(), 27, 3, 64) not in wi.table
    f, b = wi.table[(ws[2].data_ptr(), 1, 128, 64)]
    assert f is not None and b is not None
    f, b = wi.table[(ws[3].data_ptr(), 8, 32, 64)]
    assert f is not None and b is None
    f, b = wi.table[(ws[4].data_ptr(), 27, 64, 32)]
    assert f is None and b is not None
    # images never overlap
    spans = sorted((r[1], r[1] + 6 * r[2] * r[3] * r[4]) for r in d)
    assert all(a[1] <= b[0] for a, b in zip(spans, spans[1:]))


def test_nms_upper_triangle_block_enumeration():
    """csrc/nms.hip k_nms_mask (r4): block t of the grid works on the block pair (rb, cb), rb <= cb, t = cb (cb + 1) / 2 + rb, found
    from t with a float32 square root and two correcting loops.  The same arithmetic here in numpy float32: a bijection onto the
    upper triangle for every grid the entry point accepts (up to 1 024 column blocks = 65 536 boxes per segment)."""
    nb = 1024
    t = np.arange(nb * (nb + 1) // 2, dtype=np.int64)
    cb = ((np.sqrt(np.float32(8.0) * t.astype(np.float32) + np.float32(1.0)) - np.float32(1.0)) * np.float32(0.5)).astype(np.int64)
    for _ in range(4):                                   # the kernel's while loops: at most one step each way is ever needed
        cb = np.where((cb + 1) * (cb + 2) // 2 <= t, cb + 1, cb)
    for _ in range(4):
        cb = np.where(cb * (cb + 1) // 2 > t, cb - 1, cb)
    rb = t - cb * (cb + 1) // 2
    assert (rb >= 0).all() and (rb <= cb).all() and (cb < nb).all()
    assert np.unique(cb * nb + rb).size == t.size
    # one correcting step suffices
    cb0 = ((np.sqrt(np.float32(8.0) * t.astype(np.float32) + np.float32(1.0)) - np.float32(1.0)) * np.float32(0.5)).astype(np.int64)
    assert np.abs(cb0 - cb).max() <= 1


def test_committed_profile_tables_follow_from_the_committed_traces():
    """profiles/r4_kernel_stats.md is tools/kernel_stats.py over the two committed rocprofv3 --stats CSVs, and bench.py's
    `roofline.traffic` is what profiles/r4_traffic.json holds: the judged summaries cannot drift from their raw data."""
    import importlib.util
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('kernel_stats', os.path.join(root, 'tools', 'kernel_stats.py'))
    ks = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ks)
    md = open(os.path.join(root, 'profiles', 'r4_kernel_stats.md')).read()
    for name in ('r4_kernel_stats.csv', 'r4_kernel_stats_no_overlap.csv'):
        assert ks.table(os.path.join(root, 'profiles', name), 30.0) in md, name
    traffic = json.load(open(os.path.join(root, 'profiles', 'r4_traffic.json')))
    bench_line = json.load(open(os.path.join(root, 'profiles', 'r4_bench_n1.json')))
    assert bench_line['roofline']['traffic'] == round(traffic['hbm_bytes_per_launch'])
    assert abs(traffic['hbm_bytes_per_launch'] - traffic['fetch_bytes_per_launch'] - traffic['write_bytes_per_launch']) <= 1
    r = bench_line['roofline']
    assert abs(r['frac'] - r['achieved'] / r['peak']) < 1e-3


def _check_profile_take(tag, must_match_source):
    import importlib.util
    import json
    from fcaf3d_amd.build import source_hash
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    meta_path = os.path.join(root, 'profiles', f'{tag}_meta.json')
    if not os.path.exists(meta_path):
        pytest.skip(f'no {tag} profiles committed yet')
    meta = json.load(open(meta_path))
    if must_match_source:
        assert meta['kernel_source_sha16'] == source_hash(), f'csrc/ changed after the {tag} profiles were taken: take them again'
    spec = importlib.util.spec_from_file_location('kernel_stats', os.path.join(root, 'tools', 'kernel_stats.py'))
    ks = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ks)
    md = open(os.path.join(root, 'profiles', f'{tag}_kernel_stats.md')).read()
    for name in (f'{tag}_kernel_stats.csv', f'{tag}_kernel_stats_no_overlap.csv'):
        assert ks.table(os.path.join(root, 'profiles', name), float(meta['steps_profiled']), anon=bool(meta.get('anon_names'))) in md, name
    bench_line = json.load(open(os.path.join(root, 'profiles', f'{tag}_bench_n1.json')))
    assert bench_line['config']['kernel_source_sha16'] == meta['kernel_source_sha16']
    return meta, bench_line


def test_r5_profiles_are_one_consistent_take():
    """profiles/r5_meta.json: the r5 kernel table is tools/kernel_stats.py over the committed CSVs and the r5 bench line carries the
    hash the profiles were taken on (r5's kernels; r6 changed csrc/, its own take is checked below)"""
    _check_profile_take('r5', must_match_source=False)


def test_r6_profiles_were_taken_on_the_committed_kernels():
    """profiles/r6_meta.json records the hash of csrc/ (fcaf3d_amd.build.source_hash) every r6 profile was taken on; a kernel change
    after the profiles makes this fail (the r4 weight-gradient rows predated the final routing).  The line's
    `roofline.traffic` is read from the traffic file of the same take."""
    import json
    meta, line = _check_profile_take('r6', must_match_source=True)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    traffic = json.load(open(os.path.join(root, 'profiles', 'r6_traffic.json')))
    assert line['roofline']['traffic'] == round(traffic['hbm_bytes_per_launch'])
