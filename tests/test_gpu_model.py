"""-m gpu: the detector end to end (HIP path through the reference's plugin API) against the CPU oracle
on identical synthetic scenes and identical weights; loss kernels and NMS against the reference goldens.
Tolerances: integer outputs exact; fp32 1e-4 relative to tensor scale (gradients after ~40 layers: 1e-3)."""
import copy
import os

import numpy as np
import pytest
import torch

import fcaf3d_amd as fa
from fcaf3d_amd.synthetic import make_scene
from oracle import bev, loss_oracle as lo, model_oracle as MO

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), 'golden')


def _dev():
    assert torch.cuda.is_available()
    return torch.device('cuda:0')


def _rel(a, b):
    a = a.detach().cpu().double(); b = b.detach().cpu().double()
    assert a.shape == b.shape, (a.shape, b.shape)
    if b.numel() == 0:
        return 0.0
    return float((a - b).abs().max()) / max(1e-3, float(b.abs().max()))


def _build(name, voxel_size, n_levels, seed=0, **head_over):
    torch.manual_seed(seed)
    cfg = fa.get_config(name, voxel_size=voxel_size)
    m = cfg.model
    m.backbone['n_outs'] = n_levels
    m.neck_with_head['in_channels'] = (64, 128, 256, 512)[:n_levels]
    m.neck_with_head.assigner['n_scales'] = n_levels
    for k, v in head_over.items():
        m.neck_with_head[k] = v
    model = fa.build_detector(m, train_cfg=m.get('train_cfg'), test_cfg=m.get('test_cfg'))
    return model, m


def _scenes(seeds, **kw):
    out = [make_scene(s, **kw) for s in seeds]
    return [o[0] for o in out], [o[1] for o in out], [o[2] for o in out]


def _to_gpu_batch(pts, gts, labs, dev):
    return dict(points=[torch.from_numpy(p).to(dev) for p in pts],
                gt_bboxes_3d=[fa.DepthInstance3DBoxes(torch.from_numpy(g), origin=(.5, .5, .5)) for g in gts],
                gt_labels_3d=[torch.from_numpy(l).to(dev) for l in labs],
                img_metas=[dict(box_type_3d=fa.DepthInstance3DBoxes) for _ in pts])


def _oracle_params(model):
    return {k: v.detach().cpu().clone().requires_grad_(v.dtype.is_floating_point)
            for k, v in model.state_dict().items()}


SUNRGBD_KW = dict(rotated=True, single_view=True, rgb_unit=True, n_boxes=6, n_classes=10)


from oracle.record import RecordDecisions as _RecordDecisions


# rows of the deepest stage: a (scene, level-4) set of 109-862 voxels — BatchNorm statistics over so few rows amplify
# rounding differences of the same arithmetic by the conditioning of 1 / sigma (printed apart; r6: the same 1e-4 bound as everywhere,
# arbitrated by the fp64 oracle where the fp32 oracle is further away than that)
DEEP = ('backbone.layer4.', 'neck_with_head.up_block_3.', 'neck_with_head.out_block_3.')


@pytest.mark.parametrize('name,levels,B,n_points,kw,x6', [
    ('fcaf3d_scannet-3d-18class', 1, 1, 20000, {}, True),                       # BASELINE config 1 (plumbing)
    ('fcaf3d_scannet-3d-18class', 4, 2, 30000, {}, True),                       # 4 levels, 2 scenes
    ('fcaf3d_scannet-3d-18class', 4, 2, 30000, {}, False),                      # ... on the fp32 MFMA route (FC_X6=0)
    ('fcaf3d_sunrgbd-3d-10class', 2, 1, 20000, dict(rotated=True, n_boxes=6, n_classes=10), True),   # rotated IoU loss
    ('fcaf3d_scannet-3d-18class', 4, 1, 100000, {}, True),                      # BASELINE config 2 at FULL size
    ('fcaf3d_scannet-3d-18class', 4, 1, 100000, {}, False),                     # ... on the fp32 MFMA route
    ('fcaf3d_sunrgbd-3d-10class', 4, 1, 100000, SUNRGBD_KW, True),              # BASELINE config 3 at FULL size (4 levels, rotated)
    ('fcaf3d_sunrgbd-3d-10class', 4, 1, 100000, SUNRGBD_KW, False),
])
def test_forward_train_parity(name, levels, B, n_points, kw, x6):
    """Forward outputs, losses and EVERY parameter gradient of the detector against the CPU oracle on identical scenes and
    weights, for the split-bf16 route (default) and the fp32 MFMA route.

    Gradients are compared with EQUAL DISCRETE DECISIONS: the HIP forward's ReLU signs and max-pool
    arg-max rows are recorded and the fp32 oracle is run with those decisions (oracle.model_oracle.DecisionTape), so that both
    sides differentiate the same piecewise-smooth function.  Then every gradient must agree to 1e-4 of the tensor's scale with the fp32 oracle, or — where the fp32
    oracle itself is further than that from the fp64 gradient — be as close to the fp64 gradient as the fp32 oracle is: no envelope.  r3 compared gradients across DIFFERENT decisions (a pre-activation of
    +-1e-8 falls on either side of zero in any two fp32 implementations) and needed a 6e-2 envelope for it, which a 5 %
    defect in one weight-gradient variant would have passed; what was attributed to flips then is counted here: the number of
    elements where the oracle's own pre-activation has the other sign is printed and bounded, with the magnitude of those
    pre-activations (fp32 rounding level)."""
    import fcaf3d_amd.functional as Fn
    dev = _dev()
    model, m = _build(name, 0.02, levels)
    P = _oracle_params(model)
    model = model.to(dev).train()
    pts, gts, labs = _scenes(range(10, 10 + B), n_points=n_points, **kw)
    x6_0 = Fn.X6
    Fn.X6 = x6
    try:
        # forward outputs
        out_o = MO.extract_feat(P, m, pts)
        out_g = [list(x) for x in model.extract_feat([torch.from_numpy(p).to(dev) for p in pts], None)]
        for kind in range(4):
            for l in range(levels):
                for b in range(B):
                    if kind == 3:
                        assert torch.equal(out_g[kind][l][b].cpu(), out_o[kind][l][b]), 'points (voxel corners) must be exact'
                    else:
                        assert _rel(out_g[kind][l][b], out_o[kind][l][b]) < 1e-4, (kind, l, b)
        # losses + gradients (fresh forward so that BN running stats are touched once per path)
        model.zero_grad()
        with _RecordDecisions(model) as rec:
            losses_g = model(return_loss=True, **_to_gpu_batch(pts, gts, labs, dev))
        sum(losses_g.values()).backward()
        torch.cuda.synchronize()
    finally:
        Fn.X6 = x6_0
    tape = rec.tape()
    MO.TAPE = tape
    try:
        losses_o = MO.forward_train(P, m, pts, gts, labs)
    finally:
        MO.TAPE = None
    assert tape.i == len(tape.relu), 'the oracle must have consumed every recorded ReLU'
    for k in ('loss_centerness', 'loss_bbox', 'loss_cls'):
        assert _rel(losses_g[k], losses_o[k]) < 1e-4, (k, float(losses_g[k]), float(losses_o[k]))
    sum(losses_o.values()).backward()
    flips = [f for f in tape.flips if f[1]]
    n_dec = sum(f[2] for f in tape.flips)
    print(f'{name} L={levels} B={B} n={n_points} x6={x6}: {tape.total_flips()} of {n_dec} decisions differ between the HIP forward '
          f'and the fp32 oracle\'s own pre-activations: ' + ', '.join(f'{s}: {n} (|pre| <= {mx:.1e})' for s, n, _, mx in flips))
    # a flipped decision sits on a pre-activation at fp32 rounding level of the layer's scale (values are O(1) after a norm)
    assert all(mx < 1e-4 for _, _, _, mx in flips), flips
    assert tape.total_flips() <= 1e-5 * n_dec + 8, tape.total_flips()
    errs = {k: _rel(p.grad, P[k].grad) for k, p in model.named_parameters()}
    worst = max(errs, key=errs.get)
    shallow = {k: v for k, v in errs.items() if not k.startswith(DEEP)}
    deep = {k: v for k, v in errs.items() if k.startswith(DEEP)}
    ws, wd = max(shallow, key=shallow.get), (max(deep, key=deep.get) if deep else None)
    print(f'   gradients vs the fp32 oracle with equal decisions: worst {errs[worst]:.2e} ({worst}); outside the deepest stage '
          f'{shallow[ws]:.2e} ({ws}); deepest stage {deep[wd] if wd else 0:.2e} ({wd}); median {np.median(list(errs.values())):.2e}')
    # r6: north_star's bound is 1e-4 EVERYWHERE.  Every tensor beyond it against the fp32 oracle — the deepest
    # stage included, which r3-r5 passed at 1e-3 without a second opinion — is arbitrated by the fp64 oracle with the same
    # decisions: is it the HIP path, or is the fp32 ORACLE itself that far from the exact gradient there?  (BatchNorm over the
    # 109-862 rows of a (scene, level-4) set amplifies the rounding of the same arithmetic by 1 / sigma; the head's 1x1 kernels sum
    # mixed-sign products over every location of the batch: 1.6e-4 on cls_conv.kernel at 2 x 30k points on both convolution
    # routes.)  The HIP gradient must be within 1e-4 of the fp64 gradient or as close to it as the fp32 oracle's own gradient
    # is (within 2x).
    over = [k for k, v in errs.items() if v >= 1e-4]
    rows = []
    if over:
        P64 = {k: (v.detach().double().requires_grad_(True) if v.dtype.is_floating_point else v) for k, v in P.items()}
        MO.TAPE = tape.replay()
        try:
            sum(MO.forward_train(P64, m, pts, gts, labs).values()).backward()
        finally:
            MO.TAPE = None
        named = dict(model.named_parameters())
        for k in over:
            e_hip, e_o = _rel(named[k].grad, P64[k].grad), _rel(P[k].grad, P64[k].grad)
            rows.append((k, errs[k], e_hip, e_o))
            print(f'   {k}: vs the fp32 oracle {errs[k]:.2e}; vs the fp64 oracle (same decisions): HIP {e_hip:.2e}, fp32 oracle {e_o:.2e}')
    _note_parity(f'{name} L={levels} B={B} n={n_points} x6={x6}', errs, worst, rows)
    for k, e32, e_hip, e_o in rows:
        assert e_hip < max(1e-4, 2.0 * e_o), (k, e_hip, e_o)


def _note_parity(case, errs, worst, rows):
    """the table profiles/r6_notes.md quotes: appended to gpurun_out/parity_table.md when the tests run through tools/lease.sh"""
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out')
    if not os.path.isdir(out):
        return
    with open(os.path.join(out, 'parity_table.md'), 'a') as fh:
        fh.write(f'| {case} | {len(errs)} | {errs[worst]:.2e} ({worst}) | {float(np.median(list(errs.values()))):.2e} | '
                 + ('; '.join(f'{k}: fp32-oracle {a:.1e}, HIP-vs-fp64 {b:.1e}, oracle-vs-fp64 {c:.1e}' for k, a, b, c in rows) or 'none') + ' |\n')


def test_bottleneck_backbone_parity_depth50():
    """MEResNet3D depth 50 (ME Bottleneck blocks: 1x1 - 3x3 - 1x1 with a 4x expansion, me_resnet.py:114-119; no FCAF3D config
    uses it, the class supports it): forward_train losses and gradients against the oracle, 2 levels (256 / 512 channels into the
    neck), one 12k-point scene."""
    dev = _dev()
    torch.manual_seed(2)
    cfg = fa.get_config('fcaf3d_scannet-3d-18class', voxel_size=0.02)
    m = cfg.model
    m.backbone['depth'] = 50
    m.backbone['n_outs'] = 2
    m.neck_with_head['in_channels'] = (256, 512)
    m.neck_with_head.assigner['n_scales'] = 2
    model = fa.build_detector(m, train_cfg=m.get('train_cfg'), test_cfg=m.get('test_cfg'))
    assert type(model.backbone.layer1[0]).__name__ == 'Bottleneck' and len(model.backbone.layer1) == 4
    P = _oracle_params(model)
    model = model.to(dev).train()
    pts, gts, labs = _scenes([91], n_points=12000)
    with _RecordDecisions(model) as rec:
        losses_g = model(return_loss=True, **_to_gpu_batch(pts, gts, labs, dev))
    MO.TAPE = tape = rec.tape()
    try:
        losses_o = MO.forward_train(P, m, pts, gts, labs)          # with the HIP forward's ReLU / arg-max decisions
    finally:
        MO.TAPE = None
    for k in ('loss_centerness', 'loss_bbox', 'loss_cls'):
        assert _rel(losses_g[k], losses_o[k]) < 1e-4, (k, float(losses_g[k]), float(losses_o[k]))
    sum(losses_g.values()).backward()
    sum(losses_o.values()).backward()
    errs = {k: _rel(p.grad, P[k].grad) for k, p in model.named_parameters()}
    worst = max(errs, key=errs.get)
    print(f'depth 50: {tape.total_flips()} decisions differ; gradient error vs the fp32 oracle with equal decisions: worst '
          f'{errs[worst]:.2e} ({worst}), median {np.median(list(errs.values())):.2e}')
    assert errs[worst] < 1e-3 and np.median(list(errs.values())) < 1e-4, (worst, errs[worst])


def test_async_map_stream_is_bitwise_equivalent():
    """Coordinate work on the side HIP stream (bench mode) must not change a single bit."""
    dev = _dev()
    model, m = _build('fcaf3d_scannet-3d-18class', 0.02, 3)
    model = model.to(dev).train()
    pts, gts, labs = _scenes([41, 42, 43], n_points=30000)
    outs = []
    import fcaf3d_amd.functional as Fn
    for mode in (False, True, True):
        model.async_maps = mode
        Fn.WGRAD_ASYNC = mode                # weight gradients on their own stream as well (bench mode)
        model.zero_grad()
        losses = model(return_loss=True, **_to_gpu_batch(pts, gts, labs, dev))
        sum(losses.values()).backward()
        outs.append(([float(v) for v in losses.values()],
                     model.backbone.layer1[0].conv1.kernel.grad.clone(), model.neck_with_head.reg_conv.kernel.grad.clone()))
    torch.cuda.synchronize()
    Fn.WGRAD_ASYNC = False
    for o in outs[1:]:
        assert o[0] == outs[0][0]
        assert torch.equal(o[1], outs[0][1]) and torch.equal(o[2], outs[0][2])


def test_target_assignment_is_stable_beside_concurrent_convolutions():
    """The batched target assignment on a side stream while the main stream runs forward + backward (the bench's overlap of
    step i + 1's coordinate stream with step i's backward) must return bitwise the same targets every time.  r3: with the
    split-bf16 convolutions on the main stream ~30 % of such calls came back with a few dozen rows assigned against a STALE
    kth / best entry — the hand-off tables sit at the same workspace address in every call and a CU's vector L1 still held
    the previous call's line (tools/trace_det.py); csrc/assign.hip now reads them past the L1."""
    dev = _dev()
    model, m = _build('fcaf3d_scannet-3d-18class', 0.02, 4)
    model = model.to(dev).train()
    pts, gts, labs = _scenes([61, 62, 63, 64], n_points=60000)
    batch = _to_gpu_batch(pts, gts, labs, dev)
    head = model.neck_with_head
    seen = {}
    t0 = head._targets

    def grab(cmaps, gtb, gtl, pre=None):
        seen['args'], seen['pre'] = (cmaps, gtb, gtl), pre
        return t0(cmaps, gtb, gtl, pre)
    head._targets = grab
    sum(model(return_loss=True, **batch).values()).backward()
    head._targets = t0
    cmaps, gtb, gtl = seen['args']
    ref = t0(cmaps, gtb, gtl)
    assert seen['pre'] is not None, 'the native plan did not hand its head arrays to the assignment'
    via_plan = t0(cmaps, gtb, gtl, seen['pre'])                   # locations / order written by csrc/plan.hip == the torch-built ones
    torch.cuda.synchronize()
    for k in ('pts', 'scene', 'ct', 'bt', 'labels', 'inv_pos', 'inv_den'):
        assert torch.equal(via_plan[k], ref[k]), k
    side = torch.cuda.Stream(device=dev)
    outs = []
    for rep in range(3):
        model.zero_grad(set_to_none=True)
        losses = model(return_loss=True, **batch)                 # main stream: enqueue a step's worth of convolutions ...
        with torch.cuda.stream(side):                             # ... and assign beside it, again and again
            for _ in range(8):
                outs.append(t0(cmaps, gtb, gtl))
        sum(losses.values()).backward()
    torch.cuda.synchronize()
    for o in outs:
        for k in ('ct', 'bt', 'labels', 'inv_pos', 'inv_den'):
            assert torch.equal(o[k], ref[k]), k


def _fake_cmap(scene_ids, dev):
    from fcaf3d_amd.sparse import CoordMap
    c = torch.zeros((len(scene_ids), 4), dtype=torch.int32, device=dev)
    c[:, 0] = torch.as_tensor(scene_ids, dtype=torch.int32)
    cm = CoordMap.__new__(CoordMap)
    cm.coords, cm.batch_size, cm.n, cm._perm = c, int(max(scene_ids)) + 1 if len(scene_ids) else 1, len(scene_ids), None
    return cm


def test_hip_assigner_vs_reference_goldens():
    """csrc/assign.hip (whole batch, 4 launches) == the reference's Fcaf3DAssigner.assign goldens."""
    dev = _dev()
    d = np.load(os.path.join(G, 'assigner.npz'))
    for ci in range(int(d['n_cases'])):
        Lv = int(d[f'c{ci}_n_scales'])
        lv_pts = [torch.from_numpy(d[f'c{ci}_points{l}']).to(dev) for l in range(Lv)]
        a = fa.Fcaf3DAssigner(limit=27, topk=18, n_scales=Lv)
        pts = torch.cat(lv_pts)
        scene = torch.zeros(len(pts), dtype=torch.int32, device=dev)
        level = torch.cat([torch.full((len(p),), l, dtype=torch.int32, device=dev) for l, p in enumerate(lv_pts)])
        cmaps = [_fake_cmap([0] * len(p), dev) for p in lv_pts]
        gt = fa.DepthInstance3DBoxes(torch.from_numpy(d[f'c{ci}_gt']), origin=(.5, .5, .5))
        ct, bt, lb = a.assign_batched(pts, scene, level, cmaps, [gt], [torch.from_numpy(d[f'c{ci}_labels']).to(dev)])
        ref = d[f'c{ci}_assigned']
        assert np.array_equal(lb.cpu().numpy(), ref), ci
        pos = ref >= 0
        assert np.allclose(ct.cpu().numpy()[pos], d[f'c{ci}_centerness'][pos], atol=1e-6)
        assert np.allclose(bt.cpu().numpy()[pos], d[f'c{ci}_bbox_targets'][pos], atol=1e-6)
        assert float(ct.cpu()[~torch.from_numpy(pos)].abs().sum()) == 0.0


def test_hip_assigner_batched_vs_per_scene_torch():
    dev = _dev()
    rng = np.random.default_rng(0)
    B, Lv = 3, 3
    a = fa.Fcaf3DAssigner(limit=27, topk=18, n_scales=Lv)
    gts, labs, per_scene_pts = [], [], []
    for s in range(B):
        p, g, l = make_scene(50 + s, n_points=8000, n_boxes=[5, 0, 9][s], rotated=(s == 2))
        per_scene_pts.append([torch.from_numpy(np.unique(np.floor(p[:, :3] / (0.16 * 2 ** k)), axis=0).astype(np.float32)
                                               * np.float32(0.16 * 2 ** k)).to(dev) for k in range(Lv)])
        gts.append(fa.DepthInstance3DBoxes(torch.from_numpy(g.reshape(-1, 7)), origin=(.5, .5, .5)))
        labs.append(torch.from_numpy(l).to(dev))
    # batched layout: level-major, scenes interleaved randomly inside a level
    lvl_pts, lvl_scene = [], []
    for k in range(Lv):
        p = torch.cat([per_scene_pts[s][k] for s in range(B)])
        sc = torch.cat([torch.full((len(per_scene_pts[s][k]),), s, dtype=torch.int32) for s in range(B)])
        perm = torch.from_numpy(rng.permutation(len(p)))
        lvl_pts.append(p[perm.to(dev)]); lvl_scene.append(sc[perm])
    pts = torch.cat(lvl_pts)
    scene = torch.cat(lvl_scene).to(dev)
    level = torch.cat([torch.full((len(p),), k, dtype=torch.int32, device=dev) for k, p in enumerate(lvl_pts)])
    cmaps = [_fake_cmap(sc.tolist(), dev) for sc in lvl_scene]
    for cm in cmaps:
        cm.batch_size = B
    ct, bt, lb = a.assign_batched(pts, scene, level, cmaps, gts, labs)
    for s in range(B):
        ct_r, bt_r, lb_r = a.assign(per_scene_pts[s], gts[s], labs[s])
        rows = torch.cat([torch.nonzero((scene == s) & (level == k)).squeeze(1) for k in range(Lv)])
        # map the per-scene reference rows onto the batched rows through the coordinates
        ref_pts = torch.cat(per_scene_pts[s])
        def key(t):
            q = (t * 1000).round().long()
            return q[:, 0] + q[:, 1] * 100003 + q[:, 2] * 10000600009
        lvl_key = torch.cat([torch.full((len(per_scene_pts[s][k]),), k, device=dev) for k in range(Lv)]) * 7
        order_ref = torch.argsort(key(ref_pts) * 16 + lvl_key)
        order_bat = torch.argsort(key(pts[rows]) * 16 + level[rows].long() * 7)
        assert torch.equal(lb[rows][order_bat], lb_r[order_ref]), s
        pos = lb_r[order_ref] >= 0
        assert torch.allclose(ct[rows][order_bat][pos], ct_r[order_ref][pos], atol=1e-6)
        if len(gts[s]):
            assert torch.allclose(bt[rows][order_bat][pos], bt_r[order_ref][pos], atol=1e-6)


def test_batched_loss_equals_per_scene_loop():
    """loss() over SceneLists (one pass over all scenes) == the reference-shaped per-scene loop."""
    dev = _dev()
    model, m = _build('fcaf3d_scannet-3d-18class', 0.02, 3)
    model = model.to(dev).train()
    pts, gts, labs = _scenes([61, 62, 63], n_points=30000)
    batch = _to_gpu_batch(pts, gts, labs, dev)
    x = [list(v) for v in model.extract_feat(batch['points'], batch['img_metas'])]
    head = model.neck_with_head
    fast = head.loss(*x, batch['gt_bboxes_3d'], batch['gt_labels_3d'], batch['img_metas'])
    as_lists = [[[lvl[i] for i in range(len(lvl))] for lvl in kind] for kind in x]      # plain python lists
    slow = head.loss(*as_lists, batch['gt_bboxes_3d'], batch['gt_labels_3d'], batch['img_metas'])
    for k in fast:
        assert _rel(fast[k], slow[k]) < 2e-6, (k, float(fast[k]), float(slow[k]))
    # (through backward(): the native executor delivers parameter gradients into .grad, torch.autograd.grad does not see them)
    k = head.out_block_0[0].kernel
    model.zero_grad(set_to_none=True)
    sum(fast.values()).backward(retain_graph=True)
    g_fast = k.grad.detach().clone()
    model.zero_grad(set_to_none=True)
    sum(slow.values()).backward()
    g_slow = k.grad.detach().clone()
    assert _rel(g_fast, g_slow) < 1e-4


def test_simple_test_parity():
    dev = _dev()
    model, m = _build('fcaf3d_scannet-3d-18class', 0.02, 3)
    with torch.no_grad():
        model.neck_with_head.cls_conv.bias.fill_(0.0)         # random init would score below score_thr
        model.neck_with_head.cls_conv.kernel.normal_(0, 0.3)
    P = _oracle_params(model)
    model = model.to(dev).train()                             # batch-stat BN, as the oracle
    pts, _, _ = _scenes([21, 22], n_points=20000)
    res_o = MO.simple_test(P, m, pts)
    res_g = model(return_loss=False, points=[torch.from_numpy(p).to(dev) for p in pts],
                  img_metas=[dict(box_type_3d=fa.DepthInstance3DBoxes)] * 2)
    for (bo, so, lo_), rg in zip(res_o, res_g):
        assert len(so) > 10
        assert len(rg['scores_3d']) == len(so)
        assert torch.equal(rg['labels_3d'], lo_)
        assert _rel(rg['scores_3d'], so) < 1e-4
        got = torch.cat((rg['boxes_3d'].gravity_center, rg['boxes_3d'].tensor[:, 3:6]), 1)
        assert _rel(got, bo[:, :6]) < 1e-4


def test_iou_losses_vs_reference_goldens():
    from fcaf3d_amd.losses import IoU3DLoss, axis_aligned_iou_3d, rotated_iou_3d
    dev = _dev()
    d = np.load(os.path.join(G, 'iou3d.npz'))
    for key, fn in (('al', axis_aligned_iou_3d), ('ro', rotated_iou_3d)):
        pred = torch.from_numpy(d[f'{key}_pred']).to(dev).requires_grad_(True)
        tgt = torch.from_numpy(d[f'{key}_target']).to(dev)
        w = torch.from_numpy(d[f'{key}_w']).to(dev)
        iou = fn(pred, tgt)
        ((1 - iou) * w).sum().backward()
        assert np.allclose(iou.detach().cpu().numpy(), d[f'{key}_iou'], atol=1e-5), key
        # gradients: within 1e-4 of the gradient's scale (the north_star bar) — the rotated IoU against the reference's own code run
        # in float64 on the same boxes (tests/golden/iou3d_f64.npz, make_golden.py iou64; r4 allowed atol 2e-4 + rtol 1e-3)
        ref_g = np.load(os.path.join(G, 'iou3d_f64.npz'))['ro_grad64'] if key == 'ro' else d[f'{key}_grad'].astype(np.float64)
        err = float(np.abs(pred.grad.cpu().numpy().astype(np.float64) - ref_g).max())
        print(f'{key}: IoU gradient, max difference {err:.2e} at gradient scale {float(np.abs(ref_g).max()):.3f}')
        assert err <= 1e-4 * float(np.abs(ref_g).max()), (key, err)
    # module API incl. the zero-weight early-out (iou3d_loss.py:53-54)
    loss = IoU3DLoss(with_yaw=True)
    p = torch.from_numpy(d['ro_pred']).to(dev).requires_grad_(True)
    z = loss(p, torch.from_numpy(d['ro_target']).to(dev), weight=torch.zeros(len(p), device=dev), avg_factor=3.0)
    z.backward()
    assert float(z) == 0.0 and float(p.grad.abs().sum()) == 0.0


def test_focal_loss_vs_oracle():
    from fcaf3d_amd.losses import FocalLoss
    dev = _dev()
    g = torch.Generator().manual_seed(0)
    x = torch.randn(5000, 18, generator=g) * 3
    lab = torch.randint(-1, 18, (5000,), generator=g)
    xr = x.clone().requires_grad_(True)
    ref = lo.sigmoid_focal_loss_sum(xr, lab) / 37.0
    ref.backward()
    xg = x.to(dev).requires_grad_(True)
    out = FocalLoss(use_sigmoid=True, gamma=2.0, alpha=0.25)(xg, lab.to(dev), avg_factor=37.0)
    out.backward()
    assert _rel(out, ref) < 1e-5 and _rel(xg.grad, xr.grad) < 1e-5


def test_bev_iou_and_nms_vs_reference():
    from fcaf3d_amd import _lib as L
    from fcaf3d_amd.nms import nms_bev, nms_bev_multiclass
    dev = _dev()
    d = np.load(os.path.join(G, 'bev_iou.npz'))
    for n in ('1', '63', '64', '65', '300', '_hand'):
        b = torch.from_numpy(d[f'boxes{n}']).to(dev)
        out = torch.empty((len(b), len(b)), device=dev)
        L.call('fc_boxes_iou_bev', L.ptr(b), len(b), L.ptr(b), len(b), 1, L.ptr(out), L.stream())
        assert np.allclose(out.cpu().numpy(), d[f'iou{n}'], atol=2e-5), n
    rng = np.random.default_rng(0)
    for n in (1, 63, 64, 65, 1000, 4000):
        for rotated in (True, False):
            c = rng.uniform(0, 12, (n, 3)); s = rng.uniform(0.3, 2.0, (n, 3))
            yaw = rng.uniform(-3.14, 3.14, (n, 1)) if rotated else np.zeros((n, 1))
            boxes = np.concatenate([c, s, yaw], 1).astype(np.float32)
            scores = rng.permutation(n).astype(np.float32) / n
            ref = bev.nms(boxes, scores, 0.5, rotated)
            got = nms_bev(torch.from_numpy(boxes).to(dev), torch.from_numpy(scores).to(dev), 0.5, rotated)
            assert np.array_equal(got.cpu().numpy(), ref), (n, rotated, len(ref))
    # all classes in one launch == the per-class loop of fcaf3d_neck_with_head.py:336-353
    n, C = 700, 5
    boxes = np.concatenate([rng.uniform(0, 6, (n, 3)), rng.uniform(0.3, 2.0, (n, 3)), rng.uniform(-3, 3, (n, 1))], 1).astype(np.float32)
    sc = rng.random((n, C)).astype(np.float32); sc[rng.random((n, C)) < 0.5] = 0.0
    idx, cls = nms_bev_multiclass(torch.from_numpy(boxes).to(dev), torch.from_numpy(sc).to(dev), 0.01, 0.5, True)
    exp_i, exp_c = [], []
    for c_ in range(C):
        ids = np.nonzero(sc[:, c_] > 0.01)[0]
        k = bev.nms(boxes[ids], sc[ids, c_], 0.5, True)
        exp_i.append(ids[k]); exp_c.append(np.full(len(k), c_))
    assert np.array_equal(idx.cpu().numpy(), np.concatenate(exp_i)) and np.array_equal(cls.cpu().numpy(), np.concatenate(exp_c))


def test_pcdet_named_entry_points():
    """the names the reference imports (mmdet3d/ops/pcdet_nms/__init__.py): (keep, None) tuples, pre_maxsize, and the
    IoU-matrix helpers of pcdet_nms_utils.py:28-78"""
    from fcaf3d_amd.nms import boxes_iou3d_gpu, boxes_iou_bev, pcdet_nms_gpu, pcdet_nms_normal_gpu
    dev = _dev()
    rng = np.random.default_rng(4)
    n = 500
    boxes = np.concatenate([rng.uniform(0, 8, (n, 3)), rng.uniform(0.3, 2.0, (n, 3)), rng.uniform(-3, 3, (n, 1))], 1).astype(np.float32)
    scores = rng.permutation(n).astype(np.float32) / n
    bt, st = torch.from_numpy(boxes).to(dev), torch.from_numpy(scores).to(dev)
    keep, extra = pcdet_nms_gpu(bt, st, 0.4)
    assert extra is None and keep.dtype == torch.long
    assert np.array_equal(keep.cpu().numpy(), bev.nms(boxes, scores, 0.4, True))
    keep, _ = pcdet_nms_gpu(bt, st, 0.4, pre_maxsize=100)
    top = np.argsort(-scores, kind='stable')[:100]
    assert np.array_equal(keep.cpu().numpy(), top[bev.nms(boxes[top], scores[top], 0.4, True)])
    keep, _ = pcdet_nms_normal_gpu(bt, st, 0.4)
    assert np.array_equal(keep.cpu().numpy(), bev.nms(boxes, scores, 0.4, False))
    a, b = boxes[:60], boxes[60:130]
    iou = bev.iou_matrix(a, b, True)
    assert np.allclose(boxes_iou_bev(bt[:60], bt[60:130]).cpu().numpy(), iou, atol=2e-5)
    sa, sb = (a[:, 3] * a[:, 4])[:, None], (b[:, 3] * b[:, 4])[None]
    ov = iou * (sa + sb) / (1 + iou)
    oh = np.clip(np.minimum(a[:, 2:3] + a[:, 5:6] / 2, (b[:, 2] + b[:, 5] / 2)[None])
                 - np.maximum(a[:, 2:3] - a[:, 5:6] / 2, (b[:, 2] - b[:, 5] / 2)[None]), 0, None)
    o3 = ov * oh
    ref3 = o3 / np.clip((a[:, 3] * a[:, 4] * a[:, 5])[:, None] + (b[:, 3] * b[:, 4] * b[:, 5])[None] - o3, 1e-6, None)
    assert np.allclose(boxes_iou3d_gpu(bt[:60], bt[60:130]).cpu().numpy(), ref3, atol=5e-5)


def test_pruning_path_bites():
    """pts_threshold smaller than the level sizes -> interpolation + top-k + MinkowskiPruning run."""
    dev = _dev()
    model, m = _build('fcaf3d_scannet-3d-18class', 0.02, 3, pts_threshold=1500)
    with torch.no_grad():                                     # spread the scores so that top-k has no near-ties
        model.neck_with_head.cls_conv.kernel.normal_(0, 0.5)
    P = _oracle_params(model)
    model = model.to(dev).train()
    pts, gts, labs = _scenes([31, 32], n_points=20000)
    out_o = MO.extract_feat(P, m, pts)
    out_g = [list(x) for x in model.extract_feat([torch.from_numpy(p).to(dev) for p in pts], None)]
    assert len(out_o[3][0][0]) == 1500                       # finest level was pruned to the threshold
    for l in range(3):
        for b in range(2):
            # top-k ties aside, the kept coordinate SETS agree; compare in (x,y,z)-canonical order
            po, pg = out_o[3][l][b].numpy(), out_g[3][l][b].cpu().numpy()
            assert len(po) == len(pg)
            so, sg = np.lexsort(po.T[::-1]), np.lexsort(pg.T[::-1])
            assert np.array_equal(po[so], pg[sg])
            assert _rel(out_g[2][l][b][torch.from_numpy(sg).to(dev)], out_o[2][l][b][torch.from_numpy(so)]) < 1e-4


def test_full_size_config5_s3dis_pruning_live():
    """BASELINE config 5 (S3DIS-shape: 500 000 points, 12 x 10 m room, 5 classes) at FULL size with the config's real
    pts_threshold: the finest neck level exceeds it, so interpolation + per-scene top-k + MinkowskiPruning run for real.
    extract_feat against the oracle: kept coordinate sets exact (canonical order: top-k ties are order-dependent),
    head outputs 1e-4."""
    from fcaf3d_amd.synthetic import WORKLOADS
    dev = _dev()
    model, m = _build('fcaf3d_s3dis-3d-5class', 0.02, 4)
    with torch.no_grad():                                     # spread the scores so that top-k has no near-ties
        model.neck_with_head.cls_conv.kernel.normal_(0, 0.5)
    P = _oracle_params(model)
    model = model.to(dev).train()
    pts, gts, labs = _scenes([51], **WORKLOADS['s3dis-500k']['scene'])
    assert pts[0].shape[0] == 500000
    with torch.no_grad():
        out_o = MO.extract_feat({k: v.detach() for k, v in P.items()}, m, pts)
        out_g = [list(x) for x in model.extract_feat([torch.from_numpy(p).to(dev) for p in pts], None)]
    thr = m.neck_with_head['pts_threshold']
    assert len(out_o[3][0][0]) == thr, 'the finest level must have been pruned to pts_threshold'
    for l in range(4):
        po, pg = out_o[3][l][0].numpy(), out_g[3][l][0].cpu().numpy()
        assert len(po) == len(pg), (l, len(po), len(pg))
        so, sg = np.lexsort(po.T[::-1]), np.lexsort(pg.T[::-1])
        assert np.array_equal(po[so], pg[sg]), f'level {l}: kept coordinate sets differ'
        for kind in range(3):
            assert _rel(out_g[kind][l][0][torch.from_numpy(sg).to(dev)], out_o[kind][l][0][torch.from_numpy(so)]) < 1e-4, (kind, l)


def test_full_size_config5_backward_with_equal_decisions():
    """BASELINE config 5 at FULL size, forward_train + BACKWARD: one 500 000-point S3DIS-shaped scene, pruning live at the real
    pts_threshold, so the gradients flow through MinkowskiPruning's gather / scatter (fcaf3d_neck_with_head.py:110-126), the
    interpolation-selected rows and the 100 000-row level-0 maps.  The fp32 CPU oracle runs HERE (~2 minutes) with the HIP
    forward's discrete decisions — ReLU signs, max-pool arg-max rows and the top-k kept set of `_prune` — so every parameter
    gradient is held to 1e-4 of its scale (1e-3 in the deepest stage).  r3 compared against a stored digest of an oracle run
    with its OWN decisions and needed a 6e-2 envelope for the flips; that digest is gone."""
    from fcaf3d_amd.synthetic import WORKLOADS
    dev = _dev()
    torch.manual_seed(0)
    cfg = fa.get_config('fcaf3d_s3dis-3d-5class', voxel_size=0.02)
    m = cfg.model
    model = fa.build_detector(m, train_cfg=m.get('train_cfg'), test_cfg=m.get('test_cfg'))
    with torch.no_grad():                                     # spread the scores so that top-k has no near-ties
        model.neck_with_head.cls_conv.kernel.normal_(0, 0.5)
    P = _oracle_params(model)
    model = model.to(dev).train()
    pts, gts, labs = _scenes([51], **WORKLOADS['s3dis-500k']['scene'])
    with _RecordDecisions(model) as rec:
        losses_g = model(return_loss=True, **_to_gpu_batch(pts, gts, labs, dev))
    sum(losses_g.values()).backward()
    torch.cuda.synchronize()
    tape = rec.tape()
    assert len(tape.prune) >= 1, 'pruning must be live at the config\'s pts_threshold'
    MO.TAPE = tape
    try:
        losses_o = MO.forward_train(P, m, pts, gts, labs)
    finally:
        MO.TAPE = None
    assert tape.i == len(tape.relu) and tape.ip == len(tape.prune)
    for k in ('loss_centerness', 'loss_bbox', 'loss_cls'):
        assert _rel(losses_g[k], losses_o[k]) < 1e-4, (k, float(losses_g[k]), float(losses_o[k]))
    sum(losses_o.values()).backward()
    flips = [f for f in tape.flips if f[1]]
    print(f'config 5: {tape.total_flips()} of {sum(f[2] for f in tape.flips)} decisions differ: '
          + ', '.join(f'{s}: {n} (|pre| <= {mx:.1e})' for s, n, _, mx in flips))
    assert all(mx < 1e-4 for _, _, _, mx in flips), flips
    errs = {k: _rel(p.grad, P[k].grad) for k, p in model.named_parameters()}
    shallow = {k: v for k, v in errs.items() if not k.startswith(DEEP)}
    deep = {k: v for k, v in errs.items() if k.startswith(DEEP)}
    ws, wd = max(shallow, key=shallow.get), max(deep, key=deep.get)
    print(f'   gradients vs the fp32 oracle with equal decisions: outside the deepest stage {shallow[ws]:.2e} ({ws}); deepest stage '
          f'{deep[wd]:.2e} ({wd}); median {np.median(list(errs.values())):.2e}')
    assert shallow[ws] < 1e-4, (ws, shallow[ws])
    assert deep[wd] < 1e-3, (wd, deep[wd])


def test_out_of_range_coordinates_raise():
    """a stray far-away / non-finite point must not silently alias another voxel through the 16-bit fields of
    the packed hash key: the voxelisation reports it."""
    dev = _dev()
    model, m = _build('fcaf3d_scannet-3d-18class', 0.02, 1)
    model = model.to(dev).train()
    pts, _, _ = _scenes([3], n_points=5000)
    for bad in (5.0e4, float('inf')):                          # 50 km at 2 cm voxels = 2.5 M voxels; and inf
        p = pts[0].copy()
        p[17, 0] = bad
        with pytest.raises(ValueError, match='voxel coordinate outside'):
            model.extract_feat([torch.from_numpy(p).to(dev)], None)
    model.extract_feat([torch.from_numpy(pts[0]).to(dev)], None)        # the clean cloud still goes through


def test_iou_loss_zero_weight_rows_cannot_poison_gradients():
    """a zero-weight (background) row whose own IoU derivative is not finite must contribute an exact zero."""
    dev = _dev()
    for with_yaw, d in ((False, 6), (True, 7)):
        loss_fn = fa.build_loss(dict(type='IoU3DLoss', with_yaw=with_yaw, loss_weight=1.0))
        pred = torch.tensor([[0.0, 0, 0, 1, 1, 1, 0.1][:d], [0.0, 0, 0, 3e19, 3e19, 3e19, 0.0][:d]], device=dev, requires_grad=True)
        tgt = torch.tensor([[0.1, 0, 0, 1, 1, 1, 0.0][:d], [0.0, 0, 0, 1, 1, 1, 0.0][:d]], device=dev)
        w = torch.tensor([1.0, 0.0], device=dev)
        loss = loss_fn(pred, tgt, weight=w, avg_factor=1.0)
        loss.backward()
        assert torch.isfinite(loss) and torch.isfinite(pred.grad).all(), (with_yaw, loss, pred.grad)
        assert float(pred.grad[1].abs().max()) == 0.0 and float(pred.grad[0].abs().max()) > 0.0


def test_head_with_more_than_55_classes():
    """1 + n_reg + n_classes > 64 fused head columns (e.g. ScanNet200) takes the unfused split — same maths."""
    dev = _dev()
    model, m = _build('fcaf3d_scannet-3d-18class', 0.02, 2, n_classes=70)
    P = _oracle_params(model)
    model = model.to(dev).train()
    pts, gts, labs = _scenes([61], n_points=12000, n_classes=70)
    out_o = MO.extract_feat(P, m, pts)
    out_g = [list(x) for x in model.extract_feat([torch.from_numpy(p).to(dev) for p in pts], None)]
    for kind in range(3):
        for l in range(2):
            assert _rel(out_g[kind][l][0], out_o[kind][l][0]) < 1e-4, (kind, l)
    assert out_g[2][0][0].shape[1] == 70
    losses_g = model(return_loss=True, **_to_gpu_batch(pts, gts, labs, dev))
    losses_o = MO.forward_train(P, m, pts, gts, labs)
    for k in losses_o:
        assert _rel(losses_g[k], losses_o[k]) < 1e-4, k
    sum(losses_g.values()).backward()
    assert torch.isfinite(model.neck_with_head.cls_conv.kernel.grad).all()


def test_mmcv_layout_checkpoint_reproduces_detections_on_gpu(tmp_path):
    """weights travel through a checkpoint in mmcv's on-disk layout ({'meta', 'state_dict'} with DDP's
    `module.` prefix, as the released .pth files and tools/test.py:172 use it) and give the same detections on the GPU as
    the model that wrote them, and as the CPU oracle running on the checkpoint's own tensors."""
    from collections import OrderedDict
    dev = _dev()
    src, m = _build('fcaf3d_scannet-3d-18class', 0.02, 2, seed=5)
    with torch.no_grad():
        src.neck_with_head.cls_conv.bias.fill_(0.0)
        src.neck_with_head.cls_conv.kernel.normal_(0, 0.3)
        for mod in src.modules():                                   # non-trivial running statistics, as after training
            if isinstance(mod, torch.nn.BatchNorm1d):
                mod.running_mean.normal_(0, 0.05)
                mod.running_var.uniform_(0.5, 1.5)
    path = str(tmp_path / 'epoch_12.pth')
    ckpt = dict(meta=dict(epoch=12, iter=1812, mmdet3d_version='0.8.0'),
                state_dict=OrderedDict(('module.' + k, v.clone()) for k, v in src.state_dict().items()), optimizer=dict())
    torch.save(ckpt, path)
    dst, _ = _build('fcaf3d_scannet-3d-18class', 0.02, 2, seed=6)      # different random weights
    loaded = fa.load_checkpoint(dst, path, map_location='cpu', strict=True)
    assert loaded['meta']['epoch'] == 12
    pts, _, _ = _scenes([81, 82], n_points=20000)
    gp = [torch.from_numpy(p).to(dev) for p in pts]
    metas = [dict(box_type_3d=fa.DepthInstance3DBoxes)] * 2
    with torch.no_grad():
        r_src = src.to(dev).eval()(return_loss=False, points=gp, img_metas=metas)
        r_dst = dst.to(dev).eval()(return_loss=False, points=gp, img_metas=metas)
    P = {k[len('module.'):]: v for k, v in torch.load(path, weights_only=False)['state_dict'].items()}
    MO.TRAINING = False
    try:
        res_o = MO.simple_test(P, m, pts)
    finally:
        MO.TRAINING = True
    for a, b, (bo, so, lo_) in zip(r_src, r_dst, res_o):
        assert len(a['scores_3d']) > 5
        assert torch.equal(a['scores_3d'], b['scores_3d']) and torch.equal(a['labels_3d'], b['labels_3d'])
        assert torch.equal(a['boxes_3d'].tensor, b['boxes_3d'].tensor)
        assert len(so) == len(b['scores_3d']) and torch.equal(b['labels_3d'], lo_) and _rel(b['scores_3d'], so) < 1e-4
    # save_checkpoint writes the same layout back (weights on the CPU, prefix stripped)
    out = str(tmp_path / 'resaved.pth')
    fa.save_checkpoint(dst, out, meta=dict(epoch=12))
    again = torch.load(out, weights_only=False)
    assert set(again) >= {'meta', 'state_dict'} and all(not v.is_cuda for v in again['state_dict'].values())
    assert all(torch.equal(again['state_dict'][k], v.cpu()) for k, v in dst.state_dict().items())


def test_three_step_training_trajectory_vs_oracle():
    """three optimisation steps of the reference's recipe (AdamW 1e-3 / 1e-4, grad-clip 10 —
    fcaf3d_amd/runner.py TrainStep) on the HIP path against the same three steps on the CPU oracle with torch's own
    (unfused) AdamW: the loss trajectories agree."""
    from fcaf3d_amd.runner import TrainStep
    dev = _dev()
    model, m = _build('fcaf3d_scannet-3d-18class', 0.02, 2, seed=3)
    cfg = fa.get_config('fcaf3d_scannet-3d-18class', voxel_size=0.02)
    P = _oracle_params(model)
    model = model.to(dev).train()
    tr = TrainStep.from_config(model, cfg)
    leaves = [v for v in P.values() if v.requires_grad]
    opt = torch.optim.AdamW(leaves, lr=cfg.optimizer.lr, weight_decay=cfg.optimizer.weight_decay)
    traj_g, traj_o = [], []
    for step in range(3):
        pts, gts, labs = _scenes([90 + step], n_points=12000)
        loss, _ = tr(_to_gpu_batch(pts, gts, labs, dev))
        traj_g.append(float(loss))
        opt.zero_grad()
        lo_ = sum(MO.forward_train(P, m, pts, gts, labs).values())
        lo_.backward()
        torch.nn.utils.clip_grad_norm_(leaves, cfg.optimizer_config.grad_clip.max_norm)
        opt.step()
        traj_o.append(float(lo_))
    print('loss trajectory HIP', traj_g, 'oracle', traj_o)
    assert abs(traj_g[0] - traj_o[0]) <= 1e-4 * abs(traj_o[0])
    for a, b in zip(traj_g[1:], traj_o[1:]):
        assert abs(a - b) <= 2e-2 * abs(b), (traj_g, traj_o)


def test_fused_head_loss_equals_the_three_loss_modules():
    """csrc/loss.hip k_fcaf3d_loss_* (focal + centerness BCE + decode + axis-aligned IoU with per-row scene weights, two
    launches forward, one backward) == FocalLoss + CrossEntropyLoss(use_sigmoid) + IoU3DLoss on `_bbox_pred_to_bbox`, the
    modules the reference's config names (fcaf3d_neck_with_head.py:24-34) — which are themselves checked against the oracle
    by test_forward_train_parity: losses to 1e-6, every parameter gradient to 1e-5 of its scale; unequal loss weights so that
    each incoming gradient is exercised; a scene without boxes included."""
    dev = _dev()
    runs = []
    for fused in (True, False):
        model, m = _build('fcaf3d_scannet-3d-18class', 0.02, 2, seed=6,
                          loss_cls=dict(type='FocalLoss', use_sigmoid=True, gamma=2.0, alpha=0.25, loss_weight=0.7),
                          loss_bbox=dict(type='IoU3DLoss', with_yaw=False, loss_weight=1.3),
                          loss_centerness=dict(type='CrossEntropyLoss', use_sigmoid=True, loss_weight=0.9))
        model = model.to(dev).train()
        model.neck_with_head.fused_loss = fused
        pts, gts, labs = _scenes([61, 62, 63], n_points=12000)
        gts[1] = gts[1][:0]; labs[1] = labs[1][:0]                       # a scene without ground truth
        losses = model(return_loss=True, **_to_gpu_batch(pts, gts, labs, dev))
        (losses['loss_cls'] * 1.0 + losses['loss_bbox'] * 2.0 + losses['loss_centerness'] * 0.5).backward()
        runs.append(({k: float(v) for k, v in losses.items()}, [p.grad.detach().clone() for p in model.parameters()],
                     [k for k, _ in model.named_parameters()]))
    (l1, g1, names), (l2, g2, _) = runs
    for k in l1:
        assert abs(l1[k] - l2[k]) <= 1e-6 * max(1.0, abs(l2[k])), (k, l1[k], l2[k])
    rels = sorted(((_rel(a, b), k) for a, b, k in zip(g1, g2, names)), reverse=True)
    assert rels[0][0] < 1e-5, rels[:6]


def test_flat_adamw_equals_torch_adamw_with_clip():
    """csrc/optim.hip over flat buffers (fc_grad_norm + fc_adamw_step) == torch.nn.utils.clip_grad_norm_ +
    torch.optim.AdamW (the calls mmcv's OptimizerHook makes for configs/fcaf3d/fcaf3d.py:30-31) on identical
    gradients, over 5 steps with the clip biting on some and not on others; optimizer state round-trips in torch's
    state_dict layout."""
    from fcaf3d_amd.flat import FlatParams
    from fcaf3d_amd.runner import FlatAdamW
    dev = _dev()
    g = torch.Generator().manual_seed(11)
    shapes = [(27, 64, 64), (128,), (1, 18), (), (8, 256, 128), (3,), (27, 3, 64)]
    ref = [torch.nn.Parameter(torch.randn(s, generator=g) * 0.1) for s in shapes]
    mine = [torch.nn.Parameter(p.detach().clone().to(dev)) for p in ref]
    flat = FlatParams(mine)
    assert all(p.data_ptr() == flat.data.data_ptr() + 4 * o for p, o in zip(mine, flat.offsets))
    opt_r = torch.optim.AdamW(ref, lr=1e-3, weight_decay=1e-4)
    opt_m = FlatAdamW(flat, lr=1e-3, weight_decay=1e-4)
    for step in range(5):
        scale = (40.0, 0.01, 3.0, 25.0, 0.5)[step]           # global norm above and below max_norm = 10
        grads = [torch.randn(s, generator=g) * scale for s in shapes]
        for p, q, gr in zip(ref, mine, grads):
            p.grad = gr.clone()
            q.grad = gr.to(dev)                              # produced outside the flat buffer: gather() copies it in
        if step == 2:
            mine[3].grad = None                              # a parameter without a gradient this step: zeros
            ref[3].grad = torch.zeros(())
        norm_r = torch.nn.utils.clip_grad_norm_(ref, 10.0)
        opt_r.step()
        norm_m = opt_m.step(max_norm=10.0)
        assert abs(float(norm_m) - float(norm_r)) <= 1e-5 * float(norm_r), (float(norm_m), float(norm_r))
        for p, q in zip(ref, mine):
            assert _rel(q, p) < 2e-6, (step, p.shape, _rel(q, p))
    sd = opt_m.state_dict()
    assert set(sd['state'][0]) == {'step', 'exp_avg', 'exp_avg_sq'} and sd['param_groups'][0]['lr'] == 1e-3
    for i, st in opt_r.state_dict()['state'].items():
        assert _rel(sd['state'][i]['exp_avg'], st['exp_avg']) < 2e-6 and _rel(sd['state'][i]['exp_avg_sq'], st['exp_avg_sq']) < 2e-6
    opt2 = FlatAdamW(FlatParams([torch.nn.Parameter(p.detach().clone()) for p in mine]), lr=1e-3, weight_decay=1e-4)
    opt2.load_state_dict(sd)
    assert opt2.steps == 5 and torch.equal(opt2.exp_avg, opt_m.exp_avg) and torch.equal(opt2.exp_avg_sq, opt_m.exp_avg_sq)


def test_train_step_flat_buffers_equal_per_tensor_path():
    """TrainStep with parameters / gradients in flat buffers + csrc/optim.hip (the default on the GPU) against the same
    step with per-tensor gradients + torch's clip_grad_norm_ / fused AdamW (flat=False): identical losses step for step,
    the same parameters after the first step (2e-5 of their scale) and, after 3 steps, parameters within 10 % of ONE
    AdamW step (lr) for all but 1e-5 of the elements — the only room left is Adam's own conditioning: the gradient of a BatchNorm bias is a sum that
    cancels to ~1e-3 of its terms, a 1e-7 difference in the clip coefficient of step 1 moves it by percents at step 2,
    and Adam normalises every element's update to ~lr whatever the gradient's size (measured: 4.5e-5 on
    backbone.layer2.0.norm1.bn.bias).  With the weight gradients on their own stream the conv
    kernels' gradients ARE their slices of the flat buffer (no copy)."""
    import fcaf3d_amd.functional as Fn
    from fcaf3d_amd.runner import TrainStep
    dev = _dev()
    cfg = fa.get_config('fcaf3d_scannet-3d-18class', voxel_size=0.02)
    runs = []
    for flat in (True, False):
        model, m = _build('fcaf3d_scannet-3d-18class', 0.02, 2, seed=4)
        model = model.to(dev).train()
        tr = TrainStep.from_config(model, cfg, flat=flat)
        assert (tr.flat is not None) == flat
        Fn.WGRAD_ASYNC = flat
        try:
            losses = []
            for step in range(3):
                pts, gts, labs = _scenes([70 + step, 80 + step], n_points=10000)
                loss, _ = tr(_to_gpu_batch(pts, gts, labs, dev))
                losses.append(float(loss))
                if step == 0:
                    first = [p.detach().clone() for p in model.parameters()]
                if flat:
                    k = model.backbone.layer1[0].conv1.kernel
                    assert k.grad.data_ptr() == tr.flat.grad_view(k).data_ptr(), 'conv weight gradient was copied'
        finally:
            Fn.WGRAD_ASYNC = False
        runs.append((losses, [p.detach().clone() for p in model.parameters()], float(tr.last_grad_norm), first))
        names = [k for k, _ in model.named_parameters()]
    (l1, p1, n1, f1), (l2, p2, n2, f2) = runs
    assert l1[0] == l2[0]
    assert all(abs(a - b) <= 1e-5 * abs(b) for a, b in zip(l1, l2)), (l1, l2)
    assert abs(n1 - n2) <= 1e-5 * n2
    rels = sorted(((_rel(a, b), k) for a, b, k in zip(f1, f2, names)), reverse=True)
    assert rels[0][0] < 2e-5, rels[:8]          # fp32 rounding of lr / (1 - beta1) between the two AdamW kernels: 7e-6 measured
    lr = cfg.optimizer.lr
    # ... which also means that an element whose gradient is within the 1e-5 noise of zero can take its +-lr step in the
    # other direction (seen with the split-bf16 convolutions: ONE element of backbone.layer2.1.conv1.kernel at 0.75 lr): bound
    # the elements that are off by more than 0.1 lr to 1e-5 of all elements, and every element by what two differing Adam
    # steps can produce
    n_off = sum(int(((a - b).abs() > 0.1 * lr).sum()) for a, b in zip(p1, p2))
    n_all = sum(a.numel() for a in p1)
    absd = sorted(((float((a - b).abs().max()), k) for a, b, k in zip(p1, p2, names)), reverse=True)
    # (r4: the SAME counts come out of the per-operator path and of the native executor — their parameters are bit-identical
    # after three steps, tests/test_gpu_exec.py — so this is the two optimizers' business: a 1.2e-7 parameter difference after
    # step 1 flips a ReLU decision or two in step 2, one tensor's gradient then differs by 1.8e-2, and Adam turns near-zero
    # gradient elements by +-lr: 61 elements after step 2, 1 468 of 4.8 M after step 3 with the round-to-nearest split)
    assert n_off <= 1e-3 * n_all, (n_off, n_all, absd[:8])
    assert absd[0][0] < 3 * lr, absd[:8]


def test_multiclass_nms_route_equals_per_class_loop():
    """`_nms` (all classes in one sort + one pair of launches) returns exactly what the reference's per-class loop
    (`_nms_per_class`, fcaf3d_neck_with_head.py:332-374) returns: boxes, scores, labels, order."""
    dev = _dev()
    rng = np.random.default_rng(3)
    for name, rotated in (('fcaf3d_scannet-3d-18class', False), ('fcaf3d_sunrgbd-3d-10class', True)):
        model, m = _build(name, 0.02, 2)
        head = model.neck_with_head
        n, C = 3000, head.n_classes
        centre = rng.uniform(0, 4, (n, 3))
        size = rng.uniform(0.2, 1.5, (n, 3))
        cols = [centre, size] + ([rng.uniform(-3.14, 3.14, (n, 1))] if rotated else [])
        boxes = torch.from_numpy(np.concatenate(cols, 1).astype(np.float32)).to(dev)
        scores = torch.from_numpy(rng.uniform(0, 1, (n, C)).astype(np.float32) ** 4).to(dev)
        meta = dict(box_type_3d=fa.DepthInstance3DBoxes)
        b1, s1, l1 = head._nms(boxes, scores, meta)
        b2, s2, l2 = head._nms_per_class(boxes, scores, meta)
        assert len(s1) == len(s2) and len(s1) > 50
        assert torch.equal(l1, l2) and torch.equal(s1, s2) and torch.equal(b1.tensor, b2.tensor)
        # nothing above the threshold: empty result of the right types
        b0, s0, l0 = head._nms(boxes, torch.zeros_like(scores), meta)
        assert len(s0) == 0 and l0.dtype == torch.long and b0.tensor.shape[0] == 0


def test_batched_get_bboxes_equals_per_scene_loop():
    """get_bboxes over the whole batch (one segmented top-k sort, one NMS over every (scene, class) segment, one read-back)
    returns exactly what the reference's per-scene / per-level loop returns (fcaf3d_neck_with_head.py:205-253)."""
    dev = _dev()
    for name, seeds, npts in (('fcaf3d_scannet-3d-18class', [31, 32, 33], 30000), ('fcaf3d_sunrgbd-3d-10class', [34, 35], 20000)):
        model, m = _build(name, 0.02, 3)
        with torch.no_grad():
            model.neck_with_head.cls_conv.bias.fill_(0.0)
            model.neck_with_head.cls_conv.kernel.normal_(0, 0.3)
        model = model.to(dev).train()
        pts, _, _ = _scenes(seeds, n_points=npts)
        kw = dict(points=[torch.from_numpy(p).to(dev) for p in pts], img_metas=[dict(box_type_3d=fa.DepthInstance3DBoxes)] * len(pts))
        head = model.neck_with_head
        assert head.batched_decode
        with torch.no_grad():
            res_b = model(return_loss=False, **kw)
            head.batched_decode = False
            try:
                res_s = model(return_loss=False, **kw)
            finally:
                head.batched_decode = True
        for rb, rs in zip(res_b, res_s):
            assert len(rs['scores_3d']) > 10 and len(rb['scores_3d']) == len(rs['scores_3d'])
            assert torch.equal(rb['labels_3d'], rs['labels_3d'])
            assert torch.equal(rb['scores_3d'], rs['scores_3d'])
            assert torch.equal(rb['boxes_3d'].tensor, rs['boxes_3d'].tensor)
            assert rb['boxes_3d'].box_dim == rs['boxes_3d'].box_dim and rb['boxes_3d'].with_yaw == rs['boxes_3d'].with_yaw


def test_eval_mode_inference_and_running_stats():
    """model.eval(): BatchNorm runs on the running statistics that training steps accumulated
    (nn.BatchNorm1d semantics inside ME.MinkowskiBatchNorm): the buffers, and the detections, match the oracle."""
    dev = _dev()
    model, m = _build('fcaf3d_scannet-3d-18class', 0.02, 2)
    with torch.no_grad():
        model.neck_with_head.cls_conv.bias.fill_(0.0)
        model.neck_with_head.cls_conv.kernel.normal_(0, 0.3)
    model = model.to(dev).train()
    pts, gts, labs = _scenes([71, 72], n_points=20000)
    # two training forwards update running_mean / running_var / num_batches_tracked
    ref_bn = torch.nn.BatchNorm1d(64)
    for _ in range(2):
        model(return_loss=True, **_to_gpu_batch(pts, gts, labs, dev))
    bn = model.backbone.layer1[0].norm1.bn
    assert int(bn.num_batches_tracked) == 2
    P = _oracle_params(model)
    # replay the first BN of layer1 on the CPU to check the momentum / unbiased-variance update
    x = MO.SP(*MO.mo.sparse_tensor(*MO.mo.batch_sparse_collate([p[:, :3] / np.float32(0.02) for p in pts],
                                                                [p[:, 3:] / np.float32(255.) for p in pts])), 1)
    x = MO.SP(x.C, torch.from_numpy(x.F), 1)
    P0 = {k: v.detach() for k, v in P.items()}
    h = MO.conv(x, P0['backbone.conv1.0.kernel'], 3, 2)
    f = MO.mo.instance_norm(h.F, h.C[:, 0], P0['backbone.conv1.1.weight'], P0['backbone.conv1.1.bias'])
    h = MO.SP(h.C, torch.relu(f), h.stride, h.cache)
    oc, ocache = MO._strided(h, 2)
    h = MO.SP(oc, MO.mo.max_pool(h.F, MO._kmap(h, oc, 2, 'down')), h.stride * 2, ocache)
    pre = MO.conv(h, P0['backbone.layer1.0.conv1.kernel'], 3, 2).F
    ref_bn.train()
    ref_bn(pre); ref_bn(pre)
    assert _rel(bn.running_mean, ref_bn.running_mean) < 1e-4 and _rel(bn.running_var, ref_bn.running_var) < 1e-4
    # eval-mode detections vs the oracle on running statistics
    model.eval()
    MO.TRAINING = False
    try:
        res_o = MO.simple_test(P, m, pts)
    finally:
        MO.TRAINING = True
    with torch.no_grad():
        res_g = model(return_loss=False, points=[torch.from_numpy(p).to(dev) for p in pts],
                      img_metas=[dict(box_type_3d=fa.DepthInstance3DBoxes)] * 2)
    for (bo, so, lo_), rg in zip(res_o, res_g):
        assert len(so) > 5 and len(rg['scores_3d']) == len(so)
        assert torch.equal(rg['labels_3d'], lo_) and _rel(rg['scores_3d'], so) < 1e-4


def test_full_size_properties_config2():
    """BASELINE config 2 at full size (100k points/scene, 4 levels, 2 cm) — size-independent properties:
    bitwise determinism run to run, invariance of the losses to a permutation of the input points (row order
    changes, sets do not), voxel-count sanity and kernel-map symmetry (nbr_t of a same-set map is its own flip)."""
    dev = _dev()
    model, m = _build('fcaf3d_scannet-3d-18class', 0.02, 4)
    model = model.to(dev).train()
    pts, gts, labs = _scenes([81, 82], n_points=100000)
    batch = _to_gpu_batch(pts, gts, labs, dev)
    l1 = model(return_loss=True, **batch)
    l2 = model(return_loss=True, **batch)
    assert all(float(l1[k]) == float(l2[k]) for k in l1), 'two identical steps must agree bit for bit'
    # idempotence of de-duplication: feeding only the first point of every voxel (same order) changes nothing
    rng = np.random.default_rng(0)
    firsts = []
    for p in pts:
        q = np.floor(p[:, :3] / np.float32(0.02)).astype(np.int64)
        _, idx = np.unique(q, axis=0, return_index=True)
        firsts.append(p[np.sort(idx)])
    l_first = model(return_loss=True, **_to_gpu_batch(firsts, gts, labs, dev))
    assert all(float(l_first[k]) == float(l1[k]) for k in l1), 'duplicates after the first occurrence must not matter'
    # ... and permuting those unique points only permutes rows: the losses move by fp32 summation order only
    perm_pts = [p[rng.permutation(len(p))] for p in firsts]
    l3 = model(return_loss=True, **_to_gpu_batch(perm_pts, gts, labs, dev))
    for k in l1:
        assert _rel(l3[k], l1[k]) < 1e-4, (k, float(l3[k]), float(l1[k]))
    # coordinate-level properties on the real maps
    coords, feats = model.voxelize(batch['points'])
    from fcaf3d_amd.sparse import SparseTensor
    x = SparseTensor(feats, coordinates=coords, batch_size=2)
    n0 = x.cmap.n
    assert 2 * 85000 < n0 < 2 * 100000                                  # ~92.6k voxels per scene
    uniq = torch.unique(x.C, dim=0)
    assert uniq.shape[0] == n0                                           # de-duplicated
    lvl = x.cmap.strided(2).strided(2).strided(2)                        # stride 8 (backbone level 1)
    km = lvl.kernel_map(lvl, 3)
    assert torch.equal(km.nbr_t, km.nbr.flip(0))                         # same-set k3 map: transpose == offset flip
    centre = km.nbr[13]
    assert torch.equal(centre, torch.arange(lvl.n, device=dev, dtype=torch.int32))   # every voxel is its own centre
    # linearity of the sparse convolution on the full-size map
    w = torch.randn(27, 64, 64, device=dev)
    a, b = torch.randn(lvl.n, 64, device=dev), torch.randn(lvl.n, 64, device=dev)
    import fcaf3d_amd.functional as Fn
    lhs = Fn.sparse_conv(2.0 * a + b, w, km, lvl.n)
    rhs = 2.0 * Fn.sparse_conv(a, w, km, lvl.n) + Fn.sparse_conv(b, w, km, lvl.n)
    assert _rel(lhs, rhs) < 1e-5


def test_indoor_eval_hip_iou_vs_reference_golden():
    """the step after the path: indoor_eval with its IoU matrices from the HIP kernel reproduces the
    reference's mAP / mAR on the golden annotations"""
    from fcaf3d_amd.evaluation import indoor_eval
    from tests.test_oracle_golden import _indoor_eval_case
    _dev()
    d = np.load(os.path.join(G, 'indoor_eval.npz'))
    for case in (0, 1):
        gt_annos, dt_annos, label2cat, want = _indoor_eval_case(d, case)
        got = indoor_eval(gt_annos, dt_annos, (0.25, 0.5), label2cat)
        for k in want:
            assert abs(got[k] - want[k]) < 1e-4, (case, k, got[k], want[k])


def test_pipeline_feeds_the_detector_on_device(tmp_path):
    """the step before the path: .bin -> LoadPointsFromFile -> GlobalAlignment -> IndoorPointSample /
    RandomFlip3D / GlobalRotScaleTrans, all on the GPU, into forward_train; same transforms on the CPU give the same scene"""
    from fcaf3d_amd import pipelines as pl
    dev = _dev()
    pts, gt, labels = make_scene(21, n_points=30000)
    pts = pts.copy(); pts[:, 3:] *= 255.0 if pts[:, 3:].max() <= 1.0 else 1.0
    path = str(tmp_path / 'scene.bin')
    pts.astype(np.float32).tofile(path)
    th = 0.3
    A = np.eye(4, dtype=np.float32); A[:2, :2] = [[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]]; A[:3, 3] = [0.2, -0.1, 0.05]
    boxes = fa.DepthInstance3DBoxes(torch.from_numpy(gt), origin=(.5, .5, .5)).tensor
    aug = pl.TrainAugment(num_points=20000, with_yaw=False)
    outs = []
    for device in ('cpu', dev):
        p = pl.global_alignment(pl.load_points_from_file(path, device=device), A)
        params = dict(flip_h=True, flip_v=False, angle=0.05, scale=1.07, trans=[0.1, -0.05, 0.02])
        q, b = pl.flip_bev(p, boxes.to(device), 'horizontal', False)
        q, b = pl.rot_scale_trans(q, b, params['angle'], params['scale'], params['trans'], False)
        outs.append((q.cpu(), b.cpu()))
    assert torch.allclose(outs[0][0], outs[1][0], atol=1e-5) and torch.allclose(outs[0][1], outs[1][1], atol=1e-5)
    g = torch.Generator(device=dev).manual_seed(5)
    p = pl.global_alignment(pl.load_points_from_file(path, device=dev), A)
    q, b, params = aug(p, boxes.to(dev), g)
    assert q.shape == (20000, 6) and q.is_cuda
    model, m = _build('fcaf3d_scannet-3d-18class', 0.02, 2)
    model = model.to(dev).train()
    losses = model(return_loss=True, points=[q], gt_bboxes_3d=[fa.DepthInstance3DBoxes(b)], gt_labels_3d=[torch.from_numpy(labels).to(dev)],
                   img_metas=[dict(box_type_3d=fa.DepthInstance3DBoxes)])
    assert all(torch.isfinite(v) for v in losses.values())


def test_fused_augment_voxelize_vs_reference_goldens():
    """fc_augment_voxelize (GlobalAlignment / IndoorPointSample / RandomFlip3D / GlobalRotScaleTrans fused with the
    voxelisation, csrc/coords.hip) against tests/golden/pipeline.npz — the outputs of the reference's OWN DepthPoints /
    DepthInstance3DBoxes rotate / flip / scale / translate and GlobalAlignment on the same inputs (generated by importing
    them, tests/golden/make_golden.py::gen_pipeline): augmented points within 1e-6 of their scale (flips: exact), voxel
    coordinates equal to floor(golden / voxel_size) except for points that lie within 1e-5 of a cell face."""
    from fcaf3d_amd import pipelines as pl
    dev = _dev()
    d = np.load(os.path.join(G, 'pipeline.npz'))
    vs = 0.02

    def check(raw, want, params, align=None, idx=None, exact=False):
        lazy = pl.LazyAugmentedPoints(torch.from_numpy(raw).to(dev), None if idx is None else torch.from_numpy(idx).to(dev), params, align)
        got = lazy.materialize().cpu().numpy()
        if idx is not None:
            want = want[idx]
        if exact:
            assert np.array_equal(got, want)
        else:
            assert np.abs(got - want).max() <= 1e-6 * max(1.0, np.abs(want[:, :3]).max()), np.abs(got - want).max()
        n = len(want)
        coords = torch.empty((n, 4), dtype=torch.int32, device=dev)
        feats = torch.empty((n, 3), dtype=torch.float32, device=dev)
        lazy.voxelize_into(3, vs, 255.0, coords, feats)
        c = coords.cpu().numpy()
        ref_c = np.floor(want[:, :3].astype(np.float32) / np.float32(vs)).astype(np.int32)
        frac = want[:, :3] / vs - np.floor(want[:, :3] / vs)
        near_face = (np.minimum(frac, 1 - frac) < 1e-3).any(1)
        assert (c[:, 0] == 3).all() and np.array_equal(c[~near_face, 1:], ref_c[~near_face]) and near_face.mean() < 0.02
        assert np.array_equal(feats.cpu().numpy(), (want[:, 3:] / np.float32(255.0)).astype(np.float32))

    for case in (0, 1):
        raw = d[f'c{case}_points']
        angle, scale, tx, ty, tz = d[f'c{case}_params']
        check(raw, d[f'c{case}_rst_points'], dict(angle=float(angle), scale=float(scale), trans=(tx, ty, tz)))
        check(raw, d[f'c{case}_flip_horizontal_points'], dict(flip_h=True), exact=True)
        check(raw, d[f'c{case}_flip_vertical_points'], dict(flip_v=True), exact=True)
        idx = np.random.default_rng(case).permutation(len(raw))[:137].astype(np.int32)       # IndoorPointSample: a row gather
        check(raw, d[f'c{case}_rst_points'], dict(angle=float(angle), scale=float(scale), trans=(tx, ty, tz)), idx=idx)
    check(d['align_points_in'], d['align_points_out'], dict(), align=d['align_matrix'])


def test_lazy_augmentation_through_the_detector_equals_materialised_points():
    """TrainAugment.lazy: the detector voxelises the RAW scene with the drawn augmentation in one pass
    (SingleStageSparse3DDetector.voxelize -> fc_augment_voxelize) and gets the coordinates / features / losses it gets from
    the materialised augmented cloud; the GT boxes equal the ones TrainAugment.__call__ produces for the same draws."""
    from fcaf3d_amd import pipelines as pl
    dev = _dev()
    pts, gt, labels = make_scene(23, n_points=30000)
    raw = torch.from_numpy(pts).to(dev)
    boxes = fa.DepthInstance3DBoxes(torch.from_numpy(gt), origin=(.5, .5, .5)).tensor.to(dev)
    aug = pl.TrainAugment(num_points=20000, with_yaw=False)
    lazy, b_lazy, params = aug.lazy(raw, boxes, torch.Generator(device=dev).manual_seed(9))
    q, b_ref, params2 = aug(raw, boxes, torch.Generator(device=dev).manual_seed(9))
    assert params == params2 and torch.equal(b_lazy, b_ref)
    mat = lazy.materialize()
    assert torch.allclose(mat, q, atol=2e-6), float((mat - q).abs().max())
    model, m = _build('fcaf3d_scannet-3d-18class', 0.02, 2)
    model = model.to(dev).train()
    c1, f1 = model.voxelize([lazy])
    c2, f2 = model.voxelize([mat])
    assert torch.equal(c1, c2) and torch.equal(f1, f2)
    kw = dict(gt_bboxes_3d=[fa.DepthInstance3DBoxes(b_lazy)], gt_labels_3d=[torch.from_numpy(labels).to(dev)],
              img_metas=[dict(box_type_3d=fa.DepthInstance3DBoxes)])
    l1 = model(return_loss=True, points=[lazy], **kw)
    l2 = model(return_loss=True, points=[mat], **kw)
    assert all(float(l1[k]) == float(l2[k]) for k in l1)


def test_indoor_eval_reference_test_vectors_hip():
    """known-answer vectors of the reference's tests/test_metrics/test_indoor_eval.py, IoU matrices from the HIP kernel"""
    from tests.test_oracle_golden import _check_ref_indoor_eval
    _dev()
    _check_ref_indoor_eval(None)


def test_bf16_fast_mode_against_the_oracle():
    """the flagged NON-PARITY bf16 fast mode (fc_set_bf16_fast: operands rounded to bf16, one MFMA product) held
    against the ORACLE's losses, not only against the repo's own exact route: a two-level detector on one 12k-point scene — every loss
    within 2e-2 of oracle/model_oracle.py (bf16 carries 8 significand bits; measured ~1e-3), and visibly off the parity bound."""
    import fcaf3d_amd._lib as L
    from fcaf3d_amd.synthetic import make_scene
    from oracle import model_oracle as MO
    dev = _dev()
    torch.manual_seed(0)
    cfg = fa.get_config('fcaf3d_scannet-3d-18class', voxel_size=0.02)
    m = cfg.model
    m.backbone['n_outs'] = 2
    m.neck_with_head['in_channels'] = (64, 128)
    m.neck_with_head.assigner['n_scales'] = 2
    model = fa.build_detector(m, train_cfg=m.get('train_cfg'), test_cfg=m.get('test_cfg'))
    P = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model = model.to(dev).train()
    p, g, l = make_scene(5, n_points=12000)
    batch = dict(points=[torch.from_numpy(p).to(dev)], gt_bboxes_3d=[fa.DepthInstance3DBoxes(torch.from_numpy(g), origin=(.5, .5, .5))],
                 gt_labels_3d=[torch.from_numpy(l).to(dev)], img_metas=[dict(box_type_3d=fa.DepthInstance3DBoxes)])
    with torch.no_grad():
        ref = {k: float(v) for k, v in MO.forward_train(P, m, [p], [g], [l]).items()}
        exact = {k: float(v) for k, v in model(return_loss=True, **batch).items()}
        L.lib().fc_set_bf16_fast(1)
        try:
            fast = {k: float(v) for k, v in model(return_loss=True, **batch).items()}
        finally:
            L.lib().fc_set_bf16_fast(0)
    print('losses: oracle', ref, 'exact route', exact, 'bf16 fast mode', fast)
    worst = 0.0
    for k in ref:
        assert abs(exact[k] - ref[k]) <= 5e-4 * max(1.0, abs(ref[k])), (k, exact[k], ref[k])      # (anchor only: the parity tests hold this to 1e-4)
        e = abs(fast[k] - ref[k]) / max(1.0, abs(ref[k]))
        worst = max(worst, e)
        assert e <= 2e-2, (k, fast[k], ref[k])
    assert worst > 1e-6, 'the fast mode is expected to differ from the oracle beyond fp32 rounding'
