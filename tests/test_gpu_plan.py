"""-m gpu: the native coordinate phase (fcaf3d_amd/plan.py + csrc/plan.hip: fc_plan_levels + fc_plan_maps, two read-backs per step)
against the per-operator coordinate phase (fcaf3d_amd/sparse.py + csrc/coords.hip, itself held bit-exact against oracle/me_oracle.py
by tests/test_gpu_ops.py): every coordinate set, voxel hash LOOKUP, kernel map, derived table, union row and head array must be
identical — integer work, bit for bit (indices / kernel maps bit-exact)."""
import numpy as np
import pytest
import torch

import fcaf3d_amd as fa
from fcaf3d_amd import _lib as L
from fcaf3d_amd import plan as PL
from fcaf3d_amd import sparse as SP
from fcaf3d_amd.synthetic import make_scene

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available()
    return torch.device('cuda:0')


def _build(name='fcaf3d_scannet-3d-18class', levels=4, voxel_size=0.02, pts_threshold=None):
    torch.manual_seed(0)
    cfg = fa.get_config(name, voxel_size=voxel_size)
    m = cfg.model
    m.backbone['n_outs'] = levels
    m.neck_with_head['in_channels'] = (64, 128, 256, 512)[:levels]
    m.neck_with_head.assigner['n_scales'] = levels
    if pts_threshold is not None:
        m.neck_with_head['pts_threshold'] = pts_threshold
    return fa.build_detector(m, train_cfg=m.get('train_cfg'), test_cfg=m.get('test_cfg'))


def _batch(seeds, dev, n_points, **kw):
    sc = [make_scene(s, n_points=n_points, **kw) for s in seeds]
    return ([torch.from_numpy(s[0]).to(dev) for s in sc],
            [fa.DepthInstance3DBoxes(torch.from_numpy(s[1]), origin=(.5, .5, .5)) for s in sc],
            [torch.from_numpy(s[2]).to(dev) for s in sc])


def _eq(a, b, what):
    assert a.shape == b.shape, (what, a.shape, b.shape)
    assert torch.equal(a, b), what


def _pairs_eq(p, q, what):
    (pi, po, pos, cnt), (qi, qo, qpos, qcnt) = p, q
    _eq(cnt, qcnt, what + ' cnt')
    _eq(pos, qpos, what + ' pos')
    for k, c in enumerate(cnt.cpu().tolist()):
        _eq(pi[k, :c], qi[k, :c], f'{what} in[{k}]')
        _eq(po[k, :c], qo[k, :c], f'{what} out[{k}]')


def _maps_eq(a, b, backward, what):
    assert (a.K, a.n_in, a.n_out, a.sort_rows, a.use_pairs) == (b.K, b.n_in, b.n_out, b.sort_rows, b.use_pairs), what
    _eq(a.nbr, b.nbr, what + ' nbr')
    if a.K != 27:
        if backward:
            _eq(a.nbr_t, b.nbr_t, what + ' nbr_t')
        return
    if a.use_pairs and a.n_out <= SP.PAIR_CONV_ROWS:
        _pairs_eq(a.pairs(), b.pairs(), what + ' pairs')
        assert a.pair_tiles() == b.pair_tiles(), what
    else:
        (ta, ia), (tb, ib) = a.sorted_fwd(), b.sorted_fwd()
        _eq(ta, tb, what + ' sorted table')
        assert (ia is None) == (ib is None), what
        if ia is not None:
            _eq(ia, ib, what + ' sorted order')
    if backward:
        _eq(a.nbr_t, b.nbr_t, what + ' nbr_t')
        if a.use_pairs:
            _pairs_eq(a.pairs(), b.pairs(), what + ' pairs (wgrad)')
        if a.use_pairs and a.n_in <= SP.PAIR_CONV_ROWS:
            _pairs_eq(a.pairs_t(), b.pairs_t(), what + ' pairs_t')
            assert a.pair_tiles(True) == b.pair_tiles(True), what
        else:
            (ta, ia), (tb, ib) = a.sorted_bwd(), b.sorted_bwd()
            _eq(ta, tb, what + ' sorted_t table')
            if ia is not None:
                _eq(ia, ib, what + ' sorted_t order')
    for key in ((True, True), (True, False)) if backward else ((True, False),):
        da, db = a.desc(*key), b.desc(*key)
        assert (da[0], da[1], da[2], da[13], da[18], da[19]) == (db[0], db[1], db[2], db[13], db[18], db[19]), (what, key, da, db)
        assert [bool(v) for v in da[3:13]] == [bool(v) for v in db[3:13]], (what, key)


def _lookup_all(cm, q):
    """row of every query coordinate through the set's voxel hash (the table LAYOUT may differ, the mapping may not)"""
    nbr = torch.empty((1, q.shape[0]), dtype=torch.int32, device=q.device)
    offs = torch.zeros((1, 3), dtype=torch.int32, device=q.device)
    L.call('fc_kernel_map', L.ptr(q.contiguous()), q.shape[0], L.ptr(cm.keys), L.ptr(cm.vals), cm.cap, L.ptr(offs), 1, L.ptr(nbr), L.stream())
    return nbr[0]


def _walk(det, cm0, backward):
    """the object graph both coordinate phases leave behind, in a fixed order"""
    nl = min(det.backbone.n_outs, 4)
    m1 = cm0.strided(2); m2 = m1.strided(2)
    sets, maps = [cm0, m1, m2], [('stem', cm0.kernel_map(m1, 3)), ('pool', m1.kernel_map(m2, 2))]
    prev, lv = m2, []
    for li in range(1, nl + 1):
        mi = prev.strided(2)
        maps += [(f'down{li}', prev.kernel_map(mi, 3)), (f'ds{li}', prev.kernel_map(mi, 1)), (f'same{li}', mi.kernel_map(mi, 3))]
        sets.append(mi); lv.append(mi); prev = mi
    return sets, maps, lv


@pytest.mark.parametrize('levels,seeds,n_points,voxel,training', [
    (4, (1, 2), 30000, 0.02, True),
    (4, (3, 4, 5), 100000, 0.02, True),            # BASELINE config 2 shape: mask-sorted tables (>= 8 192 rows) and pair lists side by side
    (4, (6,), 100000, 0.01, True),                 # literal 1 cm
    (2, (7, 8), 30000, 0.02, True),
    (1, (9,), 20000, 0.02, True),                  # BASELINE config 1
    (4, (10, 11), 30000, 0.02, False),             # inference: no backward tables
])
def test_native_plan_equals_the_per_operator_coordinate_phase(levels, seeds, n_points, voxel, training):
    dev = _dev()
    det = _build(levels=levels, voxel_size=voxel).to(dev)
    det.train(training)
    pts, gtb, gtl = _batch(seeds, dev, n_points)
    with torch.set_grad_enabled(training):
        PL.ENABLED = False
        try:
            xa = det._sparse_input(pts, (gtb, gtl) if training else None)
            assert det._step_plan is None
            ha = det.plan_maps(xa.cmap)
            prep_a = getattr(det.neck_with_head, '_prepared', None)
        finally:
            PL.ENABLED = True
        xb = det._sparse_input(pts, (gtb, gtl) if training else None)
        sp = det._step_plan
        assert sp is not None and sp.structured, 'the native plan did not run'
        hb = sp.head_maps
        prep_b = getattr(det.neck_with_head, '_prepared', None)
        backward = training
        _eq(xa.F, xb.F, 'level-0 features')
        sa, ma, la = _walk(det, xa.cmap, backward)
        sb, mb, lb = _walk(det, xb.cmap, backward)
        for i, (a, b) in enumerate(zip(sa, sb)):
            assert (a.n, a.stride, a.batch_size) == (b.n, b.stride, b.batch_size), i
            _eq(a.coords, b.coords, f'set {i} coords')
            assert a.scene_counts == b.scene_counts, i
            assert a.cap == b.cap, i
            rows = _lookup_all(b, a.coords)
            _eq(rows, torch.arange(a.n, dtype=torch.int32, device=dev), f'set {i} hash')
            far = a.coords.clone(); far[:, 1] += 30001 * a.stride
            assert int((_lookup_all(b, far) >= 0).sum()) == 0
        for (na, a), (nb, b) in zip(ma, mb):
            _maps_eq(a, b, backward and na not in ('stem', 'pool'), na)
        assert (ha is None) == (hb is None)
        if ha is not None:
            assert len(ha) == len(hb) == levels
            for l, (a, b) in enumerate(zip(ha, hb)):
                _eq(a.coords, b.coords, f'head level {l}')
                assert a.scene_counts == b.scene_counts
                for pa, pb in zip(a.decomposition_permutations, b.decomposition_permutations):
                    _eq(pa, pb, f'head level {l} permutation')
            xa_, xb_ = la[-1], lb[-1]
            for i in range(levels - 2, -1, -1):
                ga, gb = xa_.generate(), xb_.generate()
                _maps_eq(ga.kernel_map(ga, 3), gb.kernel_map(gb, 3), backward, f'gsame{i}')
                ua, ra, swa = la[i].union(ga)
                ub, rb, swb = lb[i].union(gb)
                assert swa and swb and ua is ga and ub is gb
                _eq(ra, rb, f'union rows {i}')
                xa_, xb_ = ga, gb
        if training and ha is not None:
            ta, tb = prep_a[2], prep_b[2]
            for k in ('pts', 'scene', 'ct', 'bt', 'labels', 'posf', 'inv_pos', 'inv_den'):
                _eq(ta[k], tb[k], f'targets {k}')
    torch.cuda.synchronize()


def test_native_plan_reports_pruning_and_out_of_range_points():
    dev = _dev()
    det = _build(levels=4, pts_threshold=3000).to(dev).train()
    pts, gtb, gtl = _batch((21, 22), dev, 60000)
    PL.ENABLED = False
    try:
        det._sparse_input(pts, (gtb, gtl))
        want = det._prune_level
    finally:
        PL.ENABLED = True
    det._sparse_input(pts, (gtb, gtl))
    assert det._step_plan is not None and det._prune_level == want and want is not None
    bad = [p.clone() for p in pts]
    bad[1][17, 0] = 1e9
    with pytest.raises(ValueError):
        det._sparse_input(bad, (gtb, gtl))
    torch.cuda.synchronize()


def test_native_plan_training_step_equals_the_per_operator_phase():
    """losses and every parameter gradient of a full step, native plan vs per-operator coordinate phase: the same maps feed the same
    kernels, so the results are bit-identical"""
    dev = _dev()
    det = _build(levels=4).to(dev).train()
    pts, gtb, gtl = _batch((31, 32), dev, 40000)
    batch = dict(points=pts, gt_bboxes_3d=gtb, gt_labels_3d=gtl, img_metas=[dict(box_type_3d=fa.DepthInstance3DBoxes)] * 2)
    res = []
    for native in (False, True):
        PL.ENABLED = native
        try:
            det.zero_grad(set_to_none=True)
            losses = det(return_loss=True, **batch)
            sum(losses.values()).backward()
            torch.cuda.synchronize()
            res.append(({k: float(v) for k, v in losses.items()}, {k: p.grad.clone() for k, p in det.named_parameters()}))
        finally:
            PL.ENABLED = True
    assert res[0][0] == res[1][0], (res[0][0], res[1][0])
    for k in res[0][1]:
        assert torch.equal(res[0][1][k], res[1][1][k]), k


@pytest.mark.parametrize('n', [1, 63, 4096, 4097, 70001, 300000])
def test_argsort27_is_the_stable_argsort(n):
    dev = _dev()
    g = torch.Generator(device='cpu').manual_seed(n)
    # few distinct masks (as the occupancy masks are) and the full 27-bit range
    for hi in (37, 1 << 27):
        keys = torch.randint(0, hi, (n,), generator=g, dtype=torch.int32).to(dev)
        order = torch.empty(n, dtype=torch.int32, device=dev)
        ws = torch.empty(L.query('fc_argsort27_ws_bytes', n), dtype=torch.uint8, device=dev)
        L.call('fc_argsort27', L.ptr(keys), n, L.ptr(order), L.ptr(ws), ws.numel(), L.stream())
        want = torch.sort(keys, stable=True).indices.to(torch.int32)
        assert torch.equal(order, want)


@pytest.mark.parametrize('native', [True, False])
def test_a_step_leaves_no_device_memory_to_the_cyclic_collector(native):
    """coordinate sets, kernel maps and the arenas they view are freed by reference count when a step ends (sparse.CoordMap: caches
    keyed weakly by the other set, a generated set holds its parent weakly): with CPython's cyclic collector OFF the allocated device
    memory does not grow from step to step — r6 found 0.66 GB per 8-scene step waiting for a collection (three reference cycles), and
    +36.8 GB reserved over 1 000 steps; both coordinate paths (the native plan and the per-operator phase)"""
    import gc
    dev = _dev()
    det = _build(levels=4).to(dev).train()
    pts, gtb, gtl = _batch((41, 42), dev, 30000)
    batch = dict(points=pts, gt_bboxes_3d=gtb, gt_labels_3d=gtl, img_metas=[dict(box_type_3d=fa.DepthInstance3DBoxes)] * 2)

    def step():
        det.zero_grad(set_to_none=True)
        losses = det(return_loss=True, **batch)
        sum(losses.values()).backward()
    PL.ENABLED = native
    try:
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        gc.collect()
        gc.disable()
        try:
            step()
            torch.cuda.synchronize()
            m0 = torch.cuda.memory_allocated()
            for _ in range(4):
                step()
            torch.cuda.synchronize()
            m1 = torch.cuda.memory_allocated()
        finally:
            gc.enable()
    finally:
        PL.ENABLED = True
    assert m1 - m0 <= 4 << 20, f'{(m1 - m0) / 2 ** 20:.1f} MB of device memory per 4 steps are held by reference cycles'
