"""CPU: oracle.model_oracle.DecisionTape — the instrument of the decision-controlled gradient parity test
(tests/test_gpu_model.py::test_gradient_parity_with_equal_decisions).  Replaying the oracle's OWN decisions must not change
a bit of its losses or gradients; replaying them into the fp64 oracle must remove the fp32-vs-fp64 decision flips."""
import numpy as np
import torch

import fcaf3d_amd as fa
from fcaf3d_amd.synthetic import make_scene
from oracle import model_oracle as MO


def _setup():
    torch.manual_seed(0)
    cfg = fa.get_config('fcaf3d_scannet-3d-18class', voxel_size=0.02)
    m = cfg.model
    m.backbone['n_outs'] = 2
    m.neck_with_head['in_channels'] = (64, 128)
    m.neck_with_head.assigner['n_scales'] = 2
    model = fa.build_detector(m, train_cfg=m.get('train_cfg'), test_cfg=m.get('test_cfg'))
    P = {k: v.detach().clone().requires_grad_(v.dtype.is_floating_point) for k, v in model.state_dict().items()}
    p, g, l = make_scene(5, n_points=6000)
    return m, P, [p], [g], [l]


def _run(P, m, pts, gts, labs, tape):
    for v in P.values():
        if v.dtype.is_floating_point:
            v.grad = None
    MO.TAPE = tape
    try:
        losses = MO.forward_train(P, m, pts, gts, labs)
    finally:
        MO.TAPE = None
    sum(losses.values()).backward()
    return {k: float(v) for k, v in losses.items()}, {k: v.grad.clone() for k, v in P.items() if v.dtype.is_floating_point and v.grad is not None}


def test_replaying_own_decisions_is_the_identity():
    m, P, pts, gts, labs = _setup()
    l0, g0 = _run(P, m, pts, gts, labs, None)
    rec = MO.DecisionTape()
    l1, g1 = _run(P, m, pts, gts, labs, rec)
    assert l0 == l1 and all(torch.equal(g0[k], g1[k]) for k in g0), 'recording must not change the oracle'
    assert len(rec.relu) == 1 + 2 * (3 + 4) and rec.pool is not None          # stem + norm1 / norm2 of layer1 (3) and layer2 (4)
    tape = rec.replay()
    l2, g2 = _run(P, m, pts, gts, labs, tape)
    assert tape.i == len(tape.relu) and tape.total_flips() == 0
    assert l0 == l2 and all(torch.equal(g0[k], g2[k]) for k in g0), 'replaying the own decisions must be the identity'


def test_fp32_decisions_replayed_into_the_fp64_oracle():
    m, P, pts, gts, labs = _setup()
    rec = MO.DecisionTape()
    _, g32 = _run(P, m, pts, gts, labs, rec)
    P64 = {k: (v.detach().double().requires_grad_(True) if v.dtype.is_floating_point else v) for k, v in P.items()}
    tape = rec.replay()
    _, g64 = _run(P64, m, pts, gts, labs, tape)
    # with equal decisions what separates the two is rounding of the arithmetic alone
    errs = {k: float((g32[k].double() - g64[k]).abs().max() / max(1e-3, float(g64[k].abs().max()))) for k in g32}
    assert max(errs.values()) < 1e-3, max(errs, key=errs.get)
    assert np.median(list(errs.values())) < 1e-5
