"""CPU: the operator lists of the native executor (fcaf3d_amd/executor.py) are built from the module graph without a GPU —
structure checks of what fc_exec (csrc/exec.hip) will walk: operand indices in range, every trainable tensor has a place its
gradient is written to, forward / backward operator counts follow the model (the numerics are the GPU tests' business:
tests/test_gpu_exec.py compares the executor with the per-operator path bit for bit)."""
import numpy as np
import pytest
import torch

import fcaf3d_amd as fa
from fcaf3d_amd import executor as E
from fcaf3d_amd import nn as MEnn


def _model(levels=4, name='fcaf3d_scannet-3d-18class'):
    torch.manual_seed(0)
    cfg = fa.get_config(name, voxel_size=0.02)
    m = cfg.model
    m.backbone['n_outs'] = levels
    m.neck_with_head['in_channels'] = (64, 128, 256, 512)[:levels]
    m.neck_with_head.assigner['n_scales'] = levels
    return fa.build_detector(m, train_cfg=m.get('train_cfg'), test_cfg=m.get('test_cfg'))


@pytest.mark.parametrize('levels,wgrad_async,head_overlap', [(4, False, False), (4, True, True), (2, True, True), (1, False, False)])
def test_training_program_structure(levels, wgrad_async, head_overlap):
    det = _model(levels)
    assert E.supported(det)
    p = E.NetProgram(det, True, wgrad_async, head_overlap)
    convs = [m for m in det.modules() if isinstance(m, MEnn.MinkowskiConvolution)]
    gents = [m for m in det.modules() if isinstance(m, MEnn.MinkowskiGenerativeConvolutionTranspose)]
    n_conv = len(convs) - 3 - 1                       # the three 1x1 head kernels run as one packed GEMM per level; the stem has its own operator
    n_gemm = len(gents) + levels                      # generative convolutions + the packed head GEMM of every level
    assert p.n_conv_f == n_conv + n_gemm
    assert p.n_conv_b == n_conv + n_gemm              # one backward-data launch per forward launch (the stem's input needs none)
    f, b = p.ops_f, p.ops_b
    assert int((b[:, 0] == E.OP_WGRAD).sum()) == n_conv + n_gemm and int((b[:, 0] == E.OP_STEM_WGRAD).sum()) == 1
    n_bn = len([m for m in det.modules() if isinstance(m, MEnn.MinkowskiBatchNorm)])
    assert int((f[:, 0] == E.OP_BN_FWD).sum()) == n_bn == int((b[:, 0] == E.OP_BN_BWD).sum())
    # every parameter that requires a gradient is written by some operator (or by the head's bias / scale reductions)
    nh = det.neck_with_head
    direct = {id(nh.cls_conv.bias)} | {id(s.scale) for s in nh.scales}
    reached = {off for _, off in p.grad_refs} | {int(o) // 4 for o in p._small_goff}
    for prm in det.parameters():
        assert id(prm) in direct or p._goff[id(prm)] in reached, 'a parameter has no gradient destination'
    # streams and events: cross-stream operators only in the overlapped program, every wait has its record
    for ops in (f, b):
        assert set(np.unique(ops[:, 1])) <= ({0, 1, 2} if (wgrad_async or head_overlap) and levels > 1 or wgrad_async else {0})
        rec = set(ops[ops[:, 0] == E.OP_RECORD][:, 2])
        assert set(ops[ops[:, 0] == E.OP_WAIT][:, 2]) <= rec
    if not (wgrad_async or head_overlap):
        assert not (f[:, 0] == E.OP_RECORD).any() and not (b[:, 0] == E.OP_RECORD).any()


def test_inference_program_has_no_backward_and_no_saved_statistics():
    det = _model(4).eval()
    p = E.NetProgram(det, False, False, True)
    assert len(p.ops_b) == 0 and not p.arena['b']
    bn = p.ops_f[p.ops_f[:, 0] == E.OP_BN_FWD]
    assert (bn[:, 18] == 0).all() and (bn[:, 12] == -1).all()          # eval mode: running statistics, nothing saved


def test_bottleneck_and_wide_heads_fall_back():
    torch.manual_seed(0)
    cfg = fa.get_config('fcaf3d_scannet-3d-18class', voxel_size=0.02)
    m = cfg.model
    m.backbone['depth'] = 50
    m.backbone['n_outs'] = 2
    m.neck_with_head['in_channels'] = (256, 512)
    m.neck_with_head.assigner['n_scales'] = 2
    assert not E.supported(fa.build_detector(m, train_cfg=m.get('train_cfg'), test_cfg=m.get('test_cfg')))
    cfg = fa.get_config('fcaf3d_scannet-3d-18class', voxel_size=0.02)
    cfg.model.neck_with_head['n_classes'] = 80
    assert not E.supported(fa.build_detector(cfg.model, train_cfg=cfg.model.get('train_cfg'), test_cfg=cfg.model.get('test_cfg')))


def test_batchnorm_fusions_are_wired_consistently():
    """r5: the static links of the BatchNorm fusions — every forward BatchNorm names the convolution that wrote its input and that
    convolution names a statistics table; a backward BatchNorm that names a producer names a backward-data convolution whose result
    has the layer's shape and which carries the layer's input / statistics; a second gradient contribution is either handed to the
    BatchNorm (gy2) or added by OP_ADD, never both; nothing is linked when FC_BN_FUSE is off."""
    import fcaf3d_amd.functional as Fn
    det = _model(4)
    for wgrad_async, head_overlap in ((True, True), (False, False)):
        p = E.NetProgram(det, True, wgrad_async, head_overlap)
        f, b = p.ops_f, p.ops_b
        bn_f = f[f[:, 0] == E.OP_BN_FWD]
        assert (bn_f[:, 19] > 0).all(), 'every training-mode BatchNorm takes its statistics from a producer'
        for row in bn_f:
            prod = f[row[19] - 1]
            assert prod[0] == E.OP_CONV and prod[5] == 0 and prod[10] > 0 and prod[6] == row[2], 'the producer wrote the BatchNorm input'
            assert prod[9] == row[4] * row[20] and row[20] in (1, 8)       # columns = groups x channels
        assert int(((f[:, 0] == E.OP_CONV) & (f[:, 10] > 0)).sum()) == len(bn_f)
        bn_b = b[b[:, 0] == E.OP_BN_BWD]
        linked = bn_b[bn_b[:, 18] > 0]
        assert len(linked) >= 36 and len(bn_b) == len(bn_f)
        for row in linked:
            prod = b[row[18] - 1]
            assert prod[0] == E.OP_CONV and prod[5] == 1 and prod[10] > 0
            assert prod[11] - 1 == row[2] and prod[12] == row[7] and prod[13] == row[8] and prod[9] == row[6]     # layer input, mean, var, channels
            last = row[17] - 1 if row[17] > 0 else row[4]                # the contribution that arrived last = the producer's result
            assert prod[6] == last
            if row[17] > 0:
                assert prod[18] - 1 == row[4], 'the earlier contribution rides along as `add`'
            assert (prod[19] > 0) == (row[3] >= 0), "act' from the output exactly where the layer had a residual"
        # stream order: a linked producer precedes its BatchNorm and sits on the same stream
        idx = {tuple(r): i for i, r in enumerate(map(tuple, b))}
        for row in linked:
            assert row[18] - 1 < idx[tuple(row)] and b[row[18] - 1][1] == row[1]
    Fn.BN_FUSE = False
    try:
        p0 = E.NetProgram(det, True, True, True)
        assert not (p0.ops_f[:, 10][p0.ops_f[:, 0] == E.OP_CONV] > 0).any() and not (p0.ops_f[:, 19][p0.ops_f[:, 0] == E.OP_BN_FWD] > 0).any()
        bb = p0.ops_b[p0.ops_b[:, 0] == E.OP_BN_BWD]
        assert not (bb[:, 17] > 0).any() and not (bb[:, 18] > 0).any()
        assert int((p0.ops_b[:, 0] == E.OP_ADD).sum()) > int((p.ops_b[:, 0] == E.OP_ADD).sum())
    finally:
        Fn.BN_FUSE = True
