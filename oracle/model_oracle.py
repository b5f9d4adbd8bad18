"""ORACLE — test infrastructure only.  The whole FCAF3D forward (+ autograd backward) on the CPU,
composed from oracle/me_oracle.py and oracle/loss_oracle.py, driven by a plain state_dict with the
reference's parameter names.  Follows
  mmdet3d/models/detectors/single_stage_sparse.py:32-59,
  mmdet3d/models/backbones/me_resnet.py:14-99 (+ MinkowskiEngine BasicBlock, Appendix A.7),
  mmdet3d/models/dense_heads/fcaf3d_neck_with_head.py:94-279, :128-253, :332-374.
Used by the parity tests and as bench.py's `cpu_baseline` ("port": MinkowskiEngine's CPU algorithm —
hash-map kernel maps, per-offset gather -> GEMM -> scatter-add — restated; ME itself is not installable)."""
import numpy as np
import torch

from . import bev
from . import loss_oracle as lo
from . import me_oracle as mo

LAYERS = {14: (1, 1, 1, 1), 18: (2, 2, 2, 2), 34: (3, 4, 6, 3), 50: (4, 3, 6, 3), 101: (3, 4, 23, 3)}
BOTTLENECK = (50, 101)   # MinkowskiEngine.modules.resnet_block.Bottleneck (me_resnet.py:114-119): 1x1 -> 3x3 (stride) -> 1x1 (x4)
TRAINING = True     # False: BatchNorm uses the running statistics of the state_dict (model.eval())


class DecisionTape:
    """The DISCRETE decisions of one forward pass of the backbone, in execution order: the sign pattern of every ReLU
    (stem, norm1 / norm2 of every BasicBlock / Bottleneck) and the arg-max row of the stem's max-pooling.  With `TAPE` set
    the oracle takes these decisions from the tape instead of from its own pre-activations (the forward value of a flipped
    element changes by its pre-activation, ~1e-8 at fp32 rounding level; the BACKWARD pass then differentiates the same
    piecewise-linear branch as the path the tape was recorded from) and counts where its own decision would have differed.
    tests/test_gpu_model.py records the tape from the HIP forward: gradient parity with equal decisions is a statement about
    the kernels' arithmetic, not about which side of zero a 1e-8 pre-activation fell on.  (The neck's ELU is C1: no decision.)"""

    def __init__(self, relu_masks=None, pool_arg=None, prune_kept=None):
        """no arguments: RECORD the oracle's own decisions (a tape to replay, e.g. into the fp64 oracle)"""
        self.recording = relu_masks is None
        self.relu = list(relu_masks or [])    # bool (N, C) tensors: output > 0
        self.pool = pool_arg                  # int64 (n_out, C): input row each pooled value came from
        self.prune = list(prune_kept or [])   # per pruned neck level, in execution order: (k, 4) int coordinates kept by the top-k
        self.i = 0
        self.ip = 0
        self.flips = []                       # (site, disagreeing elements, elements, max |own pre-activation| among them)

    def replay(self):
        return DecisionTape(self.relu, self.pool, self.prune)

    def prune_forced(self, coords, own_mask, site):
        """the per-scene top-k selection of `_prune` (fcaf3d_neck_with_head.py:110-126): keep the recorded coordinate set"""
        if self.recording:
            self.prune.append(coords[own_mask].copy())
            return own_mask
        kept = self.prune[self.ip]
        self.ip += 1
        mask = np.isin(mo.pack_keys(coords), mo.pack_keys(kept))
        assert int(mask.sum()) == len(kept), (site, 'a recorded voxel is not in the oracle\'s set')
        self.flips.append((site, int((mask != own_mask).sum()), len(mask), 0.0))
        return mask

    def relu_forced(self, pre, site):
        if self.recording:
            self.relu.append(pre.detach() > 0)
            self.i += 1
            return torch.relu(pre)
        mask = self.relu[self.i]
        self.i += 1
        assert mask.shape == pre.shape, (site, mask.shape, pre.shape)
        own = pre.detach() > 0
        diff = own != mask
        n = int(diff.sum())
        self.flips.append((site, n, pre.numel(), float(pre.detach().abs()[diff].max()) if n else 0.0))
        return torch.where(mask, pre, torch.zeros_like(pre))

    def pool_forced(self, feats, nbr, site):
        if self.recording:
            K, n_out = nbr.shape
            neg = torch.full((1, feats.shape[1]), -float('inf'), dtype=feats.dtype)
            idx = torch.from_numpy(np.where(nbr >= 0, nbr, len(feats)).astype(np.int64))
            g = torch.cat([feats.detach(), neg])[idx.reshape(-1)].reshape(K, n_out, -1)
            self.pool = idx.t().gather(1, g.max(0).indices)               # (n_out, C): the row that holds the maximum
            return mo.max_pool(feats, nbr)
        own = mo.max_pool(feats.detach(), nbr)
        arg = self.pool.long()
        out = feats.gather(0, arg)
        diff = out.detach() != own
        n = int(diff.sum())
        self.flips.append((site, n, out.numel(), float((out.detach() - own).abs()[diff].max()) if n else 0.0))
        return out

    def total_flips(self):
        return sum(f[1] for f in self.flips)


TAPE = None          # a DecisionTape: take the backbone's ReLU / arg-max decisions from it


class SP:
    """coords (N,4) numpy int32, feats (N,C) torch, tensor stride, kernel-map cache shared per coordinate set"""

    def __init__(self, coords, feats, stride, cache=None):
        self.C, self.F, self.stride = coords, feats, stride
        self.cache = cache if cache is not None else {}


def _kmap(x, out_coords, ks, tag):
    key = (tag, ks)
    if key not in x.cache:
        x.cache[key] = mo.kernel_map(x.C, out_coords, mo.kernel_offsets(ks, x.stride))
    return x.cache[key]


def _strided(x, s):
    key = ('stride', s)
    if key not in x.cache:
        x.cache[key] = (mo.stride_coords(x.C, x.stride, s), {})
    return x.cache[key]


def conv(x, w, ks, s=1):
    if s == 1:
        return SP(x.C, mo.conv(x.F, w, _kmap(x, x.C, ks, 'same')), x.stride, x.cache)
    oc, ocache = _strided(x, s)
    return SP(oc, mo.conv(x.F, w, _kmap(x, oc, ks, 'down')), x.stride * s, ocache)


def bn(x, P, pre, act=None, residual=None):
    if TRAINING:
        f = mo.batch_norm(x.F, P[pre + '.bn.weight'], P[pre + '.bn.bias'])
    else:
        f = torch.nn.functional.batch_norm(x.F, P[pre + '.bn.running_mean'], P[pre + '.bn.running_var'],
                                           P[pre + '.bn.weight'], P[pre + '.bn.bias'], False, 0.1, 1e-5)
    if residual is not None:
        f = f + residual
    if act == 'relu':
        f = TAPE.relu_forced(f, pre) if TAPE is not None else torch.relu(f)
    elif act == 'elu':
        f = torch.nn.functional.elu(f)
    return SP(x.C, f, x.stride, x.cache)


def backbone(x, P, depth=34, n_outs=4):
    x = conv(x, P['backbone.conv1.0.kernel'], 3, 2)
    f = mo.instance_norm(x.F, x.C[:, 0], P['backbone.conv1.1.weight'], P['backbone.conv1.1.bias'])
    x = SP(x.C, TAPE.relu_forced(f, 'backbone.conv1.1') if TAPE is not None else torch.relu(f), x.stride, x.cache)
    oc, ocache = _strided(x, 2)
    pool_nbr = _kmap(x, oc, 2, 'down')
    x = SP(oc, TAPE.pool_forced(x.F, pool_nbr, 'backbone.conv1.3') if TAPE is not None else mo.max_pool(x.F, pool_nbr),
           x.stride * 2, ocache)
    outs = []
    def k3d(w):                      # ME stores a kernel_size = 1, stride = 1 kernel as (Cin, Cout)
        return w if w.dim() == 3 else w.unsqueeze(0)
    for li in range(n_outs):
        for j in range(LAYERS[depth][li]):
            pre = f'backbone.layer{li + 1}.{j}'
            if depth in BOTTLENECK:
                # ME BottleneckBase.forward: conv1 (1x1) - norm - relu - conv2 (3x3, the block's stride) - norm - relu -
                # conv3 (1x1, 4 x planes) - norm, + downsample(x) (first block of a layer: 1x1 stride-2 conv + norm), relu
                s = 2 if j == 0 else 1
                res = bn(conv(x, k3d(P[pre + '.downsample.0.kernel']), 1, 2), P, pre + '.downsample.1') if j == 0 else x
                out = bn(conv(x, k3d(P[pre + '.conv1.kernel']), 1, 1), P, pre + '.norm1', 'relu')
                out = bn(conv(out, P[pre + '.conv2.kernel'], 3, s), P, pre + '.norm2', 'relu')
                out = conv(out, k3d(P[pre + '.conv3.kernel']), 1, 1)
                x = bn(out, P, pre + '.norm3', 'relu', residual=res.F)
                continue
            if j == 0:
                res = bn(conv(x, P[pre + '.downsample.0.kernel'], 1, 2), P, pre + '.downsample.1')
                out = bn(conv(x, P[pre + '.conv1.kernel'], 3, 2), P, pre + '.norm1', 'relu')
            else:
                res = x
                out = bn(conv(x, P[pre + '.conv1.kernel'], 3, 1), P, pre + '.norm1', 'relu')
            out = conv(out, P[pre + '.conv2.kernel'], 3, 1)
            x = bn(out, P, pre + '.norm2', 'relu', residual=res.F)
        outs.append(x)
    return outs


class _HeadMM(torch.autograd.Function):
    """x (N, C) @ w (C, k) of the three 1x1 head convolutions (fcaf3d_neck_with_head.py:257-263).  The WEIGHT gradient x^T g
    sums mixed-sign products over every location of the batch (1e5-1e6 rows): torch's fp32 CPU GEMM leaves it 1.6e-4 of the
    tensor's scale away from the exact value on 2 x 30k points (r4, against the fp64 oracle with equal decisions: the HIP
    gradient sits at 7.5e-7) — so the oracle, the yardstick, accumulates this one reduction in fp64."""

    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        return x @ w

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        return g @ w.t(), (x.double().t() @ g.double()).to(w.dtype)


def neck_head(inputs, P, voxel_size, pts_threshold, n_reg_outs):
    """-> per level (fine..coarse): lists over scenes of centerness, bbox_pred, cls_score, points"""
    B = int(max(x.C[:, 0].max() for x in inputs)) + 1
    outs = []
    x = inputs[-1]
    scores = None
    for i in range(len(inputs) - 1, -1, -1):
        pre = 'neck_with_head.'
        if i < len(inputs) - 1:
            up = pre + f'up_block_{i + 1}'
            gc = mo.gen_conv_transpose_coords(x.C, x.stride)
            g = SP(gc, mo.gen_conv_transpose(x.F, P[up + '.0.kernel']), x.stride // 2)
            g = bn(g, P, up + '.1', 'elu')
            g = bn(conv(g, P[up + '.3.kernel'], 3, 1), P, up + '.4', 'elu')
            uc, uf = mo.union_add(inputs[i].C, inputs[i].F, g.C, g.F)
            x = SP(uc, uf, g.stride)
            if pts_threshold >= 0:
                with torch.no_grad():
                    sc = mo.features_at_coordinates(scores.C, scores.F, scores.stride, uc.astype(np.float32))[:, 0]
                    mask = np.zeros(len(uc), bool)
                    for b in range(B):
                        rows = np.nonzero(uc[:, 0] == b)[0]
                        k = min(len(rows), pts_threshold)
                        ids = torch.topk(sc[torch.from_numpy(rows)], k, sorted=False).indices.numpy()
                        mask[rows[ids]] = True
                    if TAPE is not None and (TAPE.recording or TAPE.ip < len(TAPE.prune)) and not mask.all():
                        mask = TAPE.prune_forced(uc, mask, f'prune level {i}')
                if not mask.all():
                    pc, pf = mo.prune(uc, uf, mask)
                    x = SP(pc, pf, g.stride)
        ob = pre + f'out_block_{i}'
        out = bn(conv(x, P[ob + '.0.kernel'], 3, 1), P, ob + '.1', 'elu')
        centerness = _HeadMM.apply(out.F, P[pre + 'centerness_conv.kernel'])
        cls = _HeadMM.apply(out.F, P[pre + 'cls_conv.kernel']) + P[pre + 'cls_conv.bias']
        reg = _HeadMM.apply(out.F, P[pre + 'reg_conv.kernel'])
        bbox = torch.cat([torch.exp(reg[:, :6] * P[pre + f'scales.{i}.scale']), reg[:, 6:]], 1)
        scores = SP(out.C, cls.detach().max(1, keepdim=True).values, out.stride)
        lv = [[], [], [], []]
        for b in range(B):
            rows = torch.from_numpy(np.nonzero(out.C[:, 0] == b)[0])
            lv[0].append(centerness[rows]); lv[1].append(bbox[rows]); lv[2].append(cls[rows])
            lv[3].append(torch.from_numpy(out.C[rows.numpy(), 1:].astype(np.float32)) * voxel_size)
        outs.append(lv)
    outs = outs[::-1]
    return [[o[k] for o in outs] for k in range(4)]      # [kind][level][scene]


def extract_feat(P, cfg, points):
    vs = cfg['voxel_size']
    nh = cfg['neck_with_head']
    c, f = mo.batch_sparse_collate([p[:, :3] / np.float32(vs) for p in points],
                                   [p[:, 3:] / np.float32(255.) for p in points])
    uc, uf = mo.sparse_tensor(c, f)
    x = SP(uc, torch.from_numpy(uf).to(P['backbone.conv1.0.kernel'].dtype), 1)   # fp64 params => fp64 oracle
    feats = backbone(x, P, cfg['backbone']['depth'], cfg['backbone'].get('n_outs', 4))
    return neck_head(feats, P, nh['voxel_size'], nh['pts_threshold'], nh['n_reg_outs'])


def forward_train(P, cfg, points, gt_boxes, gt_labels):
    """points: list of (n,6) numpy; gt_boxes: list of (m,7) numpy gravity-centre boxes; -> dict of 3 losses"""
    nh = cfg['neck_with_head']
    cent, bbox, cls, pts = extract_feat(P, cfg, points)
    with_yaw = nh.get('loss_bbox', {}).get('with_yaw', True)
    a = nh['assigner']
    lc, lb, ls = [], [], []
    for i in range(len(points)):
        p_lv = [lvl[i] for lvl in pts]
        with torch.no_grad():
            ct, bt, lab = lo.assign(p_lv, torch.from_numpy(gt_boxes[i]), torch.from_numpy(gt_labels[i]),
                                    a['limit'], a['topk'], a['n_scales'])
        centerness = torch.cat([lvl[i] for lvl in cent]); bp = torch.cat([lvl[i] for lvl in bbox])
        cs = torch.cat([lvl[i] for lvl in cls]); pp = torch.cat(p_lv)
        pos = torch.nonzero(lab >= 0).squeeze(1)
        n_pos = max(float(len(pos)), 1.0)
        ls.append(lo.sigmoid_focal_loss_sum(cs, lab) / n_pos)
        if len(pos) > 0:
            tgt_c = ct[pos]
            denorm = max(float(tgt_c.sum()), 1e-6)
            lc.append(lo.bce_with_logits(centerness[pos], tgt_c[:, None]).sum() / n_pos)
            dec = lo.bbox_pred_to_bbox(pp[pos], bp[pos], nh.get('yaw_parametrization', 'fcaf3d'))
            iou = lo.rotated_iou_3d(dec, bt[pos]) if with_yaw else lo.axis_aligned_iou(dec, bt[pos])
            lb.append(((1 - iou) * tgt_c).sum() / denorm)
        else:
            lc.append(centerness[pos].sum()); lb.append(bp[pos].sum())
    return dict(loss_centerness=torch.stack(lc).mean(), loss_bbox=torch.stack(lb).mean(),
                loss_cls=torch.stack(ls).mean())


def simple_test(P, cfg, points):
    """-> per scene (boxes (k,7|6) gravity-centre, scores (k,), labels (k,))  (get_bboxes + _nms)"""
    nh = cfg['neck_with_head']
    tc = cfg['test_cfg']
    with torch.no_grad():
        cent, bbox, cls, pts = extract_feat(P, cfg, points)
    res = []
    for i in range(len(points)):
        mb, ms = [], []
        for l in range(len(cent)):
            sc = torch.sigmoid(cls[l][i]) * torch.sigmoid(cent[l][i])
            bp, pt = bbox[l][i], pts[l][i]
            if len(sc) > tc['nms_pre'] > 0:
                ids = sc.max(1)[0].topk(tc['nms_pre'])[1]
                bp, sc, pt = bp[ids], sc[ids], pt[ids]
            mb.append(lo.bbox_pred_to_bbox(pt, bp, nh.get('yaw_parametrization', 'fcaf3d'))); ms.append(sc)
        boxes, scores = torch.cat(mb), torch.cat(ms)
        yaw = boxes.shape[1] == 7
        b7 = boxes if yaw else torch.cat([boxes, torch.zeros_like(boxes[:, :1])], 1)
        ob, os_, ol = [], [], []
        for c in range(scores.shape[1]):
            ids = scores[:, c] > tc['score_thr']
            if not ids.any():
                continue
            cb, csc = b7[ids], scores[ids, c]
            keep = torch.from_numpy(bev.nms(cb.numpy(), csc.numpy(), tc['iou_thr'], rotated=yaw))
            ob.append(cb[keep]); os_.append(csc[keep]); ol.append(torch.full((len(keep),), c, dtype=torch.long))
        if ob:
            res.append((torch.cat(ob), torch.cat(os_), torch.cat(ol)))
        else:
            res.append((boxes.new_zeros((0, 7)), boxes.new_zeros(0), torch.zeros(0, dtype=torch.long)))
    return res
