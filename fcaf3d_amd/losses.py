"""Loss modules of the FCAF3D head with the reference's registry names and call signatures:
  * IoU3DLoss(with_yaw, reduction, loss_weight)              — mmdet3d/models/losses/iou3d_loss.py:38-75
  * FocalLoss(use_sigmoid, gamma, alpha, reduction, loss_weight)   — mmdet FocalLoss over mmcv's
    sigmoid_focal_loss op (defaults at fcaf3d_neck_with_head.py:29-34)
  * CrossEntropyLoss(use_sigmoid=True, ...)                   — mmdet (fcaf3d_neck_with_head.py:24-27)
The per-element math runs in csrc/loss.hip; reductions follow mmdet's `weight_reduce_loss`
(`sum / avg_factor` when an avg_factor is given with reduction='mean')."""
import torch
from torch import nn

from . import _lib as L
from .registry import LOSSES


def _reduce(loss, weight, reduction, avg_factor):
    if weight is not None:
        loss = loss * weight
    if avg_factor is None:
        if reduction == 'mean':
            return loss.mean()
        if reduction == 'sum':
            return loss.sum()
        return loss
    if reduction == 'mean':
        return loss.sum() / avg_factor
    if reduction == 'none':
        return loss
    raise ValueError('avg_factor can not be used with reduction="sum"')


class _FocalFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels, row_weight, gamma, alpha):
        if not logits.is_cuda:
            raise RuntimeError('sigmoid focal loss runs on the GPU only (HIP)')
        logits = logits.contiguous()
        labels = labels.to(torch.int64).contiguous()
        w = row_weight.to(torch.float32).contiguous() if row_weight is not None else None
        n, C = logits.shape
        rows = torch.empty(n, dtype=torch.float32, device=logits.device)
        L.call('fc_focal_loss_fwd', L.ptr(logits), L.ptr(labels), L.ptr(w), n, C, float(gamma), float(alpha), L.ptr(rows),
               L.stream())
        ctx.save_for_backward(logits, labels, w)
        ctx.cfg = (float(gamma), float(alpha))
        return rows.sum()

    @staticmethod
    def backward(ctx, g):
        logits, labels, w = ctx.saved_tensors
        gamma, alpha = ctx.cfg
        n, C = logits.shape
        gx = torch.empty_like(logits)
        gs = g.reshape(1).to(torch.float32).contiguous()
        L.call('fc_focal_loss_bwd', L.ptr(logits), L.ptr(labels), L.ptr(w), n, C, gamma, alpha, L.ptr(gs), L.ptr(gx),
               L.stream())
        return gx, None, None, None, None


def sigmoid_focal_loss_sum(logits, labels, gamma=2.0, alpha=0.25, row_weight=None):
    """Σ_n w_n Σ_c focal(logits[n,c]); labels (N,) in {-1, 0..C-1}, -1 = background."""
    return _FocalFn.apply(logits, labels, row_weight, gamma, alpha)


@LOSSES.register_module()
class FocalLoss(nn.Module):
    def __init__(self, use_sigmoid=True, gamma=2.0, alpha=0.25, reduction='mean', loss_weight=1.0):
        super().__init__()
        assert use_sigmoid is True, 'Only sigmoid focal loss supported now.'
        self.gamma, self.alpha, self.reduction, self.loss_weight = gamma, alpha, reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None):
        """weight: optional per-sample weight (N,), as mmdet's FocalLoss."""
        assert reduction_override in (None, 'mean', 'sum')
        reduction = reduction_override or self.reduction
        total = sigmoid_focal_loss_sum(pred, target, self.gamma, self.alpha, row_weight=weight)
        if reduction == 'mean':
            total = total / (avg_factor if avg_factor is not None else pred.numel())
        return self.loss_weight * total


@LOSSES.register_module()
class CrossEntropyLoss(nn.Module):
    def __init__(self, use_sigmoid=False, use_mask=False, reduction='mean', class_weight=None, loss_weight=1.0):
        super().__init__()
        assert use_sigmoid and not use_mask and class_weight is None, 'the FCAF3D head uses the sigmoid (BCE) form only'
        self.reduction, self.loss_weight = reduction, loss_weight

    def forward(self, cls_score, label, weight=None, avg_factor=None, reduction_override=None):
        reduction = reduction_override or self.reduction
        loss = torch.nn.functional.binary_cross_entropy_with_logits(cls_score, label.float(), reduction='none')
        return self.loss_weight * _reduce(loss, weight, reduction, avg_factor)


class _AlignedIoUFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target):
        if not pred.is_cuda:
            raise RuntimeError('3D IoU runs on the GPU only (HIP)')
        pred = pred.contiguous()
        target = target.contiguous()
        n = pred.shape[0]
        iou = torch.empty(n, dtype=torch.float32, device=pred.device)
        dpred = torch.empty((n, 6), dtype=torch.float32, device=pred.device)
        L.call('fc_aiou3d_fwd_bwd', L.ptr(pred), L.ptr(target), target.shape[1], n, 1e-6, L.ptr(iou), L.ptr(dpred),
               L.stream())
        ctx.save_for_backward(dpred)
        return iou

    @staticmethod
    def backward(ctx, g):
        dpred, = ctx.saved_tensors
        # rows that do not reach the loss (zero weight => g == 0) must give an exact zero even when their own
        # derivative is not finite (a background row whose predicted sizes overflow): 0 * NaN would poison every
        # gradient upstream
        g = g[:, None]
        return torch.where(g != 0, g * dpred, torch.zeros_like(dpred)), None


def axis_aligned_iou_3d(pred, target):
    """IoU of (n,6) boxes [cx,cy,cz,w,l,h] with targets (n,>=6) — iou3d_loss.py:21-35."""
    assert pred.shape[1] == 6 and target.shape[1] >= 6
    return _AlignedIoUFn.apply(pred, target)


class _RotatedIoUFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, weight):
        if not pred.is_cuda:
            raise RuntimeError('3D IoU runs on the GPU only (HIP)')
        pred = pred.contiguous()
        target = target.contiguous()
        weight = weight.contiguous() if weight is not None else None
        n = pred.shape[0]
        iou = torch.empty(n, dtype=torch.float32, device=pred.device)
        dpred = torch.empty((n, 7), dtype=torch.float32, device=pred.device)
        L.call('fc_riou3d_fwd_bwd', L.ptr(pred), L.ptr(target), L.ptr(weight), n, L.ptr(iou), L.ptr(dpred), L.stream())
        ctx.save_for_backward(dpred)
        return iou

    @staticmethod
    def backward(ctx, g):
        dpred, = ctx.saved_tensors
        g = g[:, None]
        return torch.where(g != 0, g * dpred, torch.zeros_like(dpred)), None, None


def rotated_iou_3d(pred, target, weight=None):
    """cal_iou_3d of (n,7) boxes [cx,cy,cz,w,l,h,yaw] — rotated_iou/oriented_iou_loss.py:86-109.
    Rows whose `weight` is <= 0 are skipped by the kernel (they cannot contribute to the loss)."""
    assert pred.shape[1] == 7 and target.shape[1] == 7
    return _RotatedIoUFn.apply(pred, target, weight)


def sort_v(vertices, mask, num_valid):
    """Drop-in for `cuda_op.cuda_ext.sort_v` of the Rotated_IoU extension (call site:
    rotated_iou/box_intersection_2d.py:147): vertices (B,N,24,2) f32 centred on the mean of the valid ones,
    mask (B,N,24) bool, num_valid (B,N) int -> (B,N,9) int32 indices of the intersection polygon in angular order."""
    if not vertices.is_cuda:
        raise RuntimeError('fcaf3d_amd ops run on the GPU only (HIP); got a CPU tensor')
    B, N = vertices.shape[:2]
    v = vertices.detach().float().contiguous()
    m = mask.to(torch.uint8).contiguous()
    nv = num_valid.to(torch.int32).contiguous()
    idx = torch.empty((B, N, 9), dtype=torch.int32, device=v.device)
    L.call('fc_sort_v', L.ptr(v), L.ptr(m), L.ptr(nv), B * N, L.ptr(idx), L.stream())
    return idx


@LOSSES.register_module()
class IoU3DLoss(nn.Module):
    """loss_weight * Σ w·(1 − IoU3D) / avg_factor  (iou3d_loss.py:38-75)."""

    def __init__(self, with_yaw=True, reduction='mean', loss_weight=1.0):
        super().__init__()
        self.with_yaw, self.reduction, self.loss_weight = with_yaw, reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None, **kwargs):
        assert reduction_override in (None, 'none', 'mean', 'sum')
        reduction = reduction_override or self.reduction
        if weight is not None and weight.dim() > 1:
            weight = weight.mean(-1)
        if pred.shape[0] == 0:
            return pred.sum() * 0.0
        iou = rotated_iou_3d(pred, target, weight) if self.with_yaw else axis_aligned_iou_3d(pred, target)
        loss = 1 - iou
        if weight is not None:
            # rows with zero weight contribute exactly zero (value and gradient), which also covers the
            # reference's early-out `if not torch.any(weight > 0): return pred.sum() * weight.sum()`
            loss = torch.where(weight > 0, loss, torch.zeros_like(loss))
        return self.loss_weight * _reduce(loss, weight, reduction, avg_factor)


class _FusedHeadLossFn(torch.autograd.Function):
    """The three losses of the FCAF3D head over all locations of the batch in two launches forward, one backward
    (csrc/loss.hip k_fcaf3d_loss_*; yaw-less heads).  Same values as FocalLoss + CrossEntropyLoss(use_sigmoid) + IoU3DLoss on
    `_bbox_pred_to_bbox` with per-row weights — kept as the cross-check (`Fcaf3DNeckWithHead.fused_loss = False`)."""

    @staticmethod
    def forward(ctx, bbox_pred, centerness, cls_score, points, ct, bt, labels, scene, inv_pos, inv_den, cfg):
        gamma, alpha, lw_cls, lw_cent, lw_bbox = cfg
        if not bbox_pred.is_cuda:
            raise RuntimeError('the fused head loss runs on the GPU only (HIP)')
        n, C = cls_score.shape
        dev = bbox_pred.device
        t = [x.contiguous() for x in (points, bbox_pred, centerness, cls_score, ct, bt, labels, scene, inv_pos, inv_den)]
        out = [torch.empty((), dtype=torch.float32, device=dev) for _ in range(3)]
        ws = L.workspace(L.query('fc_fcaf3d_loss_ws_bytes', n), dev)
        L.call('fc_fcaf3d_loss_fwd', *[L.ptr(x) for x in t], n, C, float(gamma), float(alpha), float(lw_cls), float(lw_cent),
               float(lw_bbox), L.ptr(out[0]), L.ptr(out[1]), L.ptr(out[2]), L.ptr(ws), ws.numel(), L.stream())
        ctx.save_for_backward(*t)
        ctx.cfg = cfg
        ctx.set_materialize_grads(False)
        return out[0], out[1], out[2]               # loss_cls, loss_centerness, loss_bbox

    @staticmethod
    def backward(ctx, g_cls, g_cent, g_bbox):
        t = ctx.saved_tensors
        gamma, alpha, lw_cls, lw_cent, lw_bbox = ctx.cfg
        n, C = t[3].shape
        dev = t[1].device
        gs = [g.reshape(1).to(torch.float32).contiguous() if g is not None else None for g in (g_cls, g_cent, g_bbox)]
        d_cls = torch.empty((n, C), dtype=torch.float32, device=dev)
        d_cent = torch.empty((n, 1), dtype=torch.float32, device=dev)
        d_bbox = torch.empty((n, 6), dtype=torch.float32, device=dev)
        L.call('fc_fcaf3d_loss_bwd', *[L.ptr(x) for x in t], n, C, float(gamma), float(alpha), float(lw_cls), float(lw_cent),
               float(lw_bbox), L.ptr(gs[0]), L.ptr(gs[1]), L.ptr(gs[2]), L.ptr(d_cls), L.ptr(d_cent), L.ptr(d_bbox), L.stream())
        return d_bbox, d_cent, d_cls, None, None, None, None, None, None, None, None


def fused_head_loss(bbox_pred, centerness, cls_score, points, ct, bt, labels, scene, inv_pos, inv_den, gamma, alpha,
                    lw_cls, lw_centerness, lw_bbox):
    """-> (loss_cls, loss_centerness, loss_bbox); bbox_pred (N,6), centerness (N,1), cls_score (N,C)."""
    return _FusedHeadLossFn.apply(bbox_pred, centerness, cls_score, points, ct, bt, labels.to(torch.int64),
                                  scene.to(torch.int32), inv_pos, inv_den, (gamma, alpha, lw_cls, lw_centerness, lw_bbox))
