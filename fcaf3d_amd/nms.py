"""BEV NMS front-end with the call semantics of mmdet3d/ops/pcdet_nms/pcdet_nms_utils.py:86-117
(`nms_gpu` / `nms_normal_gpu`: sort by score, suppress by BEV IoU, return indices into the input),
running on csrc/nms.hip.  `nms_bev_multiclass` does every class of a scene in one pair of launches."""
import torch

from . import _lib as L


def _run(boxes_seg, counts, thresh, rotated):
    """boxes_seg (nseg, stride, 7) sorted by score inside each segment; counts (nseg,) int32 (device)."""
    nseg, stride, _ = boxes_seg.shape
    dev = boxes_seg.device
    keep = torch.empty((nseg, stride), dtype=torch.int32, device=dev)
    kcount = torch.zeros(nseg, dtype=torch.int32, device=dev)
    ws = L.workspace(L.query('fc_nms_bev_ws_bytes', nseg, stride), dev)
    L.call('fc_nms_bev', L.ptr(boxes_seg), L.ptr(counts), nseg, stride, float(thresh), int(bool(rotated)),
           L.ptr(ws), ws.numel(), L.ptr(keep), L.ptr(kcount), L.stream())
    return keep, kcount


def nms_bev(boxes, scores, thresh, rotated=True):
    """boxes (N,7) [x,y,z,dx,dy,dz,heading], scores (N,) -> LongTensor of kept indices, by descending score."""
    assert boxes.shape[1] == 7
    if not boxes.is_cuda:
        raise RuntimeError('BEV NMS runs on the GPU only (HIP)')
    n = boxes.shape[0]
    if n == 0:
        return torch.zeros(0, dtype=torch.long, device=boxes.device)
    order = scores.sort(0, descending=True)[1]
    sorted_boxes = boxes[order].contiguous().unsqueeze(0)
    counts = torch.tensor([n], dtype=torch.int32, device=boxes.device)
    keep, kcount = _run(sorted_boxes, counts, thresh, rotated)
    k = int(kcount.item())
    return order[keep[0, :k].long()].contiguous()


def nms_bev_multiclass(boxes, scores, score_thr, iou_thr, rotated):
    """All classes at once.  boxes (N,7), scores (N,C).  Returns (box_index, class) of the survivors in
    the order the reference's per-class loop produces them (class-major, descending score)."""
    n, C = scores.shape
    dev = boxes.device
    masked = torch.where(scores > score_thr, scores, scores.new_full((1,), -1.0)).t().contiguous()   # (C,N)
    sorted_scores, order = masked.sort(dim=1, descending=True, stable=True)      # stable: ties keep candidate order (as the batched route)
    counts = (sorted_scores > score_thr).sum(dim=1).to(torch.int32)
    seg_boxes = boxes[order.reshape(-1)].reshape(C, n, 7).contiguous()
    keep, kcount = _run(seg_boxes, counts, iou_thr, rotated)
    ar = torch.arange(n, device=dev)[None, :]
    valid = ar < kcount[:, None]
    cls, pos = torch.nonzero(valid, as_tuple=True)              # class-major, ascending position
    idx = order[cls, keep[cls, pos].long()]
    return idx, cls


# ---- the reference's names (mmdet3d/ops/pcdet_nms/__init__.py: pcdet_nms_gpu / pcdet_nms_normal_gpu) ---------------
def pcdet_nms_gpu(boxes, scores, thresh, pre_maxsize=None, **kwargs):
    """pcdet_nms_utils.py:86-101: rotated BEV NMS; returns (indices into the input by descending score, None)."""
    assert boxes.shape[1] == 7
    if pre_maxsize is not None:
        order = scores.sort(0, descending=True)[1][:pre_maxsize]
        return order[nms_bev(boxes[order], scores[order], thresh, rotated=True)].contiguous(), None
    return nms_bev(boxes, scores, thresh, rotated=True), None


def pcdet_nms_normal_gpu(boxes, scores, thresh, **kwargs):
    """pcdet_nms_utils.py:104-117: axis-aligned BEV NMS (heading ignored)."""
    assert boxes.shape[1] == 7
    return nms_bev(boxes, scores, thresh, rotated=False), None


def boxes_iou_bev(boxes_a, boxes_b, rotated=True):
    """pcdet_nms_utils.py:28-41: (N,M) BEV IoU of (.,7) boxes [x,y,z,dx,dy,dz,heading]."""
    assert boxes_a.shape[1] == boxes_b.shape[1] == 7
    if not boxes_a.is_cuda:
        raise RuntimeError('BEV IoU runs on the GPU only (HIP)')
    a, b = boxes_a.float().contiguous(), boxes_b.float().contiguous()
    out = torch.zeros((a.shape[0], b.shape[0]), dtype=torch.float32, device=a.device)
    if a.shape[0] and b.shape[0]:
        L.call('fc_boxes_iou_bev', L.ptr(a), a.shape[0], L.ptr(b), b.shape[0], int(bool(rotated)), L.ptr(out), L.stream())
    return out


def boxes_iou3d_gpu(boxes_a, boxes_b):
    """pcdet_nms_utils.py:44-78: rotated BEV overlap x height overlap / 3D union.  The BEV overlap area is recovered
    from the IoU matrix (iou = ov / (Sa + Sb - ov)  =>  ov = iou (Sa + Sb) / (1 + iou))."""
    iou = boxes_iou_bev(boxes_a, boxes_b, rotated=True)
    sa = (boxes_a[:, 3] * boxes_a[:, 4]).view(-1, 1)
    sb = (boxes_b[:, 3] * boxes_b[:, 4]).view(1, -1)
    ov_bev = iou * (sa + sb) / (1 + iou)
    a_max = (boxes_a[:, 2] + boxes_a[:, 5] / 2).view(-1, 1); a_min = (boxes_a[:, 2] - boxes_a[:, 5] / 2).view(-1, 1)
    b_max = (boxes_b[:, 2] + boxes_b[:, 5] / 2).view(1, -1); b_min = (boxes_b[:, 2] - boxes_b[:, 5] / 2).view(1, -1)
    ov_h = torch.clamp(torch.min(a_max, b_max) - torch.max(a_min, b_min), min=0)
    ov3 = ov_bev * ov_h
    vol_a = (boxes_a[:, 3] * boxes_a[:, 4] * boxes_a[:, 5]).view(-1, 1)
    vol_b = (boxes_b[:, 3] * boxes_b[:, 4] * boxes_b[:, 5]).view(1, -1)
    return ov3 / torch.clamp(vol_a + vol_b - ov3, min=1e-6)
