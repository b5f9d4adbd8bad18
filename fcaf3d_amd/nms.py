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
    sorted_scores, order = masked.sort(dim=1, descending=True)
    counts = (sorted_scores > score_thr).sum(dim=1).to(torch.int32)
    seg_boxes = boxes[order.reshape(-1)].reshape(C, n, 7).contiguous()
    keep, kcount = _run(seg_boxes, counts, iou_thr, rotated)
    ar = torch.arange(n, device=dev)[None, :]
    valid = ar < kcount[:, None]
    cls, pos = torch.nonzero(valid, as_tuple=True)              # class-major, ascending position
    idx = order[cls, keep[cls, pos].long()]
    return idx, cls
