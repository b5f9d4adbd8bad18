"""mAP / recall of indoor 3D detections — the step AFTER the hot path: what the reference's
`indoor_eval` (mmdet3d/core/evaluation/indoor_eval.py:55-309) reports for ScanNet / SUN RGB-D / S3DIS.

Same inputs, same result keys (`<cat>_AP_0.25`, `mAP_0.25`, `<cat>_rec_0.25`, `mAR_0.25`, ...), same matching rule
(detections in descending confidence; a detection is a true positive for a threshold iff its best-overlapping GT box of
the same class and scene exceeds the threshold and is not yet taken; VOC 'area' AP).  The pairwise 3D IoU of a scene is
one call of the HIP rotated-BEV kernel (`fcaf3d_amd.nms.boxes_iou3d_gpu`) instead of a per-box host loop; a different
`iou_fn(pred (n,7), gt (m,7)) -> (n,m)` on gravity-centre boxes can be passed (the CPU tests pass the oracle's).
"""
import numpy as np
import torch


def average_precision(recalls, precisions, mode='area'):
    """VOC AP (indoor_eval.py:7-52) of one curve (n,) or of several (num_scales, n) -> float32 (num_scales,).
    'area': area under the monotone precision envelope; '11points': mean of the best precision at recall >= 0, .1, ... 1.
    NB the reference divides by 11 INSIDE its loop over scales, so with S scales entry i ends up divided S - i times;
    kept, because its own test vector (tests/test_metrics/test_indoor_eval.py:183-188) encodes it."""
    r = np.atleast_2d(np.asarray(recalls, np.float64))
    p = np.atleast_2d(np.asarray(precisions, np.float64))
    assert r.shape == p.shape and r.ndim == 2
    S = r.shape[0]
    ap = np.zeros(S, np.float32)
    if mode == 'area':
        for i in range(S):
            mrec = np.concatenate(([0.0], r[i], [1.0]))
            mpre = np.concatenate(([0.0], p[i], [0.0]))
            mpre = np.maximum.accumulate(mpre[::-1])[::-1]            # envelope
            step = np.nonzero(mrec[1:] != mrec[:-1])[0]
            ap[i] = np.sum((mrec[step + 1] - mrec[step]) * mpre[step + 1])
    elif mode == '11points':
        for i in range(S):
            for thr in np.arange(0, 1 + 1e-3, 0.1):
                sel = p[i, r[i] >= thr]
                ap[i] += sel.max() if sel.size else 0.0
            ap /= 11
    else:
        raise ValueError('Unrecognized mode, only "area" and "11points" are supported')
    return ap


def _gravity7(boxes):
    """DepthInstance3DBoxes-like (bottom-centre .tensor) or (n,6|7) gravity-centre array -> (n,7) float32 gravity-centre"""
    if hasattr(boxes, 'tensor'):
        t = boxes.tensor.detach().float().cpu().clone()
        if t.dim() == 1:
            t = t[None]
        t[:, 2] = t[:, 2] + t[:, 5] * 0.5
        return t
    a = np.asarray(boxes.cpu() if hasattr(boxes, 'cpu') else boxes, np.float32)
    if a.size == 0:
        return torch.zeros((0, 7))
    t = torch.from_numpy(a.reshape(-1, a.shape[-1]).copy())
    if t.shape[1] == 6:
        t = torch.cat((t, t.new_zeros(t.shape[0], 1)), 1)
    return t


def _default_iou(pred, gt):
    if not torch.cuda.is_available():
        raise RuntimeError('indoor_eval computes its IoU matrices on the GPU (HIP); pass iou_fn= for a CPU evaluation')
    from .nms import boxes_iou3d_gpu
    dev = torch.device('cuda', torch.cuda.current_device())
    return boxes_iou3d_gpu(pred.to(dev), gt.to(dev)).cpu().numpy()


def eval_det_cls(pred, gt, iou_thr, iou_fn=None):
    """One class.  pred: {scene: (boxes (n,7) gravity-centre, scores (n,))}, gt: {scene: boxes (m,7)}.
    -> [(recall, precision, ap)] per threshold (indoor_eval.py:55-160)."""
    iou_fn = iou_fn or _default_iou
    npos = sum(len(b) for b in gt.values())
    scene_of, conf, best_iou, best_j = [], [], [], []
    for sid, (boxes, scores) in pred.items():
        n = len(boxes)
        if n == 0:
            continue
        g = gt.get(sid)
        if g is not None and len(g) > 0:
            iou = np.asarray(iou_fn(boxes, g), np.float64)
            j = iou.argmax(1)                                  # first maximum, as the reference's strict '>' scan
            bi = iou[np.arange(n), j]
        else:
            j = np.zeros(n, np.int64)
            bi = np.full(n, -np.inf) if g is None or len(g) == 0 else np.zeros(n)
        scene_of += [sid] * n
        conf.append(np.asarray(scores, np.float64).reshape(-1))
        best_iou.append(bi)
        best_j.append(j)
    if not scene_of:
        conf_all = np.zeros(0); best_iou = np.zeros(0); best_j = np.zeros(0, np.int64)
    else:
        conf_all = np.concatenate(conf); best_iou = np.concatenate(best_iou); best_j = np.concatenate(best_j)
    order = np.argsort(-conf_all)
    out = []
    for thr in iou_thr:
        taken = {sid: np.zeros(len(b), bool) for sid, b in gt.items()}
        tp = np.zeros(len(order)); fp = np.zeros(len(order))
        for d, i in enumerate(order):
            sid = scene_of[i]
            if best_iou[i] > thr and not taken[sid][best_j[i]]:
                tp[d] = 1.0
                taken[sid][best_j[i]] = True
            else:
                fp[d] = 1.0
        ctp, cfp = np.cumsum(tp), np.cumsum(fp)
        # (a class that is detected but has no GT box anywhere gives 0/0 = nan in the reference; 0 here)
        recall = ctp / float(npos) if npos else ctp * 0.0
        precision = ctp / np.maximum(ctp + cfp, np.finfo(np.float64).eps)
        out.append((recall, precision, average_precision(recall, precision)))
    return out


def indoor_eval(gt_annos, dt_annos, metric, label2cat, logger=None, box_type_3d=None, box_mode_3d=None, iou_fn=None):
    """indoor_eval.py:205-309.  gt_annos[i]: dict(gt_num, gt_boxes_upright_depth (m,6|7) gravity-centre, class (m,));
    dt_annos[i]: dict(boxes_3d (box object with bottom-centre .tensor, or (n,7) gravity-centre), scores_3d, labels_3d).
    box_type_3d / box_mode_3d are accepted for signature compatibility (boxes are Depth-mode already)."""
    assert len(dt_annos) == len(gt_annos)
    pred, gt = {}, {}
    for sid, (g, d) in enumerate(zip(gt_annos, dt_annos)):
        labels = np.asarray(d['labels_3d'].cpu() if hasattr(d['labels_3d'], 'cpu') else d['labels_3d']).astype(np.int64)
        scores = np.asarray(d['scores_3d'].cpu() if hasattr(d['scores_3d'], 'cpu') else d['scores_3d'], np.float64)
        boxes = _gravity7(d['boxes_3d'])
        for c in np.unique(labels):
            m = labels == c
            pred.setdefault(int(c), {})[sid] = (boxes[torch.from_numpy(m)], scores[m])
            gt.setdefault(int(c), {}).setdefault(sid, torch.zeros((0, 7)))     # the reference registers the class in gt too
        if g['gt_num'] != 0:
            gb = _gravity7(g['gt_boxes_upright_depth'])
            gl = np.asarray(g['class']).astype(np.int64)
            for c in np.unique(gl):
                gt.setdefault(int(c), {})[sid] = gb[torch.from_numpy(gl == c)]
    rec, prec, ap = [{} for _ in metric], [{} for _ in metric], [{} for _ in metric]
    for c in gt:
        if c in pred:
            res = eval_det_cls(pred[c], gt[c], metric, iou_fn)
        for i in range(len(metric)):
            if c in pred:
                rec[i][c], prec[i][c], ap[i][c] = res[i]
            else:
                rec[i][c] = prec[i][c] = ap[i][c] = np.zeros(1)
    ret = {}
    lines = []
    for i, thr in enumerate(metric):
        for c in ap[i]:
            ret[f'{label2cat[c]}_AP_{thr:.2f}'] = float(ap[i][c][0])
        ret[f'mAP_{thr:.2f}'] = float(np.mean([v[0] for v in ap[i].values()])) if ap[i] else float('nan')
        recs = []
        for c in rec[i]:
            r = float(rec[i][c][-1]) if len(rec[i][c]) else 0.0
            ret[f'{label2cat[c]}_rec_{thr:.2f}'] = r
            recs.append(r)
        ret[f'mAR_{thr:.2f}'] = float(np.mean(recs)) if recs else float('nan')
        lines.append(f'mAP_{thr:.2f} {ret[f"mAP_{thr:.2f}"]:.4f}  mAR_{thr:.2f} {ret[f"mAR_{thr:.2f}"]:.4f}')
    if logger is not None:
        (logger.info if hasattr(logger, 'info') else print)('\n'.join(lines))
    return ret
