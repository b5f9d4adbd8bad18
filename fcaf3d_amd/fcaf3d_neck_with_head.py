"""FCAF3D neck + anchor-free head, loss, target assignment and box decoding/NMS driver.

Host-side mirror of mmdet3d/models/dense_heads/fcaf3d_neck_with_head.py (class names, constructor
kwargs, method names, return structures and parameter names are the reference's; the body runs on
the HIP operators of this package).  Differences that do not change results:
  * norm+activation pairs and the three 1x1 head convolutions are fused into single kernel passes;
  * the loss is evaluated with positive-MASKS instead of `nonzero` index lists (no host syncs), and
    the two per-scene `reduce_mean` scalars (reference :179, :187) of ALL scenes of the batch travel in
    ONE all-reduce — numerically identical.
"""
import math
import os

import numpy as np
import torch
from torch import nn

from . import _lib as L
from . import functional as Fn
from . import nn as MEnn
from .dist import reduce_mean
from .nms import _run as _nms_run
from .nms import nms_bev, nms_bev_multiclass
from .registry import BBOX_ASSIGNERS, HEADS, build_assigner, build_loss
from .sparse import SparseTensor


class SceneList:
    """What the reference returns as a python list of per-scene tensors (`x[permutation]` for every
    decomposition permutation, fcaf3d_neck_with_head.py:266-275), kept as ONE (N,C) tensor on its coordinate
    map; the per-scene tensors are materialised only when indexed.  `loss()` consumes `.full` directly."""

    def __init__(self, full, cmap, parent=None):
        """full: the (N, C) tensor, or a callable that makes it on first use; parent: the all-levels tensor this level is a row
        range of (executor path: the loss then reads the parent instead of concatenating the levels)"""
        self._full, self.cmap, self.parent = full, cmap, parent
        self._items = {}

    @property
    def full(self):
        if callable(self._full):
            self._full = self._full()
        return self._full

    def __len__(self):
        return self.cmap.batch_size

    def __getitem__(self, i):
        if i < 0:
            i += len(self)
        if not 0 <= i < len(self):
            raise IndexError(i)
        if i not in self._items:
            self._items[i] = self.full[self.cmap.decomposition_permutations[i]]
        return self._items[i]

    def __iter__(self):
        return (self[i] for i in range(len(self)))


class Scale(nn.Module):
    """mmcv.cnn.Scale"""

    def __init__(self, scale=1.0):
        super().__init__()
        self.scale = nn.Parameter(torch.tensor(scale, dtype=torch.float))

    def forward(self, x):
        return x * self.scale


def bias_init_with_prob(prior_prob):
    return float(-math.log((1 - prior_prob) / prior_prob))


@HEADS.register_module()
class Fcaf3DNeckWithHead(nn.Module):
    def __init__(self,
                 n_classes,
                 in_channels,
                 out_channels,
                 n_reg_outs,
                 voxel_size,
                 pts_threshold,
                 assigner,
                 yaw_parametrization='fcaf3d',
                 loss_centerness=dict(type='CrossEntropyLoss', use_sigmoid=True, loss_weight=1.0),
                 loss_bbox=dict(type='IoU3DLoss', loss_weight=1.0),
                 loss_cls=dict(type='FocalLoss', use_sigmoid=True, gamma=2.0, alpha=0.25, loss_weight=1.0),
                 train_cfg=None,
                 test_cfg=None):
        super().__init__()
        self.voxel_size = voxel_size
        self.yaw_parametrization = yaw_parametrization
        self.assigner = build_assigner(assigner)
        self.loss_centerness = build_loss(loss_centerness)
        self.loss_bbox = build_loss(loss_bbox)
        self.loss_cls = build_loss(loss_cls)
        self.train_cfg = train_cfg
        self.test_cfg = test_cfg
        self.pts_threshold = pts_threshold
        self.n_classes, self.n_reg_outs = n_classes, n_reg_outs
        self._init_layers(in_channels, out_channels, n_reg_outs, n_classes)

    # ---- module graph (reference :49-86; parameter names per SURVEY.md Appendix B) -----------------
    @staticmethod
    def _make_block(in_channels, out_channels):
        return nn.Sequential(
            MEnn.MinkowskiConvolution(in_channels, out_channels, kernel_size=3, dimension=3),
            MEnn.MinkowskiBatchNorm(out_channels),
            MEnn.MinkowskiELU())

    @staticmethod
    def _make_up_block(in_channels, out_channels):
        return nn.Sequential(
            MEnn.MinkowskiGenerativeConvolutionTranspose(in_channels, out_channels, kernel_size=2, stride=2, dimension=3),
            MEnn.MinkowskiBatchNorm(out_channels),
            MEnn.MinkowskiELU(),
            MEnn.MinkowskiConvolution(out_channels, out_channels, kernel_size=3, dimension=3),
            MEnn.MinkowskiBatchNorm(out_channels),
            MEnn.MinkowskiELU())

    def _init_layers(self, in_channels, out_channels, n_reg_outs, n_classes):
        self.pruning = MEnn.MinkowskiPruning()
        for i in range(len(in_channels)):
            if i > 0:
                self.__setattr__(f'up_block_{i}', self._make_up_block(in_channels[i], in_channels[i - 1]))
            self.__setattr__(f'out_block_{i}', self._make_block(in_channels[i], out_channels))
        self.centerness_conv = MEnn.MinkowskiConvolution(out_channels, 1, kernel_size=1, dimension=3)
        self.reg_conv = MEnn.MinkowskiConvolution(out_channels, n_reg_outs, kernel_size=1, dimension=3)
        self.cls_conv = MEnn.MinkowskiConvolution(out_channels, n_classes, kernel_size=1, bias=True, dimension=3)
        self.scales = nn.ModuleList([Scale(1.) for _ in range(len(in_channels))])

    def init_weights(self):
        nn.init.normal_(self.centerness_conv.kernel, std=.01)
        nn.init.normal_(self.reg_conv.kernel, std=.01)
        nn.init.normal_(self.cls_conv.kernel, std=.01)
        nn.init.constant_(self.cls_conv.bias, bias_init_with_prob(.01))

    # ---- forward (reference :94-108) ---------------------------------------------------------------
    # out_block_i + forward_single(i) of every level but the finest depend only on that level's neck tensor, while the
    # neck itself goes on to the next finer level: they run on a second HIP stream (r3), so that the ramp / tail of one
    # branch's launches is filled by the other's.  autograd replays each node on the stream of its forward, so the
    # backward pass overlaps the same way.  Results are identical (same kernels, same inputs).
    head_overlap = os.environ.get('FC_HEAD_OVERLAP', '1') != '0'
    _head_streams = {}

    def _head_stream(self, device):
        idx = device.index if device.index is not None else torch.cuda.current_device()
        if idx not in self._head_streams:
            self._head_streams[idx] = torch.cuda.Stream(device=idx)
        return self._head_streams[idx]

    def forward(self, x):
        outs = []
        inputs = x
        x = inputs[-1]
        scores = None
        dev = x.F.device
        side = self._head_stream(dev) if (self.head_overlap and dev.type == 'cuda' and len(inputs) > 1) else None
        main = torch.cuda.current_stream(dev) if side is not None else None
        self._side_busy = False
        wk = self._packed_head_kernel()
        for i in range(len(inputs) - 1, -1, -1):
            if i < len(inputs) - 1:
                x = MEnn.run_sequential(getattr(self, f'up_block_{i + 1}'), x)
                x = inputs[i] + x
                x = self._prune(x, scores, (main, side))
            if side is not None and i > 0:
                # Every table the two branches share is built HERE, on the main stream, before the fork: this level's k3
                # kernel map (out_block_i reads it on the head stream; the next up_block derives the generated set's map
                # from it on the main stream, sparse.CoordMap.kernel_map) and its derived tables (backward replays each
                # branch on the stream of its forward).  Planned steps find them prefetched (a cache hit); with pruning
                # live nothing was, and a table first built under the head stream would be read by the main stream
                # without any ordering.
                x.cmap.kernel_map(x.cmap, 3).prefetch(self.training and torch.is_grad_enabled())
                side.wait_stream(main)                       # x is complete on the main stream
                x.F.record_stream(side)
                with torch.cuda.stream(side):
                    out = MEnn.run_sequential(getattr(self, f'out_block_{i}'), x)
                    out = self.forward_single(out, self.scales[i], wk)
                self._side_busy = True
            else:
                out = MEnn.run_sequential(getattr(self, f'out_block_{i}'), x)
                out = self.forward_single(out, self.scales[i], wk)
            scores = out[-1]
            outs.append(out[:-1])
        if side is not None and self._side_busy:
            main.wait_stream(side)
            for o in outs[:-1]:
                for v in o:
                    v.full.record_stream(main)               # allocated under the side stream, consumed by the loss on main
            self._side_busy = False
        return zip(*outs[::-1])

    def _prune(self, x, scores, streams=(None, None)):
        """Keep, per scene, the pts_threshold voxels with the largest interpolated parent score (:110-126)."""
        if self.pts_threshold < 0:
            return x
        perms = x.decomposition_permutations
        if all(len(p) <= self.pts_threshold for p in perms):
            return x                        # top-k of everything keeps everything
        main, side = streams
        if side is not None and getattr(self, '_side_busy', False):
            main.wait_stream(side)          # the parent level's scores come from the head branch
            scores.F.record_stream(main)
        with torch.no_grad():
            interpolated = scores.features_at_coordinates(x.C).squeeze(1)
            mask = torch.zeros(len(interpolated), dtype=torch.bool, device=interpolated.device)
            kept = 0
            for perm in perms:
                k = min(len(perm), self.pts_threshold)
                ids = torch.topk(interpolated[perm], k, sorted=False).indices
                # (NOT `mask[perm[ids]] = True`: indexing-assignment of a Python scalar makes a CPU tensor of it and copies it to
                # the device — a pageable host -> device copy, which blocks the host until everything enqueued before it on the
                # stream has run: r5 host profile, 6.3 ms per call, the whole network body.  index_fill_ takes the scalar as a
                # kernel argument)
                mask.index_fill_(0, perm[ids], True)
                kept += k
        # the number of kept rows is known on the host (top-k indices of a scene are distinct): MinkowskiPruning does not read the
        # count back, so the host does not wait for the network body that produced the scores (r4: the one synchronisation that
        # kept BASELINE config 5 and the 1 cm configuration host-bound, 18 ms of enqueue per step)
        return self.pruning(x, mask, expect_n=kept)

    def _packed_head_kernel(self):
        """[centerness | reg | cls] kernels side by side, zero-padded to a multiple of 64 columns (the MFMA tile)."""
        ws = [self.centerness_conv.kernel, self.reg_conv.kernel, self.cls_conv.kernel]
        pad = (-sum(w.shape[1] for w in ws)) % 64
        if pad:
            ws.append(ws[0].new_zeros(ws[0].shape[0], pad))
        return torch.cat(ws, dim=1).unsqueeze(0)

    def forward_single(self, x, scale, packed_kernel=None):
        """Three 1x1 convs as ONE 128 -> (1 + n_reg + n_cls) GEMM, padded to a multiple of 64 columns for
        the MFMA tile (reference :256-279).  `packed_kernel`: the packed operator, built once per forward pass by
        `forward` for all levels (r3: it used to be rebuilt per level — 3 launches forward and 2 backward each)."""
        n_c, n_r = self.n_classes, self.n_reg_outs
        w = (packed_kernel if packed_kernel is not None else self._packed_head_kernel())[0]
        y = Fn.sparse_conv(x.F, w.unsqueeze(0), None, x.F.shape[0])
        if w.shape[1] <= 64:
            # centerness | exp(scale * reg[:, :6]), reg[:, 6:] | cls + bias, and the max class logit `_prune` interpolates:
            # one fused pass forward, one backward (csrc/head.hip)
            centerness, bbox_pred, cls_score, cls_max = Fn.head_split(y, self.cls_conv.bias, scale.scale, n_r, n_c)
        else:
            # more than 64 fused columns (> 55 classes, e.g. ScanNet200): the fused epilogue kernel holds one 64-column row
            # per wave, so split with plain tensor ops — same values, the reference's own formulation (:264-270)
            centerness = y[:, :1]
            reg = y[:, 1:1 + n_r]
            bbox_pred = torch.cat((torch.exp(scale.scale * reg[:, :6]), reg[:, 6:]), dim=1)
            cls_score = y[:, 1 + n_r:1 + n_r + n_c] + self.cls_conv.bias
            cls_max = cls_score.detach().max(dim=1, keepdim=True).values
        prune_scores = SparseTensor(cls_max, coordinate_map_key=x.cmap)

        points = x.C[:, 1:].float() * self.voxel_size          # voxel corner, as the reference (:276-277)
        cm = x.cmap
        return (SceneList(centerness, cm), SceneList(bbox_pred, cm), SceneList(cls_score, cm), SceneList(points, cm),
                prune_scores)

    # ---- loss (reference :128-203) -----------------------------------------------------------------
    def loss(self, centernesses, bbox_preds, cls_scores, points, gt_bboxes, gt_labels, img_metas):
        assert len(centernesses[0]) == len(bbox_preds[0]) == len(cls_scores[0]) \
            == len(points[0]) == len(img_metas) == len(gt_bboxes) == len(gt_labels)
        if all(isinstance(v, SceneList) for group in (centernesses, bbox_preds, cls_scores, points) for v in group) \
                and hasattr(self.assigner, 'assign_batched'):
            return self._loss_batched(centernesses, bbox_preds, cls_scores, points, gt_bboxes, gt_labels)
        n_img = len(img_metas)
        per_img = []
        for i in range(n_img):
            pts = [x[i] for x in points]
            with torch.no_grad():
                centerness_targets, bbox_targets, labels = self.assigner.assign(pts, gt_bboxes[i], gt_labels[i])
                pos = labels >= 0
                centerness_targets = torch.where(pos, centerness_targets, torch.zeros_like(centerness_targets))
            per_img.append(dict(
                centerness=torch.cat([x[i] for x in centernesses]),
                bbox_preds=torch.cat([x[i] for x in bbox_preds]),
                cls_scores=torch.cat([x[i] for x in cls_scores]),
                points=torch.cat(pts), pos=pos, labels=labels,
                centerness_targets=centerness_targets, bbox_targets=bbox_targets))
        # one all-reduce for the 2*n_img normalisers instead of 2 per scene
        with torch.no_grad():
            norms = torch.stack([torch.stack((d['pos'].sum().float(), d['centerness_targets'].sum()))
                                 for d in per_img])
            norms = reduce_mean(norms)
        loss_centerness, loss_bbox, loss_cls = [], [], []
        for i, d in enumerate(per_img):
            lc, lb, ls = self._loss_single(d, norms[i, 0].clamp(min=1.), norms[i, 1].clamp(min=1e-6))
            loss_centerness.append(lc)
            loss_bbox.append(lb)
            loss_cls.append(ls)
        return dict(
            loss_centerness=torch.mean(torch.stack(loss_centerness)),
            loss_bbox=torch.mean(torch.stack(loss_bbox)),
            loss_cls=torch.mean(torch.stack(loss_cls)))

    early_targets = os.environ.get('FC_EARLY_TARGETS', '1') != '0'

    def _targets(self, cmaps, gt_bboxes, gt_labels, pre=None):
        """Locations, assigned targets and per-scene normalisers of one batch — everything the loss needs that does not
        depend on the network's outputs: locations = voxel corners of the head's coordinate sets (:276-277), targets from
        the assigner, normalisers n_pos / sum of centerness targets per scene, averaged over the ranks in ONE all-reduce
        (the reference: two `reduce_mean` per scene, :179, :187)."""
        from .sparse import _rec
        B = len(gt_bboxes)
        dev = pre['pts'].device if pre is not None else cmaps[0].coords.device
        with torch.no_grad():
            if pre is not None:
                pts, scene, level = pre['pts'], pre['scene'], pre['level']      # written by the native plan (csrc/plan.hip k_plan_head_arrays)
            else:
                pts = torch.cat([cm.coords[:, 1:].float() * self.voxel_size for cm in cmaps])
                scene = torch.cat([cm.coords[:, 0] for cm in cmaps])
                level = torch.cat([torch.full((cm.n,), l, dtype=torch.int32, device=dev) for l, cm in enumerate(cmaps)])
            ct, bt, labels = self.assigner.assign_batched(pts, scene, level, cmaps, gt_bboxes, gt_labels, pre=pre)
            posf = (labels >= 0).float()
            cols = torch.stack((posf, ct, torch.zeros_like(ct), torch.zeros_like(ct)), dim=1)
            norms = reduce_mean(Fn.seg_col_sums(cols, scene, B)[:, :2])             # (B,2): n_pos, Σ centerness
            inv_pos = 1.0 / (B * norms[:, 0].clamp(min=1.))
            inv_den = 1.0 / (B * norms[:, 1].clamp(min=1e-6))
        out = dict(pts=pts, scene=scene, ct=ct, bt=bt, labels=labels, posf=posf, inv_pos=inv_pos, inv_den=inv_den)
        _rec(*out.values())                                          # built on the coordinate stream, consumed on the main one
        return out

    def prepare_targets(self, cmaps, gt_bboxes, gt_labels, pre=None):
        """Called by the detector as soon as the head's coordinate sets exist (SingleStageSparse3DDetector._sparse_input, on
        the coordinate stream): `loss()` then finds the assignment done.  Identity of the coordinate-map objects is the key."""
        self._prepared = None
        if self.early_targets and hasattr(self.assigner, 'assign_batched') and len(cmaps) == self.assigner.n_scales \
                and (pre is not None or cmaps[0].coords.is_cuda):
            self._prepared = (tuple(id(cm) for cm in cmaps), cmaps, self._targets(cmaps, gt_bboxes, gt_labels, pre))

    def _loss_batched(self, centernesses, bbox_preds, cls_scores, points, gt_bboxes, gt_labels):
        """The same three losses as the per-scene loop (reference :140-157, :160-203), evaluated over all
        locations of all scenes at once: every row carries the weight 1/(B * normaliser of its scene), so
        the weighted sums equal the mean over scenes of the per-scene losses."""
        B = len(gt_bboxes)
        cmaps = [p.cmap for p in points]
        prep, self._prepared = getattr(self, '_prepared', None), None
        if prep is not None and prep[0] == tuple(id(cm) for cm in cmaps):
            tg = prep[2]
        else:
            tg = self._targets(cmaps, gt_bboxes, gt_labels)
        pts, scene, ct, bt, labels, posf, inv_pos, inv_den = (tg[k] for k in ('pts', 'scene', 'ct', 'bt', 'labels', 'posf',
                                                                               'inv_pos', 'inv_den'))
        def cat(lists):
            par = lists[0].parent
            if par is not None and all(v.parent is par for v in lists) and par.shape[0] == sum(v.cmap.n for v in lists):
                return par                                  # the levels ARE consecutive row ranges of one tensor (executor.py)
            return torch.cat([v.full for v in lists])
        centerness, bbox_pred, cls_score = cat(centernesses), cat(bbox_preds), cat(cls_scores)
        if self._fusable_loss(bbox_pred):
            # yaw-less head with the reference's default loss types: focal + centerness BCE + decode + axis-aligned IoU and
            # their weighted sums in 2 launches (1 backward) instead of ~40 (~40) tiny ones on the step's serial spine
            from .losses import fused_head_loss
            lc, lce, lb = fused_head_loss(bbox_pred, centerness, cls_score, pts, ct, bt, labels, scene, inv_pos, inv_den,
                                          self.loss_cls.gamma, self.loss_cls.alpha, self.loss_cls.loss_weight,
                                          self.loss_centerness.loss_weight, self.loss_bbox.loss_weight)
            return dict(loss_centerness=lce, loss_bbox=lb, loss_cls=lc)
        sl = scene.long()
        w_pos, w_den = inv_pos[sl], inv_den[sl]
        loss_cls = self.loss_cls(cls_score, labels, weight=w_pos, avg_factor=1.0)
        loss_centerness = self.loss_centerness(centerness, ct.unsqueeze(1), weight=(posf * w_pos).unsqueeze(1),
                                               avg_factor=1.0)
        loss_bbox = self.loss_bbox(self._bbox_pred_to_bbox(pts, bbox_pred), bt, weight=ct * w_den, avg_factor=1.0)
        return dict(loss_centerness=loss_centerness, loss_bbox=loss_bbox, loss_cls=loss_cls)

    fused_loss = os.environ.get('FC_FUSED_LOSS', '1') != '0'      # False: the three loss modules one after the other (cross-check)

    def _fusable_loss(self, bbox_pred):
        from .losses import CrossEntropyLoss, FocalLoss, IoU3DLoss
        return (self.fused_loss and bbox_pred.is_cuda and bbox_pred.shape[1] == 6
                and type(self.loss_cls) is FocalLoss and self.loss_cls.reduction == 'mean'
                and type(self.loss_centerness) is CrossEntropyLoss and self.loss_centerness.reduction == 'mean'
                and type(self.loss_bbox) is IoU3DLoss and not self.loss_bbox.with_yaw and self.loss_bbox.reduction == 'mean')

    def _loss_single(self, d, n_pos, centerness_denorm):
        posf = d['pos'].float()
        loss_cls = self.loss_cls(d['cls_scores'], d['labels'], avg_factor=n_pos)
        loss_centerness = self.loss_centerness(d['centerness'], d['centerness_targets'].unsqueeze(1),
                                               weight=posf.unsqueeze(1), avg_factor=n_pos)
        loss_bbox = self.loss_bbox(self._bbox_pred_to_bbox(d['points'], d['bbox_preds']), d['bbox_targets'],
                                   weight=d['centerness_targets'], avg_factor=centerness_denorm)
        return loss_centerness, loss_bbox, loss_cls

    # ---- inference (reference :205-253, :332-374) --------------------------------------------------
    def get_bboxes(self, centernesses, bbox_preds, cls_scores, points, img_metas, rescale=False, defer=False):
        """defer=True (SingleStageSparse3DDetector.simple_test_async): everything up to the NMS is ENQUEUED and a callable is
        returned that performs the read-backs and builds the per-scene results when called — the caller may enqueue the next
        batch first (two batches in flight: the next one's coordinate phase runs on the host while this one's forward pass and
        decode run on the GPU)"""
        assert len(centernesses[0]) == len(bbox_preds[0]) == len(cls_scores[0]) == len(points[0]) == len(img_metas)
        if self.batched_decode and self.test_cfg.nms_pre > 0 and all(
                isinstance(v, SceneList) for group in (centernesses, bbox_preds, cls_scores, points) for v in group):
            return self._get_bboxes_batched(centernesses, bbox_preds, cls_scores, points, img_metas, defer=defer)
        if defer:
            res = self.get_bboxes(centernesses, bbox_preds, cls_scores, points, img_metas, rescale)
            return lambda post=None: post(res) if post is not None else res
        results = []
        for i in range(len(img_metas)):
            results.append(self._get_bboxes_single(
                centernesses=[x[i] for x in centernesses], bbox_preds=[x[i] for x in bbox_preds],
                cls_scores=[x[i] for x in cls_scores], points=[x[i] for x in points], img_meta=img_metas[i]))
        return results

    batched_decode = True      # False: the reference's per-scene loop (_get_bboxes_single), kept as the cross-check
    NMS_WS_BUDGET = int(os.environ.get('FC_NMS_WS_MB', '512')) << 20      # bytes of suppression masks per NMS launch

    def _get_bboxes_batched(self, centernesses, bbox_preds, cls_scores, points, img_metas, defer=False):
        """get_bboxes for ALL scenes at once (r2: the per-scene, per-level loop of the reference issued ~1 500 tiny launches
        and 8 read-backs per batch of 8 scenes, 8.9 ms): scores, decode and the per-(scene, level) top-`nms_pre` selection run
        on the whole batch (one segmented sort), then every (scene, class) segment goes through ONE pair of NMS launches and
        one read-back of the survivor counts.  Candidates reach the NMS in the order of the per-scene loop (scene, level,
        then descending max score where top-k bites, row order where it does not), all sorts are stable: same results."""
        B, Lv = len(img_metas), len(centernesses)
        cfg = self.test_cfg
        dev = centernesses[0].full.device
        sc_l, mx_l, bx_l, sg_l = [], [], [], []
        for l in range(Lv):
            cm = centernesses[l].cmap
            sc = cls_scores[l].full.sigmoid() * centernesses[l].full.sigmoid()
            sc_l.append(sc)
            mx_l.append(sc.max(dim=1).values)
            bx_l.append(self._bbox_pred_to_bbox(points[l].full, bbox_preds[l].full))
            sg_l.append(cm.coords[:, 0].to(torch.int64) * Lv + l)
        scores, maxs, boxes, seg = torch.cat(sc_l), torch.cat(mx_l), torch.cat(bx_l), torch.cat(sg_l)
        N, C = scores.shape
        yaw_flag = boxes.shape[1] == 7
        boxes7 = boxes if yaw_flag else torch.cat((boxes, torch.zeros_like(boxes[:, :1])), dim=1)
        # the rows the per-scene loop keeps, gathered into the places it would put them — WITHOUT compacting them first (r4:
        # `order[keep]` is a boolean-mask gather, i.e. a device -> host synchronisation in the middle of the decode; everything
        # here is fixed-size, so the whole decode is enqueued while the forward pass still runs).  Candidate slot j of scene b
        # belongs to level l = #levels whose kept rows end at or before j, rank r inside it: source = order[start(b, l) + r]
        order, _, kept_counts = segmented_topk(seg, maxs, B * Lv, cfg.nms_pre, compact=False)
        n_max = min(Lv * cfg.nms_pre, N)                    # static bound: no read-back
        kc = kept_counts.view(B, Lv)
        ends = torch.cumsum(kc, 1)                          # (B, Lv): where each level's kept rows end in the scene's list
        counts = count_ids(seg, B * Lv)
        starts = (torch.cumsum(counts, 0) - counts).view(B, Lv)
        j = torch.arange(n_max, device=dev)[None, :].expand(B, n_max)
        lvl = (j[:, :, None] >= ends[:, None, :]).sum(-1).clamp(max=Lv - 1)
        rank = j - (ends - kc).gather(1, lvl)
        valid = j < ends[:, -1:]
        src = order[(starts.gather(1, lvl) + rank).clamp(min=0, max=N - 1)]
        P = torch.where(valid[:, :, None], scores[src], scores.new_full((), -1.0))
        PB = torch.where(valid[:, :, None], boxes7[src], boxes7.new_zeros(()))
        # ---- every (scene, class) segment through one NMS ---------------------------------------------------------------
        masked = torch.where(P > cfg.score_thr, P, P.new_full((1,), -1.0)).permute(0, 2, 1).reshape(B * C, n_max)
        sorted_scores, ordr = masked.sort(dim=1, descending=True, stable=True)
        cnts = (sorted_scores > cfg.score_thr).sum(dim=1).to(torch.int32)
        # The suppression-mask workspace of one (scene, class) segment is n_max * ceil(n_max / 64) * 8 B (2 MB at 4 000
        # candidates) and the workspace is grow-only: all B * C segments at once would pin 290 MB at B = 8, C = 18 and 3.2 GB
        # with a 200-class head.  Scenes go through the NMS in chunks whose masks fit NMS_WS_BUDGET; at the
        # benchmark's shapes that is still ONE chunk (one pair of launches).
        seg_bytes = n_max * ((n_max + 63) // 64) * 8 + n_max * 7 * 4
        per = max(1, min(B, self.NMS_WS_BUDGET // max(1, C * seg_bytes)))
        kept_l, kcount_l = [], []
        for b0 in range(0, B, per):
            b1 = min(B, b0 + per)
            o = ordr[b0 * C:b1 * C].view(b1 - b0, C, n_max, 1).expand(b1 - b0, C, n_max, 7)
            seg_boxes = PB[b0:b1, None].expand(b1 - b0, C, n_max, 7).gather(2, o)
            k_, c_ = _nms_run(seg_boxes.reshape((b1 - b0) * C, n_max, 7).contiguous(), cnts[b0 * C:b1 * C].contiguous(),
                              cfg.iou_thr, yaw_flag)
            kept_l.append(k_)
            kcount_l.append(c_)
        kept = kept_l[0] if len(kept_l) == 1 else torch.cat(kept_l)
        kcount = kcount_l[0] if len(kcount_l) == 1 else torch.cat(kcount_l)
        valid = torch.arange(n_max, device=dev)[None, :] < kcount[:, None]
        done = None
        if defer:
            done = torch.cuda.Event()
            done.record()                                  # everything this batch needs is enqueued up to here

        def finish(post=None):
            # from here on the sizes are data: `nonzero` and `tolist` wait for everything enqueued above.  Deferred (two batches
            # in flight): on the read-back stream behind THIS batch's event — a synchronisation of the main stream would wait for
            # the next batch's forward pass, which is already enqueued there
            if done is not None:
                rb = _readback_stream(dev)
                rb.wait_event(done)
                with torch.cuda.stream(rb):
                    res = collect()
                    return post(res) if post is not None else res
            res = collect()
            return post(res) if post is not None else res

        def collect():
            sc_seg, p = torch.nonzero(valid, as_tuple=True)     # (scene, class)-major, ascending position = descending score
            idx = ordr[sc_seg, kept[sc_seg, p].long()]
            out_scene, out_cls = sc_seg // C, sc_seg % C
            out_boxes = PB[out_scene, idx]
            out_scores = P[out_scene, idx, out_cls]
            sizes = count_ids(out_scene, B).tolist()          # the one read-back
            results, o = [], 0
            for i, n in enumerate(sizes):
                b = out_boxes[o:o + n]
                if yaw_flag:
                    box_dim, with_yaw = 7, True
                else:
                    box_dim, with_yaw, b = 6, False, b[:, :6]
                results.append((img_metas[i]['box_type_3d'](b, box_dim=box_dim, with_yaw=with_yaw, origin=(.5, .5, .5)),
                                out_scores[o:o + n], out_cls[o:o + n]))
                o += n
            return results
        return finish if defer else finish()

    def _get_bboxes_single(self, centernesses, bbox_preds, cls_scores, points, img_meta):
        mlvl_bboxes, mlvl_scores = [], []
        for centerness, bbox_pred, cls_score, point in zip(centernesses, bbox_preds, cls_scores, points):
            scores = cls_score.sigmoid() * centerness.sigmoid()
            max_scores, _ = scores.max(dim=1)
            if len(scores) > self.test_cfg.nms_pre > 0:
                _, ids = max_scores.topk(self.test_cfg.nms_pre)
                bbox_pred, scores, point = bbox_pred[ids], scores[ids], point[ids]
            mlvl_bboxes.append(self._bbox_pred_to_bbox(point, bbox_pred))
            mlvl_scores.append(scores)
        return self._nms(torch.cat(mlvl_bboxes), torch.cat(mlvl_scores), img_meta)

    def _bbox_pred_to_bbox(self, points, bbox_pred):
        """(dx_min,dx_max,dy_min,dy_max,dz_min,dz_max[,r6,r7]) at `points` -> (cx,cy,cz,w,l,h[,yaw]) (:281-330)."""
        if bbox_pred.shape[0] == 0:
            return bbox_pred
        d = bbox_pred
        centre = points + torch.stack((d[:, 1] - d[:, 0], d[:, 3] - d[:, 2], d[:, 5] - d[:, 4]), -1) / 2
        if d.shape[1] == 6:
            return torch.cat((centre, torch.stack((d[:, 0] + d[:, 1], d[:, 2] + d[:, 3], d[:, 4] + d[:, 5]), -1)), -1)
        if self.yaw_parametrization == 'naive':
            size = torch.stack((d[:, 0] + d[:, 1], d[:, 2] + d[:, 3], d[:, 4] + d[:, 5]), -1)
            return torch.cat((centre, size, d[:, 6:7]), -1)
        if self.yaw_parametrization == 'sin-cos':
            size = torch.stack((d[:, 0] + d[:, 1], d[:, 2] + d[:, 3], d[:, 4] + d[:, 5]), -1)
            norm = torch.pow(torch.pow(d[:, 6:7], 2) + torch.pow(d[:, 7:8], 2), 0.5)
            return torch.cat((centre, size, torch.atan2(d[:, 6:7] / norm, d[:, 7:8] / norm)), -1)
        # 'fcaf3d': (r6, r7) = ln(q)·(sin 2a, cos 2a), q = l / w
        scale = d[:, 0] + d[:, 1] + d[:, 2] + d[:, 3]
        q = torch.exp(torch.sqrt(torch.pow(d[:, 6], 2) + torch.pow(d[:, 7], 2)))
        alpha = 0.5 * torch.atan2(d[:, 6], d[:, 7])
        w = scale / (1 + q)
        return torch.cat((centre, torch.stack((w, w * q, d[:, 5] + d[:, 4], alpha), dim=-1)), -1)

    def _nms(self, bboxes, scores, img_meta):
        """Per-class score threshold + BEV NMS + re-wrap (:332-374).  All classes of the scene go through ONE sort and one
        pair of kernel launches (nms.nms_bev_multiclass: class segments, greedy scan on the device, a single read-back
        of the survivor count); the survivors come out in the order of the reference's per-class loop (class-major,
        descending score) — `_nms_per_class` is that loop, kept as the cross-check of the parity tests."""
        yaw_flag = bboxes.shape[1] == 7
        boxes7 = bboxes if yaw_flag else torch.cat((bboxes, torch.zeros_like(bboxes[:, :1])), dim=1)
        if boxes7.shape[0]:
            idx, cls = nms_bev_multiclass(boxes7, scores, self.test_cfg.score_thr, self.test_cfg.iou_thr, rotated=yaw_flag)
            nms_bboxes = boxes7[idx]
            nms_scores = scores[idx, cls]
            nms_labels = cls.to(torch.long)
        else:
            nms_bboxes = bboxes.new_zeros((0, 7))
            nms_scores = bboxes.new_zeros((0,))
            nms_labels = bboxes.new_zeros((0,), dtype=torch.long)
        if yaw_flag:
            box_dim, with_yaw = 7, True
        else:
            box_dim, with_yaw = 6, False
            nms_bboxes = nms_bboxes[:, :6]
        nms_bboxes = img_meta['box_type_3d'](nms_bboxes, box_dim=box_dim, with_yaw=with_yaw, origin=(.5, .5, .5))
        return nms_bboxes, nms_scores, nms_labels

    def _nms_per_class(self, bboxes, scores, img_meta):
        n_classes = scores.shape[1]
        yaw_flag = bboxes.shape[1] == 7
        nms_bboxes, nms_scores, nms_labels = [], [], []
        boxes7 = bboxes if yaw_flag else torch.cat((bboxes, torch.zeros_like(bboxes[:, :1])), dim=1)
        for i in range(n_classes):
            ids = scores[:, i] > self.test_cfg.score_thr
            if not ids.any():
                continue
            class_scores = scores[ids, i]
            class_bboxes = boxes7[ids]
            nms_ids = nms_bev(class_bboxes, class_scores, self.test_cfg.iou_thr, rotated=yaw_flag)
            nms_bboxes.append(class_bboxes[nms_ids])
            nms_scores.append(class_scores[nms_ids])
            nms_labels.append(bboxes.new_full(class_scores[nms_ids].shape, i, dtype=torch.long))
        if len(nms_bboxes):
            nms_bboxes = torch.cat(nms_bboxes, dim=0)
            nms_scores = torch.cat(nms_scores, dim=0)
            nms_labels = torch.cat(nms_labels, dim=0)
        else:
            nms_bboxes = bboxes.new_zeros((0, 7))
            nms_scores = bboxes.new_zeros((0,))
            nms_labels = bboxes.new_zeros((0,))
        if yaw_flag:
            box_dim, with_yaw = 7, True
        else:
            box_dim, with_yaw = 6, False
            nms_bboxes = nms_bboxes[:, :6]
        nms_bboxes = img_meta['box_type_3d'](nms_bboxes, box_dim=box_dim, with_yaw=with_yaw, origin=(.5, .5, .5))
        return nms_bboxes, nms_scores, nms_labels


_rb_streams = {}


def _readback_stream(device):
    idx = device.index if device.index is not None else torch.cuda.current_device()
    if idx not in _rb_streams:
        _rb_streams[idx] = torch.cuda.Stream(device=idx, priority=-1)
    return _rb_streams[idx]


def count_ids(ids, n):
    """torch.bincount(ids, minlength=n) for ids known to lie in [0, n), WITHOUT the device -> host synchronisation bincount
    makes to size its output (r4: in `get_bboxes` that sync made the host wait for the whole forward pass in the middle of the
    decode, 6.7 of 12.4 ms per batch of 8, and the rest of the decode was enqueued behind it).  Integer atomics: exact."""
    return torch.zeros(n, dtype=torch.int64, device=ids.device).scatter_add_(0, ids, torch.ones_like(ids))


def segmented_topk(seg, score, n_seg, k, compact=True):
    """Indices of the rows the reference's per-(scene, level) loop keeps (fcaf3d_neck_with_head.py:238-243: `if len(scores) >
    nms_pre: topk(nms_pre)`), for every segment at once: segments in ascending id; inside a segment with more than k rows the k
    best by descending score (ties: row order), inside a smaller one ALL rows in row order.  Two STABLE sorts (by descending
    score where the segment is cut, then by segment) — exact for any float32 scores (r2 packed segment and score into one
    float64 key, which merged scores below ~1e-7 into ties: ADVICE r2).  seg (N,) int64 in [0, n_seg), score (N,)."""
    N = seg.numel()
    counts = count_ids(seg, n_seg)
    big = counts > k
    row = torch.arange(N, device=seg.device)
    # rows of uncut segments keep their row order: give them a constant key, the stable sort does the rest
    key = torch.where(big[seg], -score.float(), score.new_zeros(()).float())
    o1 = torch.sort(key, stable=True).indices
    order = o1[torch.sort(seg[o1], stable=True).indices]
    starts = torch.cumsum(counts, 0) - counts
    rank = row - starts[seg[order]]
    if not compact:
        # (order, rank of every row inside its segment, rows kept per segment): the caller keeps rank < k — no boolean-mask
        # gather, hence no device -> host synchronisation
        return order, rank, counts.clamp(max=k)
    return order[rank < k]


def compute_centerness(bbox_targets):
    """sqrt( Π_axis min(d-,d+)/max(d-,d+) ) over the 6 face distances (:377-384)."""
    d = bbox_targets[..., :6].reshape(bbox_targets.shape[:-1] + (3, 2))
    lo = d.min(dim=-1).values
    hi = d.max(dim=-1).values
    return torch.sqrt(lo[..., 0] / hi[..., 0] * lo[..., 1] / hi[..., 1] * lo[..., 2] / hi[..., 2])


@BBOX_ASSIGNERS.register_module()
class Fcaf3DAssigner:
    """Target assignment of FCAF3D (:387-466): a location is positive for a GT box when it lies inside
    the (rotated) box, on the box's best pyramid level (the last level that still holds >= `limit`
    inside locations), and among the `topk` most central such locations; ties between boxes go to the
    smallest volume."""

    def __init__(self, limit, topk, n_scales):
        self.limit = limit
        self.topk = topk
        self.n_scales = n_scales

    @torch.no_grad()
    def assign_batched(self, pts, scene, level, cmaps, gt_bboxes, gt_labels, pre=None):
        """All scenes, all levels, in four kernel launches (csrc/assign.hip).
        pts (N,3) locations, scene/level (N) int32, cmaps: the coordinate map of each level (for the
        per-(level, scene) row groups).  Returns (centerness_targets (N), bbox_targets (N,7), labels (N))."""
        if not pts.is_cuda:
            raise RuntimeError('batched target assignment runs on the GPU only (HIP)')
        dev = pts.device
        B, Lv = len(gt_bboxes), len(cmaps)
        assert Lv == self.n_scales
        M = max(1, max(len(g) for g in gt_bboxes))
        # all scenes' boxes packed with a handful of launches (r2: the per-scene loop issued ~60 tiny kernels per step)
        lens = [len(g) for g in gt_bboxes]
        boxes = pts.new_zeros((B, M, 7))
        labels = torch.zeros((B, M), dtype=torch.int64, device=dev)
        if sum(lens):
            host_boxes = all(not g.tensor.is_cuda for g in gt_bboxes)
            host_labels = all(not l.is_cuda for l in gt_labels)
            if host_boxes:
                # annotations as a data loader hands them over (host tensors): packed on the host, ONE pinned upload (r6: eight
                # blocking per-scene copies + a dozen tiny launches per step on the coordinate stream)
                hb = np.zeros((B, M, 7), dtype=np.float32)
                for i, g in enumerate(gt_bboxes):
                    if lens[i]:
                        t = g.tensor.detach().numpy().astype(np.float32)
                        hb[i, :lens[i], :t.shape[1]] = t
                        hb[i, :lens[i], 2] = t[:, 2] + t[:, 5] * np.float32(0.5)      # gravity centre (DepthInstance3DBoxes.gravity_center)
                boxes = L.upload(hb, dev)
            if host_labels:
                hl = np.zeros((B, M), dtype=np.int64)
                for i, l in enumerate(gt_labels):
                    if lens[i]:
                        hl[i, :lens[i]] = l.detach().numpy().astype(np.int64)
                labels = L.upload(hl, dev)
            if not (host_boxes and host_labels):
                slot = L.upload(np.concatenate([i * M + np.arange(n) for i, n in enumerate(lens) if n]), dev)
            if not host_boxes:
                allb = torch.cat([g.tensor.to(dev) for g in gt_bboxes if len(g)])
                packed = allb.clone()
                packed[:, 2] += allb[:, 5] * 0.5                 # gravity centre (DepthInstance3DBoxes.gravity_center)
                boxes.view(B * M, 7)[slot] = packed              # (.,7): DepthInstance3DBoxes pads yaw-less boxes with a zero yaw
            if not host_labels:
                alll = torch.cat([l.to(dev) for g, l in zip(gt_bboxes, gt_labels) if len(g)])
                labels.view(B * M)[slot] = alll.to(torch.int64)
        # pinned + non_blocking: a pageable host->device copy would block the host until the whole forward has drained
        box_count = L.upload(np.asarray(lens, dtype=np.int32), dev)
        if pre is not None:
            order, seg_start = pre['order'], pre['seg_start']    # rows of a plan-made set are grouped by scene: identity order
        else:
            order, counts, off = [], [], 0
            for cm in cmaps:
                cm._decompose()
                order.append(cm._order + off)
                counts.append(cm._counts_dev)
                off += cm.n
            order = torch.cat(order).to(torch.int32)
            seg_start = torch.cat((torch.zeros(1, dtype=torch.int64, device=dev), torch.cumsum(torch.cat(counts), 0))).to(torch.int32)
        N = pts.shape[0]
        ct = torch.empty(N, dtype=torch.float32, device=dev)
        bt = torch.empty((N, 7), dtype=torch.float32, device=dev)
        lab = torch.empty(N, dtype=torch.int64, device=dev)
        pts = pts.contiguous()
        ws = L.workspace(L.query('fc_assign_ws_bytes', B, M, Lv), dev)
        L.call('fc_assign_targets', L.ptr(pts), L.ptr(scene.contiguous()), L.ptr(level.contiguous()), N, L.ptr(boxes),
               L.ptr(labels), L.ptr(box_count), B, M, Lv, L.ptr(order), L.ptr(seg_start), int(self.limit), int(self.topk),
               L.ptr(ct), L.ptr(bt), L.ptr(lab), L.ptr(ws), ws.numel(), L.stream())
        return ct, bt, lab

    @torch.no_grad()
    def assign(self, points, gt_bboxes, gt_labels):
        float_max = 1e8
        dev = points[0].device
        level = torch.cat([points[i].new_full((len(points[i]),), i) for i in range(len(points))])
        pts = torch.cat(points, dim=0)
        n_points, n_boxes = len(pts), len(gt_bboxes)
        boxes = torch.cat((gt_bboxes.gravity_center, gt_bboxes.tensor[:, 3:]), dim=1).to(dev)   # (m,7)
        volumes = gt_bboxes.volume.to(dev)
        if n_boxes == 0:
            return (pts.new_zeros(n_points), pts.new_zeros((n_points, 7)),
                    gt_labels.new_full((n_points,), -1))
        # offsets of every location from every box centre, rotated into the box frame (angle -yaw about z,
        # the reference's rotation_3d_in_axis convention: x' = x cos a + y sin a, y' = -x sin a + y cos a)
        shift = pts[:, None, :] - boxes[None, :, :3]                          # (n,m,3)
        ang = -boxes[:, 6]
        ca, sa = torch.cos(ang)[None], torch.sin(ang)[None]
        rx = shift[..., 0] * ca + shift[..., 1] * sa
        ry = -shift[..., 0] * sa + shift[..., 1] * ca
        centre = boxes[None, :, :3] + torch.stack((rx, ry, shift[..., 2]), dim=-1)
        half = boxes[None, :, 3:6] / 2
        lo = centre - boxes[None, :, :3] + half                               # distance to the "min" faces
        hi = boxes[None, :, :3] + half - centre                               # distance to the "max" faces
        targets = torch.stack((lo[..., 0], hi[..., 0], lo[..., 1], hi[..., 1], lo[..., 2], hi[..., 2],
                               boxes[None, :, 6].expand(n_points, n_boxes)), dim=-1)   # (n,m,7)
        inside = targets[..., :6].min(-1).values > 0                          # (n,m)

        # best level per box
        # inside-counts per (level, box) as one small GEMM: no boolean-mask indexing, hence no host sync
        # (exact: counts < 2^24 in fp32)
        onehot = (level[None, :] == torch.arange(self.n_scales, device=dev, dtype=level.dtype)[:, None]).float()
        per_level = onehot @ inside.float()                                                        # (L,m)
        starved = per_level < self.limit
        first_starved = torch.argmax(starved.int(), dim=0) - 1
        first_starved = first_starved.clamp(min=0)
        best = torch.where(~starved.any(dim=0), torch.full_like(first_starved, self.n_scales - 1), first_starved)
        on_best = level[:, None] == best[None, :].to(level.dtype)

        # top-k most central candidates per box
        centerness = compute_centerness(targets)
        neg = torch.full_like(centerness, -1.0)
        centerness = torch.where(inside & on_best, centerness, neg)
        kth = torch.topk(centerness, min(self.topk + 1, n_points), dim=0).values[-1]
        central = centerness > kth[None]

        vol = torch.where(inside & on_best & central, volumes[None].expand(n_points, n_boxes),
                          volumes.new_full((1,), float_max))
        min_vol, owner = vol.min(dim=1)
        labels = torch.where(min_vol == float_max, gt_labels.new_full((1,), -1), gt_labels.to(dev)[owner])
        rows = torch.arange(n_points, device=dev)
        centerness_targets = compute_centerness(targets[rows, owner])
        return centerness_targets, boxes[owner], labels


from . import executor as _executor                          # the native executor covers exactly these two methods
_executor._FORWARD_SINGLE = Fcaf3DNeckWithHead.forward_single
_executor._NECK_FORWARD = Fcaf3DNeckWithHead.forward
