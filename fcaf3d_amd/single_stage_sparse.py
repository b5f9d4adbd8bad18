"""Detector shell — host-side mirror of mmdet3d/models/detectors/single_stage_sparse.py:9-62 and the
`forward(return_loss=...)` dispatch of detectors/base.py:45-60."""
import os
import torch
from torch import nn

from . import _lib as L
from .boxes import bbox3d2result_batch
from .registry import DETECTORS, build_backbone, build_head
from .sparse import SparseTensor, _rec, on_map_stream


@DETECTORS.register_module()
class SingleStageSparse3DDetector(nn.Module):
    def __init__(self, backbone, neck_with_head, voxel_size, pretrained=False, train_cfg=None, test_cfg=None):
        super().__init__()
        self.backbone = build_backbone(backbone)
        neck_with_head = dict(neck_with_head)
        neck_with_head.update(train_cfg=train_cfg)
        neck_with_head.update(test_cfg=test_cfg)
        self.neck_with_head = build_head(neck_with_head)
        self.voxel_size = voxel_size
        self.train_cfg = train_cfg
        self.test_cfg = test_cfg
        # True: voxelisation + every coordinate/kernel map of the step are built on a side HIP stream,
        # overlapping whatever the main stream still runs (inputs must already be resident on the device)
        self.async_maps = False
        # with async_maps: True = the points were written before this step was enqueued (resident in HBM), the coordinate
        # stream need not wait for the main stream; False (default) = it waits unless sparse.mark_inputs_ready() was called
        self.inputs_resident = False
        # True: collated points are sorted along a Z-order curve before de-duplication, so that every level's row
        # order is spatially coherent (gather locality).  Only the ROW ORDER changes; per-scene results are sets.
        self.spatial_sort = False
        self.init_weights()

    def init_weights(self, pretrained=None):
        self.backbone.init_weights()
        self.neck_with_head.init_weights()

    def voxelize(self, points):
        """ME.utils.batch_sparse_collate of [(xyz / voxel_size, rgb / 255)] (reference :34-36), one fused
        kernel per scene: coords int32 (ΣN,4), feats (ΣN,3)."""
        dev = points[0].device
        total = sum(p.shape[0] for p in points)
        nfeat = points[0].shape[1] - 3
        coords = torch.empty((total, 4), dtype=torch.int32, device=dev)
        feats = torch.empty((total, nfeat), dtype=torch.float32, device=dev)
        off = 0
        for b, p in enumerate(points):
            n = p.shape[0]
            if hasattr(p, 'voxelize_into'):
                # a scene whose augmentation is drawn but not applied (pipelines.LazyAugmentedPoints): align / sample /
                # flip / rotate / scale / translate + voxelise in ONE pass over the raw points (fc_augment_voxelize)
                p.voxelize_into(b, self.voxel_size, 255.0, coords[off:], feats[off:])
                off += n
                continue
            p = p.contiguous()
            L.call('fc_voxelize', L.ptr(p), n, p.shape[1], b, float(self.voxel_size), 255.0, nfeat,
                   L.ptr(coords[off:]), L.ptr(feats[off:]), L.stream())
            off += n
        if self.spatial_sort and total:
            keys = torch.empty(total, dtype=torch.int64, device=dev)
            L.call('fc_morton_keys', L.ptr(coords), total, L.ptr(keys), L.stream())
            order = torch.argsort(keys)
            coords, feats = coords[order], feats[order]
        return coords, feats

    def prefetch(self, points, gt=True):
        """Start the coordinate phase of a LATER batch now, on a worker thread and the coordinate stream (plan.Lookahead): call it
        with the next batch's points before running the current step; `extract_feat` of exactly these tensors then finds its sets
        and maps made.  gt: whether that step will pass ground truth (training).  Returns False when the native plan does not
        cover this configuration (the step then plans in line, as without the call)."""
        from . import plan as PL
        la = self.__dict__.get('_lookahead')
        if la is None:
            la = self.__dict__['_lookahead'] = PL.Lookahead(self)
        p0 = points[0]
        if not (hasattr(p0, 'is_cuda') and p0.is_cuda):
            return False
        dev = p0.device
        from .sparse import plan_stream
        main = torch.cuda.current_stream(dev)
        want_targets = bool(gt) and hasattr(self.neck_with_head, 'prepare_targets')
        return la.submit(points, self.training, want_targets, plan_stream(dev), main, wait_main=not self.inputs_resident)

    def _sparse_input(self, points, gt=None):
        from . import plan as PL
        pl = PL.planner_of(self)
        nh = self.neck_with_head
        want_targets = gt is not None and hasattr(nh, 'prepare_targets')
        la = self.__dict__.get('_lookahead')
        sp = la.take(points, self.training, want_targets) if la is not None else None
        if sp is None and pl.applicable(points):
            sp = pl.run(points, self.training, want_targets)
        return self._use_plan(sp, points, gt)

    def _use_plan(self, sp, points, gt=None):
        """sp: the StepPlan of these points (native coordinate phase, plan.py) or None (per-operator coordinate phase)"""
        if sp is not None:
            x = sp.x
            if sp.structured:
                head_maps, self._prune_level = sp.head_maps, sp.prune_level
            else:
                head_maps = self.plan_maps(x.cmap)                 # a backbone voxel outside the generated set: unions on demand
            self._step_plan = sp
        else:
            coordinates, features = self.voxelize(points)
            x = SparseTensor(features, coordinates=coordinates, batch_size=len(points))
            _rec(x.F)
            head_maps = self.plan_maps(x.cmap)
            self._step_plan = None
        # the network body as one native call per direction (executor.py) when this step's maps fit its static operator list
        self._bound = None
        tail0 = self._prune_level == 0                     # pruning bites at the finest level only: its tail runs per operator
        if (head_maps is not None or tail0) and x.F.is_cuda:
            from . import executor
            grad_on = torch.is_grad_enabled()
            if self.training == grad_on:                   # training with gradients, or inference without: the two programs
                prog = executor.program_for(self, self.training, tail0)
                if prog is not None:
                    st = prog.bind(x, len(points), backward=self.training)
                    if st is not None:
                        self._bound = (prog, st)
        if gt is not None and head_maps is not None and hasattr(self.neck_with_head, 'prepare_targets'):
            # training: the target assignment depends on the head's LOCATIONS (coordinate sets, known now) and the ground
            # truth only — it runs here, on the coordinate stream, instead of between forward and backward (r3)
            self.neck_with_head.prepare_targets(head_maps, *gt, pre=sp.targets if sp is not None else None)
        return x

    def plan_maps(self, cm0):
        """Build, up front, every coordinate set and kernel map the step will use: they depend on the input
        coordinates only (unless pts_threshold pruning bites, where planning stops and the rest is built
        on demand).  The data-dependent sizes are read back here, while the queue holds only these small
        integer kernels — no host sync is left inside the convolution sequence."""
        bb, nh = self.backbone, self.neck_with_head
        self._prune_level = None                                   # the neck level where pts_threshold first bites (None: nowhere)
        bottleneck = getattr(bb.BLOCK, 'expansion', 1) == 4
        bwd = self.training and torch.is_grad_enabled()
        nl = min(bb.n_outs, 4)
        m1 = cm0.strided(2); cm0.kernel_map(m1, 3)                 # stem conv k3 s2
        m2 = m1.strided(2); m1.kernel_map(m2, 2)                   # max-pool k2 s2
        prev, levels = m2, []
        for _ in range(nl):
            mi = prev.strided(2)
            prev.kernel_map(mi, 3).prefetch(bwd); prev.kernel_map(mi, 1); mi.kernel_map(mi, 3).prefetch(bwd)
            if bottleneck:
                break                                              # 1x1-3x3-1x1 blocks: keep it lazy
            levels.append(mi)
            prev = mi
        if bottleneck or not levels:
            return None
        # (r5, measured and removed: reading the step's counts back in ONE copy — live pair-list tiles, union membership, per-scene
        # rows; or even every strided set made from the finest one with a single size read-back — is SLOWER than this chain of ~25
        # small read-backs: each of them lets the host go on enqueueing while the coordinate stream works, one late read-back makes
        # it wait for the whole phase.  8 scenes 362.1 vs 361.7 / 358.4 scenes/s, S3DIS 111.7 vs 108.6 / 106.0: profiles/r5_notes.md)
        x = levels[-1]
        x.scene_counts
        head_maps = [x]                                            # coordinate sets of the head's levels, coarse -> fine
        for i in range(len(levels) - 2, -1, -1):
            g = x.generate(); g.kernel_map(g, 3).prefetch(bwd)
            u, _, _ = levels[i].union(g)
            if nh.pts_threshold >= 0 and any(c > nh.pts_threshold for c in u.scene_counts):
                self._prune_level = i
                return None
            u.kernel_map(u, 3).prefetch(bwd)
            x = u
            head_maps.append(x)
        return head_maps[::-1]                                     # finest first, the order of the head's outputs

    def extract_feat(self, points, img_metas, gt=None):
        """gt (training only, optional): (gt_bboxes_3d, gt_labels_3d) — lets the target assignment start with the maps"""
        if self.async_maps:
            with on_map_stream(points[0].device, self.inputs_resident and os.environ.get('FC_MAP_WAIT') != '1'):
                x = self._sparse_input(points, gt)
        else:
            x = self._sparse_input(points, gt)
        bound, self._bound = getattr(self, '_bound', None), None
        if bound is not None:
            return self._exec_forward(*bound)
        x = self.backbone(x)
        x = self.neck_with_head(x)
        return x

    def _exec_forward(self, prog, st):
        """backbone + neck + head through the native executor; returns what `neck_with_head(backbone(x))` returns"""
        from . import executor
        from . import nn as MEnn
        from .fcaf3d_neck_with_head import SceneList
        from .sparse import SparseTensor
        res = prog.forward(st)
        cent, bbox, cls, cmax = res[:4]
        self._last_exec = (prog, st) if executor.KEEP_STATE else None
        nh = self.neck_with_head
        vs = nh.voxel_size
        outs, o = ([], [], [], []), 0
        for lvl, cm in enumerate(st['head_maps']):
            if lvl == 0 and prog.tail0:
                # the pruned finest level (fcaf3d_neck_with_head.py:104-108, :110-126): top-k of the parent level's interpolated
                # scores, MinkowskiPruning, out_block_0 and forward_single on the per-operator path, differentiable through x0
                n1 = st['head_maps'][1].n
                scores = SparseTensor(cmax[:n1], coordinate_map_key=st['head_maps'][1])
                x0 = nh._prune(SparseTensor(res[4], coordinate_map_key=cm), scores)
                out = nh.forward_single(MEnn.run_sequential(nh.out_block_0, x0), nh.scales[0])
                for k in range(4):
                    outs[k].append(out[k])
                continue
            n = cm.n
            outs[0].append(SceneList(cent[o:o + n], cm, parent=cent))
            outs[1].append(SceneList(bbox[o:o + n], cm, parent=bbox))
            outs[2].append(SceneList(cls[o:o + n], cm, parent=cls))
            outs[3].append(SceneList(lambda cm=cm: cm.coords[:, 1:].float() * vs, cm))      # voxel corners, made when asked for
            o += n
        return [tuple(v) for v in outs]

    # eval mode: True = the weights do not change between calls (inference serving): the executor keeps its pre-split weight images
    # across calls instead of rebuilding them at every forward pass.  Switching train() / eval(), load_state_dict and .to() drop them.
    static_weights = False

    def _stale_images(self):
        for pr in self.__dict__.get('_programs', {}).values():
            pr.weights_fresh = False

    def train(self, mode=True):
        self._stale_images()
        return super().train(mode)

    def load_state_dict(self, *a, **kw):
        self._stale_images()
        return super().load_state_dict(*a, **kw)

    def _apply(self, fn, *a, **kw):
        # parameters / buffers may move: the executor's address tables AND its weight images (keyed by the old data_ptr) are stale
        self.__dict__.pop('_programs', None)
        self.__dict__.pop('_exec_weights', None)
        from .flat import GENERATION
        GENERATION[0] += 1
        return super()._apply(fn, *a, **kw)

    def forward_train(self, points, gt_bboxes_3d, gt_labels_3d, img_metas):
        x = self.extract_feat(points, img_metas, (gt_bboxes_3d, gt_labels_3d))
        return self.neck_with_head.loss(*x, gt_bboxes_3d, gt_labels_3d, img_metas)

    def simple_test(self, points, img_metas, imgs=None, rescale=False):
        x = self.extract_feat(points, img_metas)
        bbox_list = self.neck_with_head.get_bboxes(*x, img_metas, rescale=rescale)
        return bbox3d2result_batch(bbox_list)

    def simple_test_async(self, points, img_metas, imgs=None, rescale=False):
        """`simple_test` in two halves for a serving loop that keeps two batches in flight: this call ENQUEUES the forward pass,
        the decode and the NMS of the batch and returns a callable; calling it performs the read-backs and returns what
        `simple_test` returns.  Enqueue the next batch before collecting this one: its coordinate phase (5 ms of host time per 8
        scenes, mostly count read-backs on the coordinate stream) then runs while the GPU is busy with this batch.  The reference's
        loop (tools/test.py -> mmdet3d/apis/test.py single_gpu_test) is synchronous per batch behind DataLoader workers."""
        x = self.extract_feat(points, img_metas)
        finish = self.neck_with_head.get_bboxes(*x, img_metas, rescale=rescale, defer=True)
        return lambda: finish(bbox3d2result_batch)

    def aug_test(self, points, img_metas, imgs=None, rescale=False):
        pass

    def forward_test(self, points, img_metas, img=None, **kwargs):
        # base.py:14-43: one (non-augmented) sample per list entry
        if isinstance(points[0], (list, tuple)):
            assert len(points) == 1, 'test-time augmentation is a stub in the reference (aug_test: pass)'
            return self.simple_test(points[0], img_metas[0], **kwargs)
        return self.simple_test(points, img_metas, **kwargs)

    def forward(self, return_loss=True, **kwargs):
        if return_loss:
            return self.forward_train(**kwargs)
        return self.forward_test(**kwargs)
