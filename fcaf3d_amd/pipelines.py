"""The step BEFORE the hot path: the reference's indoor train/test pipelines
(configs/fcaf3d/fcaf3d_scannet-3d-18class.py:16-74) as tensor functions that run where the data lives — on the GPU, on
a whole scene at once — so that feeding hundreds of scenes/s does not hang on CPU dataloader workers:

  LoadPointsFromFile        mmdet3d/datasets/pipelines/loading.py:333      -> load_points_from_file
  GlobalAlignment           transforms_3d.py:409-490                       -> global_alignment
  IndoorPointSample         transforms_3d.py:821-895                       -> indoor_point_sample
  RandomFlip3D              transforms_3d.py:59-170  (+ depth_box3d.py:178-208, depth_points.py:28-33)  -> flip_bev
  GlobalRotScaleTrans       transforms_3d.py:493-645 (+ depth_box3d.py:113-176, base_box3d.py:149-222)  -> rot_scale_trans

Boxes are Depth-mode (m,7) `[x, y, z_bottom, dx, dy, dz, yaw]` tensors (`DepthInstance3DBoxes.tensor`); points (n,3+C).
The deterministic parts take their parameters explicitly (parity-tested against the reference's own box / point classes,
tests/golden/pipeline.npz); `TrainAugment` draws them the way the reference's classes do.  Everything is plain torch:
elementwise work on one scene, nothing here deserves a kernel.
"""
import math

import numpy as np
import torch


def load_points_from_file(path, load_dim=6, use_dim=(0, 1, 2, 3, 4, 5), device=None):
    """LoadPointsFromFile(coord_type='DEPTH'): a flat float32 .bin of `load_dim` columns."""
    pts = np.fromfile(path, dtype=np.float32).reshape(-1, load_dim)[:, list(use_dim)]
    t = torch.from_numpy(np.ascontiguousarray(pts))
    return t.to(device) if device is not None else t


def _rot_z_T(angle, like):
    c, s = math.cos(angle), math.sin(angle)
    return like.new_tensor([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]]).T      # rot_mat_T of the reference


def global_alignment(points, axis_align_matrix, rotation_axis=2):
    """GlobalAlignment: points <- points @ R^T + t with the scene's 4x4 axis_align_matrix (ScanNet)."""
    m = torch.as_tensor(np.asarray(axis_align_matrix), dtype=points.dtype, device=points.device)
    assert m.shape == (4, 4), f'invalid shape {tuple(m.shape)} for axis_align_matrix'
    rot, trans = m[:3, :3], m[:3, 3]
    unit = torch.zeros(3, dtype=m.dtype, device=m.device); unit[rotation_axis] = 1.0
    ok = abs(float(torch.linalg.det(rot.double())) - 1.0) < 1e-5 and bool((rot[rotation_axis] == unit).all()) \
        and bool((rot[:, rotation_axis] == unit).all())
    assert ok, f'invalid rotation matrix {rot}'
    out = points.clone()
    out[:, :3] = points[:, :3] @ rot.T + trans
    return out


def indoor_point_sample(points, num_points, generator=None):
    """IndoorPointSample: exactly num_points rows, without replacement when the scene has enough points."""
    n = points.shape[0]
    if n >= num_points:
        idx = torch.randperm(n, generator=generator, device=points.device)[:num_points]
    else:
        idx = torch.randint(0, n, (num_points,), generator=generator, device=points.device)
    return points[idx], idx


def _corners_xy_extent(boxes, rot_T):
    """x / y extent of the rotated boxes' corners (DepthInstance3DBoxes.rotate, with_yaw=False branch)."""
    dims = boxes[:, 3:6]
    cn = torch.tensor([[0, 0], [0, 1], [1, 1], [1, 0]], dtype=boxes.dtype, device=boxes.device) - 0.5      # x,y of the 4 BEV corners
    c = dims[:, None, :2] * cn[None]                                                # (m,4,2) about the centre
    yaw = boxes[:, 6]
    cs, sn = torch.cos(yaw), torch.sin(yaw)
    # rotation_3d_in_axis(axis=2) (structures/utils.py:21-61): corners @ [[c, -s], [s, c]]
    x = c[..., 0] * cs[:, None] + c[..., 1] * sn[:, None] + boxes[:, None, 0]
    y = -c[..., 0] * sn[:, None] + c[..., 1] * cs[:, None] + boxes[:, None, 1]
    xr = x * rot_T[0, 0] + y * rot_T[1, 0]
    yr = x * rot_T[0, 1] + y * rot_T[1, 1]
    return xr.max(1).values - xr.min(1).values, yr.max(1).values - yr.min(1).values


def rotate(points, boxes, angle, with_yaw=True):
    """DepthInstance3DBoxes.rotate(angle, points): counter-clockwise about z; yaw -= angle, or (with_yaw=False) the
    axis-aligned extent of the rotated box."""
    rot_T = _rot_z_T(angle, points)
    points = points.clone(); boxes = boxes.clone()
    if boxes.shape[0]:
        old = boxes.clone()
        boxes[:, :3] = old[:, :3] @ rot_T
        if with_yaw:
            boxes[:, 6] = old[:, 6] - angle
        else:
            # the reference rotates the corners of the ALREADY moved box (self.corners after the centre update)
            nx, ny = _corners_xy_extent(torch.cat((boxes[:, :3], old[:, 3:]), 1), rot_T)
            boxes[:, 3], boxes[:, 4] = nx, ny
    points[:, :3] = points[:, :3] @ rot_T
    return points, boxes


def flip_bev(points, boxes, direction, with_yaw=True):
    """DepthInstance3DBoxes.flip: 'horizontal' negates x (yaw -> pi - yaw), 'vertical' negates y (yaw -> -yaw)."""
    assert direction in ('horizontal', 'vertical')
    points = points.clone(); boxes = boxes.clone()
    if direction == 'horizontal':
        points[:, 0] = -points[:, 0]
        boxes[:, 0] = -boxes[:, 0]
        if with_yaw:
            boxes[:, 6] = -boxes[:, 6] + math.pi
    else:
        points[:, 1] = -points[:, 1]
        boxes[:, 1] = -boxes[:, 1]
        if with_yaw:
            boxes[:, 6] = -boxes[:, 6]
    return points, boxes


def rot_scale_trans(points, boxes, angle, scale, trans, with_yaw=True):
    """GlobalRotScaleTrans with its three draws given: rotate, then scale (xyz and box sizes), then translate."""
    points, boxes = rotate(points, boxes, angle, with_yaw)
    points[:, :3] = points[:, :3] * scale
    boxes[:, :6] = boxes[:, :6] * scale
    t = torch.as_tensor(np.asarray(trans, np.float32), dtype=points.dtype, device=points.device)
    points[:, :3] = points[:, :3] + t
    boxes[:, :3] = boxes[:, :3] + t
    return points, boxes


class TrainAugment:
    """IndoorPointSample -> RandomFlip3D(sync_2d=False) -> GlobalRotScaleTrans with the ScanNet / S3DIS / SUN RGB-D
    settings of configs/fcaf3d/*.py; the draws come from a torch.Generator on the data's device."""

    def __init__(self, num_points=100000, flip_ratio_bev_horizontal=0.5, flip_ratio_bev_vertical=0.5,
                 rot_range=(-0.087266, 0.087266), scale_ratio_range=(0.9, 1.1), translation_std=(0.1, 0.1, 0.1),
                 with_yaw=False):
        self.num_points = num_points
        self.flip_h, self.flip_v = flip_ratio_bev_horizontal, flip_ratio_bev_vertical
        self.rot_range, self.scale_ratio_range = tuple(rot_range), tuple(scale_ratio_range)
        self.translation_std = tuple(translation_std) if not isinstance(translation_std, (int, float)) \
            else (translation_std,) * 3
        self.with_yaw = with_yaw

    def draw(self, generator=None, device='cpu'):
        u = torch.rand(4, generator=generator, device=device).tolist()
        g = torch.randn(3, generator=generator, device=device).tolist()
        return dict(flip_h=u[0] < self.flip_h, flip_v=u[1] < self.flip_v,
                    angle=self.rot_range[0] + u[2] * (self.rot_range[1] - self.rot_range[0]),
                    scale=self.scale_ratio_range[0] + u[3] * (self.scale_ratio_range[1] - self.scale_ratio_range[0]),
                    trans=[g[i] * self.translation_std[i] for i in range(3)])

    def __call__(self, points, boxes, generator=None):
        """points (n,3+C), boxes (m,7) Depth-mode bottom-centre -> augmented (num_points,3+C), (m,7), params"""
        points, _ = indoor_point_sample(points, self.num_points, generator)
        p = self.draw(generator, points.device)
        if p['flip_h']:
            points, boxes = flip_bev(points, boxes, 'horizontal', self.with_yaw)
        if p['flip_v']:
            points, boxes = flip_bev(points, boxes, 'vertical', self.with_yaw)
        points, boxes = rot_scale_trans(points, boxes, p['angle'], p['scale'], p['trans'], self.with_yaw)
        if points.is_cuda:
            from .sparse import mark_inputs_ready
            mark_inputs_ready(points.device)      # the detector's coordinate side stream must not read them earlier
        return points, boxes, p


class IndoorInfoDataset:
    """The annotation side of the reference's ScanNetDataset / S3DISDataset / SUNRGBDDataset
    (mmdet3d/datasets/scannet_dataset.py:70-117, custom_3d.py): an info `.pkl` (list of per-scene dicts written by
    tools/create_data.py) -> per scene: points file, axis-alignment matrix, GT boxes as Depth-mode bottom-centre (m,7)
    and labels.  `load()` returns the scene on `device`, aligned, ready for TrainAugment / the detector."""

    def __init__(self, data_root, ann_file, with_yaw=False, load_dim=6, use_dim=(0, 1, 2, 3, 4, 5)):
        import pickle
        self.data_root = data_root
        with open(ann_file, 'rb') as f:
            self.data_infos = pickle.load(f)
        self.with_yaw, self.load_dim, self.use_dim = with_yaw, load_dim, tuple(use_dim)

    def __len__(self):
        return len(self.data_infos)

    def get_ann_info(self, index):
        from .boxes import DepthInstance3DBoxes
        info = self.data_infos[index]
        ann = info.get('annos', dict(gt_num=0))              # S3DIS infos written without detection annotations
        if ann['gt_num'] != 0:
            boxes = np.asarray(ann['gt_boxes_upright_depth'], np.float32)
            labels = np.asarray(ann['class']).astype(np.int64)
        else:
            boxes = np.zeros((0, 7 if self.with_yaw else 6), np.float32)
            labels = np.zeros((0,), np.int64)
        gt = DepthInstance3DBoxes(torch.from_numpy(boxes), box_dim=boxes.shape[-1], with_yaw=self.with_yaw,
                                  origin=(0.5, 0.5, 0.5))
        out = dict(gt_bboxes_3d=gt, gt_labels_3d=torch.from_numpy(labels))
        import os
        if 'axis_align_matrix' in ann:                       # ScanNet (scannet_dataset.py:100-117); S3DIS / SUN RGB-D have none
            out['axis_align_matrix'] = np.asarray(ann['axis_align_matrix'], np.float32)
        for k in ('pts_instance_mask_path', 'pts_semantic_mask_path'):      # S3DIS / ScanNet (s3dis_dataset.py:91-99)
            if k in info:
                out[k] = os.path.join(self.data_root, info[k])
        return out

    def load(self, index, device=None, points_file=None):
        import os
        info = self.data_infos[index]
        path = points_file or os.path.join(self.data_root, info['pts_path'])
        ann = self.get_ann_info(index)
        pts = load_points_from_file(path, self.load_dim, self.use_dim, device)
        if 'axis_align_matrix' in ann:
            pts = global_alignment(pts, ann['axis_align_matrix'])
        boxes = ann['gt_bboxes_3d'].tensor
        if device is not None:
            boxes = boxes.to(device); ann['gt_labels_3d'] = ann['gt_labels_3d'].to(device)
        return pts, boxes, ann['gt_labels_3d'], dict(sample_idx=info['point_cloud']['lidar_idx'], file_name=path)


# ---- augmentation fused with the voxelisation (csrc/coords.hip k_augment_voxelize) ------------------------------------
class LazyAugmentedPoints:
    """A scene whose train-time augmentation has been DRAWN but not applied: the raw points on the device, the sample
    indices and the transform parameters.  `SingleStageSparse3DDetector.voxelize` hands it to `fc_augment_voxelize`, which
    aligns / samples / flips / rotates / scales / translates in registers and writes voxel coordinates + features
    directly — the augmented cloud (2.4 MB per scene, read once by the voxelisation and never again) is not materialised.
    `.materialize()` returns it anyway (tests, visualisation)."""

    def __init__(self, raw, sample_idx, params, align=None):
        assert raw.is_cuda, 'the fused input pipeline runs on the GPU (HIP); use TrainAugment.__call__ on the CPU'
        self.raw = raw.contiguous()
        self.sample_idx = sample_idx.to(torch.int32).contiguous() if sample_idx is not None else None
        self.params, self.align = params, align
        n = self.sample_idx.numel() if self.sample_idx is not None else raw.shape[0]
        self.shape = (n, raw.shape[1])
        self.device = raw.device

    def xform(self):
        x = np.zeros(24, np.float32)
        if self.align is not None:
            m = np.asarray(self.align, np.float32)
            x[0:9] = m[:3, :3].reshape(-1)
            x[9:12] = m[:3, 3]
            x[12] = 1.0
        p = self.params
        x[13], x[14] = float(p.get('flip_h', False)), float(p.get('flip_v', False))
        ang = p.get('angle', 0.0)
        x[15], x[16] = math.cos(ang), math.sin(ang)
        x[17] = p.get('scale', 1.0)
        x[18:21] = np.asarray(p.get('trans', (0.0, 0.0, 0.0)), np.float32)
        return x

    def voxelize_into(self, batch_idx, voxel_size, feat_div, coords, feats, points_out=None):
        from . import _lib as L
        nfeat = self.raw.shape[1] - 3
        x = self.xform()
        L.call('fc_augment_voxelize', L.ptr(self.raw), self.raw.shape[0], self.raw.shape[1], L.ptr(self.sample_idx), self.shape[0],
               x.ctypes.data, batch_idx, float(voxel_size), float(feat_div), nfeat, L.ptr(coords), L.ptr(feats),
               L.ptr(points_out), L.stream())

    def materialize(self):
        out = torch.empty(self.shape, dtype=torch.float32, device=self.device)
        coords = torch.empty((self.shape[0], 4), dtype=torch.int32, device=self.device)
        feats = torch.empty((self.shape[0], self.shape[1] - 3), dtype=torch.float32, device=self.device)
        self.voxelize_into(0, 1.0, 1.0, coords, feats, out)
        return out


def _train_augment_lazy(self, points, boxes, generator=None, align=None):
    """TrainAugment on the GPU without materialising the augmented cloud: draws the sample and the transform, moves the
    (few) GT boxes now, and returns (LazyAugmentedPoints, boxes, params) — pass the first as the scene's `points`."""
    _, idx = indoor_point_sample(points[:, :1], self.num_points, generator)
    p = self.draw(generator, points.device)
    empty = points[:0]
    if p['flip_h']:
        _, boxes = flip_bev(empty, boxes, 'horizontal', self.with_yaw)
    if p['flip_v']:
        _, boxes = flip_bev(empty, boxes, 'vertical', self.with_yaw)
    _, boxes = rot_scale_trans(empty, boxes, p['angle'], p['scale'], p['trans'], self.with_yaw)
    lazy = LazyAugmentedPoints(points, idx, p, align)
    from .sparse import mark_inputs_ready
    mark_inputs_ready(points.device)
    return lazy, boxes, p


TrainAugment.lazy = _train_augment_lazy


# ---- the reference's pipeline classes by name (configs/fcaf3d/*.py train_pipeline / test_pipeline) ----------------------
# `results` dicts carry the reference's keys (pts_filename, ann_info, points, gt_bboxes_3d, gt_labels_3d, bbox3d_fields,
# pcd_horizontal_flip, pcd_vertical_flip, pcd_rotation, pcd_scale_factor, pcd_trans); points are (n, 3+C) tensors, boxes
# `DepthInstance3DBoxes`.  Random draws come from numpy's global generator in the reference's call order, so a scene
# processed under np.random.seed(s) takes the draws the reference's classes would take.
from .registry import Registry  # noqa: E402

PIPELINES = Registry('pipeline')


@PIPELINES.register_module()
class LoadPointsFromFile:
    """mmdet3d/datasets/pipelines/loading.py:333-442 (coord_type='DEPTH')"""

    def __init__(self, coord_type='DEPTH', load_dim=6, use_dim=(0, 1, 2), shift_height=False, use_color=False,
                 file_client_args=None, device=None):
        assert coord_type == 'DEPTH' and not shift_height, 'FCAF3D loads DEPTH-mode points without a height channel'
        self.load_dim, self.use_dim, self.device = load_dim, tuple(range(use_dim)) if isinstance(use_dim, int) else tuple(use_dim), device
        assert max(self.use_dim) < load_dim

    def __call__(self, results):
        results['points'] = load_points_from_file(results['pts_filename'], self.load_dim, self.use_dim, self.device)
        return results


@PIPELINES.register_module()
class LoadAnnotations3D:
    """mmdet3d/datasets/pipelines/loading.py:456-640: the 3D boxes and labels of `ann_info` become pipeline fields
    (with_bbox_3d / with_label_3d, the only switches the FCAF3D configs use; point-wise masks are not loaded)."""

    def __init__(self, with_bbox_3d=True, with_label_3d=True, with_mask_3d=False, with_seg_3d=False, **unused):
        assert not with_mask_3d and not with_seg_3d, 'detection pipeline: no point-wise masks'
        self.with_bbox_3d, self.with_label_3d = with_bbox_3d, with_label_3d

    def __call__(self, results):
        if self.with_bbox_3d:
            results['gt_bboxes_3d'] = results['ann_info']['gt_bboxes_3d']
            results['bbox3d_fields'].append('gt_bboxes_3d')
        if self.with_label_3d:
            results['gt_labels_3d'] = results['ann_info']['gt_labels_3d']
        return results


@PIPELINES.register_module()
class GlobalAlignment:
    """transforms_3d.py:409-490"""

    def __init__(self, rotation_axis):
        self.rotation_axis = rotation_axis

    def __call__(self, results):
        assert 'axis_align_matrix' in results['ann_info'], 'axis_align_matrix is not provided in GlobalAlignment'
        results['points'] = global_alignment(results['points'], results['ann_info']['axis_align_matrix'], self.rotation_axis)
        return results


@PIPELINES.register_module()
class IndoorPointSample:
    """transforms_3d.py:821-895: np.random.choice(n, num_points, replace = n < num_points)"""

    def __init__(self, num_points):
        self.num_points = num_points

    def __call__(self, results):
        pts = results['points']
        n = pts.shape[0]
        choices = np.random.choice(n, self.num_points, replace=n < self.num_points)
        results['points'] = pts[torch.from_numpy(choices).to(pts.device)]
        return results


def _boxes_of(results):
    keys = results['bbox3d_fields']
    assert len(keys) <= 1
    return keys[0] if keys else None


@PIPELINES.register_module()
class RandomFlip3D:
    """transforms_3d.py:59-170 with sync_2d=False: one uniform draw for the (absent) 2D flip of mmdet's RandomFlip, then
    one per BEV direction."""

    def __init__(self, sync_2d=True, flip_ratio_bev_horizontal=0.0, flip_ratio_bev_vertical=0.0, **unused):
        assert not sync_2d, 'FCAF3D flips point clouds only (sync_2d=False)'
        self.h, self.v = flip_ratio_bev_horizontal, flip_ratio_bev_vertical

    def __call__(self, results):
        np.random.rand()                                        # mmdet RandomFlip.__call__: the image flip draw
        if 'pcd_horizontal_flip' not in results:
            results['pcd_horizontal_flip'] = bool(np.random.rand() < self.h)
        if 'pcd_vertical_flip' not in results:
            results['pcd_vertical_flip'] = bool(np.random.rand() < self.v)
        key = _boxes_of(results)
        for flag, direction in (('pcd_horizontal_flip', 'horizontal'), ('pcd_vertical_flip', 'vertical')):
            if results[flag]:
                b = results[key] if key else None
                t = b.tensor if b is not None else results['points'].new_zeros((0, 7))
                results['points'], t = flip_bev(results['points'], t, direction, b.with_yaw if b is not None else True)
                if b is not None:
                    b.tensor = t
        return results


@PIPELINES.register_module()
class GlobalRotScaleTrans:
    """transforms_3d.py:493-645: rotation draw (applied only when the scene has boxes, as the reference's
    `_rot_bbox_points` does when `bbox3d_fields` is set), scale draw, translation draw — in that order."""

    def __init__(self, rot_range=(-0.78539816, 0.78539816), scale_ratio_range=(0.95, 1.05), translation_std=(0, 0, 0),
                 shift_height=False):
        assert not shift_height
        self.rot_range = (-rot_range, rot_range) if isinstance(rot_range, (int, float)) else tuple(rot_range)
        self.scale_ratio_range = tuple(scale_ratio_range)
        self.translation_std = (translation_std,) * 3 if isinstance(translation_std, (int, float)) else tuple(translation_std)

    def __call__(self, results):
        key = _boxes_of(results)
        b = results[key] if key else None
        angle = float(np.random.uniform(self.rot_range[0], self.rot_range[1]))
        pts = results['points']
        t = b.tensor if b is not None else pts.new_zeros((0, 7))
        with_yaw = b.with_yaw if b is not None else True
        if b is None or len(b.tensor) != 0:
            pts, t = rotate(pts, t, angle, with_yaw)
            results['pcd_rotation'] = _rot_z_T(angle, pts)
        if 'pcd_scale_factor' not in results:
            results['pcd_scale_factor'] = float(np.random.uniform(self.scale_ratio_range[0], self.scale_ratio_range[1]))
        trans = np.random.normal(scale=np.array(self.translation_std, dtype=np.float32), size=3).T
        pts, t = rot_scale_trans(pts, t, 0.0, results['pcd_scale_factor'], trans, with_yaw)
        results['pcd_trans'] = trans
        results['points'] = pts
        if b is not None:
            b.tensor = t
        return results


@PIPELINES.register_module()
class DefaultFormatBundle3D:
    """formating.py:177-250 for point clouds: tensors stay tensors (mmcv's DataContainer wrapping is the dataloader's
    business, not the path's); labels become int64 tensors."""

    def __init__(self, class_names=None, with_gt=True, with_label=True):
        self.class_names, self.with_label = class_names, with_label

    def __call__(self, results):
        if 'gt_labels_3d' in results and not torch.is_tensor(results['gt_labels_3d']):
            results['gt_labels_3d'] = torch.as_tensor(np.asarray(results['gt_labels_3d'])).long()
        return results


@PIPELINES.register_module()
class Collect3D:
    """formating.py:253-330: the listed keys plus `img_metas` (box_type_3d, sample_idx, file name, the transform record)."""

    META = ('box_type_3d', 'sample_idx', 'pts_filename', 'pcd_horizontal_flip', 'pcd_vertical_flip', 'pcd_rotation',
            'pcd_scale_factor', 'pcd_trans')

    def __init__(self, keys, meta_keys=None):
        self.keys, self.meta_keys = tuple(keys), tuple(meta_keys) if meta_keys is not None else self.META

    def __call__(self, results):
        out = {k: results[k] for k in self.keys}
        out['img_metas'] = {k: results[k] for k in self.meta_keys if k in results}
        return out


class Compose:
    """mmdet's Compose over the classes above: `Compose(cfg.train_pipeline)(dataset.pre_pipeline(index))`"""

    def __init__(self, cfgs, device=None):
        self.transforms = []
        for c in cfgs:
            c = dict(c)
            if c['type'] == 'LoadPointsFromFile':
                c.setdefault('device', device)
            self.transforms.append(PIPELINES.build(c))

    def __call__(self, results):
        for t in self.transforms:
            results = t(results)
            if results is None:
                return None
        return results


def _pre_pipeline(self, index, points_file=None):
    """Custom3DDataset.get_data_info + pre_pipeline (custom_3d.py:96-147): the `results` dict a pipeline starts from"""
    import os
    from .boxes import DepthInstance3DBoxes
    info = self.data_infos[index]
    return dict(pts_filename=points_file or os.path.join(self.data_root, info['pts_path']),
                sample_idx=info['point_cloud']['lidar_idx'], ann_info=self.get_ann_info(index), bbox3d_fields=[],
                box_type_3d=DepthInstance3DBoxes)


IndoorInfoDataset.pre_pipeline = _pre_pipeline
