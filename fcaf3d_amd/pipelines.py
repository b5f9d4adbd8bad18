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
        ann = self.data_infos[index]['annos']
        if ann['gt_num'] != 0:
            boxes = np.asarray(ann['gt_boxes_upright_depth'], np.float32)
            labels = np.asarray(ann['class']).astype(np.int64)
        else:
            boxes = np.zeros((0, 7 if self.with_yaw else 6), np.float32)
            labels = np.zeros((0,), np.int64)
        gt = DepthInstance3DBoxes(torch.from_numpy(boxes), box_dim=boxes.shape[-1], with_yaw=self.with_yaw,
                                  origin=(0.5, 0.5, 0.5))
        align = np.asarray(ann.get('axis_align_matrix', np.eye(4)), np.float32)
        return dict(gt_bboxes_3d=gt, gt_labels_3d=torch.from_numpy(labels), axis_align_matrix=align)

    def load(self, index, device=None, points_file=None):
        import os
        info = self.data_infos[index]
        path = points_file or os.path.join(self.data_root, info['pts_path'])
        ann = self.get_ann_info(index)
        pts = load_points_from_file(path, self.load_dim, self.use_dim, device)
        pts = global_alignment(pts, ann['axis_align_matrix'])
        boxes = ann['gt_bboxes_3d'].tensor
        if device is not None:
            boxes = boxes.to(device); ann['gt_labels_3d'] = ann['gt_labels_3d'].to(device)
        return pts, boxes, ann['gt_labels_3d'], dict(sample_idx=info['point_cloud']['lidar_idx'], file_name=path)
