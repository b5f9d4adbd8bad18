"""Two-level variant of the ScanNet model (reference: configs/fcaf3d/fcaf3d_2scales_scannet-3d-18class.py — the
first two backbone stages only, 2 cm voxels): every level-dependent entry follows from `_levels`."""
_base_ = ['fcaf3d_scannet-3d-18class.py']
_levels = 2
_stage_channels = (64, 128, 256, 512)

voxel_size = 0.02
model = dict(voxel_size=voxel_size,
             backbone=dict(n_outs=_levels),
             neck_with_head=dict(voxel_size=voxel_size, in_channels=_stage_channels[:_levels],
                                 assigner=dict(n_scales=_levels)))
