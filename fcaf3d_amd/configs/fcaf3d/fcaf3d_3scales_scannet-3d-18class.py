"""Three-level variant of the ScanNet model (reference: configs/fcaf3d/fcaf3d_3scales_scannet-3d-18class.py);
voxel size stays the base config's."""
_base_ = ['fcaf3d_scannet-3d-18class.py']
_levels = 3
_stage_channels = (64, 128, 256, 512)

model = dict(backbone=dict(n_outs=_levels),
             neck_with_head=dict(in_channels=_stage_channels[:_levels], assigner=dict(n_scales=_levels)))
