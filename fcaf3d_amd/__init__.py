"""fcaf3d_amd — MI355X-native sparse-voxel hot path of FCAF3D (HIP kernels behind a C ABI).

Importing the package registers the reference's plugin names (MEResNet3D, Fcaf3DNeckWithHead,
Fcaf3DAssigner, IoU3DLoss, SingleStageSparse3DDetector, ...) in the shim registries."""
__version__ = '0.1.0'

from .registry import (BACKBONES, BBOX_ASSIGNERS, DETECTORS, HEADS, LOSSES, Config, build_assigner,  # noqa: F401
                       build_backbone, build_detector, build_head, build_loss, build_model)
from . import losses  # noqa: F401,E402
from .me_resnet import MEResNet3D  # noqa: F401,E402
from .fcaf3d_neck_with_head import Fcaf3DAssigner, Fcaf3DNeckWithHead, compute_centerness  # noqa: F401,E402
from .single_stage_sparse import SingleStageSparse3DDetector  # noqa: F401,E402
from .boxes import DepthInstance3DBoxes, bbox3d2result  # noqa: F401,E402
from .checkpoint import load_checkpoint, save_checkpoint  # noqa: F401,E402
from . import runner  # noqa: F401,E402

import os as _os

CONFIG_DIR = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), 'configs', 'fcaf3d')


def get_config(name, voxel_size=None):
    """Load configs/fcaf3d/<name>.py; `voxel_size` overrides all three places the reference sets it
    (top level, model.voxel_size, model.neck_with_head.voxel_size — fcaf3d_2scales...py:2-10)."""
    cfg = Config.fromfile(_os.path.join(CONFIG_DIR, name if name.endswith('.py') else name + '.py'))
    if voxel_size is not None:
        cfg['voxel_size'] = voxel_size
        cfg['model']['voxel_size'] = voxel_size
        cfg['model']['neck_with_head']['voxel_size'] = voxel_size
    return cfg
