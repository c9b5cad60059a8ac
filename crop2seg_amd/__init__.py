"""crop2seg_amd -- MI355X (gfx950) implementation of the Crop2Seg backbone hot path.

U-TAE / TimeUNet_v1 / W-TAE forward + backward as hand-written HIP kernels behind a C ABI
(include/c2s_hip.h, libc2s_hip.so), with the reference's model-constructor / forward() / state_dict surface.
"""
from .backbones.modules import UTAE, WTAE, TimeUNet_v1  # noqa: F401
from .learning.utils import get_model, weight_init  # noqa: F401

__all__ = ["UTAE", "WTAE", "TimeUNet_v1", "get_model", "weight_init"]
