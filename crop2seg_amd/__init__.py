"""crop2seg_amd -- MI355X (gfx950) implementation of the Crop2Seg backbone hot path.

U-TAE / TimeUNet_v1 / W-TAE forward + backward as hand-written HIP kernels behind a C ABI
(include/c2s_hip.h, libc2s_hip.so), with the reference's model-constructor / forward() / state_dict surface.
"""
import os as _os

# Two HIP streams per step + the streams RCCL creates need more than the runtime's default 4 hardware queues, or the side
# stream shares a queue with the main one.  Only effective if HIP is not initialised yet.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from .backbones.modules import UTAE, WTAE, TimeUNet_v1  # noqa: E402,F401
from .learning.utils import get_model, weight_init  # noqa: E402,F401

__all__ = ["UTAE", "WTAE", "TimeUNet_v1", "get_model", "weight_init"]
