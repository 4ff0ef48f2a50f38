"""panovlm_amd — MI355X-native association + residual/Jacobian engine behind PanoVLM's
LidarFeatureAssociate / CostFunction / Optimization call surface.

The compute path is libpvlm.so (hand-written HIP for gfx950 behind the C ABI of include/pvlm.h).
There is NO CPU fallback: importing the bindings without the built library, or creating a
context without a GPU, raises.
"""
from .api import (  # noqa: F401
    Context, ResidualSet, NormalEq, BundleSet, Scan, Comm, RingBatch, PvlmError, lib_path, load_library,
    POINT2PLANE_METER, POINT2PLANE_ANGLE, POINT2LINE_METER, POINT2LINE_ANGLE, PLANE2PLANE_GLOBAL, PLANE_IOU,
    FLAG_NORMALIZE_DISTANCE, FLAG_ASSOC_KEEP_INDICES, FLAG_ASSOC_EXACT_FIT, LOSS_NONE, LOSS_HUBER, PAIR_BLOCK, STRIDE, ABI_SYMBOLS,
)

__all__ = ["Context", "ResidualSet", "NormalEq", "BundleSet", "Scan", "RingBatch", "PvlmError", "lib_path", "load_library"]
