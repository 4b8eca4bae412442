"""splatter_a_video_amd -- MI355X (gfx950) native differentiable video-Gaussian rasterizer.

Drop-in for the reference's ``dptr.gs`` operator package: hand-written HIP kernels in
``csrc/`` behind the C ABI of ``include/splat_hip.h``, reached through ctypes.  See DESIGN.md.
"""
from . import gs  # noqa: F401

__version__ = "0.1.0"
