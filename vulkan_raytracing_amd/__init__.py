"""MI355X-native (gfx950) ray-tracing stage — drop-in for the VK_KHR_ray_tracing_pipeline path of
mcan1999/vulkan-raytracing (rgen/rchit/rmiss shaders + driver BLAS/TLAS build and traversal).

The product is the C-ABI library ``librt_mi355x.so`` (include/rt_api.h, csrc/).  This package is the
thin Python mirror used by tests and bench.py: ``api`` binds the C ABI with ctypes, ``host`` binds
the C++ host-side code (OBJ/MTL ingest, camera, animation, skybox decode), ``tiling`` holds the
multi-GPU band sharding.  Nothing here computes rays on the CPU: if the HIP library is missing or
no gfx950 device is present, construction raises.
"""
from . import api, host, tiling  # noqa: F401
from .api import RtContext, RtError  # noqa: F401
