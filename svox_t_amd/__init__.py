"""svox_t_amd -- svox_t's differentiable volume-rendering hot path on AMD
MI355X (gfx950): hand-written HIP kernels behind a C ABI (include/svoxt.h),
with the reference's `N3Tree` / `VolumeRenderer` / autograd surface on top.

    import svox_t_amd as svox_t
    tree = svox_t.N3Tree(...); r = svox_t.VolumeRenderer(tree)
    out = r(features, svox_t.Rays(origins, dirs, viewdirs))

Importing this package loads libsvoxt_hip.so; build it first with
`python svox_t_amd/build.py` (there is no CPU fallback).
"""
from svox_t_amd.helpers import DataFormat, LocalIndex, N3TreeView  # noqa: F401
from svox_t_amd.svox import (N3Tree, blend_transformation_matrix, get_transformation_matrix,  # noqa: F401
                            warp_vertices)
from svox_t_amd.renderer import NDCConfig, Rays, VolumeRenderer  # noqa: F401

__version__ = "0.1.0"
__all__ = ["N3Tree", "N3TreeView", "VolumeRenderer", "Rays", "NDCConfig",
           "DataFormat", "LocalIndex", "get_transformation_matrix", "warp_vertices",
           "blend_transformation_matrix"]
