"""gsplat_amd — MI355X (gfx950 / CDNA4) native backend for gsplat's differentiable Gaussian rasterizer.

Public surface = the hot path of nerfstudio-project/gsplat with the same names and contracts:
``rasterization``, ``fully_fused_projection``, ``isect_tiles``, ``isect_offset_encode``,
``rasterize_to_pixels``, ``spherical_harmonics``, ``quat_scale_to_covar_preci`` (+ 2DGS when built).
Importing the package loads ``csrc/libgsplat_amd.so`` (hand-written HIP kernels behind a C ABI,
``include/gsplat_amd.h``) and defines ``torch.ops.gsplat.*``; there is no CPU fallback.
"""
from ._wrapper import (  # noqa: F401
    fully_fused_projection,
    isect_offset_encode,
    isect_tiles,
    quat_scale_to_covar_preci,
    rasterize_to_pixels,
    spherical_harmonics,
)
from .rendering import rasterization  # noqa: F401
from . import distributed  # noqa: F401

__version__ = "0.1.0"


def build_config() -> dict:
    """Same keys as the reference's ``_C.build_config()`` (gsplat/cuda/ext.cpp:83-97)."""
    from . import csrc_shim

    return csrc_shim.build_config()
