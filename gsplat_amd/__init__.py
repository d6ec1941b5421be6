"""gsplat_amd — MI355X (gfx950 / CDNA4) native backend for gsplat's differentiable Gaussian rasterizer.

Public surface = the hot path of nerfstudio-project/gsplat with the same names and contracts:
``rasterization``, ``fully_fused_projection``, ``isect_tiles``, ``isect_offset_encode``,
``rasterize_to_pixels``, ``spherical_harmonics``, ``quat_scale_to_covar_preci`` (+ 2DGS when built).
Importing the package loads ``csrc/libgsplat_amd.so`` (hand-written HIP kernels behind a C ABI,
``include/gsplat_amd.h``) and defines ``torch.ops.gsplat.*``; there is no CPU fallback.
"""
import importlib

# Lazy attribute loading: `import gsplat_amd.csrc_shim` (the drop-in `gsplat.csrc` module) must
# define the ops WITHOUT attaching our autograd — under the reference's Python its own `_wrapper.py` does that.
_LAZY = {
    "fully_fused_projection": "_wrapper", "isect_offset_encode": "_wrapper", "isect_tiles": "_wrapper",
    "quat_scale_to_covar_preci": "_wrapper", "rasterize_to_pixels": "_wrapper", "spherical_harmonics": "_wrapper",
    "fully_fused_projection_2dgs": "_wrapper", "rasterize_to_pixels_2dgs": "_wrapper", "proj": "_wrapper",
    "spherical_harmonics_l0": "_wrapper", "spherical_harmonics_l1_plus": "_wrapper",
    "rasterize_num_contributing_gaussians": "_wrapper", "rasterize_contributing_gaussian_ids": "_wrapper",
    "rasterize_top_contributing_gaussian_ids": "_wrapper",
    "build_sparse_tile_layout": "_wrapper",
    "isect_tiles_sparse": "_wrapper",
    "rasterize_to_pixels_sparse": "_wrapper",
    "rasterize_num_contributing_gaussians_sparse": "_wrapper",
    "rasterize_contributing_gaussian_ids_sparse": "_wrapper",
    "rasterize_top_contributing_gaussian_ids_sparse": "_wrapper",
    "fully_fused_projection_with_ut": "_wrapper", "rasterize_to_pixels_eval3d": "_wrapper", "rasterize_to_pixels_eval3d_extra": "_wrapper", "world_to_cam": "_wrapper", "has_3dgs": "_wrapper", "has_2dgs": "_wrapper", "has_3dgut": "_wrapper",
    "has_adam": "_wrapper", "has_reloc": "_wrapper", "has_losses": "_wrapper", "has_camera_wrappers": "_wrapper",
    "rasterize_to_indices_in_range": "_wrapper", "rasterize_to_indices_in_range_2dgs": "_wrapper",
    "rasterization": "rendering", "rasterization_2dgs": "rendering", "distributed": "distributed",
    "IsectPathMemory": "_cabi",  # per-caller memory of the intersection's path choice (include/gsplat_amd.h)
    # the training step around the rasterizer
    "SelectiveAdam": "optimizers", "compute_relocation": "relocation", "DefaultStrategy": "strategy",
    "MCMCStrategy": "strategy", "strategy": "strategy", "optimizers": "optimizers", "relocation": "relocation",
    # on-disk formats
    "export_splats": "exporter", "exporter": "exporter", "PngCompression": "compression", "compression": "compression",
}


def __getattr__(name):
    mod = _LAZY.get(name)
    if mod is None:
        raise AttributeError(f"module 'gsplat_amd' has no attribute '{name}'")
    m = importlib.import_module(f"{__name__}.{mod}")
    val = m if name == mod else getattr(m, name)
    globals()[name] = val
    return val


def __dir__():
    return sorted(list(globals()) + list(_LAZY))


__version__ = "0.1.0"


def build_config() -> dict:
    """Same keys as the reference's ``_C.build_config()`` (gsplat/cuda/ext.cpp:83-97)."""
    from . import csrc_shim

    return csrc_shim.build_config()
