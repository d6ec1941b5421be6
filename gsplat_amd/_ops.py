"""torch.ops.gsplat.* — the reference's dispatcher boundary, implemented on the MI355X C-ABI.

Importing this module defines the reference's operator schemas (verbatim from
``gsplat/cuda/ext.cpp:983-1213`` for the ops on the hot path) in the ``gsplat`` namespace and
registers implementations for the ``CUDA`` dispatch key (HIP tensors carry that key on ROCm).
Each implementation owns what the reference's C++ host op owns — shape checks, output
allocation from the caching allocator, stream lookup, error translation — and hands raw device
pointers to ``libgsplat_amd.so`` (``include/gsplat_amd.h``). Autograd is attached separately
(``_autograd.py``), mirroring the reference's split between ``ext.cpp`` and ``_wrapper.py``.

There is no CPU implementation: calling these ops with CPU tensors raises.
"""
from __future__ import annotations

import math
import os
import threading
from typing import Optional, Tuple

import torch
from torch import Tensor

from . import _cabi
from ._cabi import call, ptr, ptr_strided

NS = "gsplat"

# If the reference's own extension is already loaded in this process, its TORCH_LIBRARY owns the
# namespace; we then FRAGMENT-define only the missing ops (none, normally) and override impls.
_lib_def = torch.library.Library(NS, "FRAGMENT")
_lib_impl = torch.library.Library(NS, "IMPL", "CUDA")
_lib_impl_autograd = torch.library.Library(NS, "IMPL", "AutogradCUDA")  # whole-pipeline ops only (see COMPOSITE_SCHEMAS)



def _load_torch_classes() -> Optional[str]:
    """Load libgsplat_amd_torch.so (csrc/torch_classes.cpp): the torch custom classes that the composite schemas below
    name by type (``torch.classes.gsplat.UnscentedTransformParameters`` ...). Returns None on success, else the reason
    the composite ops are not defined (every stage op stays available without it)."""
    import os

    path = os.environ.get("GSPLAT_AMD_TORCH_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc",
                                                                  "libgsplat_amd_torch.so")
    try:
        torch.classes.gsplat.UnscentedTransformParameters  # already registered (e.g. by the reference's own extension)
        return None
    except RuntimeError:
        pass
    if not os.path.exists(path):
        return f"{path} not built (make -C gsplat_amd/csrc torch)"
    try:
        torch.classes.load_library(path)
    except OSError as e:
        return f"cannot load {path}: {e}"
    _read_compiled_ops(path)
    return None


_COMPILED_ISECT = False  # torch.ops.gsplat_amd.isect_fused_{begin,finish} available (set by _read_compiled_ops)

# ---- longest tile list: a hint from the orchestrator to the compositing ops ------------------------------------------------
# rendering.py learns the longest tile list from the intersection's pinned host words; the reference's op schemas have no
# room for it, so it travels as a per-thread hint around the op call: above SEG_MIN_LONGEST the forward / backward cut long
# lists into segments that separate workgroups composite (csrc/raster3d_seg.hip). No hint (0) = one workgroup per tile.
SEG_LEN = int(os.environ.get("GSPLAT_AMD_SEG_LEN", "768"))  # 0 switches segmenting off (A/B); 768: profiles/r11_ab.md #2
SEG_MIN_LONGEST = 2 * SEG_LEN if SEG_LEN > 0 else 1 << 62  # lower bound of the cut (csrc/raster3d_seg.hip: seg_cut_for)


def _seg_cut(n_isects: int, n_images: int, tw: int, th: int) -> int:
    """Lists longer than this are cut into slices: max(2 slices, 3 x the mean list) - segments are for outliers."""
    return _cabi._lib.gsx_raster3d_seg_cut(int(n_isects), int(n_images), int(tw), int(th), SEG_LEN) if SEG_LEN > 0 else 1 << 62

_ROWS_FILL_TORCH = os.environ.get("GSPLAT_AMD_ROWS_FILL", "") == "torch"  # A/B switch: torch.zeros in front of the compositing backward
_hint = __import__("threading").local()
_set_hint_compiled = None  # gsx_torch_set_long_tile_hint of libgsplat_amd_torch.so (the compiled op bodies read it)


def long_tile_hint() -> int:
    return getattr(_hint, "longest", 0)


def long_tile_hint_of_call() -> int:
    """What the caller set around the op call in flight - still readable after the op body consumed long_tile_hint()
    (the autograd setup_context stores it for the backward)."""
    return getattr(_hint, "of_call", 0)


def set_long_tile_hint(longest: int) -> None:
    """Called by the WRAPPER around an op call (and with 0 in its finally block)."""
    _hint.longest = _hint.of_call = int(longest)
    if _set_hint_compiled is not None:
        _set_hint_compiled(int(longest))


# The compositing kernels' 48-byte array-of-structures rows (csrc/raster3d.hpp: Raster3DArgs::splat_rows) travel like the long-list
# hint: rasterization() has them written by its SH forward (gsx_sh_fwd_rows) and the WRAPPER announces them around the op call -
# the reference's op schema has no room for a thirteenth tensor. A body only uses rows announced for ITS means2d.
_set_rows_compiled = None  # gsx_torch_set_splat_rows of libgsplat_amd_torch.so


def set_splat_rows_hint(rows: Optional[Tensor], means2d: Optional[Tensor]) -> None:
    _hint.rows = _hint.rows_of_call = None if rows is None else (rows, means2d.data_ptr())
    if _set_rows_compiled is not None:
        _set_rows_compiled(0 if rows is None else rows.data_ptr(), 0 if rows is None else means2d.data_ptr())


def splat_rows_hint_of_call() -> Optional[Tensor]:
    r = getattr(_hint, "rows_of_call", None)
    return None if r is None else r[0]


def _consume_splat_rows_hint(means2d: Tensor, D: int) -> Optional[Tensor]:
    r = getattr(_hint, "rows", None)
    _hint.rows = None
    if r is None or D != 3 or r[1] != means2d.data_ptr() or r[0].shape[0] * 2 != means2d.numel():
        return None
    return r[0]


def _consume_long_tile_hint() -> int:
    """Called by an op BODY: returns the hint and clears it for nested calls, but leaves long_tile_hint_of_call() alone - the
    autograd setup_context runs after the body and stores it for the backward."""
    longest = getattr(_hint, "longest", 0)
    _hint.longest = 0
    return longest

# Stage-level callers (isect_tiles -> isect_offset_encode -> rasterize_to_pixels without rasterization() in between) have no
# orchestrator to carry the hint: the intersection notes the longest list of its result and a compositing call without a hint
# looks its flatten_ids up. A note is keyed by the IDENTITY of the tensor's storage, never by an address the allocator may hand
# out again (csrc/torch_ops.cpp keeps the notes - weak references to the StorageImpl - when the compiled shim is loaded, so that
# Python and compiled bodies see the same ones; the fallback below keeps the noted tensors of its 16-entry ring alive instead).
_notes_compiled = False
_notes_py: list = []  # [((StorageImpl address, storage offset, numel), the tensor (kept alive), longest)], newest last
_notes_lock = __import__("threading").Lock()


def _note_key(t: Tensor):
    st = t.untyped_storage()
    return (st._cdata, t.storage_offset(), t.numel())


def _note_longest(flatten_ids: Tensor, longest: int) -> None:
    if flatten_ids.numel() == 0:
        return
    if _notes_compiled:
        torch.ops.gsplat_amd.note_longest(flatten_ids, int(longest))
        return
    # fallback (no compiled shim, or GSPLAT_AMD_LIB set): keyed by the StorageImpl's address + view, with a STRONG reference
    # to the tensor in a 16-entry ring - the address cannot be handed out again while its note is alive, and the note does not
    # depend on torch preserving the Python wrapper of a storage between calls
    key = _note_key(flatten_ids)
    with _notes_lock:
        _notes_py[:] = [e for e in _notes_py if e[0] != key][-15:]
        _notes_py.append((key, flatten_ids, int(longest)))


def _lookup_longest(flatten_ids: Tensor) -> int:
    if _notes_compiled:
        return int(torch.ops.gsplat_amd.lookup_longest(flatten_ids))
    if flatten_ids.numel() == 0:
        return 0
    key = _note_key(flatten_ids)
    with _notes_lock:
        for k, _keep, longest in _notes_py:
            if k == key:
                return longest
    return 0


# The segment workspace of a compositing forward, for the backward over the same lists (csrc/raster3d_seg.hip:
# gsx_raster3d_bwd_seg_reuse). The notes live in libgsplat_amd_torch.so (csrc/torch_ops.cpp: keyed by the identity of last_ids,
# shared with the compiled op bodies); without the compiled shim the backward simply runs its pre-pass.
_SEG_REUSE = os.environ.get("GSPLAT_AMD_SEG_REUSE", "1") not in ("0", "")


def _note_seg_workspace(last_ids: Tensor, ws: Tensor, n_isects: int, D: int, inputs) -> None:
    """`inputs`: (means2d, conics, colors, opacities, isect_offsets, flatten_ids) as the op received them - the note only serves
    a backward that brings the same tensors, unwritten since (address + version counter)."""
    if _SEG_REUSE and _notes_compiled and hasattr(torch.ops.gsplat_amd, "note_seg_workspace"):
        torch.ops.gsplat_amd.note_seg_workspace(last_ids, ws, int(n_isects), int(D), SEG_LEN, list(inputs))


def _lookup_seg_workspace(last_ids: Tensor, n_isects: int, D: int, inputs) -> Optional[Tensor]:
    if _SEG_REUSE and _notes_compiled and hasattr(torch.ops.gsplat_amd, "lookup_seg_workspace"):
        return torch.ops.gsplat_amd.lookup_seg_workspace(last_ids, int(n_isects), int(D), SEG_LEN, list(inputs))
    return None


COMPILED_OPS: frozenset = frozenset()  # ops whose CUDA-key body is C++ (csrc/torch_ops.cpp) rather than a function of this file


def _read_compiled_ops(path: str) -> None:
    """libgsplat_amd_torch.so also carries compiled op bodies (csrc/torch_ops.cpp: TORCH_LIBRARY_IMPL(gsplat, CUDA) for the
    hot stage ops, INTEGRATION.md route B). It names them through gsx_torch_compiled_ops(); _register() below keeps the
    Python body of every OTHER op. GSPLAT_AMD_COMPILED_OPS=0 puts the Python bodies back (A/B of the host overhead)."""
    global COMPILED_OPS
    import ctypes

    global _notes_compiled
    _notes_compiled = hasattr(torch.ops, "gsplat_amd") and hasattr(torch.ops.gsplat_amd, "note_longest")
    if os.environ.get("GSPLAT_AMD_LIB"):
        # An A/B build of the kernel library is in use (tools/mkvariant.sh): the compiled bodies are linked against the
        # DEFAULT libgsplat_amd.so and would silently run its kernels instead - keep every op on the ctypes path.
        return
    try:
        fn = ctypes.CDLL(path).gsx_torch_compiled_ops
    except (OSError, AttributeError):
        return
    fn.restype = ctypes.c_char_p
    COMPILED_OPS = frozenset(fn().decode().split())
    global _COMPILED_ISECT, _set_hint_compiled
    try:
        _set_hint_compiled = ctypes.CDLL(path).gsx_torch_set_long_tile_hint
        global _set_rows_compiled
        _set_rows_compiled = ctypes.CDLL(path).gsx_torch_set_splat_rows
        _set_rows_compiled.argtypes = [ctypes.c_uint64, ctypes.c_uint64]
        _set_rows_compiled.restype = None
        _set_hint_compiled.argtypes, _set_hint_compiled.restype = [ctypes.c_int64], None
    except (OSError, AttributeError):
        _set_hint_compiled = None
    _COMPILED_ISECT = (hasattr(torch.ops, "gsplat_amd") and hasattr(torch.ops.gsplat_amd, "isect_fused_begin")
                       and os.environ.get("GSPLAT_AMD_COMPILED_OPS", "1") not in ("0", "")
                       and os.environ.get("GSPLAT_AMD_COMPILED_ISECT", "1") not in ("0", ""))  # A/B switch


COMPOSITE_UNAVAILABLE = _load_torch_classes()

SCHEMAS = {
    # gsplat/cuda/ext.cpp:984-991
    "quat_scale_to_covar_preci": "(Tensor quats, Tensor scales, bool compute_covar, bool compute_preci, bool triu) -> (Tensor?, Tensor?)",
    "quat_scale_to_covar_preci_bwd": "(Tensor quats, Tensor scales, bool triu, Tensor? v_covars, Tensor? v_precis) -> (Tensor, Tensor)",
    # ext.cpp:994-1002
    "spherical_harmonics": "(int degrees_to_use, Tensor means, Tensor viewmats, Tensor coeffs, Tensor? masks, Tensor? batch_ids, Tensor? camera_ids, Tensor? gaussian_ids, Tensor? viewmats_rs=None) -> Tensor",
    "spherical_harmonics_bwd": "(int degrees_to_use, Tensor means, Tensor viewmats, Tensor coeffs, Tensor? masks, Tensor? batch_ids, Tensor? camera_ids, Tensor? gaussian_ids, Tensor? viewmats_rs, Tensor v_colors, bool compute_v_means, bool compute_v_viewmats, bool compute_v_viewmats_rs) -> (Tensor, Tensor?, Tensor?, Tensor?)",
    # ext.cpp:1022-1027
    "intersect_tile": "(Tensor means2d, Tensor radii, Tensor depths, Tensor? conics, Tensor? opacities, Tensor? image_ids, Tensor? gaussian_ids, int? n_images, int tile_size, int tile_width, int tile_height, bool sort, bool segmented) -> (Tensor, Tensor, Tensor)",
    "intersect_offset": "(Tensor isect_ids, int I, int tile_width, int tile_height) -> Tensor",
    # ext.cpp:1052-1077
    "projection_ewa_3dgs_fused": "(Tensor means, Tensor? covars, Tensor? quats, Tensor? scales, Tensor? opacities, Tensor viewmats, Tensor Ks, int image_width, int image_height, float eps2d, float near_plane, float far_plane, float radius_clip, bool calc_compensations, int camera_model) -> (Tensor, Tensor, Tensor, Tensor, Tensor?)",
    "projection_ewa_3dgs_fused_bwd": "(Tensor means, Tensor? covars, Tensor? quats, Tensor? scales, Tensor viewmats, Tensor Ks, int image_width, int image_height, float eps2d, int camera_model, Tensor radii, Tensor conics, Tensor? compensations, Tensor v_means2d, Tensor v_depths, Tensor v_conics, Tensor? v_compensations, bool viewmats_requires_grad) -> (Tensor, Tensor?, Tensor?, Tensor?, Tensor?)",
    "projection_ewa_3dgs_packed": "(Tensor means, Tensor? covars, Tensor? quats, Tensor? scales, Tensor? opacities, Tensor viewmats, Tensor Ks, int image_width, int image_height, float eps2d, float near_plane, float far_plane, float radius_clip, bool sparse_grad, bool calc_compensations, int camera_model) -> (Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor?)",
    "projection_ewa_3dgs_packed_bwd": "(Tensor means, Tensor? covars, Tensor? quats, Tensor? scales, Tensor viewmats, Tensor Ks, int image_width, int image_height, float eps2d, int camera_model, bool sparse_grad, Tensor batch_ids, Tensor camera_ids, Tensor gaussian_ids, Tensor conics, Tensor? compensations, Tensor v_means2d, Tensor v_depths, Tensor v_conics, Tensor? v_compensations, bool viewmats_requires_grad) -> (Tensor, Tensor?, Tensor?, Tensor?, Tensor?)",
    # ext.cpp:1079-1089
    "rasterize_to_pixels_3dgs": "(Tensor means2d, Tensor conics, Tensor colors, Tensor opacities, Tensor? backgrounds, Tensor? masks, int image_width, int image_height, int tile_size, Tensor isect_offsets, Tensor flatten_ids, bool packed, bool absgrad) -> (Tensor, Tensor, Tensor, Tensor)",
    "rasterize_to_pixels_3dgs_bwd": "(Tensor means2d, Tensor conics, Tensor colors, Tensor opacities, Tensor? backgrounds, Tensor? masks, Tensor tile_offsets, Tensor flatten_ids, Tensor render_alphas, Tensor last_ids, int image_width, int image_height, int tile_size, bool absgrad, Tensor v_render_colors, Tensor v_render_alphas, bool compute_v_backgrounds) -> (Tensor?, Tensor, Tensor, Tensor, Tensor, Tensor?)",
    # ext.cpp:1003-1014 (split SH), 1043-1050 (proj), 1105-1109 / 1200-1204 (rasterize_to_indices)
    "spherical_harmonics_l0": "(Tensor sh0) -> Tensor",
    "spherical_harmonics_l0_bwd": "(Tensor sh0, Tensor v_colors) -> Tensor",
    "spherical_harmonics_l1_plus": "(int degrees_to_use, Tensor means, Tensor viewmats, Tensor shN, Tensor? masks, Tensor? batch_ids, Tensor? camera_ids, Tensor? gaussian_ids, Tensor? viewmats_rs=None) -> Tensor",
    "spherical_harmonics_l1_plus_bwd": "(int degrees_to_use, Tensor means, Tensor viewmats, Tensor shN, Tensor? masks, Tensor? batch_ids, Tensor? camera_ids, Tensor? gaussian_ids, Tensor? viewmats_rs, Tensor v_colors, bool compute_v_means, bool compute_v_viewmats, bool compute_v_viewmats_rs) -> (Tensor, Tensor?, Tensor?, Tensor?)",
    "projection_ewa_simple": "(Tensor means, Tensor covars, Tensor Ks, int width, int height, int camera_model) -> (Tensor, Tensor)",
    "projection_ewa_simple_bwd": "(Tensor means, Tensor covars, Tensor Ks, int width, int height, int camera_model, Tensor v_means2d, Tensor v_covars2d) -> (Tensor, Tensor)",
    "rasterize_to_indices_3dgs": "(int range_start, int range_end, Tensor transmittances, Tensor means2d, Tensor conics, Tensor opacities, int image_width, int image_height, int tile_size, Tensor tile_offsets, Tensor flatten_ids) -> (Tensor, Tensor, Tensor)",
    "rasterize_to_indices_2dgs": "(int range_start, int range_end, Tensor transmittances, Tensor means2d, Tensor ray_transforms, Tensor opacities, int image_width, int image_height, int tile_size, Tensor tile_offsets, Tensor flatten_ids) -> (Tensor, Tensor, Tensor)",
    # ext.cpp:1163-1199 (2DGS)
    "projection_2dgs_fused": "(Tensor means, Tensor quats, Tensor scales, Tensor viewmats, Tensor Ks, int image_width, int image_height, float eps2d, float near_plane, float far_plane, float radius_clip) -> (Tensor, Tensor, Tensor, Tensor, Tensor)",
    "projection_2dgs_fused_bwd": "(Tensor means, Tensor quats, Tensor scales, Tensor viewmats, Tensor Ks, int image_width, int image_height, Tensor radii, Tensor ray_transforms, Tensor v_means2d, Tensor v_depths, Tensor v_ray_transforms, Tensor v_normals, bool viewmats_requires_grad) -> (Tensor, Tensor, Tensor, Tensor?)",
    "projection_2dgs_packed": "(Tensor means, Tensor quats, Tensor scales, Tensor viewmats, Tensor Ks, int image_width, int image_height, float near_plane, float far_plane, float radius_clip, bool sparse_grad) -> (Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor)",
    "projection_2dgs_packed_bwd": "(Tensor means, Tensor quats, Tensor scales, Tensor viewmats, Tensor Ks, int image_width, int image_height, bool sparse_grad, Tensor batch_ids, Tensor camera_ids, Tensor gaussian_ids, Tensor ray_transforms, Tensor v_means2d, Tensor v_depths, Tensor v_ray_transforms, Tensor v_normals, bool viewmats_requires_grad) -> (Tensor, Tensor, Tensor, Tensor?)",
    "rasterize_to_pixels_2dgs": "(Tensor means2d, Tensor ray_transforms, Tensor colors, Tensor opacities, Tensor normals, Tensor densify, Tensor? backgrounds, Tensor? masks, int image_width, int image_height, int tile_size, Tensor tile_offsets, Tensor flatten_ids, bool packed, bool absgrad, bool distloss) -> (Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor)",
    "rasterize_to_pixels_2dgs_bwd": "(Tensor means2d, Tensor ray_transforms, Tensor colors, Tensor opacities, Tensor normals, Tensor densify, Tensor? backgrounds, Tensor? masks, Tensor tile_offsets, Tensor flatten_ids, Tensor render_colors, Tensor render_alphas, Tensor last_ids, Tensor median_ids, int image_width, int image_height, int tile_size, bool absgrad, Tensor v_render_colors, Tensor v_render_alphas, Tensor v_render_normals, Tensor v_render_distort, Tensor v_render_median, bool compute_v_backgrounds) -> (Tensor?, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor?)",
    # query rasterizers, dense tile layout: ext.cpp:1111-1134
    "rasterize_num_contributing_gaussians": "(Tensor means2d, Tensor conics, Tensor opacities, Tensor tile_offsets, Tensor flatten_ids, int image_width, int image_height, int tile_size) -> (Tensor, Tensor)",
    "rasterize_contributing_gaussian_ids": "(Tensor means2d, Tensor conics, Tensor opacities, Tensor tile_offsets, Tensor flatten_ids, int image_width, int image_height, int tile_size, Tensor num_contributing_gaussians) -> (Tensor, Tensor)",
    "rasterize_top_contributing_gaussian_ids": "(Tensor means2d, Tensor conics, Tensor opacities, Tensor tile_offsets, Tensor flatten_ids, int image_width, int image_height, int tile_size, int num_depth_samples) -> (Tensor, Tensor)",
    # sparse pixel sets: ext.cpp:1028-1036, 1090-1104, 1115-1140
    "intersect_tile_sparse": "(Tensor means2d, Tensor radii, Tensor depths, Tensor? image_ids, Tensor tile_mask, Tensor active_tiles, int I, int tile_size, int tile_width, int tile_height) -> (Tensor, Tensor)",
    "build_sparse_tile_layout": "(Tensor pixels, Tensor image_ids, int n_images, int tile_size, int tile_width, int tile_height) -> (Tensor, Tensor, Tensor, Tensor, Tensor)",
    "rasterize_to_pixels_sparse": "(Tensor means2d, Tensor conics, Tensor colors, Tensor opacities, Tensor? backgrounds, Tensor? masks, Tensor image_ids, int image_width, int image_height, int tile_size, int tile_width, int tile_height, Tensor active_tiles, Tensor tile_offsets, Tensor flatten_ids, Tensor tile_pixel_mask, Tensor tile_pixel_cumsum, Tensor pixel_map, bool packed, bool absgrad) -> (Tensor, Tensor, Tensor, Tensor)",
    "rasterize_to_pixels_sparse_bwd": "(Tensor means2d, Tensor conics, Tensor colors, Tensor opacities, Tensor? backgrounds, Tensor? masks, Tensor image_ids, Tensor active_tiles, Tensor tile_offsets, Tensor flatten_ids, Tensor tile_pixel_mask, Tensor tile_pixel_cumsum, Tensor pixel_map, Tensor render_alphas, Tensor last_ids, int image_width, int image_height, int tile_size, int tile_width, int tile_height, bool absgrad, Tensor v_render_colors, Tensor v_render_alphas, bool compute_v_backgrounds) -> (Tensor?, Tensor, Tensor, Tensor, Tensor, Tensor?)",
    "rasterize_num_contributing_gaussians_sparse": "(Tensor means2d, Tensor conics, Tensor opacities, int image_width, int image_height, int tile_size, int tile_width, int tile_height, Tensor active_tiles, Tensor tile_offsets, Tensor flatten_ids, Tensor tile_pixel_mask, Tensor tile_pixel_cumsum, Tensor pixel_map) -> (Tensor, Tensor)",
    "rasterize_contributing_gaussian_ids_sparse": "(Tensor means2d, Tensor conics, Tensor opacities, int image_width, int image_height, int tile_size, int tile_width, int tile_height, Tensor active_tiles, Tensor tile_offsets, Tensor flatten_ids, Tensor tile_pixel_mask, Tensor tile_pixel_cumsum, Tensor pixel_map, Tensor num_contributing_gaussians) -> (Tensor, Tensor)",
    "rasterize_top_contributing_gaussian_ids_sparse": "(Tensor means2d, Tensor conics, Tensor opacities, int image_width, int image_height, int tile_size, int tile_width, int tile_height, int num_depth_samples, Tensor active_tiles, Tensor tile_offsets, Tensor flatten_ids, Tensor tile_pixel_mask, Tensor tile_pixel_cumsum, Tensor pixel_map) -> (Tensor, Tensor)",
    # fused feature-row assembly (ext.cpp:1015-1020)
    "assemble_proj_features_unpacked_fwd": "(int degrees_to_use, int B, int C, int N, int Dc, int E, int color_post, int extra_post, bool has_depth, bool depth_is_zero, bool extra_has_c, Tensor means, Tensor viewmats, Tensor? viewmats_rs, Tensor coeffs, Tensor? extra, Tensor? depths, Tensor? masks, Tensor(a!) out, Tensor(b!)? relu_mask) -> ()",
    # training-step ops around the rasterizer: ext.cpp:1217-1221, 1224-1227, 1256-1258
    "adam": "(Tensor(a!) param, Tensor param_grad, Tensor(b!) exp_avg, Tensor(c!) exp_avg_sq, Tensor? valid, float lr, float b1, float b2, float eps) -> ()",
    "relocation": "(Tensor opacities, Tensor scales, Tensor ratios, Tensor binoms, int n_max, float min_opacity=0.0) -> (Tensor, Tensor)",
    "distort_camera_rays": "(Tensor rays, Tensor h_poly, Tensor v_poly, Tensor h_inv_poly, Tensor v_inv_poly, int reference_poly, bool inverse) -> Tensor",
    "eval_bivariate_poly": "(Tensor x, Tensor y, Tensor poly_coeffs, int order) -> Tensor",
    "mcmc_perturb_positions": "(Tensor(a!) positions, Tensor quats, Tensor scales, Tensor opacities, Tensor noise, float noise_scale, float t=0.005, float k=100.) -> ()",
}

# The whole-pipeline ops that gsplat.rasterization() / rasterization_2dgs() call (ext.cpp:1144-1159, 1205-1212). Their
# schemas name torch custom classes, so they are only defined once csrc/torch_classes.cpp is loaded. Registered on the
# CUDA key and, like the reference (Rendering.cpp:1972-1980), on AutogradCUDA with the same function, so that the stage
# ops inside record their own autograd nodes.
COMPOSITE_SCHEMAS = {
    "rasterization_3dgs": "(Tensor means, Tensor? covars, Tensor? quats, Tensor? scales, Tensor opacities, Tensor? colors, Tensor viewmats, Tensor Ks, int image_width, int image_height, int tile_size, float eps2d, float near_plane, float far_plane, float radius_clip, Tensor? backgrounds, bool packed, bool sparse_grad, bool absgrad, bool calc_compensations, bool rasterize_mode_is_classic, int camera_model, bool segmented, int channel_chunk, bool has_color, int sh_degree, Tensor? extra_signals, int extra_signals_sh_degree, bool append_depth, bool expected_depth, bool with_eval3d, bool with_ut, Tensor? rays, Tensor? viewmats_rs, __torch__.torch.classes.gsplat.UnscentedTransformParameters ut_params, int rolling_shutter, Tensor? radial_coeffs, Tensor? tangential_coeffs, Tensor? thin_prism_coeffs, __torch__.torch.classes.gsplat.FThetaCameraDistortionParameters ftheta_coeffs, __torch__.torch.classes.gsplat.RowOffsetStructuredSpinningLidarModelParametersExt? lidar_coeffs, __torch__.torch.classes.gsplat.BivariateWindshieldModelParameters? external_distortion_params, bool global_z_order, bool use_hit_distance, bool return_normals, int renderer_config, str? process_group_name, int world_size) -> (Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, int, int)",
    "rasterization_2dgs": "(Tensor means, Tensor quats, Tensor scales, Tensor opacities, Tensor colors, Tensor viewmats, Tensor Ks, int image_width, int image_height, int tile_size, float eps2d, float near_plane, float far_plane, float radius_clip, Tensor? backgrounds, bool packed, bool sparse_grad, bool absgrad, bool distloss, int? sh_degree, str render_mode, str depth_mode) -> (Tensor, Tensor, Tensor, Tensor?, Tensor, Tensor, Tensor, Tensor?, Tensor?, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, int, int, int)",
}

# Ops whose schemas name the custom classes but carry no gradient (CUDA key only).
CLASS_SCHEMAS = {
    "intersect_tile_lidar": "(__torch__.torch.classes.gsplat.RowOffsetStructuredSpinningLidarModelParametersExt lidar, Tensor means2d, Tensor radii, Tensor depths, Tensor? image_ids, Tensor? gaussian_ids, int? n_images, bool sort, bool segmented) -> (Tensor, Tensor, Tensor)",
    # Unscented-Transform projection of 3DGUT (ext.cpp:1230-1239)
    # from-world compositing of 3DGUT (ext.cpp:1241-1252); differentiable (the body builds its own autograd graph)
    "rasterize_to_pixels_from_world_3dgs": "(Tensor means, Tensor quats, Tensor scales, Tensor colors, Tensor opacities, Tensor? backgrounds, Tensor? masks, int image_width, int image_height, int tile_size, Tensor viewmats0, Tensor? viewmats1, Tensor Ks, int camera_model, __torch__.torch.classes.gsplat.UnscentedTransformParameters ut_params, int rs_type, Tensor? rays, Tensor? radial_coeffs, Tensor? tangential_coeffs, Tensor? thin_prism_coeffs, __torch__.torch.classes.gsplat.FThetaCameraDistortionParameters ftheta_coeffs, __torch__.torch.classes.gsplat.RowOffsetStructuredSpinningLidarModelParametersExt? lidar_coeffs, __torch__.torch.classes.gsplat.BivariateWindshieldModelParameters? external_distortion_params, Tensor tile_offsets, Tensor flatten_ids, bool return_sample_counts, bool use_hit_distance, bool return_normals, int renderer_config, bool return_last_ids, bool unsafe_masked_tile_outputs=False) -> (Tensor, Tensor, Tensor?, Tensor?, Tensor?)",
    "projection_ut_3dgs_fused": "(Tensor means, Tensor quats, Tensor scales, Tensor? opacities, Tensor viewmats0, Tensor? viewmats1, Tensor Ks, int image_width, int image_height, float eps2d, float near_plane, float far_plane, float radius_clip, bool calc_compensations, int camera_model, bool global_z_order, __torch__.torch.classes.gsplat.UnscentedTransformParameters? ut_params, int rs_type, Tensor? radial_coeffs, Tensor? tangential_coeffs, Tensor? thin_prism_coeffs, __torch__.torch.classes.gsplat.FThetaCameraDistortionParameters? ftheta_coeffs, __torch__.torch.classes.gsplat.RowOffsetStructuredSpinningLidarModelParametersExt? lidar_coeffs, __torch__.torch.classes.gsplat.BivariateWindshieldModelParameters? external_distortion_params) -> (Tensor, Tensor, Tensor, Tensor, Tensor?)",
}

_impls = {}


def _op(name):
    def deco(fn):
        _impls[name] = fn
        return fn

    return deco


def _check_f32(**tensors):
    for k, t in tensors.items():
        if t is not None and t.dtype != torch.float32:
            raise TypeError(f"gsplat_amd: {k} must be float32 (got {t.dtype}); the gfx950 kernels compute in fp32")


def _c(t: Optional[Tensor]) -> Optional[Tensor]:
    return None if t is None else t.contiguous()


def _row_view(t: Tensor, width: int):
    """(tensor, row stride in elements) for a [..., width] float tensor whose rows are `width` contiguous elements at
    a uniform stride — e.g. a column view of the AoS gradient buffer returned by rasterize_to_pixels_3dgs_bwd — else a
    contiguous copy with stride `width`. The kernels read such views in place (no gather copy)."""
    if t.is_contiguous():
        return t, width
    if t.dim() >= 2 and t.shape[-1] == width and t.stride(-1) == 1 and t.numel() > 0:
        rs = t.stride(-2)
        ok = rs >= width
        for d in range(t.dim() - 3, -1, -1):
            if t.shape[d] != 1 and t.stride(d) != t.stride(d + 1) * t.shape[d + 1]:
                ok = False
        if ok:
            return t, rs
    return t.contiguous(), width


def _elem_view(t: Tensor):
    """(tensor, element stride) for a float tensor whose elements, taken in row-major order, sit at ONE uniform stride - a
    contiguous tensor (stride 1) or a single column of the compositing backward's gradient rows (their row stride) - else a
    contiguous copy. The kernels read such a column in place."""
    if t.is_contiguous():
        return t, 1
    if t.dim() >= 1 and t.numel() > 0:
        st = t.stride(-1) if t.shape[-1] != 1 else None
        ok, expect = True, None
        for d in range(t.dim() - 1, -1, -1):
            if t.shape[d] == 1:
                continue
            if expect is None:
                st, expect = t.stride(d), t.stride(d) * t.shape[d]
            elif t.stride(d) != expect:
                ok = False
                break
            else:
                expect *= t.shape[d]
        if ok and st is not None and st >= 1:
            return t, int(st)
    return t.contiguous(), 1


def _common_row_views(ts, widths):
    """Three gradient tensors for the 2DGS projection backward: (tensors, common row stride or 0). If every one is a
    column view with the SAME row stride (views of one AoS gradient buffer) they are used in place, otherwise they are
    made contiguous (stride 0 = "contiguous rows" in the C-ABI)."""
    views = [_row_view(t, w) for t, w in zip(ts, widths)]
    strides = {st for (_t, st), w in zip(views, widths) if st != w}
    if len(strides) == 1 and all(st != w for (_t, st), w in zip(views, widths)):
        return [t for t, _ in views], strides.pop()
    return [t.contiguous() for t in ts], 0


_row_map_lock = threading.Lock()
_row_map_cache: list = []  # at most one (key, id tensors, row_map): see _packed_row_map


def clear_row_map_cache() -> None:
    """Drops the cached row map (and with it the references that keep a step's id tensors alive). The packed projection
    backward - the last of the two kernels of a step that walk the map - calls this when it is done, and rasterization() when a
    new forward pass starts: the map only ever serves the packed backward kernels of ONE step."""
    with _row_map_lock:
        _row_map_cache.clear()


def _packed_row_map(batch_ids, camera_ids, gaussian_ids, B: int, C: int, N: int) -> Tensor:
    """int32 [B*C*N]: packed row of every (batch, camera, gaussian), -1 where the pair is not stored. The packed
    backward kernels walk it Gaussian-major (one thread per Gaussian, no atomics). The SH backward and the projection
    backward of a step ask for the same map: the last one is kept, keyed by the id tensors' storage and version - and the
    entry HOLDS those tensors, so that their addresses cannot be handed to other tensors while it is alive."""
    ids = (batch_ids.contiguous(), camera_ids.contiguous(), gaussian_ids.contiguous())
    key = tuple((t.data_ptr(), t.numel(), t._version) for t in ids) + (B, C, N)
    with _row_map_lock:
        for k, _, rm in _row_map_cache:
            if k == key:
                return rm
    row_map = torch.empty(B * C * N, device=gaussian_ids.device, dtype=torch.int32)
    call("gsx_packed_row_map", ptr(ids[0]), ptr(ids[1]), ptr(ids[2]), gaussian_ids.shape[0], B, C, N, ptr(row_map))
    with _row_map_lock:
        _row_map_cache[:] = [(key, ids, row_map)]
    return row_map


def bits_for_count(count: int) -> int:
    return (count - 1).bit_length() if count > 1 else 0


# ----------------------------------------------------------------------------------------------
# quat_scale_to_covar_preci
# ----------------------------------------------------------------------------------------------
def _qs_dtype(*tensors) -> bool:
    """float32 or float64 throughout (the reference dispatches this op over both, QuatScaleToCovarCUDA.cu:145); True = float64."""
    dts = {t.dtype for t in tensors if t is not None}
    if dts == {torch.float64}:
        return True
    if dts != {torch.float32}:
        raise TypeError(f"gsplat_amd: quat_scale_to_covar_preci takes float32 or float64 tensors of ONE type (got {sorted(map(str, dts))})")
    return False


@_op("quat_scale_to_covar_preci")
def quat_scale_to_covar_preci(quats, scales, compute_covar, compute_preci, triu):
    f64 = _qs_dtype(quats, scales)
    batch = quats.shape[:-1]
    if quats.shape[-1] != 4 or scales.shape != batch + (3,):
        raise ValueError(f"quat_scale_to_covar_preci: bad shapes {tuple(quats.shape)} / {tuple(scales.shape)}")
    quats, scales = quats.contiguous(), scales.contiguous()
    n = math.prod(batch)
    tail = (6,) if triu else (3, 3)
    covars = torch.empty(batch + tail, device=quats.device, dtype=quats.dtype) if compute_covar else None
    precis = torch.empty(batch + tail, device=quats.device, dtype=quats.dtype) if compute_preci else None
    call("gsx_quat_scale_to_covar_fwd_f64" if f64 else "gsx_quat_scale_to_covar_fwd", ptr(quats), ptr(scales), n, int(triu),
         ptr(covars), ptr(precis))
    return covars, precis


@_op("quat_scale_to_covar_preci_bwd")
def quat_scale_to_covar_preci_bwd(quats, scales, triu, v_covars, v_precis):
    f64 = _qs_dtype(quats, scales, v_covars, v_precis)
    quats, scales = quats.contiguous(), scales.contiguous()
    n = math.prod(quats.shape[:-1])
    v_quats, v_scales = torch.empty_like(quats), torch.empty_like(scales)
    call("gsx_quat_scale_to_covar_bwd_f64" if f64 else "gsx_quat_scale_to_covar_bwd", ptr(quats), ptr(scales), n, int(triu),
         ptr(_c(v_covars)), ptr(_c(v_precis)), ptr(v_quats), ptr(v_scales))
    return v_quats, v_scales


# ----------------------------------------------------------------------------------------------
# spherical harmonics
# ----------------------------------------------------------------------------------------------
def _sh_dims(means, viewmats, coeffs, gaussian_ids):
    packed = gaussian_ids is not None
    C = viewmats.shape[-3]
    N = means.shape[-2]
    B = math.prod(means.shape[:-2])
    K, D = coeffs.shape[-2], coeffs.shape[-1]
    return packed, B, C, N, K, D


_SH_MAX_DEGREE = 4


def _check_sh_inputs(degrees_to_use, means, viewmats, coeffs, masks, batch_ids, camera_ids, gaussian_ids, omit_l0=False,
                     gathered=True):
    """The reference's input contract, message for message (SphericalHarmonics.cpp:38-131, check_spherical_harmonics_inputs);
    `gathered=False` (private in-place layout: [N, K, D] rows read through gaussian_ids) skips the nnz row count."""
    if not 0 <= degrees_to_use <= _SH_MAX_DEGREE:
        raise ValueError(f"degrees_to_use must be between 0 and {_SH_MAX_DEGREE}, got {degrees_to_use}")
    if means.dim() < 2 or means.shape[-1] != 3:
        raise ValueError(f"means must have shape [..., N, 3], got {tuple(means.shape)}")
    if viewmats.dim() != means.dim() + 1 or viewmats.shape[-2:] != (4, 4):
        raise ValueError(f"viewmats must have shape [..., C, 4, 4], got {tuple(viewmats.shape)}")
    if means.shape[:-2] != viewmats.shape[:-3]:
        raise ValueError("means and viewmats batch dimensions must match")
    if coeffs.dim() != 3:
        raise ValueError(f"coeffs must have shape [N, K, D] or [nnz, K, D], got {tuple(coeffs.shape)}")
    if coeffs.shape[-1] < 1:
        raise ValueError(f"coeffs last dim D must be >= 1, got {coeffs.shape[-1]}")
    if (degrees_to_use + 1) ** 2 - (1 if omit_l0 else 0) > coeffs.shape[-2]:
        raise ValueError(f"degrees_to_use requires more SH coefficients than provided; degree {degrees_to_use}, "
                         f"coeffs shape {tuple(coeffs.shape)}")
    ids = (batch_ids, camera_ids, gaussian_ids)
    packed = any(t is not None for t in ids)
    if packed and not all(t is not None for t in ids):
        raise ValueError("batch_ids, camera_ids, and gaussian_ids must either all be provided or all be None")
    if packed:
        nnz = coeffs.shape[0] if gathered else gaussian_ids.numel()
        for t in ids:
            if t.dim() != 1 or t.numel() != nnz:
                raise ValueError("packed ID tensors must have shape [nnz]")
            if t.dtype != torch.int64:
                raise ValueError("packed ID tensors must be int64")
        if masks is not None and (masks.dim() != 1 or masks.numel() != nnz):
            raise ValueError("packed masks must have shape [nnz]")
        if not gathered and coeffs.shape[0] != means.shape[-2]:
            raise ValueError(f"coefficient rows are indexed by Gaussian: expected {means.shape[-2]} rows, got {coeffs.shape[0]}")
    else:
        if means.shape[-2] != coeffs.shape[0]:
            raise ValueError("means N must match coeffs N in dense mode")
        if masks is not None and tuple(masks.shape) != tuple(viewmats.shape[:-2]) + (means.shape[-2],):
            raise ValueError("dense masks must have shape [..., C, N]")


def _sh_rs_viewmats(viewmats, viewmats_rs):
    """Rolling-shutter SH (reference SphericalHarmonics.cuh:40-65): the view direction is mean + offset with the camera
    offset R^T t AVERAGED over the two shutter endpoints. An equivalent global-shutter view matrix (identity rotation,
    t = that average) lets the kernels run unchanged."""
    if tuple(viewmats_rs.shape) != tuple(viewmats.shape):
        raise ValueError("viewmats_rs must match viewmats shape")
    _check_f32(viewmats_rs=viewmats_rs)

    def offset(vm):
        return torch.einsum("...ji,...j->...i", vm[..., :3, :3], vm[..., :3, 3])

    syn = torch.zeros_like(viewmats)
    for d in range(4):
        syn[..., d, d] = 1.0
    syn[..., :3, 3] = 0.5 * (offset(viewmats) + offset(viewmats_rs))
    return syn


def _sh_rs_split(v_syn, viewmats, viewmats_rs, want, want_rs):
    """Gradient of the synthetic matrix -> the two endpoints: v_syn[:3, 3] = S = sum over rows of v_dir (identity rotation);
    offset_j = sum_i R_ij t_i, each endpoint weighs 1/2: v_R = t (x) S / 2, v_t = R S / 2."""
    S = v_syn[..., :3, 3]

    def back(vm):
        g = torch.zeros_like(vm)
        g[..., :3, :3] = 0.5 * vm[..., :3, 3][..., :, None] * S[..., None, :]
        g[..., :3, 3] = 0.5 * torch.einsum("...ij,...j->...i", vm[..., :3, :3], S)
        return g

    return (back(viewmats) if want else None), (back(viewmats_rs) if want_rs else None)


@_op("spherical_harmonics")
def spherical_harmonics(degrees_to_use, means, viewmats, coeffs, masks, batch_ids, camera_ids, gaussian_ids,
                        viewmats_rs=None, *, _gathered: bool = True, _radii=None, _post: bool = False, _splat=None):
    """Private keywords (used by rendering.py, not part of the reference schema): `_gathered=False` reads [N,K,D]
    coefficients through gaussian_ids; `_radii` masks rows by radii > 0 instead of a bool tensor; `_post` fuses the
    orchestrator's `clamp_min(colors + 0.5, 0)`; `_splat=(means2d, conics, opacities, rows)` (D == 3, float32 coefficients): the
    kernel also writes the compositing kernels' 48-byte array-of-structures row of every live row into `rows` [n_rows, 12]
    (gsx_sh_fwd_rows)."""
    if viewmats_rs is not None:
        viewmats = _sh_rs_viewmats(viewmats, viewmats_rs)
    _check_sh_inputs(degrees_to_use, means, viewmats, coeffs, masks, batch_ids, camera_ids, gaussian_ids, gathered=_gathered)
    if coeffs.dtype == torch.float16:
        # half coefficients, float arithmetic and colours (reference SphericalHarmonicsCUDA.cu:609-638): the band kernels
        # read [N, K, 3] half rows in place; gathered packed rows / D != 3 widen first (rare layouts)
        _check_f32(means=means, viewmats=viewmats)
        packed = gaussian_ids is not None
        if _band_kernels_apply(coeffs) and _radii is None and not _post and (not packed or not _gathered):
            return _sh_band_fwd(degrees_to_use, 0, means, viewmats, coeffs, masks, batch_ids, camera_ids, gaussian_ids)
        return spherical_harmonics(degrees_to_use, means, viewmats, coeffs.float(), masks, batch_ids, camera_ids, gaussian_ids,
                                   None, _gathered=_gathered, _radii=_radii, _post=_post)
    _check_f32(means=means, viewmats=viewmats, coeffs=coeffs)
    packed, B, C, N, K, D = _sh_dims(means, viewmats, coeffs, gaussian_ids)
    means, viewmats, coeffs, masks = means.contiguous(), viewmats.contiguous(), coeffs.contiguous(), _c(masks)
    if coeffs.dim() != 3:
        raise ValueError(f"coeffs must have shape [N, K, D] or [nnz, K, D], got {tuple(coeffs.shape)}")
    fn, tail = "gsx_sh_fwd", ()
    if _splat is not None and D == 3:
        m2, con, op, rows_out = _splat
        n_rows = gaussian_ids.shape[0] if packed else B * C * N
        if rows_out.shape != (n_rows, 12) or not rows_out.is_contiguous() or m2.numel() != 2 * n_rows or con.numel() != 3 * n_rows \
                or op.numel() != n_rows:
            raise ValueError("spherical_harmonics(_splat=...): means2d / conics / opacities / rows do not cover the SH rows")
        fn, tail = "gsx_sh_fwd_rows", (ptr(m2.contiguous()), ptr(con.contiguous()), ptr(op.contiguous()), ptr(rows_out))
    if packed:
        nnz = gaussian_ids.shape[0]
        colors = torch.empty((nnz, D), device=means.device, dtype=means.dtype)
        call(fn, degrees_to_use, ptr(means), ptr(viewmats), ptr(coeffs), ptr(masks), ptr(_c(batch_ids)),
             ptr(_c(camera_ids)), ptr(_c(gaussian_ids)), B, C, N, nnz, int(_gathered), K, D, ptr(_c(_radii)),
             int(_post), ptr(colors), *tail)
    else:
        if coeffs.shape[0] != N:
            raise ValueError("means N must match coeffs N in dense mode")
        colors = torch.empty(viewmats.shape[:-2] + (N, D), device=means.device, dtype=means.dtype)
        call(fn, degrees_to_use, ptr(means), ptr(viewmats), ptr(coeffs), ptr(masks), None, None, None,
             B, C, N, -1, 1, K, D, ptr(_c(_radii)), int(_post), ptr(colors), *tail)
    return colors


@_op("spherical_harmonics_bwd")
def spherical_harmonics_bwd(degrees_to_use, means, viewmats, coeffs, masks, batch_ids, camera_ids, gaussian_ids,
                            viewmats_rs, v_colors, compute_v_means, compute_v_viewmats, compute_v_viewmats_rs,
                            *, _gathered: bool = True, _radii=None, _post_colors=None):
    if viewmats_rs is not None:
        v_co, v_me, v_syn, _ = spherical_harmonics_bwd(
            degrees_to_use, means, _sh_rs_viewmats(viewmats, viewmats_rs), coeffs, masks, batch_ids, camera_ids, gaussian_ids,
            None, v_colors, compute_v_means, compute_v_viewmats or compute_v_viewmats_rs, False, _gathered=_gathered,
            _radii=_radii, _post_colors=_post_colors)
        v_vm, v_rs = (None, None) if v_syn is None else _sh_rs_split(v_syn, viewmats, viewmats_rs, compute_v_viewmats,
                                                                     compute_v_viewmats_rs)
        return v_co, v_me, v_vm, v_rs
    if coeffs.dtype == torch.float16:  # see spherical_harmonics: v_coeffs comes back in the coefficients' own type
        packed = gaussian_ids is not None
        if _band_kernels_apply(coeffs) and _radii is None and _post_colors is None and (not packed or not _gathered):
            v_co, v_me, v_vm = _sh_band_bwd(degrees_to_use, 0, means, viewmats, coeffs, masks, batch_ids, camera_ids,
                                            gaussian_ids, v_colors, compute_v_means, compute_v_viewmats)
            return v_co, v_me, v_vm, None
        v_co, v_me, v_vm, v_rs = spherical_harmonics_bwd(
            degrees_to_use, means, viewmats, coeffs.float(), masks, batch_ids, camera_ids, gaussian_ids, None, v_colors,
            compute_v_means, compute_v_viewmats, False, _gathered=_gathered, _radii=_radii, _post_colors=_post_colors)
        return v_co.half(), v_me, v_vm, v_rs
    packed, B, C, N, K, D = _sh_dims(means, viewmats, coeffs, gaussian_ids)
    means, viewmats, coeffs, masks = means.contiguous(), viewmats.contiguous(), coeffs.contiguous(), _c(masks)
    v_colors, vc_stride = _row_view(v_colors, D)  # may be a column view of the compositing kernel's gradient rows
    # packed rows read through gaussian_ids with D == 3: walk them Gaussian-major through a row map — v_coeffs / v_means
    # are then written once per Gaussian (no atomics when several cameras see a Gaussian, no 4*K*D*N-byte zero fill)
    row_map = None
    if packed and not _gathered and D == 3 and N > 0 and gaussian_ids.shape[0] > 0:
        row_map = _packed_row_map(batch_ids, camera_ids, gaussian_ids, B, C, N)
    need_zero = packed and not _gathered and row_map is None
    v_coeffs = torch.zeros_like(coeffs) if need_zero else torch.empty_like(coeffs)
    # the Gaussian-major D == 3 kernels store v_means for every (b, g) (sh3_bwd_dense_kernel); the other paths accumulate
    v_means = None
    if compute_v_means:
        full_write = D == 3 and N > 0 and (not packed or row_map is not None)
        v_means = torch.empty_like(means) if full_write else torch.zeros_like(means)
    nnz = gaussian_ids.shape[0] if packed else -1
    # pose gradient: the kernel also returns d(loss)/d(view direction) per row; dir = mean + R^T t, so
    # v_R = t (x) sum_rows v_dir and v_t = R sum_rows v_dir per camera (small host-side tensors)
    v_dirs = None
    if compute_v_viewmats:
        n_rows = nnz if packed else B * C * N
        v_dirs = torch.zeros((n_rows, 3), device=means.device, dtype=means.dtype)
    call("gsx_sh_bwd", degrees_to_use, ptr(means), ptr(viewmats), ptr(coeffs), ptr(masks), ptr(_c(batch_ids)),
         ptr(_c(camera_ids)), ptr(_c(gaussian_ids)), B, C, N, nnz, int(_gathered) if packed else 1, K, D,
         ptr(_c(_radii)), ptr(_c(_post_colors)), ptr_strided(v_colors), vc_stride, ptr(row_map), ptr(v_coeffs),
         ptr(v_means), ptr(v_dirs))
    v_viewmats = None
    if compute_v_viewmats:
        if packed:
            S = torch.zeros((B * C, 3), device=means.device, dtype=means.dtype)
            S.index_add_(0, batch_ids * C + camera_ids, v_dirs)
        else:
            S = v_dirs.view(B * C, N, 3).sum(dim=1)
        vm = viewmats.reshape(B * C, 4, 4)
        R, t = vm[:, :3, :3], vm[:, :3, 3]
        v_vm = torch.zeros_like(vm)
        v_vm[:, :3, :3] = t[:, :, None] * S[:, None, :]
        v_vm[:, :3, 3] = torch.einsum("cij,cj->ci", R, S)
        v_viewmats = v_vm.reshape(viewmats.shape)
    return v_coeffs, v_means, v_viewmats, None


# ----------------------------------------------------------------------------------------------
# tile intersection
# ----------------------------------------------------------------------------------------------
def _scan_i32(x: Tensor) -> Tensor:
    """Inclusive int64 prefix sum of an int32 tensor (flattened)."""
    n = x.numel()
    out = torch.empty(n, device=x.device, dtype=torch.int64)
    ws = torch.empty(max(_cabi.scan_workspace_bytes(n), 8), device=x.device, dtype=torch.uint8)
    call("gsx_scan_i32", ptr(x), n, ptr(out), ptr(ws), ws.numel())
    return out


def _isect_fused_count(st, tile_mask):
    """Count half of the fused intersection through the C-ABI: the tile-owner-major path (csrc/isect_binned.hip) when
    st.binned, else the Gaussian-major one (csrc/isect_fused.hip). The grand total is written by the last kernel straight
    into the pinned host word st.host_total (no copy kernel)."""
    means2d, radii, depths, conics, opacities, _ = st.args
    tile_size, tile_width, tile_height = st.geom[:3]
    dev = means2d.device
    tpg = None if st.tiles_per_gauss is None else ptr(st.tiles_per_gauss)
    if st.binned:
        st.count_ws = torch.empty(_cabi.isect_binned_count_workspace_bytes(st.rows, st.I, tile_width, tile_height), device=dev,
                                  dtype=torch.uint8)
        call("gsx_isect_binned_count", ptr(means2d), ptr(radii), ptr(depths), ptr(conics), ptr(opacities), ptr(tile_mask),
             st.rows, st.I, tile_size, tile_width, tile_height, tpg, ptr(st.offsets), _cabi.ptr_host(st.host_total),
             _cabi.ptr_host(st.host_total) + 8, ptr(st.count_ws), st.count_ws.numel())
    else:
        st.count_ws = torch.empty(_cabi.isect_fused_count_workspace_bytes(st.rows, st.I, tile_width, tile_height), device=dev,
                                  dtype=torch.uint8)
        call("gsx_isect_fused_count", ptr(means2d), ptr(radii), ptr(conics), ptr(opacities), ptr(tile_mask), st.rows, st.I,
             tile_size, tile_width, tile_height, tpg, ptr(st.offsets), _cabi.ptr_host(st.host_total),
             _cabi.ptr_host(st.host_total) + 8, ptr(st.count_ws), st.count_ws.numel())


def _isect_fused_emit(st, tile_mask, n_isects):
    """Emit + sort half (after the host read n_isects); returns (isect_ids, flatten_ids)."""
    means2d, radii, depths, conics, opacities, _ = st.args
    tile_size, tile_width, tile_height = st.geom[:3]
    dev = means2d.device
    isect_ids = torch.empty(n_isects, device=dev, dtype=torch.int64)
    flatten_ids = torch.empty(n_isects, device=dev, dtype=torch.int32)
    if n_isects == 0:
        return isect_ids, flatten_ids
    if st.binned:
        ws = torch.empty(_cabi.isect_binned_emit_workspace_bytes(n_isects), device=dev, dtype=torch.uint8)
        call("gsx_isect_binned_emit_sort", st.rows, st.I, tile_size, tile_width, tile_height, ptr(st.count_ws),
             st.count_ws.numel(), ptr(st.offsets), n_isects, int(isect_max_tile_len(st)), ptr(isect_ids), ptr(flatten_ids),
             ptr(ws), ws.numel())
    else:
        ws = torch.empty(_cabi.isect_fused_emit_workspace_bytes(n_isects, st.I, tile_width, tile_height), device=dev,
                         dtype=torch.uint8)
        call("gsx_isect_fused_emit_sort", ptr(means2d), ptr(radii), ptr(depths), ptr(conics), ptr(opacities), ptr(tile_mask),
             st.rows, st.I, tile_size, tile_width, tile_height, ptr(st.count_ws), st.count_ws.numel(), ptr(st.offsets),
             n_isects, ptr(isect_ids), ptr(flatten_ids), ptr(ws), ws.numel())
    return isect_ids, flatten_ids


def _isect_fused_total(st, tile_mask):
    """Host sync on the count; reruns the Gaussian-major count when the binned path reports GSX_ISECT_RETRY (-2)."""
    n_isects = int(st.host_total[0].item())
    if st.binned and n_isects == -2:
        st.binned = False
        _cabi._lib.gsx_isect_binned_note_retry(st.rows, st.I, st.geom[1], st.geom[2])  # not tried again for the next 63 calls
        _isect_fused_count(st, tile_mask)
        torch.cuda.current_stream(st.args[0].device).synchronize()
        n_isects = int(st.host_total[0].item())
    if n_isects >= 2**31:
        raise RuntimeError(f"intersect_tile: {n_isects} intersections overflow the int32 index space")
    return n_isects


class _IsectPending:
    """State between the two halves of intersect_tile (see isect_begin)."""
    __slots__ = ("args", "tiles_per_gauss", "cum", "host_total", "event", "rows", "n_per", "I", "geom", "sort",
                 "fused", "binned", "count_ws", "offsets", "n_dev")


def isect_begin(means2d, radii, depths, conics, opacities, image_ids, gaussian_ids, n_images, tile_size,
                tile_width, tile_height, sort, segmented) -> "_IsectPending":
    """First half of intersect_tile: count tiles per Gaussian, prefix-sum, and START the device->host read of the
    total (pinned buffer + event) without waiting for it. The caller may enqueue independent work (the orchestrator
    runs the SH kernels here) before isect_finish() blocks on the event, so the host round trip for the exact output
    length (reference: the `.item()` at Intersect.cpp:258-259) no longer idles the GPU."""
    f64 = means2d.dtype == torch.float64
    if f64:
        # float64 rows (the reference dispatches this op over float and double): radius boxes in double, keys carry the
        # depth narrowed to float32; the exact ellipse test is fp32 only
        if conics is not None or opacities is not None:
            raise TypeError("gsplat_amd: intersect_tile with float64 rows supports the radius-box test only "
                            "(conics / opacities select the exact test, which is computed in fp32)")
        depths = depths.to(torch.float64)
    else:
        _check_f32(means2d=means2d, depths=depths, conics=conics, opacities=opacities)
    packed = image_ids is not None
    if packed and segmented:
        # the reference refuses the combination (Intersect.cpp:207-211: its packed segment offsets collapse to one segment);
        # `segmented` changes nothing here (the per-tile sort gives the same order), but the contract is the reference's
        raise RuntimeError("segmented sort is not supported for packed inputs")
    means2d, radii, depths = means2d.contiguous(), radii.contiguous(), depths.contiguous()
    if radii.dtype != torch.int32:
        radii = radii.to(torch.int32)
    conics, opacities, image_ids = _c(conics), _c(opacities), _c(image_ids)
    if packed:
        if n_images is None:
            raise ValueError("n_images is required when packed")
        rows, n_per, I = means2d.shape[0], 1, int(n_images)
        out_shape = (rows,)
    else:
        image_dims = means2d.shape[:-2]
        I, n_per = math.prod(image_dims), means2d.shape[-2]
        rows = I * n_per
        out_shape = tuple(means2d.shape[:-1])
    tile_bits, image_bits = bits_for_count(tile_width * tile_height), bits_for_count(I)
    if tile_bits + image_bits > 32:
        raise RuntimeError(
            f"intersect_tile: tile id bits ({tile_bits}) + image id bits ({image_bits}) exceed the 32 bits "
            "available above the depth in the 64-bit sort key"
        )
    dev = means2d.device
    st = _IsectPending()
    st.args = (means2d, radii, depths, conics, opacities, image_ids)
    st.rows, st.n_per, st.I, st.sort = rows, n_per, I, sort
    st.geom = (tile_size, tile_width, tile_height, tile_bits, image_bits)
    st.tiles_per_gauss = torch.empty(out_shape, device=dev, dtype=torch.int32)
    st.cum = st.host_total = st.event = st.count_ws = st.offsets = st.n_dev = None
    st.fused = st.binned = False
    if rows == 0:
        return st
    # sort=True: fused path (csrc/isect_fused.hip) — per-(chunk, tile) histogram while counting, emission straight into
    # tile segments, offsets as a by-product; dense rows of any image count, packed rows of a single image
    st.fused = bool(sort) and not f64 and _cabi.isect_fused_supported(I, tile_width, tile_height, packed)
    if st.fused and _COMPILED_ISECT:
        # compiled halves (csrc/torch_ops.cpp): same launches, ~30 us less interpreter time per step, and the count comes
        # back through a polled pinned word instead of an event
        st.tiles_per_gauss, st.offsets, st.count_ws, st.host_total = torch.ops.gsplat_amd.isect_fused_begin(
            means2d, radii, depths, conics, opacities, rows, I, tile_size, tile_width, tile_height, list(out_shape))
        st.event = "polled"
        return st
    st.host_total = torch.zeros(2, dtype=torch.int64, pin_memory=True)  # [n_isects, longest tile list]
    if st.fused:
        st.offsets = torch.empty(I * tile_width * tile_height, device=dev, dtype=torch.int32)
        st.binned = _cabi.isect_binned_should_try(rows, I, tile_width, tile_height, packed)  # once; st carries it
        _isect_fused_count(st, None)
        st.event = torch.cuda.Event()
        st.event.record()
        return st
    if f64:
        call("gsx_isect_count_f64", ptr(means2d), ptr(radii), ptr(image_ids), rows, n_per, I, tile_size, tile_width,
             tile_height, ptr(st.tiles_per_gauss))
    else:
        call("gsx_isect_count", ptr(means2d), ptr(radii), ptr(conics), ptr(opacities), ptr(image_ids), rows, n_per, I,
             tile_size, tile_width, tile_height, ptr(st.tiles_per_gauss))
    st.cum = _scan_i32(st.tiles_per_gauss)
    st.host_total[:1].copy_(st.cum[-1:], non_blocking=True)
    st.event = torch.cuda.Event()
    st.event.record()
    return st


def isect_max_tile_len(st: "_IsectPending") -> int:
    """Length of the longest tile list (0 when the path taken does not report it). Valid after isect_finish()."""
    if st.host_total is None or st.host_total.numel() < 2 or not st.fused:
        return 0
    return int(st.host_total[1].item())


def isect_finish(st: "_IsectPending"):
    """Second half of intersect_tile: wait for the total, allocate exact-length outputs, emit (key, value) pairs, sort."""
    means2d, radii, depths, conics, opacities, image_ids = st.args
    tile_size, tile_width, tile_height, tile_bits, image_bits = st.geom
    rows, n_per, I = st.rows, st.n_per, st.I
    dev = means2d.device
    tiles_per_gauss = st.tiles_per_gauss
    if rows == 0:
        return (tiles_per_gauss, torch.empty(0, device=dev, dtype=torch.int64),
                torch.empty(0, device=dev, dtype=torch.int32))
    if st.event == "polled":  # compiled second half: waits for the count (the one host round trip), allocates, emits, sorts
        isect_ids, flatten_ids = torch.ops.gsplat_amd.isect_fused_finish(
            means2d, radii, depths, conics, opacities, rows, I, tile_size, tile_width, tile_height, st.count_ws, st.offsets,
            st.host_total, tiles_per_gauss)
        return tiles_per_gauss, isect_ids, flatten_ids
    st.event.synchronize()  # host sync: exact-length outputs (reference: Intersect.cpp:258-259)
    if st.fused:
        n_isects = _isect_fused_total(st, None)
        isect_ids, flatten_ids = _isect_fused_emit(st, None, n_isects)
        _note_longest(flatten_ids, isect_max_tile_len(st))
        return tiles_per_gauss, isect_ids, flatten_ids
    n_isects = int(st.host_total[0].item())
    cum = st.cum
    if n_isects >= 2**31:
        raise RuntimeError(f"intersect_tile: {n_isects} intersections overflow the int32 index space")
    isect_ids = torch.empty(n_isects, device=dev, dtype=torch.int64)
    flatten_ids = torch.empty(n_isects, device=dev, dtype=torch.int32)
    if n_isects == 0:
        return tiles_per_gauss, isect_ids, flatten_ids
    if means2d.dtype == torch.float64:
        call("gsx_isect_emit_f64", ptr(means2d), ptr(radii), ptr(depths), ptr(image_ids), ptr(cum), rows, n_per, I,
             tile_size, tile_width, tile_height, ptr(isect_ids), ptr(flatten_ids))
    else:
        call("gsx_isect_emit", ptr(means2d), ptr(radii), ptr(depths), ptr(conics), ptr(opacities), ptr(image_ids),
             ptr(cum), rows, n_per, I, tile_size, tile_width, tile_height, ptr(isect_ids), ptr(flatten_ids))
    if st.sort and _cabi.tile_sort_supported(I, tile_width, tile_height):
        keys_s, vals_s = torch.empty_like(isect_ids), torch.empty_like(flatten_ids)
        ws = torch.empty(_cabi.tile_sort_workspace_bytes(n_isects, I, tile_width, tile_height), device=dev,
                         dtype=torch.uint8)
        call("gsx_isect_tile_sort", ptr(isect_ids), ptr(flatten_ids), n_isects, I, tile_width, tile_height,
             ptr(keys_s), ptr(vals_s), ptr(ws), ws.numel())
        isect_ids, flatten_ids = keys_s, vals_s
    elif st.sort:
        keys_alt, vals_alt = torch.empty_like(isect_ids), torch.empty_like(flatten_ids)
        ws = torch.empty(_cabi.sort_workspace_bytes(n_isects), device=dev, dtype=torch.uint8)
        in_alt = _cabi.sort_pairs(isect_ids, flatten_ids, keys_alt, vals_alt, n_isects, 32 + tile_bits + image_bits, ws)
        if in_alt:
            isect_ids, flatten_ids = keys_alt, vals_alt
    return tiles_per_gauss, isect_ids, flatten_ids


@_op("intersect_tile")
def intersect_tile(means2d, radii, depths, conics, opacities, image_ids, gaussian_ids, n_images, tile_size,
                   tile_width, tile_height, sort, segmented):
    return isect_finish(isect_begin(means2d, radii, depths, conics, opacities, image_ids, gaussian_ids, n_images,
                                    tile_size, tile_width, tile_height, sort, segmented))


def _lidar_tiling_args(lidar, dev):
    """(fields of view, direction, tiling sizes, the two tables on `dev`) of a RowOffsetStructuredSpinningLidarModelParametersExt."""
    cdf_el = lidar.cdf_elevation.to(device=dev, dtype=torch.int32).contiguous()
    raycdf = lidar.cdf_dense_ray_mask.to(device=dev, dtype=torch.int32).contiguous()
    if cdf_el.dim() != 1 or raycdf.dim() != 2 or raycdf.shape[0] != cdf_el.shape[0]:
        raise RuntimeError("lidar tiling: cdf_elevation [R + 1] and cdf_dense_ray_mask [R + 1, A + 1] expected")
    return ((float(lidar.fov_horiz_rad.start), float(lidar.fov_horiz_rad.span), float(lidar.fov_vert_rad.start),
             float(lidar.fov_vert_rad.span), int(lidar.spinning_direction), int(lidar.n_bins_azimuth), int(lidar.n_bins_elevation),
             int(raycdf.shape[1] - 1), int(raycdf.shape[0] - 1), ptr(cdf_el), ptr(raycdf)), (cdf_el, raycdf))


@_op("intersect_tile_lidar")
def intersect_tile_lidar(lidar, means2d, radii, depths, image_ids, gaussian_ids, n_images, sort, segmented):
    """gsplat::intersect_tile_lidar (Intersect.cpp:388-520): tiles of a spinning lidar's angular tiling touched by every
    Gaussian's (azimuth x elevation) box; same outputs as intersect_tile."""
    packed = means2d.dim() == 2
    if packed:
        if image_ids is None or gaussian_ids is None:
            raise RuntimeError("When packed is set, image_ids and gaussian_ids must be provided.")
        if n_images is None:
            raise RuntimeError("n_images is required when means2d is packed ([nnz, 2]).")
        if segmented:
            raise RuntimeError("segmented sort is not supported for packed inputs")
        I = int(n_images)
        if means2d.shape[-1] != 2 or tuple(radii.shape) != tuple(means2d.shape) or tuple(depths.shape) != (means2d.shape[0],):
            raise RuntimeError(f"means2d must be [nnz, 2] with matching radii / depths, got {tuple(means2d.shape)}")
        n_per = 1
    else:
        if means2d.dim() < 2 or means2d.shape[-1] != 2:
            raise RuntimeError(f"means2d must be [..., N, 2], got {tuple(means2d.shape)}")
        if tuple(radii.shape) != tuple(means2d.shape):
            raise RuntimeError(f"radii must be [..., N, 2] matching means2d, got {tuple(radii.shape)}")
        if tuple(depths.shape) != tuple(means2d.shape[:-1]):
            raise RuntimeError(f"depths must be [..., N], got {tuple(depths.shape)}")
        I = math.prod(means2d.shape[:-2])
        n_per = means2d.shape[-2]
    n_tiles = int(lidar.n_bins_azimuth) * int(lidar.n_bins_elevation)
    tile_bits, image_bits = bits_for_count(n_tiles), bits_for_count(I)
    if tile_bits + image_bits > 32:
        raise RuntimeError(f"intersect_tile_lidar: (image, tile) id packing needs {tile_bits + image_bits} bits but only 32 are "
                           f"available (I={I}, n_tiles={n_tiles}).")
    dev = means2d.device
    # the reference's second instantiation (scalar_t = double, IntersectTileLidar.cu:479-493) loads every value into float and
    # narrows the depth of the key to float (:185-186, :387-392): the double call IS the float call on narrowed inputs
    if means2d.dtype == torch.float64:
        means2d = means2d.to(torch.float32)
    if depths.dtype == torch.float64:
        depths = depths.to(torch.float32)
    _check_f32(means2d=means2d, depths=depths)
    means2d, depths = means2d.contiguous(), depths.contiguous()
    radii = radii.contiguous()
    if radii.dtype not in (torch.int32, torch.float32):
        radii = radii.to(torch.float32) if radii.is_floating_point() else radii.to(torch.int32)
    r_i, r_f = (ptr(radii), None) if radii.dtype == torch.int32 else (None, ptr(radii))
    rows = depths.numel()
    tiles_per_gauss = torch.empty(depths.shape, device=dev, dtype=torch.int32)
    empty = (tiles_per_gauss, torch.empty(0, device=dev, dtype=torch.int64), torch.empty(0, device=dev, dtype=torch.int32))
    if rows == 0:
        return empty
    targs, _keep = _lidar_tiling_args(lidar, dev)
    n_per_arg = max(int(n_per), 1)
    call("gsx_isect_lidar_count", ptr(means2d), r_i, r_f, rows, n_per_arg, *targs, ptr(tiles_per_gauss))
    cum = torch.cumsum(tiles_per_gauss.reshape(-1), 0, dtype=torch.int64)
    n_isects = int(cum[-1].item())  # host sync: exact-length outputs, like the reference (Intersect.cpp:470-480)
    if n_isects >= 2**31:
        raise RuntimeError(f"intersect_tile_lidar: {n_isects} intersections overflow the int32 index space")
    if n_isects == 0:
        return empty
    isect_ids = torch.empty(n_isects, device=dev, dtype=torch.int64)
    flatten_ids = torch.empty(n_isects, device=dev, dtype=torch.int32)
    img = image_ids.contiguous().to(torch.int64) if packed else None
    call("gsx_isect_lidar_emit", ptr(means2d), r_i, r_f, ptr(depths), ptr(img), ptr(cum), rows, n_per_arg, I, *targs,
         ptr(isect_ids), ptr(flatten_ids))
    if sort:
        tw, th = int(lidar.n_bins_azimuth), int(lidar.n_bins_elevation)
        if _cabi.tile_sort_supported(I, tw, th):
            keys_s, vals_s = torch.empty_like(isect_ids), torch.empty_like(flatten_ids)
            ws = torch.empty(_cabi.tile_sort_workspace_bytes(n_isects, I, tw, th), device=dev, dtype=torch.uint8)
            call("gsx_isect_tile_sort", ptr(isect_ids), ptr(flatten_ids), n_isects, I, tw, th, ptr(keys_s), ptr(vals_s), ptr(ws),
                 ws.numel())
            isect_ids, flatten_ids = keys_s, vals_s
        else:
            keys_alt, vals_alt = torch.empty_like(isect_ids), torch.empty_like(flatten_ids)
            ws = torch.empty(_cabi.sort_workspace_bytes(n_isects), device=dev, dtype=torch.uint8)
            if _cabi.sort_pairs(isect_ids, flatten_ids, keys_alt, vals_alt, n_isects, 32 + tile_bits + image_bits, ws):
                isect_ids, flatten_ids = keys_alt, vals_alt
    return tiles_per_gauss, isect_ids, flatten_ids


@_op("intersect_offset")
def intersect_offset(isect_ids, I, tile_width, tile_height):
    isect_ids = isect_ids.contiguous()
    offsets = torch.empty((I, tile_height, tile_width), device=isect_ids.device, dtype=torch.int32)
    call("gsx_isect_offsets", ptr(isect_ids), isect_ids.numel(), I, tile_width, tile_height, ptr(offsets))
    return offsets


# ----------------------------------------------------------------------------------------------
# projection
# ----------------------------------------------------------------------------------------------
def _proj_dims(means, viewmats):
    batch_dims = means.shape[:-2]
    return batch_dims, math.prod(batch_dims), viewmats.shape[-3], means.shape[-2]


def _f64_instantiation(fn):
    """The reference dispatches the projection ops over float AND double (AT_DISPATCH_FLOATING_TYPES, ProjectionEWA3DGSFused.cu:260,
    686; ProjectionEWA3DGSPacked.cu:344, 733). In its double instantiation only the MEMORY type is double: the kernels load every
    value into glm float vectors / matrices (include/Common.h:65-70), compute in float and widen the results on store. Same here:
    double tensors are narrowed, the fp32 kernels run, floating outputs are widened (csrc/torch_ops.cpp does the same)."""
    import functools

    @functools.wraps(fn)
    def body(means, *args, **kw):
        if means.dtype != torch.float64:
            return fn(means, *args, **kw)
        down = lambda t: t.to(torch.float32) if isinstance(t, Tensor) and t.dtype == torch.float64 else t  # noqa: E731
        up = lambda t: t.to(torch.float64) if isinstance(t, Tensor) and t.dtype == torch.float32 else t  # noqa: E731
        out = fn(down(means), *[down(a) for a in args], **{k: down(v) for k, v in kw.items()})
        return tuple(up(t) for t in out)

    return body


def _check_proj_inputs(means, covars, quats, scales, viewmats, Ks):
    _check_f32(means=means, covars=covars, quats=quats, scales=scales, viewmats=viewmats, Ks=Ks)
    if covars is None and (quats is None or scales is None):
        raise ValueError("projection: either covars or (quats, scales) must be given")


@_op("projection_ewa_3dgs_fused")
@_f64_instantiation
def projection_ewa_3dgs_fused(means, covars, quats, scales, opacities, viewmats, Ks, image_width, image_height,
                              eps2d, near_plane, far_plane, radius_clip, calc_compensations, camera_model):
    _check_proj_inputs(means, covars, quats, scales, viewmats, Ks)
    batch_dims, B, C, N = _proj_dims(means, viewmats)
    means, viewmats, Ks = means.contiguous(), viewmats.contiguous(), Ks.contiguous()
    covars, quats, scales, opacities = _c(covars), _c(quats), _c(scales), _c(opacities)
    dev, dt = means.device, means.dtype
    shape = tuple(batch_dims) + (C, N)
    radii = torch.empty(shape + (2,), device=dev, dtype=torch.int32)
    means2d = torch.empty(shape + (2,), device=dev, dtype=dt)
    depths = torch.empty(shape, device=dev, dtype=dt)
    conics = torch.empty(shape + (3,), device=dev, dtype=dt)
    comps = torch.empty(shape, device=dev, dtype=dt) if calc_compensations else None
    call("gsx_project_ewa_fwd", ptr(means), ptr(covars), ptr(None if covars is not None else quats),
         ptr(None if covars is not None else scales), ptr(opacities), ptr(viewmats), ptr(Ks), B, C, N, image_width,
         image_height, eps2d, near_plane, far_plane, radius_clip, int(camera_model), ptr(radii), ptr(means2d),
         ptr(depths), ptr(conics), ptr(comps))
    return radii, means2d, depths, conics, comps


@_op("projection_ewa_3dgs_fused_bwd")
@_f64_instantiation
def projection_ewa_3dgs_fused_bwd(means, covars, quats, scales, viewmats, Ks, image_width, image_height, eps2d,
                                  camera_model, radii, conics, compensations, v_means2d, v_depths, v_conics,
                                  v_compensations, viewmats_requires_grad, *, _v_view_opacities=None):
    """`_v_view_opacities` (private, gsplat_amd's own autograd only): the cotangent of the per-view opacities [..., C, N]; the
    kernel sums it over the views and a sixth value, v_opacities [..., N], is returned."""
    batch_dims, B, C, N = _proj_dims(means, viewmats)
    means, viewmats, Ks = means.contiguous(), viewmats.contiguous(), Ks.contiguous()
    covars, quats, scales = _c(covars), _c(quats), _c(scales)
    v_means = torch.empty_like(means)
    v_covars = v_quats = v_scales = None
    if covars is not None:
        v_covars = torch.empty_like(covars)
    else:
        v_quats, v_scales = torch.empty_like(quats), torch.empty_like(scales)
    v_viewmats = torch.zeros_like(viewmats) if viewmats_requires_grad else None
    v_means2d, m2_stride = _row_view(v_means2d, 2)
    v_conics, con_stride = _row_view(v_conics, 3)
    head = (ptr(means), ptr(covars), ptr(None if covars is not None else quats),
            ptr(None if covars is not None else scales), ptr(viewmats), ptr(Ks), B, C, N, image_width, image_height,
            eps2d, int(camera_model), ptr(radii.contiguous()), ptr(conics.contiguous()), ptr(_c(compensations)),
            ptr_strided(v_means2d), m2_stride, ptr(_c(v_depths)), ptr_strided(v_conics), con_stride,
            ptr(_c(v_compensations)))
    if _v_view_opacities is not None:
        v_view, opac_stride = _elem_view(_v_view_opacities)
        v_opacities = torch.empty(tuple(batch_dims) + (N,), device=means.device, dtype=means.dtype)
        call("gsx_project_ewa_bwd_opac", *head, ptr_strided(v_view), opac_stride, ptr(v_means), ptr(v_covars), ptr(v_quats),
             ptr(v_scales), ptr(v_viewmats), ptr(v_opacities))
        return v_means, v_covars, v_quats, v_scales, v_viewmats, v_opacities
    call("gsx_project_ewa_bwd", *head, ptr(v_means), ptr(v_covars), ptr(v_quats), ptr(v_scales), ptr(v_viewmats))
    return v_means, v_covars, v_quats, v_scales, v_viewmats


_PACKED_ROW_BYTES = 64  # 3 int64 ids + radii + means2d + depth + conic (+ compensation) per packed row
_PACKED_PREALLOC_LIMIT = 1 << 30  # upper-bound row buffers are only used below this size
_PACKED_COMPACT_ABOVE = 1 << 28  # ... and their unused tails are given back when they exceed this many bytes


@_op("projection_ewa_3dgs_packed")
@_f64_instantiation
def projection_ewa_3dgs_packed(means, covars, quats, scales, opacities, viewmats, Ks, image_width, image_height,
                               eps2d, near_plane, far_plane, radius_clip, sparse_grad, calc_compensations,
                               camera_model):
    _check_proj_inputs(means, covars, quats, scales, viewmats, Ks)
    batch_dims, B, C, N = _proj_dims(means, viewmats)
    means, viewmats, Ks = means.contiguous(), viewmats.contiguous(), Ks.contiguous()
    covars, quats, scales, opacities = _c(covars), _c(quats), _c(scales), _c(opacities)
    dev, dt = means.device, means.dtype
    q = None if covars is not None else quats
    s = None if covars is not None else scales
    total = B * C * N
    common = (ptr(means), ptr(covars), ptr(q), ptr(s), ptr(opacities), ptr(viewmats), ptr(Ks), B, C, N, image_width,
              image_height, eps2d, near_plane, far_plane, radius_clip, int(camera_model))

    def outputs(rows):
        return (torch.empty(rows, device=dev, dtype=torch.int64), torch.empty(rows, device=dev, dtype=torch.int64),
                torch.empty(rows, device=dev, dtype=torch.int64), torch.zeros(B * C + 1, device=dev, dtype=torch.int32),
                torch.empty((rows, 2), device=dev, dtype=torch.int32), torch.empty((rows, 2), device=dev, dtype=dt),
                torch.empty((rows,), device=dev, dtype=dt), torch.empty((rows, 3), device=dev, dtype=dt),
                torch.empty((rows,), device=dev, dtype=dt) if calc_compensations else None)

    if total == 0:
        return outputs(0)
    # Rows are placed from BLOCK counts (csrc/projection.hip: PackedBlocks): one int32 per 256 (image, Gaussian) pairs, scanned
    # by one workgroup that stores the row count straight into a pinned host word - no per-pair flags, no cumsum tensor.
    n_blocks = _cabi._lib.gsx_project_packed_blocks(total)
    blocks = torch.empty((2, n_blocks), device=dev, dtype=torch.int32)
    host_nnz = torch.full((1,), -1, dtype=torch.int64).pin_memory()
    call("gsx_project_ewa_packed_count_blocks", *common, int(calc_compensations), ptr(blocks[0]), ptr(blocks[1]), None,
         host_nnz.data_ptr())
    ev = torch.cuda.Event()
    ev.record()
    prealloc = total * _PACKED_ROW_BYTES <= _PACKED_PREALLOC_LIMIT
    if prealloc:
        # The write pass only needs the DEVICE-side offsets: enqueue it into row buffers sized for the upper bound (every
        # pair visible) before the host learns nnz, and hand out the first nnz rows. Scenes whose upper bound would not be
        # small next to the model keep the exact-length path below - saving that memory is what packed rows are for.
        bufs = outputs(total)
        call("gsx_project_ewa_packed_write_blocks", *common, ptr(blocks[1]), *[ptr(t) for t in bufs])
    ev.synchronize()  # host round trip: exact-length COO outputs (reference: Projection.cpp:928-941)
    nnz = int(host_nnz.item())
    if prealloc:
        # a view pins the whole upper-bound buffer for as long as the step (and its autograd graph) holds the rows: copy the
        # heads out and let the big buffers go only when that is a real amount of memory - seven copies cost 32 us of kernels
        # and as much host time, which left the GPU idle behind the write pass (c3 at 25 % visibility: packed 0.86 ms per
        # step against dense 0.80, profiles/r08_ab.md #28)
        compact = (total - nnz) * _PACKED_ROW_BYTES > _PACKED_COMPACT_ABOVE
        return tuple(t if (t is None or i == 3) else (t[:nnz].clone() if compact else t[:nnz]) for i, t in enumerate(bufs))
    bufs = outputs(nnz)
    call("gsx_project_ewa_packed_write_blocks", *common, ptr(blocks[1]), *[ptr(t) for t in bufs])
    return bufs


@_op("projection_ewa_3dgs_packed_bwd")
@_f64_instantiation
def projection_ewa_3dgs_packed_bwd(means, covars, quats, scales, viewmats, Ks, image_width, image_height, eps2d,
                                   camera_model, sparse_grad, batch_ids, camera_ids, gaussian_ids, conics,
                                   compensations, v_means2d, v_depths, v_conics, v_compensations,
                                   viewmats_requires_grad, *, _v_view_opacities=None):
    """`_v_view_opacities` (private, gsplat_amd's own autograd only): the cotangent of the packed rows' opacities [nnz]; a
    sixth value, v_opacities [..., N], is returned - summed in the Gaussian-major kernel where that one runs."""
    batch_dims, B, C, N = _proj_dims(means, viewmats)
    means, viewmats, Ks = means.contiguous(), viewmats.contiguous(), Ks.contiguous()
    covars, quats, scales = _c(covars), _c(quats), _c(scales)
    nnz = gaussian_ids.shape[0]
    v_viewmats = torch.zeros_like(viewmats) if viewmats_requires_grad else None
    v_means2d, m2_stride = _row_view(v_means2d, 2)
    v_conics, con_stride = _row_view(v_conics, 3)
    head = (ptr(means), ptr(covars), ptr(None if covars is not None else quats),
            ptr(None if covars is not None else scales), ptr(viewmats), ptr(Ks), B, C, N, image_width, image_height,
            eps2d, int(camera_model), nnz, ptr(batch_ids.contiguous()), ptr(camera_ids.contiguous()),
            ptr(gaussian_ids.contiguous()), ptr(conics.contiguous()), ptr(_c(compensations)),
            ptr_strided(v_means2d), m2_stride, ptr(_c(v_depths)), ptr_strided(v_conics), con_stride,
            ptr(_c(v_compensations)))
    def scatter_opacities():  # where no Gaussian-major kernel runs: what autograd does for opacities[ids] (an index_add)
        flat = torch.zeros(B * N, device=means.device, dtype=means.dtype)
        flat.index_add_(0, batch_ids * N + gaussian_ids if B > 1 else gaussian_ids, _v_view_opacities.reshape(-1))
        return flat.reshape(tuple(batch_dims) + (N,))

    if sparse_grad:
        # COO gradients exactly as the reference builds them (Projection.cpp:1125-1200): the kernel writes one [nnz, .] row per
        # packed row, indices = gaussian_ids, coalesced iff a single image (every Gaussian appears at most once). No dense
        # [N, .] tensor is allocated and nothing is read back. rasterization() refuses batch dimensions with sparse_grad
        # (Rendering.cpp:278); a stage-level caller with batch dimensions gets a two-row index (batch, gaussian).
        def rows(width):
            return torch.empty((nnz, width), device=means.device, dtype=means.dtype)

        r_means = rows(3)
        r_covars = rows(6) if covars is not None else None
        r_quats = rows(4) if covars is None else None
        r_scales = rows(3) if covars is None else None
        call("gsx_project_ewa_packed_bwd_rows", *head, ptr(r_means), ptr(r_covars), ptr(r_quats), ptr(r_scales),
             ptr(v_viewmats))
        flat = len(batch_dims) == 0
        indices = gaussian_ids.unsqueeze(0) if flat else torch.stack([batch_ids, gaussian_ids])
        coalesced = B * C == 1

        def coo(vals, like):
            if vals is None:
                return None
            size = ((N,) if flat else (B, N)) + (like.shape[-1],)
            sp = torch.sparse_coo_tensor(indices, vals, size=size, is_coalesced=coalesced)
            return sp if len(batch_dims) <= 1 else sp.to_dense().reshape(like.shape)

        res = (coo(r_means, means), coo(r_covars, covars), coo(r_quats, quats), coo(r_scales, scales), v_viewmats)
        clear_row_map_cache()  # the SH backward of this step (which runs first) may have left its map: this is the last consumer
        return res if _v_view_opacities is None else res + (scatter_opacities(),)
    # several images: walk the packed rows Gaussian-major through a row map (each output row written once, no atomics; the map
    # is the one the SH backward asked for); a single image: every Gaussian has at most one row and the row-major kernel
    # stores without atomics into zero-filled outputs - the Gaussian-major walk is equal there at 25 % visibility (35 us either
    # way) and 0.5 ms slower on the 49 M-Gaussian scene (fps_bwd 616 -> 463), where it visits 49 M Gaussians for 2 M rows.
    # Round 5: with at least half of the Gaussians visible the single image goes Gaussian-major too - the three zero fills of
    # the row-major route (40 B per Gaussian, 16.5 us at c3) are what it saves.
    gaussian_major = B * C > 1 or 2 * nnz >= N
    row_map = _packed_row_map(batch_ids, camera_ids, gaussian_ids, B, C, N) if (gaussian_major and nnz > 0 and N > 0) else None
    alloc = torch.empty_like if row_map is not None else torch.zeros_like
    v_means = alloc(means)
    v_covars = v_quats = v_scales = None
    if covars is not None:
        v_covars = alloc(covars)
    else:
        v_quats, v_scales = alloc(quats), alloc(scales)
    v_opacities = None
    if _v_view_opacities is not None:
        v_view, opac_stride = _elem_view(_v_view_opacities)
        # [..., N]: written once per Gaussian through the row map, accumulated into zeros without one
        v_opacities = (torch.empty if row_map is not None else torch.zeros)(tuple(batch_dims) + (N,), device=means.device, dtype=means.dtype)
        call("gsx_project_ewa_packed_bwd_opac", *head, ptr_strided(v_view), opac_stride, ptr(row_map), ptr(v_means), ptr(v_covars),
             ptr(v_quats), ptr(v_scales), ptr(v_viewmats), ptr(v_opacities))
    else:
        call("gsx_project_ewa_packed_bwd", *head, ptr(row_map), ptr(v_means), ptr(v_covars), ptr(v_quats), ptr(v_scales),
             ptr(v_viewmats))
    # the last consumer of a step's row map (autograd runs the SH backward, created later, first): drop it here, and with it
    # the references that keep the step's id tensors alive - a stage-level caller has no next rasterization() to do that
    clear_row_map_cache()
    res = (v_means, v_covars, v_quats, v_scales, v_viewmats)
    return res if _v_view_opacities is None else res + (v_opacities,)


# ----------------------------------------------------------------------------------------------
# rasterize_to_pixels (3DGS)
# ----------------------------------------------------------------------------------------------
def _raster_dims(isect_offsets, colors):
    image_dims = tuple(isect_offsets.shape[:-2])
    return image_dims, math.prod(image_dims), isect_offsets.shape[-2], isect_offsets.shape[-1], colors.shape[-1]


@_op("rasterize_to_pixels_3dgs")
def rasterize_to_pixels_3dgs(means2d, conics, colors, opacities, backgrounds, masks, image_width, image_height,
                             tile_size, isect_offsets, flatten_ids, packed, absgrad):
    _check_f32(means2d=means2d, conics=conics, colors=colors, opacities=opacities, backgrounds=backgrounds)
    as_received = (means2d, conics, colors, opacities, isect_offsets, flatten_ids)  # key of the segment-workspace note
    image_dims, I, th, tw, D = _raster_dims(isect_offsets, colors)
    if th * tile_size < image_height or tw * tile_size < image_width:
        raise ValueError("rasterize_to_pixels: isect_offsets tile grid does not cover the image")
    if masks is not None and masks.dtype != torch.bool:
        raise TypeError("masks must be a bool tensor")
    means2d, conics, colors, opacities = (means2d.contiguous(), conics.contiguous(), colors.contiguous(),
                                          opacities.contiguous())
    backgrounds, masks = _c(backgrounds), _c(masks)
    isect_offsets, flatten_ids = isect_offsets.contiguous(), flatten_ids.contiguous()
    dev, dt = means2d.device, means2d.dtype
    renders = torch.empty(image_dims + (image_height, image_width, D), device=dev, dtype=dt)
    alphas = torch.empty(image_dims + (image_height, image_width, 1), device=dev, dtype=dt)
    last_ids = torch.empty(image_dims + (image_height, image_width), device=dev, dtype=torch.int32)
    longest = _consume_long_tile_hint() or _lookup_longest(flatten_ids)
    rows = _consume_splat_rows_hint(means2d, D)  # rasterization()'s array-of-structures rows of these Gaussians, or None
    if longest > SEG_MIN_LONGEST and longest > _seg_cut(flatten_ids.numel(), I, tw, th):
        ws = torch.empty(_cabi._lib.gsx_raster3d_seg_workspace_bytes(flatten_ids.numel(), I, tw, th, D, SEG_LEN), device=dev,
                         dtype=torch.uint8)
        call("gsx_raster3d_fwd_seg", ptr(means2d), ptr(conics), ptr(colors), ptr(opacities), ptr(backgrounds), ptr(masks),
             ptr(isect_offsets), ptr(flatten_ids), I, flatten_ids.numel(), D, image_width, image_height, tile_size, tw,
             th, ptr(renders), ptr(alphas), ptr(last_ids), SEG_LEN, ptr(ws), ws.numel())
        if D <= 4 and tile_size == 16:
            # the backward over these lists starts its slices from the sums this call left in `ws` (no pre-pass): noted
            # under the identity of last_ids, the tensor every autograd formula hands to the backward op
            _note_seg_workspace(last_ids, ws, flatten_ids.numel(), D, as_received)
    elif rows is not None:
        call("gsx_raster3d_fwd_rows", ptr(means2d), ptr(conics), ptr(colors), ptr(opacities), ptr(rows), ptr(backgrounds),
             ptr(masks), ptr(isect_offsets), ptr(flatten_ids), I, flatten_ids.numel(), D, image_width, image_height, tile_size,
             tw, th, ptr(renders), ptr(alphas), ptr(last_ids))
    else:
        call("gsx_raster3d_fwd", ptr(means2d), ptr(conics), ptr(colors), ptr(opacities), ptr(backgrounds), ptr(masks),
             ptr(isect_offsets), ptr(flatten_ids), I, flatten_ids.numel(), D, image_width, image_height, tile_size, tw,
             th, ptr(renders), ptr(alphas), ptr(last_ids))
    holder = torch.zeros_like(means2d) if absgrad else torch.empty(0, device=dev, dtype=dt)
    return renders, alphas, holder, last_ids


def _pixel_linear_strides(t):
    """(pixel stride, channel stride) in elements when the float32 tensor t [..., H, W, D] is not contiguous but still
    addresses pixel p = ((i H) + y) W + x, channel k at p * ps + k * cs (expanded scalars, channel slices of a wider
    contiguous image, ...); None when it is contiguous or needs a copy."""
    if t.is_contiguous() or t.dim() < 3:
        return None
    shape, stride = t.shape, t.stride()
    cs, ps = stride[-1], stride[-2]
    if cs < 0 or ps < 0:
        return None
    expect = ps
    for d in range(t.dim() - 2, -1, -1):
        if shape[d] != 1 and stride[d] != expect:
            return None
        expect *= shape[d]
    return int(ps), int(cs)


@_op("rasterize_to_pixels_3dgs_bwd")
def rasterize_to_pixels_3dgs_bwd(means2d, conics, colors, opacities, backgrounds, masks, tile_offsets, flatten_ids,
                                 render_alphas, last_ids, image_width, image_height, tile_size, absgrad,
                                 v_render_colors, v_render_alphas, compute_v_backgrounds):
    image_dims, I, th, tw, D = _raster_dims(tile_offsets, colors)
    as_received = (means2d, conics, colors, opacities, tile_offsets, flatten_ids)  # key of the segment-workspace note
    means2d, conics, colors, opacities = (means2d.contiguous(), conics.contiguous(), colors.contiguous(),
                                          opacities.contiguous())
    backgrounds, masks = _c(backgrounds), _c(masks)
    v_render_alphas = _c(v_render_alphas)  # None = zeros
    # ONE zero-filled array-of-structures buffer [R][6 (+2) + D] (layout: include/gsplat_amd.h, gsx_raster3d_bwd); the
    # gradients the reference returns as separate tensors are COLUMN VIEWS of it. A Gaussian's gradients share a cache
    # line, which is what makes the kernel's atomic flush cheap; projection_ewa_3dgs_*_bwd reads the views in place.
    R = opacities.numel()
    geo = 8 if absgrad else 6
    longest = _consume_long_tile_hint() or _lookup_longest(flatten_ids)  # set by the autograd formula around this call
    splat_rows = _consume_splat_rows_hint(means2d, D)
    segmented = (longest > SEG_MIN_LONGEST and not absgrad and D <= 4 and tile_size == 16
                 and longest > _seg_cut(flatten_ids.numel(), I, tw, th))
    # the per-tile launch zero-fills the rows itself (inside its tile-order kernel: gsx_raster3d_bwd_fill)
    own_fill = segmented or _ROWS_FILL_TORCH
    rows = (torch.zeros if own_fill else torch.empty)((R, geo + D), device=means2d.device, dtype=means2d.dtype)
    # autograd hands cotangents over as views (the gradient of sum() is ONE float expanded to [.., H, W, D]): the per-tile
    # launch reads any layout that is linear in the pixel index in place instead of materialising 4 D bytes per pixel
    vrc_strides = None if segmented else _pixel_linear_strides(v_render_colors)
    if vrc_strides is None:
        v_render_colors = v_render_colors.contiguous()
        vrc_strides = (-1, 1)
    if segmented:
        ws = torch.empty(_cabi._lib.gsx_raster3d_bwd_seg_workspace_bytes(flatten_ids.numel(), I, tw, th, D, SEG_LEN),
                         device=means2d.device, dtype=torch.uint8)
        fws = _lookup_seg_workspace(last_ids, flatten_ids.numel(), D, as_received)  # the forward call's workspace, if still around
        call("gsx_raster3d_bwd_seg_reuse", ptr(means2d), ptr(conics), ptr(colors), ptr(opacities), ptr(backgrounds), ptr(masks),
             ptr(tile_offsets.contiguous()), ptr(flatten_ids.contiguous()), ptr(render_alphas.contiguous()),
             ptr(last_ids.contiguous()), ptr(v_render_colors), ptr(v_render_alphas), I, flatten_ids.numel(), D,
             image_width, image_height, tile_size, tw, th, ptr(rows), geo + D, SEG_LEN, ptr(fws),
             0 if fws is None else fws.numel(), ptr(ws), ws.numel())
    else:
        # workspace for the longest-first tile order of the launch (csrc/raster3d_bwd.hip: "longest tiles first")
        ws = torch.empty(_cabi._lib.gsx_raster3d_bwd_workspace_bytes(I, tw, th), device=means2d.device, dtype=torch.uint8)
        head = (ptr(means2d), ptr(conics), ptr(colors), ptr(opacities)) + ((ptr(splat_rows),) if splat_rows is not None else ())
        call("gsx_raster3d_bwd_fill_rows" if splat_rows is not None else "gsx_raster3d_bwd_fill", *head, ptr(backgrounds), ptr(masks),
             ptr(tile_offsets.contiguous()), ptr(flatten_ids.contiguous()), ptr(render_alphas.contiguous()),
             ptr(last_ids.contiguous()), ptr_strided(v_render_colors), ptr(v_render_alphas), I, flatten_ids.numel(), D,
             image_width, image_height, tile_size, tw, th, int(bool(absgrad)), ptr(rows), geo + D, 0 if own_fill else R, vrc_strides[0],
             vrc_strides[1], ptr(ws), ws.numel())
    v_means2d, v_conics = rows[:, 0:2].view(means2d.shape), rows[:, 2:5].view(conics.shape)
    v_opacities, v_colors = rows[:, 5].view(opacities.shape), rows[:, geo:].view(colors.shape)
    v_abs = rows[:, 6:8].view(means2d.shape) if absgrad else None
    v_backgrounds = None
    if backgrounds is not None and compute_v_backgrounds:
        # sum_{h,w} v_colors * (1 - alpha)  (reference does this with torch ops too: Rasterization.cpp:567-577)
        v_backgrounds = (v_render_colors * (1.0 - render_alphas)).sum(dim=(-3, -2))
    return v_abs, v_means2d, v_conics, v_colors, v_opacities, v_backgrounds


# ----------------------------------------------------------------------------------------------
# split SH ops (reference SphericalHarmonics.cpp l0 / l1_plus): colours = Y0 * sh0 + sum_{k>=1} Y_k(dir) shN[k-1].
# l1_plus reuses the full-band kernels on a zero-padded band 0 (same arithmetic for k >= 1).
# ----------------------------------------------------------------------------------------------
_SH_C0 = 0.2820947917738781


@_op("spherical_harmonics_l0")
def spherical_harmonics_l0(sh0):
    """[N, 1, D] -> [N, D] in float32 (fp16 rows widen like the other SH ops' coefficients; SphericalHarmonics.cpp:149-262)."""
    if sh0.dim() != 3:
        raise ValueError(f"sh0 must have shape [N, 1, D], got {tuple(sh0.shape)}")
    if sh0.shape[-2] != 1:
        raise ValueError(f"sh0 must contain exactly one SH coefficient, got {tuple(sh0.shape)}")
    if sh0.shape[-1] < 1:
        raise ValueError(f"sh0 last dim D must be >= 1, got {sh0.shape[-1]}")
    if sh0.dtype not in (torch.float32, torch.float16):
        raise TypeError(f"gsplat_amd: sh0 must be float32 or float16 (got {sh0.dtype})")
    return sh0[:, 0, :].float() * _SH_C0


@_op("spherical_harmonics_l0_bwd")
def spherical_harmonics_l0_bwd(sh0, v_colors):
    if v_colors.dim() != 2 or v_colors.shape[0] != sh0.shape[0] or v_colors.shape[1] != sh0.shape[2]:
        raise ValueError(f"v_colors must have shape [N, D], got {tuple(v_colors.shape)}")
    return (v_colors * _SH_C0)[:, None, :].to(sh0.dtype).contiguous()


def _pad_band0(shN):
    return torch.cat([torch.zeros_like(shN[:, :1]), shN], dim=1).contiguous()


_BAND_DTYPES = {torch.float32: 0, torch.float16: 1}


def _sh_band_fwd(degrees_to_use, first_band, means, viewmats, coeffs, masks, batch_ids, camera_ids, gaussian_ids):
    """gsx_sh_band_fwd (csrc/sh_band.hip): bands first_band .. of [N, K - first_band, 3] coefficient rows (fp32 / fp16) read in
    place, indexed by Gaussian (packed rows through gaussian_ids)."""
    packed, B, C, N, KM, D = _sh_dims(means, viewmats, coeffs, gaussian_ids)
    means, viewmats, coeffs, masks = means.contiguous(), viewmats.contiguous(), coeffs.contiguous(), _c(masks)
    if coeffs.shape[0] != N:
        raise ValueError(f"coefficient rows are indexed by Gaussian: expected {N} rows, got {coeffs.shape[0]}")
    K = KM + first_band
    if packed:
        nnz = gaussian_ids.shape[0]
        colors = torch.empty((nnz, 3), device=means.device, dtype=means.dtype)
        call("gsx_sh_band_fwd", degrees_to_use, first_band, _BAND_DTYPES[coeffs.dtype], ptr(means), ptr(viewmats), ptr(coeffs),
             ptr(masks), ptr(_c(batch_ids)), ptr(_c(camera_ids)), ptr(_c(gaussian_ids)), B, C, N, nnz, K, ptr(colors))
    else:
        colors = torch.empty(viewmats.shape[:-2] + (N, 3), device=means.device, dtype=means.dtype)
        call("gsx_sh_band_fwd", degrees_to_use, first_band, _BAND_DTYPES[coeffs.dtype], ptr(means), ptr(viewmats), ptr(coeffs),
             ptr(masks), None, None, None, B, C, N, -1, K, ptr(colors))
    return colors


def _sh_band_bwd(degrees_to_use, first_band, means, viewmats, coeffs, masks, batch_ids, camera_ids, gaussian_ids, v_colors,
                 compute_v_means, compute_v_viewmats):
    packed, B, C, N, KM, D = _sh_dims(means, viewmats, coeffs, gaussian_ids)
    means, viewmats, coeffs, masks = means.contiguous(), viewmats.contiguous(), coeffs.contiguous(), _c(masks)
    K = KM + first_band
    nnz = gaussian_ids.shape[0] if packed else -1
    row_map = _packed_row_map(batch_ids, camera_ids, gaussian_ids, B, C, N) if (packed and nnz > 0 and N > 0) else None
    if packed and row_map is None:  # no rows at all: nothing reaches the coefficients
        return torch.zeros_like(coeffs), (torch.zeros_like(means) if compute_v_means else None), \
            (torch.zeros_like(viewmats) if compute_v_viewmats else None)
    v_coeffs = torch.empty_like(coeffs)  # every row is written by the thread that owns the Gaussian
    v_means = torch.empty_like(means) if compute_v_means else None
    v_dirs = None
    if compute_v_viewmats:
        v_dirs = torch.zeros(((nnz if packed else B * C * N), 3), device=means.device, dtype=means.dtype)
    call("gsx_sh_band_bwd", degrees_to_use, first_band, _BAND_DTYPES[coeffs.dtype], ptr(means), ptr(viewmats), ptr(coeffs),
         ptr(masks), B, C, N, nnz, K, ptr(v_colors.contiguous()), ptr(row_map), ptr(v_coeffs), ptr(v_means), ptr(v_dirs))
    v_viewmats = None
    if compute_v_viewmats:
        if packed:
            S = torch.zeros((B * C, 3), device=means.device, dtype=means.dtype)
            S.index_add_(0, batch_ids * C + camera_ids, v_dirs)
        else:
            S = v_dirs.view(B * C, N, 3).sum(dim=1)
        vm = viewmats.reshape(B * C, 4, 4)
        R, t = vm[:, :3, :3], vm[:, :3, 3]
        v_vm = torch.zeros_like(vm)
        v_vm[:, :3, :3] = t[:, :, None] * S[:, None, :]
        v_vm[:, :3, 3] = torch.einsum("cij,cj->ci", R, S)
        v_viewmats = v_vm.reshape(viewmats.shape)
    return v_coeffs, v_means, v_viewmats


def _band_kernels_apply(coeffs) -> bool:
    return coeffs.dim() == 3 and coeffs.shape[-1] == 3 and coeffs.shape[-2] >= 1 and coeffs.dtype in _BAND_DTYPES


@_op("spherical_harmonics_l1_plus")
def spherical_harmonics_l1_plus(degrees_to_use, means, viewmats, shN, masks, batch_ids, camera_ids, gaussian_ids,
                                viewmats_rs=None, *, _gathered: bool = True):
    """colours of bands l >= 1 from shN [N, K - 1, D] (packed: the reference's contract is PRE-GATHERED rows [nnz, K - 1, D],
    SphericalHarmonics.cpp:90-104). D == 3 rows indexed by Gaussian are read IN PLACE (csrc/sh_band.hip; reference
    SphericalHarmonicsL1PlusCUDA.cu:441: no concatenation with a zero band); gathered packed rows and D != 3 go through the
    general kernels on a zero-padded band 0 (same arithmetic for k >= 1). Private keyword `_gathered=False` (rendering.py):
    packed rows read [N, K - 1, 3] through gaussian_ids."""
    if viewmats_rs is not None:
        viewmats = _sh_rs_viewmats(viewmats, viewmats_rs)
    if shN.dim() != 3:
        raise ValueError(f"shN must have shape [N, K - 1, D], got {tuple(shN.shape)}")
    _check_sh_inputs(degrees_to_use, means, viewmats, shN, masks, batch_ids, camera_ids, gaussian_ids, omit_l0=True,
                     gathered=_gathered)
    _check_f32(means=means, viewmats=viewmats)
    packed = gaussian_ids is not None
    if degrees_to_use == 0 or shN.shape[-2] == 0:  # no band >= 1 is evaluated (deg-0 zero contract; shN may be [N, 0, D])
        shape = (shN.shape[0] if _gathered else gaussian_ids.shape[0], shN.shape[-1]) if packed \
            else tuple(viewmats.shape[:-2]) + (means.shape[-2], shN.shape[-1])
        return torch.zeros(shape, device=means.device, dtype=means.dtype)
    if _band_kernels_apply(shN) and not (packed and _gathered):
        return _sh_band_fwd(degrees_to_use, 1, means, viewmats, shN, masks, batch_ids, camera_ids, gaussian_ids)
    return spherical_harmonics(degrees_to_use, means, viewmats, _pad_band0(shN), masks, batch_ids, camera_ids,
                               gaussian_ids, None, _gathered=_gathered)


@_op("spherical_harmonics_l1_plus_bwd")
def spherical_harmonics_l1_plus_bwd(degrees_to_use, means, viewmats, shN, masks, batch_ids, camera_ids, gaussian_ids,
                                    viewmats_rs, v_colors, compute_v_means, compute_v_viewmats, compute_v_viewmats_rs,
                                    *, _gathered: bool = True):
    if viewmats_rs is not None:
        v_shN, v_me, v_syn, _ = spherical_harmonics_l1_plus_bwd(
            degrees_to_use, means, _sh_rs_viewmats(viewmats, viewmats_rs), shN, masks, batch_ids, camera_ids, gaussian_ids, None,
            v_colors, compute_v_means, compute_v_viewmats or compute_v_viewmats_rs, False, _gathered=_gathered)
        v_vm, v_rs = (None, None) if v_syn is None else _sh_rs_split(v_syn, viewmats, viewmats_rs, compute_v_viewmats,
                                                                     compute_v_viewmats_rs)
        return v_shN, v_me, v_vm, v_rs
    packed = gaussian_ids is not None
    if degrees_to_use == 0 or shN.shape[-2] == 0:
        return (torch.zeros_like(shN), torch.zeros_like(means) if compute_v_means else None,
                torch.zeros_like(viewmats) if compute_v_viewmats else None, None)
    if _band_kernels_apply(shN) and not (packed and _gathered):
        v_shN, v_means, v_viewmats = _sh_band_bwd(degrees_to_use, 1, means, viewmats, shN, masks, batch_ids, camera_ids,
                                                  gaussian_ids, v_colors, compute_v_means, compute_v_viewmats)
        return v_shN, v_means, v_viewmats, None
    v_coeffs, v_means, v_viewmats, v_rs = spherical_harmonics_bwd(
        degrees_to_use, means, viewmats, _pad_band0(shN), masks, batch_ids, camera_ids, gaussian_ids, viewmats_rs,
        v_colors, compute_v_means, compute_v_viewmats, compute_v_viewmats_rs, _gathered=_gathered)
    return v_coeffs[:, 1:].contiguous(), v_means, v_viewmats, v_rs


# ----------------------------------------------------------------------------------------------
# proj(): projection_ewa_simple (reference ProjectionEWASimple.cu; torch restatement _torch_impl.py:53-260)
# ----------------------------------------------------------------------------------------------
@_op("projection_ewa_simple")
def projection_ewa_simple(means, covars, Ks, width, height, camera_model):
    _check_f32(means=means, covars=covars, Ks=Ks)
    if means.shape[-1] != 3 or covars.shape[-2:] != (3, 3) or covars.shape[:-2] != means.shape[:-1]:
        raise ValueError(f"proj: bad shapes means {tuple(means.shape)} covars {tuple(covars.shape)}")
    means, covars, Ks = means.contiguous(), covars.contiguous(), Ks.contiguous()
    N = means.shape[-2]
    rows = means.numel() // 3
    means2d = torch.empty(means.shape[:-1] + (2,), device=means.device, dtype=means.dtype)
    covars2d = torch.empty(means.shape[:-1] + (2, 2), device=means.device, dtype=means.dtype)
    call("gsx_project_simple_fwd", ptr(means), ptr(covars), ptr(Ks), rows, N, width, height, int(camera_model),
         ptr(means2d), ptr(covars2d))
    return means2d, covars2d


@_op("projection_ewa_simple_bwd")
def projection_ewa_simple_bwd(means, covars, Ks, width, height, camera_model, v_means2d, v_covars2d):
    means, covars, Ks = means.contiguous(), covars.contiguous(), Ks.contiguous()
    N = means.shape[-2]
    rows = means.numel() // 3
    v_means, v_covars = torch.empty_like(means), torch.empty_like(covars)
    call("gsx_project_simple_bwd", ptr(means), ptr(covars), ptr(Ks), rows, N, width, height, int(camera_model),
         ptr(v_means2d.contiguous()), ptr(v_covars2d.contiguous()), ptr(v_means), ptr(v_covars))
    return v_means, v_covars


# ----------------------------------------------------------------------------------------------
# rasterize_to_indices (reference Rasterization.cpp:954-1100): count, exclusive cumsum, write
# ----------------------------------------------------------------------------------------------
def _raster_indices(mode, range_start, range_end, transmittances, means2d, geom, opacities, image_width, image_height,
                    tile_size, tile_offsets, flatten_ids):
    _check_f32(transmittances=transmittances, means2d=means2d, geom=geom, opacities=opacities)
    image_dims = tuple(means2d.shape[:-2])
    N = means2d.shape[-2]
    I = math.prod(image_dims)
    if transmittances.shape != image_dims + (image_height, image_width):
        raise ValueError(f"transmittances must have shape [*image_dims, image_height, image_width], got "
                         f"{tuple(transmittances.shape)}")
    th, tw = tile_offsets.shape[-2], tile_offsets.shape[-1]
    dev = means2d.device
    args = (int(mode), int(range_start), int(range_end), ptr(transmittances.contiguous()), ptr(means2d.contiguous()),
            ptr(geom.contiguous()), ptr(opacities.contiguous()), ptr(tile_offsets.contiguous()),
            ptr(flatten_ids.contiguous()), I, N, flatten_ids.numel(), image_width, image_height, tile_size, tw, th)
    cnts = torch.zeros(I * image_height * image_width, device=dev, dtype=torch.int32)
    n_elems = 0
    if cnts.numel() > 0 and flatten_ids.numel() > 0:
        call("gsx_raster_indices", *args, None, ptr(cnts), None, None)
        cum = torch.cumsum(cnts, 0, dtype=torch.int32)
        n_elems = int(cum[-1].item())
        starts = (cum - cnts).contiguous()
    gaussian_ids = torch.empty(n_elems, device=dev, dtype=torch.int64)
    combined = torch.empty(n_elems, device=dev, dtype=torch.int64)
    if n_elems:
        call("gsx_raster_indices", *args, ptr(starts), None, ptr(gaussian_ids), ptr(combined))
    pix_per_image = image_height * image_width
    return gaussian_ids, torch.remainder(combined, pix_per_image), torch.div(combined, pix_per_image,
                                                                            rounding_mode="floor")


@_op("rasterize_to_indices_3dgs")
def rasterize_to_indices_3dgs(range_start, range_end, transmittances, means2d, conics, opacities, image_width,
                              image_height, tile_size, tile_offsets, flatten_ids):
    return _raster_indices(0, range_start, range_end, transmittances, means2d, conics, opacities, image_width,
                           image_height, tile_size, tile_offsets, flatten_ids)


@_op("rasterize_to_indices_2dgs")
def rasterize_to_indices_2dgs(range_start, range_end, transmittances, means2d, ray_transforms, opacities, image_width,
                              image_height, tile_size, tile_offsets, flatten_ids):
    return _raster_indices(1, range_start, range_end, transmittances, means2d, ray_transforms, opacities, image_width,
                           image_height, tile_size, tile_offsets, flatten_ids)


# ----------------------------------------------------------------------------------------------
# 2DGS: projection (Projection.cpp 2DGS section; ext.cpp:1163-1184) and compositing (ext.cpp:1186-1199)
# ----------------------------------------------------------------------------------------------
def _check_2dgs_inputs(means, quats, scales, viewmats, Ks):
    _check_f32(means=means, quats=quats, scales=scales, viewmats=viewmats, Ks=Ks)
    N = means.shape[-2]
    if means.shape[-1] != 3 or quats.shape[-2:] != (N, 4) or scales.shape[-2:] != (N, 3):
        raise ValueError(f"projection_2dgs: bad shapes means {tuple(means.shape)} quats {tuple(quats.shape)} "
                         f"scales {tuple(scales.shape)}")


@_op("projection_2dgs_fused")
def projection_2dgs_fused(means, quats, scales, viewmats, Ks, image_width, image_height, eps2d, near_plane, far_plane,
                          radius_clip):
    _check_2dgs_inputs(means, quats, scales, viewmats, Ks)
    batch_dims, B, C, N = _proj_dims(means, viewmats)
    means, quats, scales, viewmats, Ks = (t.contiguous() for t in (means, quats, scales, viewmats, Ks))
    dev, dt = means.device, means.dtype
    shape = tuple(batch_dims) + (C, N)
    radii = torch.empty(shape + (2,), device=dev, dtype=torch.int32)
    means2d = torch.empty(shape + (2,), device=dev, dtype=dt)
    depths = torch.empty(shape, device=dev, dtype=dt)
    ray_transforms = torch.empty(shape + (3, 3), device=dev, dtype=dt)
    normals = torch.empty(shape + (3,), device=dev, dtype=dt)
    call("gsx_project_2dgs_fwd", ptr(means), ptr(quats), ptr(scales), ptr(viewmats), ptr(Ks), B, C, N, image_width,
         image_height, near_plane, far_plane, radius_clip, ptr(radii), ptr(means2d), ptr(depths), ptr(ray_transforms),
         ptr(normals))
    return radii, means2d, depths, ray_transforms, normals


@_op("projection_2dgs_fused_bwd")
def projection_2dgs_fused_bwd(means, quats, scales, viewmats, Ks, image_width, image_height, radii, ray_transforms,
                              v_means2d, v_depths, v_ray_transforms, v_normals, viewmats_requires_grad, *,
                              _v_view_opacities=None):
    """`_v_view_opacities` (private, gsplat_amd's own autograd only): the cotangent of the per-view opacities [..., C, N]; the
    kernel sums it over the views and a fifth value, v_opacities [..., N], is returned."""
    batch_dims, B, C, N = _proj_dims(means, viewmats)
    means, quats, scales, viewmats, Ks = (t.contiguous() for t in (means, quats, scales, viewmats, Ks))
    v_means, v_quats, v_scales = torch.empty_like(means), torch.empty_like(quats), torch.empty_like(scales)
    v_viewmats = torch.zeros_like(viewmats) if viewmats_requires_grad else None
    (v_means2d, v_rt, v_normals), vstride = _common_row_views(
        (v_means2d, v_ray_transforms.reshape(v_ray_transforms.shape[:-2] + (9,)), v_normals), (2, 9, 3))
    # the depth cotangent is read in place when it is one column of the gradient rows (the slice autograd cuts out of the
    # colour cotangent of a depth render mode), like the three above
    v_dep, dep_stride = (None, 1) if v_depths is None else _elem_view(v_depths)
    head = (ptr(means), ptr(quats), ptr(scales), ptr(viewmats), ptr(Ks), B, C, N,
            ptr(radii.contiguous()), ptr(ray_transforms.contiguous()), ptr_strided(v_means2d),
            None if v_dep is None else ptr_strided(v_dep), ptr_strided(v_rt), ptr_strided(v_normals), vstride, int(dep_stride))
    if _v_view_opacities is not None:
        v_view, opac_stride = _elem_view(_v_view_opacities)
        v_opacities = torch.empty(tuple(batch_dims) + (N,), device=means.device, dtype=means.dtype)
        call("gsx_project_2dgs_bwd_opac", *head, ptr_strided(v_view), opac_stride, ptr(v_means), ptr(v_quats), ptr(v_scales),
             ptr(v_viewmats), ptr(v_opacities))
        return v_means, v_quats, v_scales, v_viewmats, v_opacities
    call("gsx_project_2dgs_bwd", *head, ptr(v_means), ptr(v_quats), ptr(v_scales), ptr(v_viewmats))
    return v_means, v_quats, v_scales, v_viewmats


@_op("projection_2dgs_packed")
def projection_2dgs_packed(means, quats, scales, viewmats, Ks, image_width, image_height, near_plane, far_plane,
                           radius_clip, sparse_grad):
    _check_2dgs_inputs(means, quats, scales, viewmats, Ks)
    batch_dims, B, C, N = _proj_dims(means, viewmats)
    means, quats, scales, viewmats, Ks = (t.contiguous() for t in (means, quats, scales, viewmats, Ks))
    dev, dt = means.device, means.dtype
    total = B * C * N
    common = (ptr(means), ptr(quats), ptr(scales), ptr(viewmats), ptr(Ks), B, C, N, image_width, image_height,
              near_plane, far_plane, radius_clip)
    nnz, cum = 0, None
    if total > 0:
        visible = torch.empty(total, device=dev, dtype=torch.int32)
        call("gsx_project_2dgs_packed_count", *common, ptr(visible))
        cum = _scan_i32(visible)
        nnz = int(cum[-1].item())  # host sync: exact-length COO outputs, as the reference
    batch_ids = torch.empty(nnz, device=dev, dtype=torch.int64)
    camera_ids = torch.empty(nnz, device=dev, dtype=torch.int64)
    gaussian_ids = torch.empty(nnz, device=dev, dtype=torch.int64)
    indptr = torch.zeros(B * C + 1, device=dev, dtype=torch.int32)
    radii = torch.empty((nnz, 2), device=dev, dtype=torch.int32)
    means2d = torch.empty((nnz, 2), device=dev, dtype=dt)
    depths = torch.empty((nnz,), device=dev, dtype=dt)
    ray_transforms = torch.empty((nnz, 3, 3), device=dev, dtype=dt)
    normals = torch.empty((nnz, 3), device=dev, dtype=dt)
    if total > 0:
        call("gsx_project_2dgs_packed_write", *common, ptr(cum), nnz, ptr(batch_ids), ptr(camera_ids),
             ptr(gaussian_ids), ptr(indptr), ptr(radii), ptr(means2d), ptr(depths), ptr(ray_transforms), ptr(normals))
    return batch_ids, camera_ids, gaussian_ids, indptr, radii, means2d, depths, ray_transforms, normals


@_op("projection_2dgs_packed_bwd")
def projection_2dgs_packed_bwd(means, quats, scales, viewmats, Ks, image_width, image_height, sparse_grad, batch_ids,
                               camera_ids, gaussian_ids, ray_transforms, v_means2d, v_depths, v_ray_transforms,
                               v_normals, viewmats_requires_grad):
    batch_dims, B, C, N = _proj_dims(means, viewmats)
    means, quats, scales, viewmats, Ks = (t.contiguous() for t in (means, quats, scales, viewmats, Ks))
    nnz = gaussian_ids.shape[0]
    v_viewmats = torch.zeros_like(viewmats) if viewmats_requires_grad else None
    (v_means2d, v_rt, v_normals), vstride = _common_row_views(
        (v_means2d, v_ray_transforms.reshape(v_ray_transforms.shape[:-2] + (9,)), v_normals), (2, 9, 3))
    head = (ptr(means), ptr(quats), ptr(scales), ptr(viewmats), ptr(Ks), B, C, N, nnz,
            ptr(batch_ids.contiguous()), ptr(camera_ids.contiguous()), ptr(gaussian_ids.contiguous()),
            ptr(ray_transforms.contiguous()), ptr_strided(v_means2d), ptr(_c(v_depths)), ptr_strided(v_rt),
            ptr_strided(v_normals), vstride)
    if sparse_grad and len(batch_dims) == 0:
        # COO gradients as the reference builds them (Projection.cpp:1780-1863): [nnz, .] rows from the kernel, indices =
        # gaussian_ids, coalesced iff a single image; no dense [N, .] tensor, nothing read back
        r_means, r_quats, r_scales = (torch.empty((nnz, w), device=means.device, dtype=means.dtype) for w in (3, 4, 3))
        call("gsx_project_2dgs_packed_bwd_rows", *head, ptr(r_means), ptr(r_quats), ptr(r_scales), ptr(v_viewmats))
        indices = gaussian_ids.unsqueeze(0)

        def coo(vals, like):
            return torch.sparse_coo_tensor(indices, vals, size=like.shape, is_coalesced=(C == 1))

        clear_row_map_cache()  # a packed 2DGS step's SH backward leaves its map; nothing after this op walks it
        return coo(r_means, means), coo(r_quats, quats), coo(r_scales, scales), v_viewmats
    v_means, v_quats, v_scales = torch.zeros_like(means), torch.zeros_like(quats), torch.zeros_like(scales)
    call("gsx_project_2dgs_packed_bwd", *head, ptr(v_means), ptr(v_quats), ptr(v_scales), ptr(v_viewmats))
    clear_row_map_cache()
    return v_means, v_quats, v_scales, v_viewmats


@_op("rasterize_to_pixels_2dgs")
def rasterize_to_pixels_2dgs(means2d, ray_transforms, colors, opacities, normals, densify, backgrounds, masks,
                             image_width, image_height, tile_size, tile_offsets, flatten_ids, packed, absgrad, distloss):
    _check_f32(means2d=means2d, ray_transforms=ray_transforms, colors=colors, opacities=opacities, normals=normals,
               backgrounds=backgrounds)
    image_dims, I, th, tw, D = _raster_dims(tile_offsets, colors)
    if th * tile_size < image_height or tw * tile_size < image_width:
        raise ValueError("rasterize_to_pixels_2dgs: tile grid does not cover the image")
    if masks is not None and masks.dtype != torch.bool:
        raise TypeError("masks must be a bool tensor")
    means2d, ray_transforms, colors, opacities, normals = (t.contiguous() for t in (means2d, ray_transforms, colors,
                                                                                    opacities, normals))
    backgrounds, masks = _c(backgrounds), _c(masks)
    tile_offsets, flatten_ids = tile_offsets.contiguous(), flatten_ids.contiguous()
    dev, dt = means2d.device, means2d.dtype
    hw = image_dims + (image_height, image_width)
    renders = torch.empty(hw + (D,), device=dev, dtype=dt)
    alphas = torch.empty(hw + (1,), device=dev, dtype=dt)
    rnormals = torch.empty(hw + (3,), device=dev, dtype=dt)
    rdistort = torch.empty(hw + (1,), device=dev, dtype=dt)
    rmedian = torch.empty(hw + (1,), device=dev, dtype=dt)
    last_ids = torch.empty(hw, device=dev, dtype=torch.int32)
    median_ids = torch.empty(hw, device=dev, dtype=torch.int32)
    call("gsx_raster2d_fwd", ptr(means2d), ptr(ray_transforms), ptr(colors), ptr(opacities), ptr(normals),
         ptr(backgrounds), ptr(masks), ptr(tile_offsets), ptr(flatten_ids), I, flatten_ids.numel(), D, image_width,
         image_height, tile_size, tw, th, int(distloss), ptr(renders), ptr(alphas), ptr(rnormals), ptr(rdistort),
         ptr(rmedian), ptr(last_ids), ptr(median_ids))
    holder = torch.zeros_like(means2d) if absgrad else torch.empty(0, device=dev, dtype=dt)
    return renders, alphas, rnormals, rdistort, rmedian, holder, last_ids, median_ids


@_op("rasterize_to_pixels_2dgs_bwd")
def rasterize_to_pixels_2dgs_bwd(means2d, ray_transforms, colors, opacities, normals, densify, backgrounds, masks,
                                 tile_offsets, flatten_ids, render_colors, render_alphas, last_ids, median_ids,
                                 image_width, image_height, tile_size, absgrad, v_render_colors, v_render_alphas,
                                 v_render_normals, v_render_distort, v_render_median, compute_v_backgrounds):
    """The dispatcher schema (verbatim from the reference's ext.cpp) declares the five cotangents as non-optional tensors; this
    body ALSO accepts None for v_render_alphas / _normals / _distort / _median, but only through the private Python entry
    (`_ops.impl()`, used by gsplat_amd's own autograd node) - a dispatcher call must pass dense cotangents
    (tests/test_gpu_2dgs.py::test_dispatcher_2dgs_bwd_with_dense_cotangents)."""
    image_dims, I, th, tw, D = _raster_dims(tile_offsets, colors)
    means2d, ray_transforms, colors, opacities, normals = (t.contiguous() for t in (means2d, ray_transforms, colors,
                                                                                    opacities, normals))
    backgrounds, masks = _c(backgrounds), _c(masks)
    v_render_colors = v_render_colors.contiguous()
    # one AoS gradient buffer, zero-filled by the launch itself (gsx_raster2d_bwd_fill: inside its tile-order kernel); the
    # reference's gradient tensors are column views of it (include/gsplat_amd.h). Cotangents of outputs the loss does not use
    # arrive as None from gsplat_amd's own autograd and go to the kernels as NULL (= zeros)
    R = opacities.numel()
    geo = 19 if absgrad else 17
    rows = torch.empty((R, geo + D), device=means2d.device, dtype=means2d.dtype)
    # workspace for the longest-first tile order of the launch (csrc/tile_order.hip)
    ws = torch.empty(_cabi._lib.gsx_raster3d_bwd_workspace_bytes(I, tw, th), device=means2d.device, dtype=torch.uint8)
    call("gsx_raster2d_bwd_fill", ptr(means2d), ptr(ray_transforms), ptr(colors), ptr(opacities), ptr(normals),
         ptr(backgrounds), ptr(masks), ptr(tile_offsets.contiguous()), ptr(flatten_ids.contiguous()),
         ptr(render_colors.contiguous()), ptr(render_alphas.contiguous()), ptr(last_ids.contiguous()),
         ptr(median_ids.contiguous()), ptr(v_render_colors), ptr(_c(v_render_alphas)), ptr(_c(v_render_normals)),
         ptr(_c(v_render_distort)), ptr(_c(v_render_median)), I, flatten_ids.numel(), D, image_width,
         image_height, tile_size, tw, th, int(bool(absgrad)), ptr(rows), geo + D, R, ptr(ws), ws.numel())
    v_means2d, v_opacities = rows[:, 0:2].view(means2d.shape), rows[:, 2].view(opacities.shape)
    v_densify, v_normals = rows[:, 3:5].view(means2d.shape), rows[:, 5:8].view(normals.shape)
    v_rt = rows[:, 8:17].view(ray_transforms.shape)
    v_abs = rows[:, 17:19].view(means2d.shape) if absgrad else None
    v_colors = rows[:, geo:].view(colors.shape)
    v_backgrounds = None
    if backgrounds is not None and compute_v_backgrounds:
        v_backgrounds = (v_render_colors * (1.0 - render_alphas)).sum(dim=(-3, -2))
    return v_abs, v_means2d, v_rt, v_colors, v_opacities, v_normals, v_densify, v_backgrounds


# ----------------------------------------------------------------------------------------------
# query rasterizers (reference gsplat/cuda/_wrapper.py:1668-1941, dense variants)
# ----------------------------------------------------------------------------------------------
def _query_common(means2d, conics, opacities, tile_offsets, flatten_ids, image_width, image_height, tile_size):
    _check_f32(means2d=means2d, conics=conics, opacities=opacities)
    image_dims = tuple(tile_offsets.shape[:-2])
    I, th, tw = math.prod(image_dims), tile_offsets.shape[-2], tile_offsets.shape[-1]
    if th * tile_size < image_height or tw * tile_size < image_width:
        raise ValueError("tile grid does not cover the image")
    packed = means2d.dim() == 2
    n_per = 0 if packed else means2d.shape[-2]
    args = (ptr(means2d.contiguous()), ptr(conics.contiguous()), ptr(opacities.contiguous()),
            ptr(tile_offsets.contiguous()), ptr(flatten_ids.contiguous()), I, flatten_ids.numel(), n_per, image_width,
            image_height, tile_size, tw, th)
    return image_dims + (image_height, image_width), args


@_op("rasterize_num_contributing_gaussians")
def rasterize_num_contributing_gaussians(means2d, conics, opacities, tile_offsets, flatten_ids, image_width,
                                         image_height, tile_size):
    hw, args = _query_common(means2d, conics, opacities, tile_offsets, flatten_ids, image_width, image_height, tile_size)
    counts = torch.empty(hw, device=means2d.device, dtype=torch.int32)
    alphas = torch.empty(hw, device=means2d.device, dtype=means2d.dtype)
    call("gsx_raster3d_num_contributing", *args, ptr(counts), ptr(alphas))
    return counts, alphas


@_op("rasterize_contributing_gaussian_ids")
def rasterize_contributing_gaussian_ids(means2d, conics, opacities, tile_offsets, flatten_ids, image_width, image_height,
                                        tile_size, num_contributing_gaussians):
    hw, args = _query_common(means2d, conics, opacities, tile_offsets, flatten_ids, image_width, image_height, tile_size)
    # padded to the largest per-pixel count (host read, as the reference: RasterizeContributingGaussianIds.cu host fn)
    kmax = int(num_contributing_gaussians.max().item()) if num_contributing_gaussians.numel() > 0 else 0
    ids = torch.full(hw + (kmax,), -1, device=means2d.device, dtype=torch.int32)
    weights = torch.zeros(hw + (kmax,), device=means2d.device, dtype=means2d.dtype)
    call("gsx_raster3d_contributing_ids", *args, kmax, ptr(ids), ptr(weights))
    return ids, weights


@_op("rasterize_top_contributing_gaussian_ids")
def rasterize_top_contributing_gaussian_ids(means2d, conics, opacities, tile_offsets, flatten_ids, image_width,
                                            image_height, tile_size, num_depth_samples):
    hw, args = _query_common(means2d, conics, opacities, tile_offsets, flatten_ids, image_width, image_height, tile_size)
    if num_depth_samples < 0:
        raise ValueError("num_depth_samples must be >= 0")
    ids = torch.empty(hw + (num_depth_samples,), device=means2d.device, dtype=torch.int32)
    weights = torch.empty(hw + (num_depth_samples,), device=means2d.device, dtype=means2d.dtype)
    call("gsx_raster3d_top_contributing", *args, num_depth_samples, ptr(ids), ptr(weights))
    return ids, weights


# ----------------------------------------------------------------------------------------------
# sparse pixel sets (reference Intersect.cpp:563-794, Rasterization.cpp:637-856, 1196-1500): render only a caller-given
# set of pixels. The layout is built once per pixel set with torch index ops on the device (not on the per-step path);
# intersection and compositing run the same HIP kernels as the dense path, restricted to the active tiles.
# ----------------------------------------------------------------------------------------------
@_op("build_sparse_tile_layout")
def build_sparse_tile_layout(pixels, image_ids, n_images, tile_size, tile_width, tile_height):
    if pixels.dim() != 2 or pixels.shape[1] != 2:
        raise ValueError("pixels must be [P, 2]")
    P, n_tiles = pixels.shape[0], tile_width * tile_height
    words = (tile_size * tile_size + 63) // 64
    dev = pixels.device
    if P == 0 or n_images == 0:
        return (torch.empty(0, device=dev, dtype=torch.int32),
                torch.zeros((n_images, tile_height, tile_width), device=dev, dtype=torch.bool),
                torch.empty((0, words), device=dev, dtype=torch.uint64),
                torch.zeros(1, device=dev, dtype=torch.int64), torch.empty(0, device=dev, dtype=torch.int64))
    if n_images * n_tiles >= 2**31:
        raise RuntimeError(f"build_sparse_tile_layout: n_images * n_tiles ({n_images * n_tiles}) must be < 2^31.")
    pos_bits = bits_for_count(tile_size * tile_size)
    row, col = pixels[:, 0].to(torch.int64), pixels[:, 1].to(torch.int64)
    img = image_ids.reshape(P).to(torch.int64)
    tile = img * n_tiles + torch.div(row, tile_size, rounding_mode="floor") * tile_width + torch.div(
        col, tile_size, rounding_mode="floor")
    in_tile = (row % tile_size) * tile_size + (col % tile_size)
    # unique keys (callers deduplicate pixels), so the argsort is deterministic: (tile, raster position in the tile)
    sorted_key, pixel_map = torch.sort((tile << pos_bits) | in_tile)
    active, counts = torch.unique_consecutive(sorted_key >> pos_bits, return_counts=True)
    cumsum = counts.cumsum(0)
    mask = torch.zeros(n_images * n_tiles, device=dev, dtype=torch.bool)
    mask[active] = True
    AT = active.shape[0]
    pos = sorted_key & ((1 << pos_bits) - 1)
    slot = torch.repeat_interleave(torch.arange(AT, device=dev), counts, output_size=P)
    bits = torch.zeros(AT * words, device=dev, dtype=torch.int64)
    # distinct bits per word: integer add == bitwise or (bit 63 wraps to the sign bit, which is the same bit pattern)
    bits.index_add_(0, slot * words + (pos >> 6), torch.ones_like(pos) << (pos & 63))
    return (active.to(torch.int32), mask.view(n_images, tile_height, tile_width), bits.view(AT, words).view(torch.uint64),
            cumsum, pixel_map)


@_op("intersect_tile_sparse")
def intersect_tile_sparse(means2d, radii, depths, image_ids, tile_mask, active_tiles, I, tile_size, tile_width,
                          tile_height):
    f64 = means2d.dtype == torch.float64  # the reference dispatches over float and double (radius boxes in double; the depth
    if not f64:                           # in the sort key is narrowed to float32): double rows take the generic kernels below
        _check_f32(means2d=means2d, depths=depths)
    if tile_mask.dtype != torch.bool:
        raise TypeError("tile_mask must be bool")
    if active_tiles.dtype != torch.int32:
        raise TypeError("active_tiles must be int32")
    packed = means2d.dim() == 2
    if packed and image_ids is None:
        raise ValueError("image_ids is required when means2d is packed ([nnz, 2]).")
    n_tiles = tile_width * tile_height
    if bits_for_count(I) + bits_for_count(n_tiles) > 32:
        raise RuntimeError(f"intersect_tile_sparse: (image, tile) id packing needs {bits_for_count(I) + bits_for_count(n_tiles)} "
                           f"bits but only 32 are available (I={I}, n_tiles={n_tiles}).")
    dev = means2d.device
    means2d, radii, depths = means2d.contiguous(), radii.contiguous(), depths.contiguous()
    if radii.dtype != torch.int32:
        radii = radii.to(torch.int32)
    tile_mask, active_tiles = tile_mask.contiguous(), active_tiles.contiguous()
    AT, rows = active_tiles.shape[0], means2d.numel() // 2
    empty = (torch.zeros(AT + 1, device=dev, dtype=torch.int32), torch.empty(0, device=dev, dtype=torch.int32))
    if rows == 0 or AT == 0:
        return empty
    sentinel = lambda n: torch.full((1,), n, device=dev, dtype=torch.int32)  # noqa: E731
    if not f64 and _cabi.isect_fused_supported(I, tile_width, tile_height, packed):
        # the dense fused path with the tile mask applied inside the walk (AABB test: conics / opacities NULL, as the
        # reference's sparse enumeration, Intersect.cpp:617-634): inactive tiles get empty segments, so the dense
        # offsets of the active tiles ARE the compacted offsets
        st = _IsectPending()
        st.args = (means2d, radii, depths, None, None, None)
        st.rows, st.I, st.geom = rows, I, (tile_size, tile_width, tile_height)
        st.tiles_per_gauss = st.cum = st.event = st.count_ws = st.n_dev = None
        st.n_per, st.sort, st.fused = 1, True, True  # every slot set: _isect_fused_emit reads st.fused through isect_max_tile_len
        st.offsets = offsets = torch.empty(I * n_tiles, device=dev, dtype=torch.int32)
        st.host_total = torch.zeros(2, dtype=torch.int64, pin_memory=True)
        st.binned = _cabi.isect_binned_should_try(rows, I, tile_width, tile_height, packed)  # once; st carries it
        _isect_fused_count(st, tile_mask)
        torch.cuda.current_stream(dev).synchronize()  # host sync: exact-length outputs (reference: Intersect.cpp:637)
        n_isects = _isect_fused_total(st, tile_mask)
        if n_isects == 0:
            return empty
        isect_ids, flatten_ids = _isect_fused_emit(st, tile_mask, n_isects)
        return torch.cat([offsets[active_tiles.long()], sentinel(n_isects)]), flatten_ids
    # packed rows of several images / tile grids beyond the fused path's LDS histogram: enumerate every tile with the
    # generic kernels, then drop the intersections of inactive tiles (order within a tile is preserved)
    if image_ids is not None:
        image_ids = image_ids.to(torch.int64)  # callers may pass int32 camera ids (reference widens too, Intersect.cpp:607-613)
    _tpg, isect_ids, flatten_ids = intersect_tile(means2d, radii, depths, None, None, image_ids, None, I, tile_size,
                                                  tile_width, tile_height, True, False)
    tile_bits = bits_for_count(n_tiles)  # key = (image << tile_bits | tile) << 32 | depth bits
    key_hi = isect_ids >> 32
    keep = tile_mask.reshape(-1)[(key_hi >> tile_bits) * n_tiles + (key_hi & ((1 << tile_bits) - 1))]
    isect_ids, flatten_ids = isect_ids[keep], flatten_ids[keep]
    offsets = intersect_offset(isect_ids, I, tile_width, tile_height).reshape(-1)
    return torch.cat([offsets[active_tiles.long()], sentinel(flatten_ids.shape[0])]), flatten_ids


def _sparse_layout_args(active_tiles, tile_offsets, flatten_ids, tile_pixel_mask, tile_pixel_cumsum, pixel_map, tile_size):
    if tile_pixel_mask.dtype not in (torch.uint64, torch.int64):
        raise TypeError("tile_pixel_mask must be uint64")
    if active_tiles.dtype != torch.int32 or tile_offsets.dtype != torch.int32 or flatten_ids.dtype != torch.int32:
        raise TypeError("active_tiles, tile_offsets and flatten_ids must be int32")
    if tile_pixel_cumsum.dtype != torch.int64 or pixel_map.dtype != torch.int64:
        raise TypeError("tile_pixel_cumsum and pixel_map must be int64")
    if pixel_map.dim() != 1:
        raise ValueError(f"pixel_map must be [P], got {tuple(pixel_map.shape)}")
    AT = active_tiles.shape[0]
    if tile_offsets.shape[0] != AT + 1:
        raise ValueError("tile_offsets must be [num_active_tiles + 1]")
    words = tile_pixel_mask.shape[1] if tile_pixel_mask.dim() == 2 else (tile_size * tile_size + 63) // 64
    return (ptr(active_tiles.contiguous()), ptr(tile_offsets.contiguous()), ptr(flatten_ids.contiguous()),
            ptr(tile_pixel_mask.contiguous()), ptr(tile_pixel_cumsum.contiguous()), ptr(pixel_map.contiguous()), AT, words)


@_op("rasterize_to_pixels_sparse")
def rasterize_to_pixels_sparse(means2d, conics, colors, opacities, backgrounds, masks, image_ids, image_width,
                               image_height, tile_size, tile_width, tile_height, active_tiles, tile_offsets, flatten_ids,
                               tile_pixel_mask, tile_pixel_cumsum, pixel_map, packed, absgrad):
    _check_f32(means2d=means2d, conics=conics, colors=colors, opacities=opacities, backgrounds=backgrounds)
    if means2d.shape[-1] != 2:
        raise ValueError(f"means2d must have shape [..., N, 2] or [nnz, 2], got {tuple(means2d.shape)}")
    if masks is not None and masks.dtype != torch.bool:
        raise TypeError("masks must be a bool tensor")
    layout = _sparse_layout_args(active_tiles, tile_offsets, flatten_ids, tile_pixel_mask, tile_pixel_cumsum, pixel_map,
                                 tile_size)
    means2d, conics, colors, opacities = (means2d.contiguous(), conics.contiguous(), colors.contiguous(),
                                          opacities.contiguous())
    backgrounds, masks = _c(backgrounds), _c(masks)
    P, D, dev, dt = pixel_map.shape[0], colors.shape[-1], means2d.device, means2d.dtype
    renders = torch.empty((P, D), device=dev, dtype=dt)
    alphas = torch.empty((P, 1), device=dev, dtype=dt)
    last_ids = torch.empty((P,), device=dev, dtype=torch.int32)
    a_t, t_off, f_ids, p_mask, p_cum, p_map, AT, words = layout
    call("gsx_raster3d_sparse_fwd", ptr(means2d), ptr(conics), ptr(colors), ptr(opacities), ptr(backgrounds), ptr(masks),
         a_t, t_off, f_ids, p_mask, p_cum, p_map, AT, words, 0, flatten_ids.numel(), D, image_width, image_height,
         tile_size, tile_width, tile_height, ptr(renders), ptr(alphas), ptr(last_ids))
    holder = torch.zeros_like(means2d) if absgrad else torch.empty(0, device=dev, dtype=dt)
    return renders, alphas, holder, last_ids


@_op("rasterize_to_pixels_sparse_bwd")
def rasterize_to_pixels_sparse_bwd(means2d, conics, colors, opacities, backgrounds, masks, image_ids, active_tiles,
                                   tile_offsets, flatten_ids, tile_pixel_mask, tile_pixel_cumsum, pixel_map, render_alphas,
                                   last_ids, image_width, image_height, tile_size, tile_width, tile_height, absgrad,
                                   v_render_colors, v_render_alphas, compute_v_backgrounds):
    layout = _sparse_layout_args(active_tiles, tile_offsets, flatten_ids, tile_pixel_mask, tile_pixel_cumsum, pixel_map,
                                 tile_size)
    means2d, conics, colors, opacities = (means2d.contiguous(), conics.contiguous(), colors.contiguous(),
                                          opacities.contiguous())
    backgrounds, masks = _c(backgrounds), _c(masks)
    v_render_colors, v_render_alphas = v_render_colors.contiguous(), _c(v_render_alphas)  # None = zeros
    R, D = opacities.numel(), colors.shape[-1]
    geo = 8 if absgrad else 6
    rows = torch.zeros((R, geo + D), device=means2d.device, dtype=means2d.dtype)  # AoS rows, see rasterize_to_pixels_3dgs_bwd
    a_t, t_off, f_ids, p_mask, p_cum, p_map, AT, words = layout
    call("gsx_raster3d_sparse_bwd", ptr(means2d), ptr(conics), ptr(colors), ptr(opacities), ptr(backgrounds), ptr(masks),
         a_t, t_off, f_ids, p_mask, p_cum, p_map, AT, words, ptr(render_alphas.contiguous()), ptr(last_ids.contiguous()),
         ptr(v_render_colors), ptr(v_render_alphas), 0, flatten_ids.numel(), D, image_width, image_height, tile_size,
         tile_width, tile_height, int(bool(absgrad)), ptr(rows), geo + D)
    v_means2d, v_conics = rows[:, 0:2].view(means2d.shape), rows[:, 2:5].view(conics.shape)
    v_opacities, v_colors = rows[:, 5].view(opacities.shape), rows[:, geo:].view(colors.shape)
    v_abs = rows[:, 6:8].view(means2d.shape) if absgrad else None
    v_backgrounds = None
    if backgrounds is not None and compute_v_backgrounds:
        # per image: sum over its requested pixels of v_colors * (1 - alpha)  (reference Rasterization.cpp:835-846)
        v_backgrounds = torch.zeros((backgrounds.shape[0], D), device=means2d.device, dtype=v_render_colors.dtype)
        v_backgrounds.index_add_(0, image_ids.to(torch.int64), v_render_colors * (1.0 - render_alphas))
    return v_abs, v_means2d, v_conics, v_colors, v_opacities, v_backgrounds


def _sparse_query_common(means2d, conics, opacities, image_width, image_height, tile_size, tile_width, tile_height,
                         active_tiles, tile_offsets, flatten_ids, tile_pixel_mask, tile_pixel_cumsum, pixel_map):
    _check_f32(means2d=means2d, conics=conics, opacities=opacities)
    a_t, t_off, f_ids, p_mask, p_cum, p_map, AT, words = _sparse_layout_args(
        active_tiles, tile_offsets, flatten_ids, tile_pixel_mask, tile_pixel_cumsum, pixel_map, tile_size)
    n_per = 0 if means2d.dim() == 2 else means2d.shape[-2]
    return (ptr(means2d.contiguous()), ptr(conics.contiguous()), ptr(opacities.contiguous()), a_t, t_off, f_ids, p_mask,
            p_cum, p_map, AT, words, 0, flatten_ids.numel(), n_per, image_width, image_height, tile_size, tile_width,
            tile_height)


@_op("rasterize_num_contributing_gaussians_sparse")
def rasterize_num_contributing_gaussians_sparse(means2d, conics, opacities, image_width, image_height, tile_size,
                                                tile_width, tile_height, active_tiles, tile_offsets, flatten_ids,
                                                tile_pixel_mask, tile_pixel_cumsum, pixel_map):
    args = _sparse_query_common(means2d, conics, opacities, image_width, image_height, tile_size, tile_width, tile_height,
                                active_tiles, tile_offsets, flatten_ids, tile_pixel_mask, tile_pixel_cumsum, pixel_map)
    P = pixel_map.shape[0]
    counts = torch.empty(P, device=means2d.device, dtype=torch.int32)
    alphas = torch.empty(P, device=means2d.device, dtype=means2d.dtype)
    call("gsx_raster3d_sparse_num_contributing", *args, ptr(counts), ptr(alphas))
    return counts, alphas


@_op("rasterize_contributing_gaussian_ids_sparse")
def rasterize_contributing_gaussian_ids_sparse(means2d, conics, opacities, image_width, image_height, tile_size,
                                               tile_width, tile_height, active_tiles, tile_offsets, flatten_ids,
                                               tile_pixel_mask, tile_pixel_cumsum, pixel_map, num_contributing_gaussians):
    args = _sparse_query_common(means2d, conics, opacities, image_width, image_height, tile_size, tile_width, tile_height,
                                active_tiles, tile_offsets, flatten_ids, tile_pixel_mask, tile_pixel_cumsum, pixel_map)
    if num_contributing_gaussians.dtype != torch.int32:
        raise ValueError("num_contributing_gaussians must have dtype int32")
    P = pixel_map.shape[0]
    kmax = int(num_contributing_gaussians.max().item()) if num_contributing_gaussians.numel() > 0 else 0
    ids = torch.full((P, kmax), -1, device=means2d.device, dtype=torch.int32)
    weights = torch.zeros((P, kmax), device=means2d.device, dtype=means2d.dtype)
    call("gsx_raster3d_sparse_contributing_ids", *args, kmax, ptr(ids), ptr(weights))
    return ids, weights


@_op("rasterize_top_contributing_gaussian_ids_sparse")
def rasterize_top_contributing_gaussian_ids_sparse(means2d, conics, opacities, image_width, image_height, tile_size,
                                                   tile_width, tile_height, num_depth_samples, active_tiles,
                                                   tile_offsets, flatten_ids, tile_pixel_mask, tile_pixel_cumsum,
                                                   pixel_map):
    args = _sparse_query_common(means2d, conics, opacities, image_width, image_height, tile_size, tile_width, tile_height,
                                active_tiles, tile_offsets, flatten_ids, tile_pixel_mask, tile_pixel_cumsum, pixel_map)
    if num_depth_samples < 0:
        raise ValueError("num_depth_samples must be >= 0")
    P = pixel_map.shape[0]
    ids = torch.full((P, num_depth_samples), -1, device=means2d.device, dtype=torch.int32)
    weights = torch.zeros((P, num_depth_samples), device=means2d.device, dtype=means2d.dtype)
    call("gsx_raster3d_sparse_top_contributing", *args, num_depth_samples, ptr(ids), ptr(weights))
    return ids, weights


# ----------------------------------------------------------------------------------------------
# training-step ops (optimizer / MCMC), reference gsplat/cuda/_wrapper.py:419-435, gsplat/relocation.py
# ----------------------------------------------------------------------------------------------
@_op("adam")
def adam(param, param_grad, exp_avg, exp_avg_sq, valid, lr, b1, b2, eps):
    _check_f32(param=param, param_grad=param_grad, exp_avg=exp_avg, exp_avg_sq=exp_avg_sq)
    for name, t in (("param", param), ("exp_avg", exp_avg), ("exp_avg_sq", exp_avg_sq)):
        if not t.is_contiguous():
            raise ValueError(f"adam: {name} is updated in place and must be contiguous")
    n_rows = param.shape[0] if param.dim() > 0 else 1
    width = param.numel() // n_rows if n_rows > 0 else 0
    if valid is not None:
        if valid.dtype != torch.bool or valid.numel() != n_rows:
            raise ValueError("adam: valid must be a bool tensor with one entry per row of param")
        valid = valid.contiguous()
    call("gsx_adam", ptr(param), ptr(param_grad.contiguous()), ptr(exp_avg), ptr(exp_avg_sq), ptr(valid), n_rows, width,
         float(lr), float(b1), float(b2), float(eps))


@_op("relocation")
def relocation(opacities, scales, ratios, binoms, n_max, min_opacity=0.0):
    _check_f32(opacities=opacities, scales=scales, binoms=binoms)
    n = opacities.shape[0]
    if scales.shape != (n, 3) or ratios.shape != (n,):
        raise ValueError("relocation: scales must be [N, 3] and ratios [N]")
    opacities, scales, binoms = opacities.contiguous(), scales.contiguous(), binoms.contiguous()
    ratios = ratios.to(torch.int32).contiguous()
    new_opacities, new_scales = torch.empty_like(opacities), torch.empty_like(scales)
    call("gsx_relocation", ptr(opacities), ptr(scales), ptr(ratios), ptr(binoms), n, int(n_max), float(min_opacity),
         ptr(new_opacities), ptr(new_scales))
    return new_opacities, new_scales


@_op("mcmc_perturb_positions")
def mcmc_perturb_positions(positions, quats, scales, opacities, noise, noise_scale, t=0.005, k=100.0):
    # the dispatcher strips trailing arguments that equal the schema's defaults before calling a Python kernel
    _check_f32(positions=positions, quats=quats, scales=scales, opacities=opacities, noise=noise)
    if not positions.is_contiguous():
        raise ValueError("mcmc_perturb_positions: positions is updated in place and must be contiguous")
    n = positions.shape[0]
    call("gsx_mcmc_perturb", ptr(positions), ptr(quats.contiguous()), ptr(scales.contiguous()),
         ptr(opacities.reshape(-1).contiguous()), ptr(noise.contiguous()), n, float(noise_scale), float(t), float(k))


# ----------------------------------------------------------------------------------------------
# fused feature-row assembly (dense rows): [SH colours | extra signals | depth] in one pass
# ----------------------------------------------------------------------------------------------
@_op("assemble_proj_features_unpacked_fwd")
def assemble_proj_features_unpacked_fwd(degrees_to_use, B, C, N, Dc, E, color_post, extra_post, has_depth, depth_is_zero,
                                        extra_has_c, means, viewmats, viewmats_rs, coeffs, extra, depths, masks, out,
                                        relu_mask):
    """Checks follow the reference host op (SphericalHarmonics.cpp:572-676). Writes ``out`` (and ``relu_mask``) in place."""
    if viewmats_rs is not None:
        viewmats = _sh_rs_viewmats(viewmats, viewmats_rs)  # view direction from the averaged camera offset
    _check_f32(means=means, viewmats=viewmats, coeffs=coeffs, out=out, extra=extra, depths=depths)
    width, lead = Dc + E + (1 if has_depth else 0), B * C * N
    if coeffs.dim() != 3 or coeffs.shape[0] != N or coeffs.shape[2] != Dc:
        raise ValueError(f"coeffs must be [N={N}, K, Dc={Dc}], got {tuple(coeffs.shape)}")
    if means.numel() != B * N * 3 or viewmats.numel() != B * C * 16:
        raise ValueError("means / viewmats numel mismatch")
    if not out.is_contiguous() or out.shape[-1] != width or out.numel() != lead * width:
        raise ValueError(f"out must be contiguous [B, C, N, {width}], got {tuple(out.shape)}")
    if not (0 <= color_post <= 2 and 0 <= extra_post <= 2):
        raise ValueError(f"bad color_post / extra_post {color_post} / {extra_post}")
    if E > 0:
        if extra is None or extra.numel() != (lead if extra_has_c else B * N) * E:
            raise ValueError("extra is required when E > 0 and must be [B, C, N, E] or [B, N, E]")
    use_depths = has_depth and not depth_is_zero
    if use_depths and (depths is None or depths.numel() != lead):
        raise ValueError("depths [B, C, N] is required when has_depth and not depth_is_zero")
    if masks is not None and masks.numel() != lead:
        raise ValueError("masks numel mismatch")
    if relu_mask is not None:
        if relu_mask.dtype != torch.bool or not relu_mask.is_contiguous() or relu_mask.numel() != lead * Dc:
            raise ValueError("relu_mask must be a contiguous bool [B, C, N, Dc] tensor")
        if color_post != 2:
            raise ValueError("relu_mask is only valid with color_post = 2 (shift + relu)")
    call("gsx_assemble_features_fwd", int(degrees_to_use), B, C, N, coeffs.shape[1], Dc, E, int(color_post),
         int(extra_post), int(has_depth), int(extra_has_c), ptr(means.contiguous()), ptr(viewmats.contiguous()),
         ptr(coeffs.contiguous()), ptr(_c(extra)) if E > 0 else None, ptr(_c(depths)) if use_depths else None,
         ptr(_c(masks)), ptr(out), ptr(relu_mask))


# ----------------------------------------------------------------------------------------------
# Unscented-Transform projection (3DGUT)
# ----------------------------------------------------------------------------------------------
def _cubic_first_positive_root(a: Tensor, b: Tensor, c: Tensor) -> Tensor:
    """Per camera: the smallest positive x with 1 + a x + b x^2 + c x^3 = 0, or inf. Closed forms by degree; a cubic with
    one real root by Cardano, with three by the cosine form."""
    inf = torch.full_like(a, float("inf"))
    degree1 = torch.where(a < 0.0, -1.0 / a, inf)
    disc2 = a * a - 4.0 * b
    den = torch.sqrt(disc2.clamp_min(0.0)) - a
    degree2 = torch.where((disc2 >= 0.0) & (den > 0.0), 2.0 / den, inf)
    cs = torch.where(c.abs() < 1e-10, torch.ones_like(c), c)  # placeholder where the cubic branch is not taken
    r = b / cs
    u = (9.0 * a * r - 2.0 * b * r * r - 27.0) / cs
    w = 3.0 * a / cs - r * r
    disc3 = u * u + 4.0 * w * w * w
    half = (torch.sqrt(disc3.clamp_min(0.0)) + u) * 0.5
    cbrt = torch.sign(half) * half.abs().pow(1.0 / 3.0)
    single = torch.where(cbrt != 0.0, (cbrt - w / torch.where(cbrt != 0.0, cbrt, torch.ones_like(cbrt)) - r) / 3.0, inf)
    single = torch.where(single > 0.0, single, inf)
    phi = torch.atan2(torch.sqrt((-disc3).clamp_min(0.0)), u) / 3.0
    amp = 2.0 * torch.sqrt((-w).clamp_min(0.0))
    triple = inf
    for turn in (-1.0, 0.0, 1.0):
        x = (amp * torch.cos(phi + turn * (2.0 * math.pi / 3.0)) - r) / 3.0
        triple = torch.minimum(triple, torch.where(x > 0.0, x, inf))
    cubic = torch.where(disc3 >= 0.0, single, triple)
    is_quadratic, is_linear = c.abs() < 1e-10, (c.abs() < 1e-10) & (b.abs() < 1e-10)
    return torch.where(is_linear, degree1, torch.where(is_quadratic, degree2, cubic))


def fisheye_max_angle(radial_coeffs: Tensor, Ks: Tensor, width: int, height: int) -> Tensor:
    """[..., C] largest ray angle an OpenCV fisheye camera projects: the first zero of the derivative of
    theta (1 + k1 theta^2 + k2 theta^4 + k3 theta^6 + k4 theta^8) - a cubic in theta^2 when k4 vanishes, otherwise Newton
    from 1.57 rad (20 steps; accepted when a step fell below 1e-6 and the result is positive) - capped at the angle of the
    image corner. Restates the constructor of the reference's fisheye model (gsplat/cuda/_torch_cameras.py:1344-1521);
    a handful of scalars per camera, evaluated with tensor ops on the cameras' device."""
    k1, k2, k3, k4 = radial_coeffs.unbind(-1)
    from_cubic = torch.sqrt(_cubic_first_positive_root(3.0 * k1, 5.0 * k2, 7.0 * k3))
    x = torch.full_like(k1, 1.57)
    settled = torch.zeros_like(k1, dtype=torch.bool)
    for _ in range(20):
        q = x * x
        value = 1.0 + q * (3.0 * k1 + q * (5.0 * k2 + q * (7.0 * k3 + q * (9.0 * k4))))
        slope = x * (6.0 * k1 + q * (20.0 * k2 + q * (42.0 * k3 + q * (72.0 * k4))))
        step = value / slope
        x = torch.where(settled, x, x - step)
        settled = settled | (step.abs() < 1e-6)
    from_newton = torch.where(settled & (x > 0.0), x, torch.full_like(x, float("inf")))
    angle = torch.where(k4.abs() < 1e-10, from_cubic, from_newton)
    fx, fy, cx, cy = Ks[..., 0, 0], Ks[..., 1, 1], Ks[..., 0, 2], Ks[..., 1, 2]
    half_w, half_h = torch.maximum(width - cx, cx), torch.maximum(height - cy, cy)
    corner = torch.sqrt(half_w * half_w + half_h * half_h)
    return torch.minimum(angle, torch.maximum(corner / fx, corner / fy))


def _lidar_fov_args(lidar):
    return (float(lidar.fov_horiz_rad.start), float(lidar.fov_horiz_rad.span), float(lidar.fov_vert_rad.start),
            float(lidar.fov_vert_rad.span))


def _lidar_angle_map(lidar, dev):
    """angles_to_columns_map as int32 on `dev` (it maps a relative (elevation, azimuth) to the column that fires there)."""
    m = lidar.angles_to_columns_map
    if m is None or m.numel() == 0:
        return None
    return m.to(device=dev, dtype=torch.int32).contiguous()


def _projection_ut_lidar(means, quats, scales, opacities, viewmats0, viewmats1, Ks, eps2d, near_plane, far_plane, radius_clip,
                         calc_compensations, global_z_order, ut_params, rs_type, lidar, radial_coeffs, tangential_coeffs,
                         thin_prism_coeffs, external_distortion_params):
    """projection_ut_3dgs_fused for a spinning lidar (gsx_project_ut_lidar_fwd): means2d / radii come out in angular pixels
    (azimuth, elevation) * 1024."""
    if radial_coeffs is not None or tangential_coeffs is not None or thin_prism_coeffs is not None \
            or external_distortion_params is not None:
        raise RuntimeError("the lidar camera model takes no distortion coefficients")
    rolling = rs_type != _ROLLING_SHUTTER_GLOBAL
    if rolling and viewmats1 is None:
        raise ValueError("a rolling shutter needs viewmats_rs (the pose at the end of the frame)")
    _check_f32(means=means, quats=quats, scales=scales, opacities=opacities, viewmats=viewmats0, viewmats_rs=viewmats1, Ks=Ks)
    batch = tuple(means.shape[:-2])
    N, C, B = means.shape[-2], viewmats0.shape[-3], math.prod(means.shape[:-2])
    alpha, beta, kappa, margin, all_valid = 0.1, 2.0, 0.0, 0.1, False  # Cameras.h:59-64
    if ut_params is not None:
        alpha, beta, kappa = float(ut_params.alpha), float(ut_params.beta), float(ut_params.kappa)
        margin, all_valid = float(ut_params.in_image_margin_factor), bool(ut_params.require_all_sigma_points_valid)
    dev, dt = means.device, means.dtype
    radii = torch.empty(batch + (C, N, 2), device=dev, dtype=torch.int32)
    means2d = torch.empty(batch + (C, N, 2), device=dev, dtype=dt)
    depths = torch.empty(batch + (C, N), device=dev, dtype=dt)
    conics = torch.empty(batch + (C, N, 3), device=dev, dtype=dt)
    comps = torch.empty(batch + (C, N), device=dev, dtype=dt) if calc_compensations else None
    amap = _lidar_angle_map(lidar, dev) if rolling else None
    mh, mw = (int(amap.shape[0]), int(amap.shape[1])) if amap is not None else (0, 0)
    call("gsx_project_ut_lidar_fwd", ptr(means.contiguous()), ptr(quats.contiguous()), ptr(scales.contiguous()),
         ptr(_c(opacities)), ptr(viewmats0.contiguous()), ptr(_c(viewmats1)) if rolling else None, ptr(Ks.contiguous()),
         *_lidar_fov_args(lidar), int(lidar.spinning_direction), ptr(amap), mh, mw, int(lidar.column_azimuths_rad.shape[0]), B, C,
         N, float(eps2d), float(near_plane), float(far_plane), float(radius_clip), int(rs_type), int(bool(global_z_order)), alpha,
         beta, kappa, margin, int(all_valid), ptr(radii), ptr(means2d), ptr(depths), ptr(conics), ptr(comps))
    return radii, means2d, depths, conics, comps


def lidar_element_rays(viewmats: Tensor, viewmats_rs: Optional[Tensor], lidar, rs_type: int = 4) -> Tensor:
    """[..., C, n_rows, n_columns, 6]: world ray of every element of a spinning lidar (gsx_lidar_rays) - the rays the reference's
    from-world kernels derive per thread for lidar elements; the zero ray outside the fields of view."""
    lead = tuple(viewmats.shape[:-2])
    dev, dt = viewmats.device, viewmats.dtype
    rolling = rs_type != 4
    if rolling and (viewmats_rs is None or viewmats_rs.shape != viewmats.shape):
        raise ValueError("a rolling shutter needs viewmats_rs of the shape of viewmats")
    rows = lidar.row_elevations_rad.to(device=dev, dtype=torch.float32).contiguous()
    cols = lidar.column_azimuths_rad.to(device=dev, dtype=torch.float32).contiguous()
    offs = lidar.row_azimuth_offsets_rad.to(device=dev, dtype=torch.float32).contiguous()
    amap = _lidar_angle_map(lidar, dev) if rolling else None
    mh, mw = (int(amap.shape[0]), int(amap.shape[1])) if amap is not None else (0, 0)
    rays = torch.empty(lead + (rows.shape[0], cols.shape[0], 6), device=dev, dtype=dt)
    call("gsx_lidar_rays", ptr(viewmats.contiguous()), ptr(_c(viewmats_rs)) if rolling else None, ptr(rows), ptr(cols), ptr(offs),
         math.prod(lead), int(rows.shape[0]), int(cols.shape[0]), *_lidar_fov_args(lidar), float(lidar.fov_eps_rad),
         int(lidar.spinning_direction), ptr(amap), mh, mw, int(rs_type), ptr(rays))
    return rays


def _lidar_virtual_layout(lidar, dev):
    """Elements of a lidar tile -> pixels of a square VIRTUAL tile, so that the from-world kernels (square pixel tiles) composite
    lidar tiles unchanged: tile t of the lidar's tiling (n_bins_elevation x n_bins_azimuth, tiles_pack_info = (first, count) into
    tiles_to_elements_map = (column, row) pairs; RasterizeToPixelsFromWorld3DGS.cuh:262-330) becomes the S x S block t of a
    virtual image, its e-th element the block's e-th pixel; the other pixels carry the zero ray (no samples). Returns
    (S, virtual width, virtual height, element index [K], virtual pixel index [K])."""
    pack = lidar.tiles_pack_info.to(device=dev, dtype=torch.int64)
    emap = lidar.tiles_to_elements_map.to(device=dev, dtype=torch.int64)
    tw, th = int(lidar.n_bins_azimuth), int(lidar.n_bins_elevation)
    if pack.shape != (tw * th, 2) or emap.dim() != 2 or emap.shape[1] != 2:
        raise RuntimeError("lidar tiling: tiles_pack_info [n_tiles, 2] and tiles_to_elements_map [n_elements, 2] expected")
    most = int(pack[:, 1].max().item()) if pack.numel() else 0
    if most > 256:
        raise NotImplementedError(f"gsplat_amd: lidar tiles of up to 256 elements are built, this tiling has {most}")
    S = 8 if most <= 64 else 16
    n_cols = int(lidar.column_azimuths_rad.shape[0])
    e = torch.arange(S * S, device=dev)
    valid = e[None, :] < pack[:, 1:2]                                        # [T, S*S]
    src = (pack[:, 0:1] + e[None, :]).clamp_(max=max(emap.shape[0] - 1, 0))  # [T, S*S] rows of the element map
    elem = emap[src]                                                         # [T, S*S, 2] (column, row)
    elem_flat = elem[..., 1] * n_cols + elem[..., 0]
    t = torch.arange(tw * th, device=dev)
    vy = (t // tw)[:, None] * S + e[None, :] // S
    vx = (t % tw)[:, None] * S + e[None, :] % S
    virt_flat = vy * (tw * S) + vx
    return S, tw * S, th * S, elem_flat[valid], virt_flat[valid]


@_op("projection_ut_3dgs_fused")
def projection_ut_3dgs_fused(means, quats, scales, opacities, viewmats0, viewmats1, Ks, image_width, image_height,
                             eps2d, near_plane, far_plane, radius_clip, calc_compensations, camera_model, global_z_order,
                             ut_params, rs_type, radial_coeffs, tangential_coeffs, thin_prism_coeffs, ftheta_coeffs,
                             lidar_coeffs, external_distortion_params):
    """gsplat::projection_ut_3dgs_fused (kernel ``ProjectionUT3DGSFused.cu``): perfect / OpenCV pinhole, orthographic, OpenCV
    fisheye and f-theta cameras, global or rolling shutter, z or Euclidean sort depth. Lidar and external distortion are
    refused, never approximated."""
    rolling = rs_type != _ROLLING_SHUTTER_GLOBAL
    if rolling and viewmats1 is None:
        raise ValueError("a rolling shutter needs viewmats_rs (the pose at the end of the frame)")
    if rs_type not in (0, 1, 2, 3, 4):
        raise ValueError(f"unknown rolling shutter type {rs_type}")
    if camera_model not in (0, 1, 2, 3, 4):
        raise NotImplementedError(f"gsplat_amd: UT projection is built for pinhole, ortho, fisheye, f-theta and lidar cameras, not "
                                  f"'{_CAMERA_MODEL_NAMES.get(camera_model, camera_model)}'")
    if (camera_model == 4) != (lidar_coeffs is not None):
        raise RuntimeError("Lidar coefficients must be given for lidar camera model" if camera_model == 4 else
                           "lidar_coeffs given but camera_model is not lidar")
    if camera_model == 4:
        return _projection_ut_lidar(means, quats, scales, opacities, viewmats0, viewmats1, Ks, eps2d, near_plane, far_plane,
                                    radius_clip, calc_compensations, global_z_order, ut_params, rs_type, lidar_coeffs,
                                    radial_coeffs, tangential_coeffs, thin_prism_coeffs, external_distortion_params)
    _check_f32(means=means, quats=quats, scales=scales, opacities=opacities, viewmats=viewmats0, viewmats_rs=viewmats1, Ks=Ks,
               radial_coeffs=radial_coeffs, tangential_coeffs=tangential_coeffs, thin_prism_coeffs=thin_prism_coeffs)
    batch = tuple(means.shape[:-2])
    N, C, B = means.shape[-2], viewmats0.shape[-3], math.prod(means.shape[:-2])
    if quats.shape != batch + (N, 4) or scales.shape != batch + (N, 3) or viewmats0.shape != batch + (C, 4, 4) \
            or Ks.shape != batch + (C, 3, 3) or (opacities is not None and opacities.shape != batch + (N,)):
        raise ValueError("projection_ut_3dgs_fused: inconsistent input shapes")
    ftheta_rec = None
    if camera_model == 3:  # f-theta: one parameter record per call, no OpenCV coefficients
        if radial_coeffs is not None or tangential_coeffs is not None or thin_prism_coeffs is not None:
            raise ValueError("the f-theta camera model takes ftheta_coeffs, not radial / tangential / thin-prism coefficients")
        if ftheta_coeffs is None:
            raise ValueError("camera_model='ftheta' needs ftheta_coeffs (FThetaCameraDistortionParameters)")
        import ctypes

        p2a, a2p = list(ftheta_coeffs.pixeldist_to_angle_poly), list(ftheta_coeffs.angle_to_pixeldist_poly)
        cde = list(ftheta_coeffs.linear_cde)
        if len(p2a) != 6 or len(a2p) != 6 or len(cde) != 3:
            raise ValueError("ftheta_coeffs: the polynomials have 6 terms, linear_cde 3")
        vals = [1.0 if int(ftheta_coeffs.reference_poly) != 0 else 0.0] + [float(v) for v in p2a] + [float(v) for v in a2p] \
            + [float(ftheta_coeffs.max_angle)] + [float(v) for v in cde]
        ftheta_rec = (ctypes.c_float * 17)(*vals)
    max_angle = None
    if camera_model == 2:  # fisheye: k1..k4 only, plus the per-camera angle limit
        if tangential_coeffs is not None or thin_prism_coeffs is not None:
            raise ValueError("the fisheye camera model takes radial_coeffs [..., C, 4] only")
        if radial_coeffs is None:
            radial_coeffs = torch.zeros(batch + (C, 4), device=means.device, dtype=means.dtype)
        if radial_coeffs.shape != batch + (C, 4):
            raise ValueError(f"fisheye radial_coeffs must have shape [..., C, 4], got {tuple(radial_coeffs.shape)}")
        max_angle = fisheye_max_angle(radial_coeffs, Ks, int(image_width), int(image_height)).contiguous()
    if radial_coeffs is not None:
        if radial_coeffs.shape[:-1] != batch + (C,) or radial_coeffs.shape[-1] not in (4, 6):
            raise ValueError(f"radial_coeffs must have shape [..., C, 6] or [..., C, 4], got {tuple(radial_coeffs.shape)}")
        if radial_coeffs.shape[-1] == 4:
            radial_coeffs = torch.nn.functional.pad(radial_coeffs, (0, 2))
    if tangential_coeffs is not None and tangential_coeffs.shape != batch + (C, 2):
        raise ValueError(f"tangential_coeffs must have shape [..., C, 2], got {tuple(tangential_coeffs.shape)}")
    if thin_prism_coeffs is not None and thin_prism_coeffs.shape != batch + (C, 4):
        raise ValueError(f"thin_prism_coeffs must have shape [..., C, 4], got {tuple(thin_prism_coeffs.shape)}")
    if camera_model == 1 and (radial_coeffs is not None or tangential_coeffs is not None or thin_prism_coeffs is not None):
        raise RuntimeError("ortho camera model does not support radial_coeffs, tangential_coeffs, or thin_prism_coeffs "
                           "parameters")
    alpha, beta, kappa, margin, all_valid = 0.1, 2.0, 0.0, 0.1, False  # Cameras.h:59-64
    if ut_params is not None:
        alpha, beta, kappa = float(ut_params.alpha), float(ut_params.beta), float(ut_params.kappa)
        margin, all_valid = float(ut_params.in_image_margin_factor), bool(ut_params.require_all_sigma_points_valid)
    dev, dt = means.device, means.dtype
    radii = torch.empty(batch + (C, N, 2), device=dev, dtype=torch.int32)
    means2d = torch.empty(batch + (C, N, 2), device=dev, dtype=dt)
    depths = torch.empty(batch + (C, N), device=dev, dtype=dt)
    conics = torch.empty(batch + (C, N, 3), device=dev, dtype=dt)
    comps = torch.empty(batch + (C, N), device=dev, dtype=dt) if calc_compensations else None
    if rolling or not global_z_order or external_distortion_params is not None:
        import ctypes

        if rolling and viewmats1.shape != viewmats0.shape:
            raise ValueError("viewmats_rs must match viewmats shape")
        head = (ptr(means.contiguous()), ptr(quats.contiguous()), ptr(scales.contiguous()),
                ptr(_c(opacities)), ptr(viewmats0.contiguous()), ptr(_c(viewmats1)) if rolling else None, ptr(Ks.contiguous()),
                ptr(_c(radial_coeffs)), ptr(_c(tangential_coeffs)), ptr(_c(thin_prism_coeffs)), ptr(max_angle),
                ctypes.addressof(ftheta_rec) if ftheta_rec is not None else None)
        tail = (B, C, N, int(image_width), int(image_height),
                float(eps2d), float(near_plane), float(far_plane), float(radius_clip), int(camera_model), int(rs_type),
                int(bool(global_z_order)), alpha, beta, kappa, margin, int(all_valid), ptr(radii), ptr(means2d), ptr(depths),
                ptr(conics), ptr(comps))
        if external_distortion_params is not None:  # behind a windshield: sigma points go through the forward polynomials
            ext = _windshield_record(external_distortion_params)
            call("gsx_project_ut_ext_fwd", *head, ctypes.addressof(ext), *tail)
        else:
            call("gsx_project_ut_rs_fwd", *head, *tail)
        return radii, means2d, depths, conics, comps
    if ftheta_rec is not None:
        import ctypes

        call("gsx_project_ut_ftheta_fwd", ptr(means.contiguous()), ptr(quats.contiguous()), ptr(scales.contiguous()),
             ptr(_c(opacities)), ptr(viewmats0.contiguous()), ptr(Ks.contiguous()), ctypes.addressof(ftheta_rec), B, C, N,
             int(image_width), int(image_height), float(eps2d), float(near_plane), float(far_plane), float(radius_clip),
             alpha, beta, kappa, margin, int(all_valid), ptr(radii), ptr(means2d), ptr(depths), ptr(conics), ptr(comps))
        return radii, means2d, depths, conics, comps
    call("gsx_project_ut_fwd", ptr(means.contiguous()), ptr(quats.contiguous()), ptr(scales.contiguous()),
         ptr(_c(opacities)), ptr(viewmats0.contiguous()), ptr(Ks.contiguous()), ptr(_c(radial_coeffs)),
         ptr(_c(tangential_coeffs)), ptr(_c(thin_prism_coeffs)), ptr(max_angle), B, C, N, int(image_width),
         int(image_height),
         float(eps2d), float(near_plane), float(far_plane), float(radius_clip), int(camera_model), alpha, beta, kappa,
         margin, int(all_valid), ptr(radii), ptr(means2d), ptr(depths), ptr(conics), ptr(comps))
    return radii, means2d, depths, conics, comps


def pinhole_pixel_rays(viewmats: Tensor, Ks: Tensor, width: int, height: int) -> Tensor:
    """[..., C, H, W, 6]: world-space origin | unit direction of the ray through every pixel centre of perfect pinhole
    cameras with a global shutter (what the reference's `_generate_rays` yields for that model,
    gsplat/cuda/_torch_impl_eval3d.py:91-132): direction = R^T normalise(K^-1 (x + 0.5, y + 0.5, 1)), origin = -R^T t."""
    dt, dev = viewmats.dtype, viewmats.device
    xs = torch.arange(width, device=dev, dtype=dt) + 0.5
    ys = torch.arange(height, device=dev, dtype=dt) + 0.5
    fx, fy, cx, cy = (Ks[..., 0, 0, None, None], Ks[..., 1, 1, None, None], Ks[..., 0, 2, None, None],
                      Ks[..., 1, 2, None, None])
    dx = ((xs[None, :] - cx) / fx).expand(Ks.shape[:-2] + (height, width))
    dy = ((ys[:, None] - cy) / fy).expand(Ks.shape[:-2] + (height, width))
    d = torch.stack([dx, dy, torch.ones_like(dx)], dim=-1)
    d = d / d.norm(dim=-1, keepdim=True)
    R, t = viewmats[..., :3, :3], viewmats[..., :3, 3]
    d_world = torch.einsum("...ji,...hwj->...hwi", R, d)
    o_world = -torch.einsum("...ji,...j->...i", R, t)
    return torch.cat([o_world[..., None, None, :].expand_as(d_world), d_world], dim=-1).contiguous()


def _rotmat_to_quat_wxyz(R: Tensor) -> Tensor:
    """glm::quat_cast of row-major rotation matrices [..., 3, 3] -> (w, x, y, z) (Cameras.cuh:85-101)."""
    m = lambda i, j: R[..., i, j]  # noqa: E731
    four = torch.stack([m(0, 0) + m(1, 1) + m(2, 2), m(0, 0) - m(1, 1) - m(2, 2), m(1, 1) - m(0, 0) - m(2, 2),
                        m(2, 2) - m(0, 0) - m(1, 1)], dim=-1)
    big = four.argmax(dim=-1)
    val = torch.sqrt(four.gather(-1, big[..., None])[..., 0] + 1.0) * 0.5
    mult = 0.25 / val
    a, b, c = (m(2, 1) - m(1, 2)) * mult, (m(0, 2) - m(2, 0)) * mult, (m(1, 0) - m(0, 1)) * mult
    d, e, f = (m(1, 0) + m(0, 1)) * mult, (m(0, 2) + m(2, 0)) * mult, (m(2, 1) + m(1, 2)) * mult
    cands = torch.stack([torch.stack([val, a, b, c], -1), torch.stack([a, val, d, e], -1), torch.stack([b, d, val, f], -1),
                         torch.stack([c, e, f, val], -1)], dim=-2)  # [..., case, 4]
    return cands.gather(-2, big[..., None, None].expand(big.shape + (1, 4)))[..., 0, :]


def _quat_rotate_wxyz(q: Tensor, v: Tensor) -> Tensor:
    """glm::rotate(q, v) = v + 2 (w (u x v) + u x (u x v))."""
    u = q[..., 1:]
    uv = torch.cross(u, v, dim=-1)
    return v + 2.0 * (q[..., :1] * uv + torch.cross(u, uv, dim=-1))


def _slerp_wxyz(q0: Tensor, q1: Tensor, t: Tensor) -> Tensor:
    """glm::slerp on the short arc (component-wise mix when nearly parallel), normalised (Cameras.cuh:362-429)."""
    cos = (q0 * q1).sum(-1, keepdim=True)
    q1 = torch.where(cos < 0, -q1, q1)
    cos = cos.abs()
    t = t[..., None]
    angle = torch.acos(cos.clamp(max=1.0))
    sin = torch.sin(angle)
    safe = torch.where(sin > 0, sin, torch.ones_like(sin))
    wa = torch.where(cos > 1.0 - 1.1920929e-07, 1.0 - t, torch.sin((1.0 - t) * angle) / safe)
    wb = torch.where(cos > 1.0 - 1.1920929e-07, t, torch.sin(t * angle) / safe)
    q = wa * q0 + wb * q1
    return q / q.norm(dim=-1, keepdim=True)


def pinhole_pixel_rays_rolling(viewmats: Tensor, viewmats_rs: Tensor, Ks: Tensor, width: int, height: int, rs_type: int) -> Tensor:
    """pinhole_pixel_rays under a ROLLING shutter: the pose of a pixel is the one interpolated (translation lerp, rotation slerp)
    at the relative frame time at which its row / column is read - floor(y) / (H - 1) top-to-bottom, floor(x) / (W - 1)
    left-to-right, (H - ceil(y)) / (H - 1) bottom-to-top, (W - ceil(x)) / (W - 1) right-to-left at the pixel centre
    (Cameras.cuh:503-546; gsplat/cuda/_torch_cameras.py:424-553). [..., C, H, W, 6]."""
    dt, dev = viewmats.dtype, viewmats.device
    xs = torch.arange(width, device=dev, dtype=dt) + 0.5
    ys = torch.arange(height, device=dev, dtype=dt) + 0.5
    if rs_type == 0:
        tm, along_rows = (torch.floor(ys) / (height - 1) if height > 1 else torch.full_like(ys, 0.5)), True
    elif rs_type == 2:
        tm, along_rows = ((height - torch.ceil(ys)) / (height - 1) if height > 1 else torch.full_like(ys, 0.5)), True
    elif rs_type == 1:
        tm, along_rows = (torch.floor(xs) / (width - 1) if width > 1 else torch.full_like(xs, 0.5)), False
    elif rs_type == 3:
        tm, along_rows = ((width - torch.ceil(xs)) / (width - 1) if width > 1 else torch.full_like(xs, 0.5)), False
    else:
        raise ValueError(f"rolling shutter type {rs_type}")
    L = tm.shape[0]
    lead = viewmats.shape[:-2]
    q0 = _rotmat_to_quat_wxyz(viewmats[..., :3, :3])[..., None, :].expand(lead + (L, 4))
    q1 = _rotmat_to_quat_wxyz(viewmats_rs[..., :3, :3])[..., None, :].expand(lead + (L, 4))
    t0, t1 = viewmats[..., None, :3, 3], viewmats_rs[..., None, :3, 3]
    tl = tm.expand(lead + (L,))
    q = _slerp_wxyz(q0, q1, tl)                                   # [..., C, L, 4] pose of every line
    t = (1.0 - tl[..., None]) * t0 + tl[..., None] * t1           # [..., C, L, 3]
    q_inv = torch.cat([q[..., :1], -q[..., 1:]], dim=-1)
    origin = _quat_rotate_wxyz(q_inv, -t)                          # camera position at the line's time
    fx, fy, cx, cy = (Ks[..., 0, 0, None, None], Ks[..., 1, 1, None, None], Ks[..., 0, 2, None, None],
                      Ks[..., 1, 2, None, None])
    dx = ((xs[None, :] - cx) / fx).expand(Ks.shape[:-2] + (height, width))
    dy = ((ys[:, None] - cy) / fy).expand(Ks.shape[:-2] + (height, width))
    d = torch.stack([dx, dy, torch.ones_like(dx)], dim=-1)
    d = d / d.norm(dim=-1, keepdim=True)
    if along_rows:
        qi, oi = q_inv[..., :, None, :], origin[..., :, None, :]   # [..., C, H, 1, .]
    else:
        qi, oi = q_inv[..., None, :, :], origin[..., None, :, :]   # [..., C, 1, W, .]
    d_world = _quat_rotate_wxyz(qi.expand(d.shape[:-1] + (4,)), d)
    return torch.cat([oi.expand_as(d_world), d_world], dim=-1).contiguous()


def _ftheta_record(ftheta_coeffs):
    """The 17-float host record the C-ABI takes for an f-theta camera: reference_poly | pixeldist_to_angle_poly[6] |
    angle_to_pixeldist_poly[6] | max_angle | linear_cde[3] (FThetaCameraDistortionParameters, Cameras.h:103-117)."""
    import ctypes

    p2a, a2p = list(ftheta_coeffs.pixeldist_to_angle_poly), list(ftheta_coeffs.angle_to_pixeldist_poly)
    cde = list(ftheta_coeffs.linear_cde)
    if len(p2a) != 6 or len(a2p) != 6 or len(cde) != 3:
        raise ValueError("ftheta_coeffs: the polynomials have 6 terms, linear_cde 3")
    vals = [1.0 if int(ftheta_coeffs.reference_poly) != 0 else 0.0] + [float(v) for v in p2a] + [float(v) for v in a2p] \
        + [float(ftheta_coeffs.max_angle)] + [float(v) for v in cde]
    return (ctypes.c_float * 17)(*vals)


def _windshield_record(params):
    """The 84-float host record the C-ABI takes for the external (windshield) distortion: horizontal | vertical | horizontal
    inverse | vertical inverse polynomial of BivariateWindshieldModelParameters, each padded to the order-5 triangular layout of
    21 coefficients (block k = the coefficients in x of y^k; pad_coefficients_to_max_order, ExternalDistortion.cuh:104-124)."""
    import ctypes

    def pad(t, what):
        vals = [float(v) for v in (t.detach().cpu().reshape(-1).tolist() if isinstance(t, Tensor) else list(t))]
        order = (-3 + int(math.sqrt(1 + 8 * len(vals)))) // 2  # compute_order
        if len(vals) not in (1, 3, 6, 10, 15, 21) or (order + 1) * (order + 2) // 2 != len(vals):
            raise RuntimeError(f"Invalid number of bivariate polynomial coefficients: {len(vals)}. Expected triangular number: "
                               f"1, 3, 6, 10, 15, or 21. ({what})")
        out, src = [], 0
        for k in range(6):
            n_src = order - k + 1 if k <= order else 0
            out += vals[src:src + n_src] + [0.0] * (6 - k - n_src)
            src += n_src
        return out

    rec = (pad(params.horizontal_poly, "horizontal_poly") + pad(params.vertical_poly, "vertical_poly")
           + pad(params.horizontal_poly_inverse, "horizontal_poly_inverse") + pad(params.vertical_poly_inverse, "vertical_poly_inverse"))
    return (ctypes.c_float * 84)(*rec)


@_op("distort_camera_rays")
def distort_camera_rays(rays, horizontal_poly, vertical_poly, horizontal_poly_inverse, vertical_poly_inverse, reference_poly,
                        inverse):
    """gsplat::distort_camera_rays (ExternalDistortionWrappers.cu:96-160): the windshield model on rays [..., 3]; `inverse`
    applies the inverse pair of polynomials."""
    import ctypes
    import types as _types

    if rays.dtype != torch.float32 or rays.dim() < 1 or rays.shape[-1] != 3:
        raise RuntimeError("rays must be float32 with shape [..., 3]")
    rec = _windshield_record(_types.SimpleNamespace(horizontal_poly=horizontal_poly, vertical_poly=vertical_poly,
                                                    horizontal_poly_inverse=horizontal_poly_inverse,
                                                    vertical_poly_inverse=vertical_poly_inverse))
    base = ctypes.addressof(rec)
    h, v = (base + 4 * 42, base + 4 * 63) if inverse else (base, base + 4 * 21)
    rays = rays.contiguous()
    out = torch.empty_like(rays)
    call("gsx_distort_camera_rays", ptr(rays), rays.numel() // 3, h, v, ptr(out))
    return out


@_op("eval_bivariate_poly")
def eval_bivariate_poly(x, y, poly_coeffs, order):
    """gsplat::eval_bivariate_poly (ExternalDistortionWrappers.cu:30-94)."""
    import ctypes
    import types as _types

    if x.dtype != torch.float32 or y.dtype != torch.float32 or x.numel() != y.numel():
        raise RuntimeError("x and y must be float32 tensors with the same number of elements")
    one = _types.SimpleNamespace(horizontal_poly=poly_coeffs, vertical_poly=[0.0], horizontal_poly_inverse=[0.0],
                                 vertical_poly_inverse=[0.0])
    rec = _windshield_record(one)
    x, y = x.contiguous(), y.contiguous()
    out = torch.empty_like(x)
    call("gsx_eval_bivariate_poly", ptr(x), ptr(y), x.numel(), ctypes.addressof(rec), ptr(out))
    return out


def camera_pixel_rays(viewmats: Tensor, viewmats_rs: Optional[Tensor], Ks: Tensor, width: int, height: int, camera_model: int = 0,
                      rs_type: int = 4, radial_coeffs: Optional[Tensor] = None,
                      tangential_coeffs: Optional[Tensor] = None, thin_prism_coeffs: Optional[Tensor] = None,
                      ftheta_coeffs=None, external_distortion_params=None) -> Tensor:
    """[..., C, H, W, 6]: world-space origin | unit direction of the ray through every pixel centre, for every built camera model
    (perfect / OpenCV-distorted pinhole, orthographic, OpenCV fisheye, f-theta) under a global or rolling shutter - what the
    reference's from-world kernels derive per thread when no `rays` are passed (RasterizeToPixelsFromWorld3DGS.cuh:349-529).
    A pixel the model cannot invert gets the zero ray (no samples). One kernel: gsx_camera_rays (csrc/projection_ut.hip)."""
    import ctypes

    lead = tuple(viewmats.shape[:-2])
    I = math.prod(lead)
    dev, dt = viewmats.device, viewmats.dtype
    rolling = rs_type != 4  # RollingShutterType.GLOBAL
    if rolling and (viewmats_rs is None or viewmats_rs.shape != viewmats.shape):
        raise ValueError("a rolling shutter needs viewmats_rs of the shape of viewmats")
    ftheta_rec, max_angle = None, None
    if camera_model == 3:
        if ftheta_coeffs is None:
            raise ValueError("camera_model='ftheta' needs ftheta_coeffs (FThetaCameraDistortionParameters)")
        if radial_coeffs is not None or tangential_coeffs is not None or thin_prism_coeffs is not None:
            raise ValueError("the f-theta camera model takes ftheta_coeffs, not radial / tangential / thin-prism coefficients")
        ftheta_rec = _ftheta_record(ftheta_coeffs)
    if camera_model == 2:
        if tangential_coeffs is not None or thin_prism_coeffs is not None:
            raise ValueError("the fisheye camera model takes radial_coeffs [..., C, 4] only")
        if radial_coeffs is None:
            radial_coeffs = torch.zeros(lead + (4,), device=dev, dtype=dt)
        max_angle = fisheye_max_angle(radial_coeffs, Ks, int(width), int(height)).contiguous()
    if camera_model == 1 and (radial_coeffs is not None or tangential_coeffs is not None or thin_prism_coeffs is not None):
        raise RuntimeError("ortho camera model does not support radial_coeffs, tangential_coeffs, or thin_prism_coeffs "
                           "parameters")
    if radial_coeffs is not None and radial_coeffs.shape[-1] == 4:
        radial_coeffs = torch.nn.functional.pad(radial_coeffs, (0, 2))
    _check_f32(viewmats=viewmats, viewmats_rs=viewmats_rs, Ks=Ks, radial_coeffs=radial_coeffs,
               tangential_coeffs=tangential_coeffs, thin_prism_coeffs=thin_prism_coeffs)
    rays = torch.empty(lead + (int(height), int(width), 6), device=dev, dtype=dt)
    head = (ptr(viewmats.contiguous()), ptr(_c(viewmats_rs)) if rolling else None, ptr(Ks.contiguous()),
            ptr(_c(radial_coeffs)), ptr(_c(tangential_coeffs)), ptr(_c(thin_prism_coeffs)), ptr(max_angle),
            ctypes.addressof(ftheta_rec) if ftheta_rec is not None else None)
    tail = (I, int(width), int(height), int(camera_model), int(rs_type), ptr(rays))
    if external_distortion_params is not None:  # behind a windshield: the camera model's ray goes through the inverse polynomials
        ext = _windshield_record(external_distortion_params)
        call("gsx_camera_rays_ext", *head, ctypes.addressof(ext), *tail)
    else:
        call("gsx_camera_rays", *head, *tail)
    return rays


class _FromWorldCompositing(torch.autograd.Function):
    """Autograd of the from-world compositing (the reference attaches it in C++: Rasterization.cpp:3266-3340 around
    rasterize_to_pixels_from_world_3dgs_bwd, kernel RasterizeToPixelsFromWorld3DGSBwd.cu). Gradients reach means / quats /
    scales / colors / opacities, the rays (origins and directions: one lane owns a pixel, six sums in registers) and the
    backgrounds (sum of T_final v_render over the pixels); cameras are constants. The kernel returns per-(image, Gaussian) rows
    [v_mean(3) | torque(3) | v_scale(3) | v_opacity | v_colors(D)]; v_quats follows from the torque in closed form (backward below).
    Validated against the gradients the reference's own autograd gives (tests/golden/eval3d_ref.npz)."""

    @staticmethod
    def forward(ctx, means, quats, scales, colors, opacities, rays, backgrounds, masks, width, height, tile_size,
                tile_offsets, flatten_ids, want_counts=False, hit_distance=False, want_normals=False, *tracked):
        # `tracked`: the op's other tensor inputs (poses, intrinsics, distortion coefficients). They receive no gradient, but - like
        # torch::autograd::Function::apply() in the reference's C++ adapter - an input that requires grad keeps the outputs on
        # the autograd graph (the reference's test_rasterize_eval3d_autograd_tracks_any_tensor_input)
        ctx.n_tracked = len(tracked)
        batch = tuple(means.shape[:-2])
        N, C, D = means.shape[-2], colors.shape[-3], colors.shape[-1]
        I = math.prod(batch) * C
        th, tw = tile_offsets.shape[-2], tile_offsets.shape[-1]
        dev, dt = means.device, means.dtype
        renders = torch.empty(batch + (C, height, width, D), device=dev, dtype=dt)
        alphas = torch.empty(batch + (C, height, width, 1), device=dev, dtype=dt)
        last_ids = torch.empty(batch + (C, height, width), device=dev, dtype=torch.int32)
        args = [t.contiguous() for t in (means, quats, scales, colors, opacities, rays)]
        bg, mk = _c(backgrounds), _c(masks)
        off, fl = tile_offsets.contiguous(), flatten_ids.contiguous()
        counts = torch.empty(batch + (C, height, width), device=dev, dtype=torch.int32) if want_counts else None
        normals = torch.empty(batch + (C, height, width, 3), device=dev, dtype=dt) if want_normals else None
        call("gsx_raster_world_fwd_ex", *[ptr(t) for t in args], ptr(bg), ptr(mk), ptr(off), ptr(fl), I, C, N, fl.numel(), D,
             int(width), int(height), int(tile_size), tw, th, int(bool(hit_distance)), ptr(renders), ptr(alphas), ptr(last_ids),
             ptr(counts), ptr(normals))
        ctx.save_for_backward(*args, off, fl, alphas, last_ids, *([bg] if bg is not None else []),
                              *([mk] if mk is not None else []))
        ctx.flags = (bg is not None, mk is not None, I, C, N, D, int(width), int(height), int(tile_size), tw, th, batch)
        ctx.extras = (bool(hit_distance), bool(want_normals))
        ctx.rays_shape = tuple(rays.shape)
        ctx.mark_non_differentiable(last_ids)
        if counts is None:
            counts = torch.empty(0, device=dev, dtype=torch.int32)
        ctx.mark_non_differentiable(counts)
        if normals is None:
            normals = torch.empty(0, device=dev, dtype=dt)
            ctx.mark_non_differentiable(normals)
        return renders, alphas, last_ids, counts, normals

    @staticmethod
    def backward(ctx, v_renders, v_alphas, _v_last, _v_counts=None, v_normals=None):
        has_bg, has_mk, I, C, N, D, width, height, tile_size, tw, th, batch = ctx.flags
        hit_distance, want_normals = ctx.extras
        want_rays = ctx.needs_input_grad[5]
        extra = hit_distance or want_normals or want_rays
        saved = list(ctx.saved_tensors)
        means, quats, scales, colors, opacities, rays, off, fl, alphas, last_ids = saved[:10]
        rest = saved[10:]
        bg = rest.pop(0) if has_bg else None
        mk = rest.pop(0) if has_mk else None
        width_rows = 10 + D  # v_mean 3 | torque 3 | v_scale 3 | v_opacity | v_colors
        rows = torch.zeros((I * N, width_rows), device=means.device, dtype=means.dtype)
        v_r = (torch.zeros_like(alphas).expand(alphas.shape[:-1] + (D,)) if v_renders is None else v_renders).contiguous()
        v_a = None if v_alphas is None else v_alphas.contiguous()
        v_rays = None
        if extra:
            v_n = None if (v_normals is None or not want_normals) else v_normals.contiguous()
            v_rays = torch.zeros(ctx.rays_shape, device=means.device, dtype=means.dtype) if want_rays else None
            call("gsx_raster_world_bwd_ex", ptr(means), ptr(quats), ptr(scales), ptr(colors), ptr(opacities), ptr(rays), ptr(bg),
                 ptr(mk), ptr(off), ptr(fl), ptr(alphas), ptr(last_ids), ptr(v_r), ptr(v_a), ptr(v_n), I, C, N, fl.numel(), D,
                 width, height, tile_size, tw, th, int(hit_distance), ptr(rows), width_rows, ptr(v_rays))
        else:
            call("gsx_raster_world_bwd", ptr(means), ptr(quats), ptr(scales), ptr(colors), ptr(opacities), ptr(rays), ptr(bg),
                 ptr(mk), ptr(off), ptr(fl), ptr(alphas), ptr(last_ids), ptr(v_r), ptr(v_a), I, C, N, fl.numel(), D, width,
                 height, tile_size, tw, th, ptr(rows), width_rows)
        B = I // C
        per = rows.view(B, C, N, width_rows)
        v_means = per[..., 0:3].sum(1).reshape(means.shape)
        v_scales = per[..., 6:9].sum(1).reshape(scales.shape)
        # torque (dL/dw for R -> exp([w]x) R) -> the raw quaternion: v_q = (2 / |q|) (0, torque) (x) q_unit, Hamilton product with
        # the scalar part first: (0, t) (x) (w, v) = (-t . v, w t + t x v)
        tq = per[..., 3:6].sum(1).reshape(means.shape)
        qn_inv = 1.0 / quats.norm(dim=-1, keepdim=True)
        qu = quats * qn_inv
        qw, qv = qu[..., :1], qu[..., 1:]
        v_quats = 2.0 * qn_inv * torch.cat([-(tq * qv).sum(-1, keepdim=True), qw * tq + torch.cross(tq, qv, dim=-1)], dim=-1)
        v_opac = per[..., 9].reshape(opacities.shape)
        v_cols = per[..., 10:10 + D].reshape(colors.shape)
        v_bg = None
        if has_bg and ctx.needs_input_grad[6]:  # render = sum + T_final background: v_background = sum over the pixels of T_final v_render
            v_bg = (v_r * (1.0 - alphas)).sum(dim=(-3, -2)).reshape(bg.shape)
        return (v_means, v_quats, v_scales, v_cols, v_opac, v_rays, v_bg) + (None,) * (9 + ctx.n_tracked)


def _from_world_lidar(means, quats, scales, colors, opacities, backgrounds, masks, image_width, image_height, viewmats0,
                      viewmats1, rs_type, rays, lidar, tile_offsets, flatten_ids, return_sample_counts, use_hit_distance,
                      return_normals, return_last_ids):
    """The from-world rasterizer for a spinning lidar: the "image" is [n_rows, n_columns] ELEMENTS, a tile of the lidar's tiling
    holds an arbitrary set of them (tiles_to_elements_map). The elements of a tile are laid out as the pixels of a square
    virtual tile (_lidar_virtual_layout), the compositing kernels run on the virtual image unchanged, and the results are
    carried back to the elements; the index moves are differentiable torch ops, so the rays' and every other gradient follow."""
    batch = tuple(means.shape[:-2])
    N, C, D = means.shape[-2], viewmats0.shape[-3], colors.shape[-1]
    lead = batch + (C,)
    I = math.prod(lead)
    dev, dt = means.device, means.dtype
    n_rows, n_cols = int(lidar.row_elevations_rad.shape[0]), int(lidar.column_azimuths_rad.shape[0])
    if (int(image_width), int(image_height)) != (n_cols, n_rows):
        raise RuntimeError(f"a lidar renders [n_rows, n_columns] = [{n_rows}, {n_cols}] elements, got height {image_height}, "
                           f"width {image_width}")
    if colors.shape != batch + (C, N, D) or opacities.shape != batch + (C, N):
        raise ValueError("eval3d takes dense rows: colors [..., C, N, D] and opacities [..., C, N]")
    P = n_rows * n_cols
    if rays is None:
        with torch.no_grad():
            rays = lidar_element_rays(viewmats0, viewmats1, lidar, int(rs_type))
    elif rays.numel() != I * P * 6 or rays.shape[-1] != 6:
        raise ValueError(f"rays must be [..., C, n_rows * n_columns, 6], got {tuple(rays.shape)}")
    _check_f32(means=means, quats=quats, scales=scales, colors=colors, opacities=opacities, rays=rays)
    S, VW, VH, eidx, vidx = _lidar_virtual_layout(lidar, dev)
    th, tw = tile_offsets.shape[-2], tile_offsets.shape[-1]
    if (tw * S, th * S) != (VW, VH):
        raise RuntimeError(f"tile_offsets must be [..., C, n_bins_elevation, n_bins_azimuth], got {tuple(tile_offsets.shape)}")
    rays_e = rays.reshape(I, P, 6)
    rays_v = torch.zeros((I, VH * VW, 6), device=dev, dtype=dt).index_copy(1, vidx, rays_e.index_select(1, eidx))
    out = _FromWorldCompositing.apply(
        means, quats, scales, colors, opacities, rays_v.reshape(lead + (VH, VW, 6)), backgrounds, masks, VW, VH, S,
        tile_offsets, flatten_ids, bool(return_sample_counts), bool(use_hit_distance), bool(return_normals),
        *[t for t in (viewmats0, viewmats1) if isinstance(t, Tensor)])
    renders_v, alphas_v, last_v, counts_v, normals_v = out

    def to_elements(x_v, k, base):  # [..., C, VH, VW(, k)] -> [..., C, n_rows, n_columns(, k)]
        x = base.index_copy(1, eidx, x_v.reshape(I, VH * VW, k).index_select(1, vidx))
        return x.reshape(lead + (n_rows, n_cols) + ((k,) if x_v.dim() > len(lead) + 2 else ()))

    base_c = torch.zeros((I, P, D), device=dev, dtype=dt)
    if backgrounds is not None:  # an element no tile holds shows its background
        base_c = base_c + backgrounds.reshape(I, 1, D)
    renders = to_elements(renders_v, D, base_c)
    alphas = to_elements(alphas_v, 1, torch.zeros((I, P, 1), device=dev, dtype=dt))
    last_ids = to_elements(last_v[..., None], 1, torch.full((I, P, 1), -1, device=dev, dtype=torch.int32))[..., 0] \
        if return_last_ids else None
    counts = to_elements(counts_v[..., None], 1, torch.zeros((I, P, 1), device=dev, dtype=torch.int32))[..., 0] \
        if return_sample_counts else None
    normals = to_elements(normals_v, 3, torch.zeros((I, P, 3), device=dev, dtype=dt)) if return_normals else None
    return renders, alphas, last_ids, counts, normals


@_op("rasterize_to_pixels_from_world_3dgs")
def rasterize_to_pixels_from_world_3dgs(means, quats, scales, colors, opacities, backgrounds, masks, image_width,
                                        image_height, tile_size, viewmats0, viewmats1, Ks, camera_model, ut_params, rs_type,
                                        rays, radial_coeffs, tangential_coeffs, thin_prism_coeffs, ftheta_coeffs,
                                        lidar_coeffs, external_distortion_params, tile_offsets, flatten_ids,
                                        return_sample_counts, use_hit_distance, return_normals, renderer_config,
                                        return_last_ids, unsafe_masked_tile_outputs=False):
    """gsplat::rasterize_to_pixels_from_world_3dgs (forward + autograd, like the reference's C++ autograd function,
    Rasterization.cpp:3266-3340): dense rows, rays either given or generated for every built camera model (global or rolling
    shutter: camera_pixel_rays), sample counts, hit distance, normals, gradients to the rays and the backgrounds. Lidar and
    external distortion are refused, never approximated."""
    if renderer_config not in (0, 1):
        raise ValueError(f"unknown renderer_config {renderer_config}")
    # renderer_config 1 (PARALLEL_BATCH, Rasterization.cpp:106-117) is a scheduling choice of the reference (its lists split over
    # several CTAs); this backend has one schedule, the results are the same
    rolling = rs_type != _ROLLING_SHUTTER_GLOBAL
    if rolling and viewmats1 is None:
        raise ValueError("a rolling shutter needs viewmats_rs (the pose at the end of the frame)")
    if (camera_model == 4) != (lidar_coeffs is not None):
        raise RuntimeError("Lidar coefficients must be given if and only if camera model is lidar")
    if camera_model == 4:
        return _from_world_lidar(means, quats, scales, colors, opacities, backgrounds, masks, image_width, image_height,
                                 viewmats0, viewmats1 if rolling else None, rs_type, rays, lidar_coeffs, tile_offsets, flatten_ids,
                                 return_sample_counts, use_hit_distance, return_normals, return_last_ids)
    if rays is None:
        if camera_model not in (0, 1, 2, 3):
            raise NotImplementedError(f"gsplat_amd: eval3d generates rays for pinhole, ortho, fisheye and f-theta cameras, not "
                                      f"'{_CAMERA_MODEL_NAMES.get(camera_model, camera_model)}'")
        with torch.no_grad():  # the pose of a pixel is the one at the time its row / column is read
            rays = camera_pixel_rays(viewmats0, viewmats1 if rolling else None, Ks, int(image_width), int(image_height),
                                     int(camera_model), int(rs_type), radial_coeffs, tangential_coeffs, thin_prism_coeffs,
                                     ftheta_coeffs if camera_model == 3 else None, external_distortion_params)
    _check_f32(means=means, quats=quats, scales=scales, colors=colors, opacities=opacities, rays=rays)
    batch = tuple(means.shape[:-2])
    N, C, D = means.shape[-2], viewmats0.shape[-3], colors.shape[-1]
    if colors.shape != batch + (C, N, D) or opacities.shape != batch + (C, N):
        raise ValueError("eval3d takes dense rows: colors [..., C, N, D] and opacities [..., C, N]")
    image_dims, I, th, tw, _ = _raster_dims(tile_offsets, colors)
    if rays.numel() != I * image_height * image_width * 6 or rays.shape[-1] != 6:
        raise ValueError(f"rays must be [..., C, H, W, 6] (or [..., C, H * W, 6]), got {tuple(rays.shape)}")
    rays = rays.reshape(batch + (C, int(image_height), int(image_width), 6))
    renders, alphas, last_ids, counts, normals = _FromWorldCompositing.apply(
        means, quats, scales, colors, opacities, rays, backgrounds, masks, int(image_width), int(image_height),
        int(tile_size), tile_offsets, flatten_ids, bool(return_sample_counts), bool(use_hit_distance), bool(return_normals),
        *[t for t in (viewmats0, viewmats1, Ks, radial_coeffs, tangential_coeffs, thin_prism_coeffs) if isinstance(t, Tensor)])
    return (renders, alphas, (last_ids if return_last_ids else None), (counts if return_sample_counts else None),
            (normals if return_normals else None))


# ----------------------------------------------------------------------------------------------
# whole-pipeline ops (what the reference's gsplat.rasterization() / rasterization_2dgs() call)
# ----------------------------------------------------------------------------------------------
_CAMERA_MODEL_NAMES = {0: "pinhole", 1: "ortho", 2: "fisheye", 3: "ftheta", 4: "lidar"}  # Common.h:75-82
_ROLLING_SHUTTER_GLOBAL = 4  # _wrapper.py RollingShutterType.GLOBAL
_SELF_DIFFERENTIABLE = ("rasterize_to_pixels_from_world_3dgs",)


def _empty(like: Tensor, dtype=None) -> Tensor:
    """The reference returns ``at::empty({0})`` for outputs a mode does not produce (Rendering.cpp:688, 713, 1348)."""
    return torch.empty(0, device=like.device, dtype=like.dtype if dtype is None else dtype)


@_op("rasterization_3dgs")
def rasterization_3dgs(means, covars, quats, scales, opacities, colors, viewmats, Ks, image_width, image_height,
                       tile_size, eps2d, near_plane, far_plane, radius_clip, backgrounds, packed, sparse_grad, absgrad,
                       calc_compensations, rasterize_mode_is_classic, camera_model, segmented, channel_chunk, has_color,
                       sh_degree, extra_signals, extra_signals_sh_degree, append_depth, expected_depth, with_eval3d,
                       with_ut, rays, viewmats_rs, ut_params, rolling_shutter, radial_coeffs, tangential_coeffs,
                       thin_prism_coeffs, ftheta_coeffs, lidar_coeffs, external_distortion_params, global_z_order,
                       use_hit_distance, return_normals, renderer_config, process_group_name, world_size):
    """gsplat::rasterization_3dgs (host ``Rendering.cpp:745-1481``): the flattened argument list of
    ``gsplat.rasterization()`` (``gsplat/rendering.py:601-650``) mapped back onto the orchestrator in ``rendering.py``.
    ``ut_params`` / ``ftheta_coeffs`` are always passed by the reference (default-constructed records) and only read by
    the 3DGUT / f-theta paths, which are rejected like every other out-of-scope argument."""
    from .rendering import rasterization

    if renderer_config != 0 and not with_eval3d:
        raise ValueError("RendererConfig PARALLEL_BATCH requires with_eval3d=True; the classic path only supports "
                         "MIXED_BATCH")
    if camera_model not in _CAMERA_MODEL_NAMES:
        raise ValueError(f"unknown camera_model id {camera_model}")
    depth = ""
    if append_depth or use_hit_distance:
        depth = ("Ed" if expected_depth else "d") if use_hit_distance else ("ED" if expected_depth else "D")
    if has_color:
        render_mode = "RGB" + (("-" if use_hit_distance else "+") + depth if depth else "")
    else:
        render_mode = depth
    rc, ra, meta = rasterization(
        means, quats, scales, opacities, colors if has_color else None, viewmats, Ks, image_width, image_height,
        near_plane=near_plane, far_plane=far_plane, radius_clip=radius_clip, eps2d=eps2d,
        sh_degree=None if sh_degree < 0 else sh_degree, packed=packed, tile_size=tile_size, backgrounds=backgrounds,
        render_mode=render_mode, sparse_grad=sparse_grad, absgrad=absgrad,
        rasterize_mode="antialiased" if calc_compensations else "classic", channel_chunk=channel_chunk,
        distributed=process_group_name is not None or world_size > 1, camera_model=_CAMERA_MODEL_NAMES[camera_model],
        segmented=segmented, covars=covars, with_ut=with_ut, with_eval3d=with_eval3d, return_normals=return_normals,
        global_z_order=global_z_order, rays=rays, radial_coeffs=radial_coeffs, tangential_coeffs=tangential_coeffs,
        thin_prism_coeffs=thin_prism_coeffs, ftheta_coeffs=ftheta_coeffs if camera_model == 3 else None,
        lidar_coeffs=lidar_coeffs, ut_params=ut_params,
        external_distortion_coeffs=external_distortion_params, viewmats_rs=viewmats_rs, rolling_shutter=rolling_shutter,
        extra_signals=extra_signals,
        extra_signals_sh_degree=None if extra_signals_sh_degree < 0 else extra_signals_sh_degree, _covars_triu=True)
    extra = meta.get("render_extra_signals")
    absgrad_holder = getattr(meta["means2d"], "absgrad", None) if absgrad else None
    ids = [meta["batch_ids"], meta["camera_ids"], meta["gaussian_ids"]]
    normals = meta.get("normals")
    return (rc, ra, _empty(rc) if extra is None else extra, _empty(rc) if normals is None else normals,
            _empty(rc) if absgrad_holder is None else absgrad_holder,
            *[_empty(rc, torch.long) if t is None else t for t in ids],
            meta["radii"], meta["means2d"], meta["depths"], meta["conics"], meta["opacities"], meta["tiles_per_gauss"],
            meta["isect_ids"], meta["flatten_ids"], meta["isect_offsets"], meta["tile_width"], meta["tile_height"])


@_op("rasterization_2dgs")
def rasterization_2dgs(means, quats, scales, opacities, colors, viewmats, Ks, image_width, image_height, tile_size,
                       eps2d, near_plane, far_plane, radius_clip, backgrounds, packed, sparse_grad, absgrad, distloss,
                       sh_degree, render_mode, depth_mode):
    """gsplat::rasterization_2dgs (host ``Rendering.cpp:1705-1960``; caller ``gsplat/rendering.py:1509-1532``)."""
    from .rendering import rasterization_2dgs as run

    rc, ra, normals, surf_normals, distort, median, meta = run(
        means, quats, scales, opacities, colors, viewmats, Ks, image_width, image_height, near_plane=near_plane,
        far_plane=far_plane, radius_clip=radius_clip, eps2d=eps2d, sh_degree=sh_degree, packed=packed,
        tile_size=tile_size, backgrounds=backgrounds, render_mode=render_mode, sparse_grad=sparse_grad, absgrad=absgrad,
        distloss=distloss, depth_mode=depth_mode)
    absgrad_holder = getattr(meta["means2d"], "absgrad", None) if absgrad else None
    return (rc, ra, normals, surf_normals, distort, median, _empty(rc) if absgrad_holder is None else absgrad_holder,
            meta["camera_ids"], meta["gaussian_ids"], meta["radii"], meta["means2d"], meta["depths"],
            meta["ray_transforms"], meta["opacities"], meta["normals"], meta["tiles_per_gauss"], meta["isect_ids"],
            meta["flatten_ids"], meta["isect_offsets"], meta["gradient_2dgs"], meta["tile_width"], meta["tile_height"],
            meta["n_cameras"])


# ----------------------------------------------------------------------------------------------
# registration
# ----------------------------------------------------------------------------------------------
def _os_environ_get(k, d):
    import os

    return os.environ.get(k, d)


class CheckError(RuntimeError, ValueError):
    """A failed argument check of an op body, as the DISPATCHER reports it. The reference's bodies use TORCH_CHECK, which
    Python sees as RuntimeError (its tests assert `pytest.raises(RuntimeError, match=...)`); this package's own Python API
    documents ValueError / TypeError. An op called through `torch.ops.gsplat.*` raises a class that is both."""


class CheckTypeError(RuntimeError, TypeError):
    pass


def _torch_check(fn):
    """The dispatcher-side face of a Python op body: ValueError / TypeError become RuntimeError subclasses (see CheckError)."""
    import functools

    @functools.wraps(fn)
    def body(*args, **kwargs):
        try:
            return fn(*args, **kwargs)
        except RuntimeError:
            raise
        except ValueError as e:
            raise CheckError(str(e)) from None
        except TypeError as e:
            raise CheckTypeError(str(e)) from None

    return body


def _register():
    for name in list(_impls):
        _impls[name] = _torch_check(_impls[name])
    for name, schema in SCHEMAS.items():
        qual = f"{NS}::{name}"
        try:
            torch._C._dispatch_find_schema_or_throw(qual, "")
            exists = True
        except RuntimeError:
            exists = False
        if not exists:
            _lib_def.define(name + schema)
        fn = _impls[name]
        if name not in COMPILED_OPS:
            _lib_impl.impl(name, fn)
        elif _os_environ_get("GSPLAT_AMD_COMPILED_OPS", "1") in ("0", ""):
            _lib_impl.impl(name, fn, allow_override=True)
    if COMPOSITE_UNAVAILABLE is None:
        for name, schema in COMPOSITE_SCHEMAS.items():
            try:
                torch._C._dispatch_find_schema_or_throw(f"{NS}::{name}", "")
            except RuntimeError:
                _lib_def.define(name + schema)
            _lib_impl.impl(name, _impls[name])
            _lib_impl_autograd.impl(name, _impls[name])
        for name, schema in CLASS_SCHEMAS.items():
            try:
                torch._C._dispatch_find_schema_or_throw(f"{NS}::{name}", "")
            except RuntimeError:
                _lib_def.define(name + schema)
            _lib_impl.impl(name, _impls[name])
            if name in _SELF_DIFFERENTIABLE:  # the body builds its own autograd graph (torch.autograd.Function inside)
                _lib_impl_autograd.impl(name, _impls[name])


_register()


def op(name: str):
    """torch.ops.gsplat.<name> (dispatcher entry)."""
    return getattr(torch.ops.gsplat, name)


def impl(name: str):
    """The Python implementation behind the op (lets internal callers pass private keyword options)."""
    return _impls[name]
