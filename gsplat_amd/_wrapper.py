"""Operator surface of the hot path — same names, arguments and error behaviour as the reference's
``gsplat/cuda/_wrapper.py`` for: ``quat_scale_to_covar_preci`` (:657-716), ``spherical_harmonics``
(:436-489), ``fully_fused_projection`` (:819-963), ``isect_tiles`` (:1196-1266),
``isect_offset_encode`` (:1328-1347), ``rasterize_to_pixels`` (:1497-1562).

Every function makes its inputs contiguous and performs one dispatcher call into
``torch.ops.gsplat.<op>`` (defined in ``_ops.py``, autograd in ``_autograd.py``).
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import torch
from torch import Tensor

from . import _autograd  # noqa: F401  (registers ops + autograd)

_ops = _autograd.fast_ops  # torch.ops.gsplat, with direct autograd Functions for the differentiable ops
from . import _ops as _impl  # noqa: E402  (python-side op bodies: begin/finish halves of intersect_tile)

CAMERA_MODELS = {"pinhole": 0, "ortho": 1, "fisheye": 2}


def _camera_model_id(camera_model: str) -> int:
    try:
        return CAMERA_MODELS[camera_model]
    except KeyError:
        raise ValueError(
            f"camera_model '{camera_model}' is not supported by the classic 3DGS path "
            "(pinhole / ortho / fisheye; ftheta and lidar belong to the 3DGUT path, out of scope)"
        ) from None


def world_to_cam(means: Tensor, covars: Tensor, viewmats: Tensor) -> Tuple[Tensor, Tensor]:
    """Gaussians from world to camera coordinates: means [..., N, 3], covars [..., N, 3, 3], viewmats [..., C, 4, 4] ->
    ([..., C, N, 3], [..., C, N, 3, 3]). Plain tensor algebra on whatever device the inputs live on — the reference
    removed its kernel for this too (``gsplat/cuda/_wrapper.py:381-416`` forwards to ``_torch_impl._world_to_cam``)."""
    batch_dims = means.shape[:-2]
    N, C = means.shape[-2], viewmats.shape[-3]
    assert means.shape == batch_dims + (N, 3), means.shape
    assert covars.shape == batch_dims + (N, 3, 3), covars.shape
    assert viewmats.shape == batch_dims + (C, 4, 4), viewmats.shape
    R, t = viewmats[..., :3, :3], viewmats[..., :3, 3]
    means_c = torch.einsum("...cij,...nj->...cni", R, means) + t[..., None, :]
    covars_c = torch.einsum("...cij,...njk,...clk->...cnil", R, covars, R)
    return means_c, covars_c


@torch.no_grad()
def fully_fused_projection_with_ut(
    means: Tensor,  # [..., N, 3]
    quats: Tensor,  # [..., N, 4]
    scales: Tensor,  # [..., N, 3]
    opacities: Optional[Tensor],  # [..., N]
    viewmats: Tensor,  # [..., C, 4, 4]
    Ks: Tensor,  # [..., C, 3, 3]
    width: int,
    height: int,
    eps2d: float = 0.3,
    near_plane: float = 0.01,
    far_plane: float = 1e10,
    radius_clip: float = 0.0,
    calc_compensations: bool = False,
    camera_model: str = "pinhole",
    ut_params=None,
    radial_coeffs: Optional[Tensor] = None,  # [..., C, 6] or [..., C, 4]
    tangential_coeffs: Optional[Tensor] = None,  # [..., C, 2]
    thin_prism_coeffs: Optional[Tensor] = None,  # [..., C, 4]
    ftheta_coeffs=None,
    lidar_coeffs=None,
    external_distortion_coeffs=None,
    rolling_shutter: int = 4,  # RollingShutterType.GLOBAL
    viewmats_rs: Optional[Tensor] = None,
    global_z_order: bool = True,
) -> Tuple[Tensor, Tensor, Tensor, Tensor, Optional[Tensor]]:
    """Projects Gaussians to 2D with the Unscented Transform — like ``fully_fused_projection`` but through a camera model
    with lens distortion; not differentiable (reference ``gsplat/cuda/_wrapper.py:2545-2633``). Returns
    (radii [..., C, N, 2], means2d, depths, conics, compensations or None). Built: pinhole (perfect or OpenCV-distorted)
    and ortho cameras with a global shutter."""
    models = {"pinhole": 0, "ortho": 1, "fisheye": 2, "ftheta": 3, "lidar": 4}
    if camera_model not in models:
        raise ValueError(f"unknown camera_model '{camera_model}'")
    c = lambda t: None if t is None else t.contiguous()  # noqa: E731
    return _ops.projection_ut_3dgs_fused(
        means.contiguous(), quats.contiguous(), scales.contiguous(), c(opacities), viewmats.contiguous(), c(viewmats_rs),
        Ks.contiguous(), width, height, eps2d, near_plane, far_plane, radius_clip, calc_compensations,
        models[camera_model], global_z_order, ut_params, int(rolling_shutter), c(radial_coeffs), c(tangential_coeffs),
        c(thin_prism_coeffs), ftheta_coeffs, lidar_coeffs, external_distortion_coeffs)


def rasterize_to_pixels_eval3d(
    means: Tensor,  # [..., N, 3]
    quats: Tensor,  # [..., N, 4]
    scales: Tensor,  # [..., N, 3]
    colors: Tensor,  # [..., C, N, channels]
    opacities: Tensor,  # [..., C, N]
    viewmats: Tensor,  # [..., C, 4, 4]
    Ks: Tensor,  # [..., C, 3, 3]
    image_width: int,
    image_height: int,
    tile_size: int,
    isect_offsets: Tensor,  # [..., C, tile_height, tile_width]
    flatten_ids: Tensor,  # [n_isects]
    backgrounds: Optional[Tensor] = None,
    masks: Optional[Tensor] = None,
    camera_model: str = "pinhole",
    ut_params=None,
    rays: Optional[Tensor] = None,  # [..., C, H, W, 6]
    radial_coeffs: Optional[Tensor] = None,
    tangential_coeffs: Optional[Tensor] = None,
    thin_prism_coeffs: Optional[Tensor] = None,
    ftheta_coeffs=None,
    lidar_coeffs=None,
    external_distortion_coeffs=None,
    rolling_shutter: int = 4,
    viewmats_rs: Optional[Tensor] = None,
    use_hit_distance: bool = False,
    return_normals: bool = False,
    renderer_config=None,
) -> Tuple[Tensor, Tensor]:
    """Like ``rasterize_to_pixels`` but every sample is the Gaussian's response along the pixel's ray in world space
    (reference ``gsplat/cuda/_wrapper.py:2263-2353``): perfect pinhole cameras or caller-provided ``rays``, global
    shutter. Differentiable w.r.t. means / quats / scales / colors / opacities (rays, cameras and backgrounds are constants,
    like the reference's from-world backward, RasterizeToPixelsFromWorld3DGSBwd.cu). Returns (render_colors, render_alphas)."""
    models = {"pinhole": 0, "ortho": 1, "fisheye": 2, "ftheta": 3, "lidar": 4}
    c = lambda t: None if t is None else t.contiguous()  # noqa: E731
    cls = torch.classes.gsplat
    out = _ops.rasterize_to_pixels_from_world_3dgs(
        means.contiguous(), quats.contiguous(), scales.contiguous(), colors.contiguous(), opacities.contiguous(),
        c(backgrounds), c(masks), image_width, image_height, tile_size, viewmats.contiguous(), c(viewmats_rs),
        Ks.contiguous(), models[camera_model], ut_params if ut_params is not None else cls.UnscentedTransformParameters(),
        int(rolling_shutter), c(rays), c(radial_coeffs), c(tangential_coeffs), c(thin_prism_coeffs),
        ftheta_coeffs if ftheta_coeffs is not None else cls.FThetaCameraDistortionParameters(), lidar_coeffs,
        external_distortion_coeffs, isect_offsets.contiguous(), flatten_ids.contiguous(), False, use_hit_distance,
        return_normals, 0 if renderer_config is None else int(renderer_config), False)
    return out[0], out[1]


def rasterize_to_pixels_eval3d_extra(means, quats, scales, colors, opacities, viewmats, Ks, image_width, image_height, tile_size,
                                     isect_offsets, flatten_ids, backgrounds=None, masks=None, camera_model="pinhole",
                                     ut_params=None, rays=None, radial_coeffs=None, tangential_coeffs=None, thin_prism_coeffs=None,
                                     ftheta_coeffs=None, lidar_coeffs=None, external_distortion_coeffs=None, rolling_shutter=4,
                                     viewmats_rs=None, return_sample_counts=False, use_hit_distance=False, return_normals=False,
                                     renderer_config=None, return_last_ids=True):
    """rasterize_to_pixels_eval3d + the optional outputs (reference ``gsplat/cuda/_wrapper.py:2356-2482``): returns
    (render_colors, render_alphas, last_ids or None, sample_counts or None, render_normals or None)."""
    models = {"pinhole": 0, "ortho": 1, "fisheye": 2, "ftheta": 3, "lidar": 4}
    c = lambda t: None if t is None else t.contiguous()  # noqa: E731
    cls = torch.classes.gsplat
    return _ops.rasterize_to_pixels_from_world_3dgs(
        means.contiguous(), quats.contiguous(), scales.contiguous(), colors.contiguous(), opacities.contiguous(),
        c(backgrounds), c(masks), image_width, image_height, tile_size, viewmats.contiguous(), c(viewmats_rs),
        Ks.contiguous(), models[camera_model], ut_params if ut_params is not None else cls.UnscentedTransformParameters(),
        int(rolling_shutter), c(rays), c(radial_coeffs), c(tangential_coeffs), c(thin_prism_coeffs),
        ftheta_coeffs if ftheta_coeffs is not None else cls.FThetaCameraDistortionParameters(), lidar_coeffs,
        external_distortion_coeffs, isect_offsets.contiguous(), flatten_ids.contiguous(), bool(return_sample_counts),
        bool(use_hit_distance), bool(return_normals), 0 if renderer_config is None else int(renderer_config), bool(return_last_ids))


def _has(feature: str) -> bool:
    from . import csrc_shim

    return bool(csrc_shim.build_config().get(feature, False))


def has_3dgs() -> bool:
    """Feature probes with the reference's names (``gsplat/cuda/_wrapper.py:261-286``), answered from build_config()."""
    return _has("3dgs")


def has_2dgs() -> bool:
    return _has("2dgs")


def has_3dgut() -> bool:
    return _has("3dgut")


def has_adam() -> bool:
    return _has("adam")


def has_reloc() -> bool:
    return _has("reloc")


def has_losses() -> bool:
    return _has("losses")


def has_camera_wrappers() -> bool:
    return _has("camera_wrappers")


def quat_scale_to_covar_preci(
    quats: Tensor,  # [..., 4]
    scales: Tensor,  # [..., 3]
    compute_covar: bool = True,
    compute_preci: bool = True,
    triu: bool = False,
) -> Tuple[Optional[Tensor], Optional[Tensor]]:
    """Converts quaternions and scales to covariance and precision matrices
    ([..., 3, 3], or [..., 6] upper-triangular when ``triu``)."""
    assert quats.shape[-1] == 4, quats.shape
    assert scales.shape[-1] == 3, scales.shape
    return _ops.quat_scale_to_covar_preci(quats.contiguous(), scales.contiguous(), compute_covar, compute_preci, triu)


def spherical_harmonics(
    degrees_to_use: int,
    means: Tensor,  # [..., N, 3]
    viewmats: Tensor,  # [..., C, 4, 4]
    coeffs: Tensor,  # [N, K, D] or [nnz, K, D]
    masks: Optional[Tensor] = None,  # [..., C, N] or [nnz]
    batch_ids: Optional[Tensor] = None,
    camera_ids: Optional[Tensor] = None,
    gaussian_ids: Optional[Tensor] = None,
    viewmats_rs: Optional[Tensor] = None,
) -> Tensor:
    """Evaluates SH colours for every (camera, gaussian) — [..., C, N, D] or [nnz, D]."""
    if masks is not None:
        masks = masks.contiguous()
    return _ops.spherical_harmonics(
        degrees_to_use, means.contiguous(), viewmats.contiguous(), coeffs.contiguous(), masks,
        None if batch_ids is None else batch_ids.contiguous(),
        None if camera_ids is None else camera_ids.contiguous(),
        None if gaussian_ids is None else gaussian_ids.contiguous(),
        None if viewmats_rs is None else viewmats_rs.contiguous(),
    )


def fully_fused_projection(
    means: Tensor,  # [..., N, 3]
    covars: Optional[Tensor],  # [..., N, 6] or None
    quats: Optional[Tensor],  # [..., N, 4] or None
    scales: Optional[Tensor],  # [..., N, 3] or None
    viewmats: Tensor,  # [..., C, 4, 4]
    Ks: Tensor,  # [..., C, 3, 3]
    width: int,
    height: int,
    eps2d: float = 0.3,
    near_plane: float = 0.01,
    far_plane: float = 1e10,
    radius_clip: float = 0.0,
    packed: bool = False,
    sparse_grad: bool = False,
    calc_compensations: bool = False,
    camera_model: str = "pinhole",
    opacities: Optional[Tensor] = None,  # [..., N] or None
):
    """Projects Gaussians to 2D. Dense: (radii, means2d, depths, conics, compensations) with shapes
    [..., C, N, *]. Packed: (batch_ids, camera_ids, gaussian_ids, indptr, radii, means2d, depths,
    conics, compensations) over the nnz visible (camera, gaussian) pairs."""
    means = means.contiguous()
    if covars is not None:
        covars = covars.contiguous()
    else:
        assert quats is not None and scales is not None, "covars or (quats, scales) required"
        quats, scales = quats.contiguous(), scales.contiguous()
    if sparse_grad:
        assert packed, "sparse_grad is only supported when packed is True"
    if opacities is not None:
        opacities = opacities.contiguous()
    viewmats, Ks = viewmats.contiguous(), Ks.contiguous()
    cm = _camera_model_id(camera_model)
    if packed:
        return _ops.projection_ewa_3dgs_packed(
            means, covars, quats, scales, opacities, viewmats, Ks, width, height, eps2d, near_plane, far_plane,
            radius_clip, sparse_grad, calc_compensations, cm)
    return _ops.projection_ewa_3dgs_fused(
        means, covars, quats, scales, opacities, viewmats, Ks, width, height, eps2d, near_plane, far_plane,
        radius_clip, calc_compensations, cm)


def fully_fused_projection_view_opacities(means, covars, quats, scales, viewmats, Ks, width, height, opacities, eps2d=0.3,
                                          near_plane=0.01, far_plane=1e10, radius_clip=0.0, calc_compensations=False,
                                          camera_model="pinhole"):
    """fully_fused_projection (dense rows) + the per-view opacities [..., C, N] as a sixth output whose gradient the projection
    backward reduces (``_autograd.ProjectionWithViewOpacities``). What gsplat_amd's rasterization() calls; not part of the
    reference's surface."""
    c = lambda t: None if t is None else t.contiguous()  # noqa: E731
    if covars is None:
        assert quats is not None and scales is not None, "covars or (quats, scales) required"
    return _autograd.ProjectionWithViewOpacities.apply(
        means.contiguous(), c(covars), None if covars is not None else c(quats), None if covars is not None else c(scales),
        opacities.contiguous(), viewmats.contiguous(), Ks.contiguous(), width, height, eps2d, near_plane, far_plane,
        radius_clip, calc_compensations, _camera_model_id(camera_model))


def fully_fused_projection_packed_row_opacities(means, covars, quats, scales, viewmats, Ks, width, height, opacities, eps2d=0.3,
                                                near_plane=0.01, far_plane=1e10, radius_clip=0.0, calc_compensations=False,
                                                camera_model="pinhole"):
    """fully_fused_projection(packed=True) + the packed rows' opacities [nnz] as a tenth output whose gradient the projection
    backward reduces (``_autograd.PackedProjectionWithViewOpacities``). What gsplat_amd's rasterization() calls; not part of
    the reference's surface."""
    c = lambda t: None if t is None else t.contiguous()  # noqa: E731
    if covars is None:
        assert quats is not None and scales is not None, "covars or (quats, scales) required"
    return _autograd.PackedProjectionWithViewOpacities.apply(
        means.contiguous(), c(covars), None if covars is not None else c(quats), None if covars is not None else c(scales),
        opacities.contiguous(), viewmats.contiguous(), Ks.contiguous(), width, height, eps2d, near_plane, far_plane,
        radius_clip, False, calc_compensations, _camera_model_id(camera_model))


@torch.no_grad()
def isect_tiles(
    means2d: Tensor,  # [..., N, 2] or [nnz, 2]
    radii: Tensor,  # [..., N, 2] or [nnz, 2]
    depths: Tensor,  # [..., N] or [nnz]
    tile_size: int,
    tile_width: int,
    tile_height: int,
    sort: bool = True,
    segmented: bool = False,
    packed: bool = False,
    n_images: Optional[int] = None,
    image_ids: Optional[Tensor] = None,
    gaussian_ids: Optional[Tensor] = None,
    conics: Optional[Tensor] = None,
    opacities: Optional[Tensor] = None,
) -> Tuple[Tensor, Tensor, Tensor]:
    """Maps projected Gaussians to the tiles they touch: (tiles_per_gauss int32, isect_ids int64 [M],
    flatten_ids int32 [M]). With ``conics`` and ``opacities`` the exact opacity-aware ellipse test is
    used, otherwise the axis-aligned box of ``radii``."""
    if packed:
        assert image_ids is not None and gaussian_ids is not None and n_images is not None
        image_ids, gaussian_ids = image_ids.contiguous(), gaussian_ids.contiguous()
    else:
        image_ids = gaussian_ids = None
    return _ops.intersect_tile(
        means2d.contiguous(), radii.contiguous(), depths.contiguous(),
        None if conics is None else conics.contiguous(), None if opacities is None else opacities.contiguous(),
        image_ids, gaussian_ids, n_images, tile_size, tile_width, tile_height, sort, segmented)


def isect_tiles_begin(means2d, radii, depths, tile_size, tile_width, tile_height, sort=True, segmented=False,
                      packed=False, n_images=None, image_ids=None, gaussian_ids=None, conics=None, opacities=None):
    """isect_tiles() split in two so that the host read of the intersection count overlaps other GPU work:
    ``pending = isect_tiles_begin(...)``, enqueue independent kernels, then ``isect_tiles_finish(pending)`` returns
    what isect_tiles() returns. (Extension of the reference surface; isect_tiles() itself is unchanged.)"""
    if packed:
        assert image_ids is not None and gaussian_ids is not None and n_images is not None
        image_ids, gaussian_ids = image_ids.contiguous(), gaussian_ids.contiguous()
    else:
        image_ids = gaussian_ids = None
    with torch.no_grad():
        return _impl.isect_begin(
            means2d.contiguous(), radii.contiguous(), depths.contiguous(),
            None if conics is None else conics.contiguous(), None if opacities is None else opacities.contiguous(),
            image_ids, gaussian_ids, n_images, tile_size, tile_width, tile_height, sort, segmented)


def isect_tiles_finish(pending) -> Tuple[Tensor, Tensor, Tensor]:
    with torch.no_grad():
        return _impl.isect_finish(pending)


@torch.no_grad()
def isect_offset_encode(isect_ids: Tensor, n_images: int, tile_width: int, tile_height: int) -> Tensor:
    """Sorted intersection ids -> per-(image, tile) start offsets, int32 [I, tile_height, tile_width]."""
    return _ops.intersect_offset(isect_ids.contiguous(), n_images, tile_width, tile_height)


def rasterize_to_pixels(
    means2d: Tensor,  # [..., N, 2] or [nnz, 2]
    conics: Tensor,  # [..., N, 3] or [nnz, 3]
    colors: Tensor,  # [..., N, channels] or [nnz, channels]
    opacities: Tensor,  # [..., N] or [nnz]
    image_width: int,
    image_height: int,
    tile_size: int,
    isect_offsets: Tensor,  # [..., tile_height, tile_width]
    flatten_ids: Tensor,  # [n_isects]
    backgrounds: Optional[Tensor] = None,  # [..., channels]
    masks: Optional[Tensor] = None,  # [..., tile_height, tile_width]
    packed: bool = False,
    absgrad: bool = False,
    _longest_tile_list: int = 0,
    _splat_rows: Optional[Tensor] = None,
) -> Tuple[Tensor, Tensor]:
    """Front-to-back alpha compositing of the depth-sorted per-tile lists. Returns
    (render_colors [..., H, W, channels], render_alphas [..., H, W, 1]). With ``absgrad`` the
    backward pass also fills ``means2d.absgrad``. ``_longest_tile_list`` (private; rendering.py passes what the
    intersection reported): above ``_ops.SEG_MIN_LONGEST`` long lists are cut into segments composited in parallel.
    0 = unknown: the ops look up what the intersection that produced ``flatten_ids`` noted (stage-level callers get the
    segments too); negative = one workgroup per tile whatever the lists look like. ``_splat_rows`` (private; three channels):
    the [R, 12] array-of-structures rows rasterization()'s SH forward wrote for these very tensors (``_ops.spherical_harmonics``
    ``_splat``) - the kernels then stage a list entry from one row instead of four arrays; results are bit-identical."""
    if backgrounds is not None:
        backgrounds = backgrounds.contiguous()
    if masks is not None:
        masks = masks.contiguous()
    means2d_c = means2d.contiguous()
    _impl.set_long_tile_hint(_longest_tile_list)
    if _splat_rows is not None:
        _impl.set_splat_rows_hint(_splat_rows, means2d_c)
    try:
        render_colors, render_alphas, means2d_absgrad, _last_ids = _ops.rasterize_to_pixels_3dgs(
            means2d_c, conics.contiguous(), colors.contiguous(), opacities.contiguous(), backgrounds, masks,
            image_width, image_height, tile_size, isect_offsets.contiguous(), flatten_ids.contiguous(), packed, absgrad)
    finally:
        _impl.set_long_tile_hint(0)  # also when the op raises: the hints belong to THIS call only
        if _splat_rows is not None:
            _impl.set_splat_rows_hint(None, None)
    if absgrad:
        means2d.absgrad = means2d_absgrad
    return render_colors, render_alphas


# ----------------------------------------------------------------------------------------------
# 2DGS (reference _wrapper.py:2633-2727, 2918-3001)
# ----------------------------------------------------------------------------------------------
def fully_fused_projection_2dgs(
    means: Tensor,  # [..., N, 3]
    quats: Tensor,  # [..., N, 4]
    scales: Tensor,  # [..., N, 3]
    viewmats: Tensor,  # [..., C, 4, 4]
    Ks: Tensor,  # [..., C, 3, 3]
    width: int,
    height: int,
    eps2d: float = 0.3,
    near_plane: float = 0.01,
    far_plane: float = 1e10,
    radius_clip: float = 0.0,
    packed: bool = False,
    sparse_grad: bool = False,
):
    """Ray-splat intersection matrices, screen-space boxes and normals of 2D Gaussians. Dense:
    (radii, means2d, depths, ray_transforms [..., C, N, 3, 3], normals [..., C, N, 3]); packed: (batch_ids,
    camera_ids, gaussian_ids, indptr, radii, means2d, depths, ray_transforms, normals) over nnz rows."""
    means, quats, scales = means.contiguous(), quats.contiguous(), scales.contiguous()
    if sparse_grad:
        assert packed, "sparse_grad is only supported when packed is True"
    viewmats, Ks = viewmats.contiguous(), Ks.contiguous()
    if packed:
        return _ops.projection_2dgs_packed(means, quats, scales, viewmats, Ks, width, height, near_plane, far_plane,
                                           radius_clip, sparse_grad)
    return _ops.projection_2dgs_fused(means, quats, scales, viewmats, Ks, width, height, eps2d, near_plane, far_plane,
                                      radius_clip)


def fully_fused_projection_2dgs_view_opacities(means, quats, scales, viewmats, Ks, width, height, opacities, eps2d=0.3,
                                               near_plane=0.01, far_plane=1e10, radius_clip=0.0):
    """fully_fused_projection_2dgs (dense rows) + the per-view opacities [..., C, N] as a sixth output whose gradient the
    projection backward reduces (``_autograd.Projection2DGSWithViewOpacities``). What gsplat_amd's rasterization_2dgs() calls;
    not part of the reference's surface."""
    return _autograd.Projection2DGSWithViewOpacities.apply(
        means.contiguous(), quats.contiguous(), scales.contiguous(), viewmats.contiguous(), Ks.contiguous(), width, height,
        eps2d, near_plane, far_plane, radius_clip, opacities.contiguous())


def rasterize_to_pixels_2dgs(
    means2d: Tensor,  # [..., N, 2] or [nnz, 2]
    ray_transforms: Tensor,  # [..., N, 3, 3] or [nnz, 3, 3]
    colors: Tensor,  # [..., N, channels]
    opacities: Tensor,  # [..., N]
    normals: Tensor,  # [..., N, 3]
    densify: Tensor,  # [..., N, 2]
    image_width: int,
    image_height: int,
    tile_size: int,
    isect_offsets: Tensor,
    flatten_ids: Tensor,
    backgrounds: Optional[Tensor] = None,
    masks: Optional[Tensor] = None,
    packed: bool = False,
    absgrad: bool = False,
    distloss: bool = False,
) -> Tuple[Tensor, Tensor, Tensor, Tensor, Tensor]:
    """Composites 2D Gaussians. Returns (render_colors, render_alphas, render_normals [..., H, W, 3],
    render_distort [..., H, W, 1], render_median [..., H, W, 1]). The last colour channel is the depth used by the
    distortion and median outputs. ``densify`` only collects a gradient (reference ``meta["gradient_2dgs"]``)."""
    if backgrounds is not None:
        backgrounds = backgrounds.contiguous()
    if masks is not None:
        masks = masks.contiguous()
    (render_colors, render_alphas, render_normals, render_distort, render_median, means2d_absgrad, _last_ids,
     _median_ids) = _ops.rasterize_to_pixels_2dgs(
        means2d.contiguous(), ray_transforms.contiguous(), colors.contiguous(), opacities.contiguous(),
        normals.contiguous(), densify.contiguous(), backgrounds, masks, image_width, image_height, tile_size,
        isect_offsets.contiguous(), flatten_ids.contiguous(), packed, absgrad, distloss)
    if absgrad:
        means2d.absgrad = means2d_absgrad
    return render_colors, render_alphas, render_normals, render_distort, render_median


# ----------------------------------------------------------------------------------------------
# proj, split SH, rasterize_to_indices (reference _wrapper.py:493-560, 750-780, 2170-2260, 3154-3210)
# ----------------------------------------------------------------------------------------------
def proj(means: Tensor, covars: Tensor, Ks: Tensor, width: int, height: int, camera_model: str = "pinhole"):
    """Camera-space Gaussians [..., C, N, 3] / [..., C, N, 3, 3] -> (means2d [..., C, N, 2], covars2d [..., C, N, 2, 2])."""
    return _ops.projection_ewa_simple(means.contiguous(), covars.contiguous(), Ks.contiguous(), width, height,
                                      _camera_model_id(camera_model))


def spherical_harmonics_l0(sh0: Tensor) -> Tensor:
    """l = 0 band only: [N, 1, D] -> [N, D]."""
    return _ops.spherical_harmonics_l0(sh0.contiguous())


def spherical_harmonics_l1_plus(degrees_to_use: int, means: Tensor, viewmats: Tensor, shN: Tensor,
                                masks: Optional[Tensor] = None, batch_ids: Optional[Tensor] = None,
                                camera_ids: Optional[Tensor] = None, gaussian_ids: Optional[Tensor] = None,
                                viewmats_rs: Optional[Tensor] = None) -> Tensor:
    """Bands l >= 1 only; shN [N, K-1, D] starts at the degree-one basis."""
    c = lambda t: None if t is None else t.contiguous()  # noqa: E731
    return _ops.spherical_harmonics_l1_plus(degrees_to_use, means.contiguous(), viewmats.contiguous(), shN.contiguous(),
                                            c(masks), c(batch_ids), c(camera_ids), c(gaussian_ids), c(viewmats_rs))


@torch.no_grad()
def rasterize_to_indices_in_range(range_start: int, range_end: int, transmittances: Tensor, means2d: Tensor,
                                  conics: Tensor, opacities: Tensor, image_width: int, image_height: int,
                                  tile_size: int, isect_offsets: Tensor, flatten_ids: Tensor):
    """(gaussian_ids, pixel_ids, image_ids) of the contributing pairs inside batches [range_start, range_end) of the
    tile lists, given the per-pixel transmittance at range_start."""
    return _ops.rasterize_to_indices_3dgs(range_start, range_end, transmittances.contiguous(), means2d.contiguous(),
                                          conics.contiguous(), opacities.contiguous(), image_width, image_height,
                                          tile_size, isect_offsets.contiguous(), flatten_ids.contiguous())


@torch.no_grad()
def rasterize_to_indices_in_range_2dgs(range_start: int, range_end: int, transmittances: Tensor, means2d: Tensor,
                                       ray_transforms: Tensor, opacities: Tensor, image_width: int, image_height: int,
                                       tile_size: int, isect_offsets: Tensor, flatten_ids: Tensor):
    return _ops.rasterize_to_indices_2dgs(range_start, range_end, transmittances.contiguous(), means2d.contiguous(),
                                          ray_transforms.contiguous(), opacities.contiguous(), image_width,
                                          image_height, tile_size, isect_offsets.contiguous(), flatten_ids.contiguous())


@torch.no_grad()
def rasterize_num_contributing_gaussians(means2d: Tensor, conics: Tensor, opacities: Tensor, tile_offsets: Tensor,
                                         flatten_ids: Tensor, image_width: int, image_height: int, tile_size: int):
    """(number of contributing Gaussians int32 [..., H, W], rendered alphas [..., H, W]) — reference
    ``_wrapper.py:1668-1707``."""
    return _ops.rasterize_num_contributing_gaussians(means2d.contiguous(), conics.contiguous(), opacities.contiguous(),
                                                     tile_offsets.contiguous(), flatten_ids.contiguous(), image_width,
                                                     image_height, tile_size)


@torch.no_grad()
def rasterize_contributing_gaussian_ids(means2d: Tensor, conics: Tensor, opacities: Tensor, tile_offsets: Tensor,
                                        flatten_ids: Tensor, image_width: int, image_height: int, tile_size: int,
                                        num_contributing_gaussians: Tensor):
    """All contributing Gaussian ids and radiance weights per pixel, front to back, padded with (-1, 0) to
    ``num_contributing_gaussians.max()`` — reference ``_wrapper.py:1776-1822``."""
    return _ops.rasterize_contributing_gaussian_ids(means2d.contiguous(), conics.contiguous(), opacities.contiguous(),
                                                    tile_offsets.contiguous(), flatten_ids.contiguous(), image_width,
                                                    image_height, tile_size, num_contributing_gaussians.contiguous())


@torch.no_grad()
def rasterize_top_contributing_gaussian_ids(means2d: Tensor, conics: Tensor, opacities: Tensor, tile_offsets: Tensor,
                                            flatten_ids: Tensor, image_width: int, image_height: int, tile_size: int,
                                            num_depth_samples: int):
    """The ``num_depth_samples`` strongest contributors (by alpha * T) per pixel, in front-to-back order —
    reference ``_wrapper.py:1895-1940``."""
    return _ops.rasterize_top_contributing_gaussian_ids(means2d.contiguous(), conics.contiguous(), opacities.contiguous(),
                                                        tile_offsets.contiguous(), flatten_ids.contiguous(), image_width,
                                                        image_height, tile_size, num_depth_samples)


# ---- sparse pixel sets (reference _wrapper.py:1351-1500, 1565-1665, 1709-2000) ----------------------------------
def build_sparse_tile_layout(pixels: Tensor, image_ids: Tensor, n_images: int, tile_size: int, tile_width: int,
                             tile_height: int):
    """Per-active-tile layout of a set of pixels ``(row, col)`` [P, 2] with image index ``image_ids`` [P] (no
    duplicates): ``(active_tiles int32 [AT], active_tile_mask bool [I, th, tw], tile_pixel_mask uint64 [AT, words],
    tile_pixel_cumsum int64 [AT] inclusive ([1] zero when P == 0), pixel_map int64 [P])`` — reference
    ``_wrapper.py:1433-1493``."""
    assert pixels.dim() == 2 and pixels.shape[1] == 2, pixels.shape
    assert image_ids.shape == (pixels.shape[0],), (image_ids.shape, pixels.shape[0])
    return _ops.build_sparse_tile_layout(pixels.contiguous(), image_ids.contiguous(), n_images, tile_size, tile_width,
                                         tile_height)


def isect_tiles_sparse(means2d: Tensor, radii: Tensor, depths: Tensor, tile_mask: Tensor, active_tiles: Tensor,
                       n_images: int, tile_size: int, tile_width: int, tile_height: int,
                       image_ids: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
    """Intersections restricted to the active tiles: ``(tile_offsets int32 [AT + 1] with the n_isects sentinel,
    flatten_ids int32 [n_isects])`` sorted by (image, tile, depth) — reference ``_wrapper.py:1351-1430``."""
    packed = means2d.dim() == 2
    if packed:
        nnz = means2d.size(0)
        assert means2d.shape == (nnz, 2) and radii.shape == (nnz, 2) and depths.shape == (nnz,), means2d.shape
        assert image_ids is not None, "image_ids is required when packed ([nnz, 2])"
        assert image_ids.shape == (nnz,), image_ids.shape
    else:
        I, N = means2d.shape[0], means2d.shape[1]
        assert means2d.shape == (I, N, 2) and radii.shape == (I, N, 2) and depths.shape == (I, N), means2d.shape
        assert I == n_images, (I, n_images)
    assert tile_mask.shape == (n_images, tile_height, tile_width), tile_mask.shape
    assert tile_mask.dtype == torch.bool, tile_mask.dtype
    assert active_tiles.dim() == 1 and active_tiles.dtype == torch.int32, (active_tiles.shape, active_tiles.dtype)
    return _ops.intersect_tile_sparse(means2d.contiguous(), radii.contiguous(), depths.contiguous(),
                                      image_ids.contiguous() if image_ids is not None else None, tile_mask.contiguous(),
                                      active_tiles.contiguous(), n_images, tile_size, tile_width, tile_height)


def rasterize_to_pixels_sparse(means2d: Tensor, conics: Tensor, colors: Tensor, opacities: Tensor, image_ids: Tensor,
                               active_tiles: Tensor, tile_offsets: Tensor, flatten_ids: Tensor, tile_pixel_mask: Tensor,
                               tile_pixel_cumsum: Tensor, pixel_map: Tensor, image_width: int, image_height: int,
                               tile_size: int, tile_width: int, tile_height: int, backgrounds: Optional[Tensor] = None,
                               masks: Optional[Tensor] = None, packed: bool = False,
                               absgrad: bool = False) -> Tuple[Tensor, Tensor]:
    """Composite only the requested pixels: ``(colors [P, channels], alphas [P, 1])`` in the order of the ``pixels``
    given to :func:`build_sparse_tile_layout` — reference ``_wrapper.py:1565-1665``. Differentiable in means2d, conics,
    colors, opacities and backgrounds."""
    render_colors, render_alphas, means2d_absgrad, _last_ids = _ops.rasterize_to_pixels_sparse(
        means2d.contiguous(), conics.contiguous(), colors.contiguous(), opacities.contiguous(),
        backgrounds.contiguous() if backgrounds is not None else None, masks.contiguous() if masks is not None else None,
        image_ids.contiguous(), image_width, image_height, tile_size, tile_width, tile_height, active_tiles.contiguous(),
        tile_offsets.contiguous(), flatten_ids.contiguous(), tile_pixel_mask.contiguous(), tile_pixel_cumsum.contiguous(),
        pixel_map.contiguous(), packed, absgrad)
    if absgrad:
        means2d.absgrad = means2d_absgrad
    return render_colors, render_alphas


@torch.no_grad()
def rasterize_num_contributing_gaussians_sparse(means2d: Tensor, conics: Tensor, opacities: Tensor, active_tiles: Tensor,
                                                tile_offsets: Tensor, flatten_ids: Tensor, tile_pixel_mask: Tensor,
                                                tile_pixel_cumsum: Tensor, pixel_map: Tensor, image_width: int,
                                                image_height: int, tile_size: int, tile_width: int, tile_height: int):
    """(count int32 [P], alpha [P]) for the requested pixels — reference ``_wrapper.py:1709-1773``."""
    return _ops.rasterize_num_contributing_gaussians_sparse(
        means2d.contiguous(), conics.contiguous(), opacities.contiguous(), image_width, image_height, tile_size,
        tile_width, tile_height, active_tiles.contiguous(), tile_offsets.contiguous(), flatten_ids.contiguous(),
        tile_pixel_mask.contiguous(), tile_pixel_cumsum.contiguous(), pixel_map.contiguous())


@torch.no_grad()
def rasterize_contributing_gaussian_ids_sparse(means2d: Tensor, conics: Tensor, opacities: Tensor, active_tiles: Tensor,
                                               tile_offsets: Tensor, flatten_ids: Tensor, tile_pixel_mask: Tensor,
                                               tile_pixel_cumsum: Tensor, pixel_map: Tensor,
                                               num_contributing_gaussians: Tensor, image_width: int, image_height: int,
                                               tile_size: int, tile_width: int, tile_height: int):
    """(ids int32 [P, K], weights [P, K]), K = max count, padded with (-1, 0) — reference ``_wrapper.py:1824-1893``."""
    return _ops.rasterize_contributing_gaussian_ids_sparse(
        means2d.contiguous(), conics.contiguous(), opacities.contiguous(), image_width, image_height, tile_size,
        tile_width, tile_height, active_tiles.contiguous(), tile_offsets.contiguous(), flatten_ids.contiguous(),
        tile_pixel_mask.contiguous(), tile_pixel_cumsum.contiguous(), pixel_map.contiguous(),
        num_contributing_gaussians.contiguous())


@torch.no_grad()
def rasterize_top_contributing_gaussian_ids_sparse(means2d: Tensor, conics: Tensor, opacities: Tensor,
                                                   active_tiles: Tensor, tile_offsets: Tensor, flatten_ids: Tensor,
                                                   tile_pixel_mask: Tensor, tile_pixel_cumsum: Tensor, pixel_map: Tensor,
                                                   image_width: int, image_height: int, tile_size: int, tile_width: int,
                                                   tile_height: int, num_depth_samples: int):
    """The strongest ``num_depth_samples`` contributors of the requested pixels, front to back — reference
    ``_wrapper.py:1942-2008``."""
    return _ops.rasterize_top_contributing_gaussian_ids_sparse(
        means2d.contiguous(), conics.contiguous(), opacities.contiguous(), image_width, image_height, tile_size,
        tile_width, tile_height, num_depth_samples, active_tiles.contiguous(), tile_offsets.contiguous(),
        flatten_ids.contiguous(), tile_pixel_mask.contiguous(), tile_pixel_cumsum.contiguous(), pixel_map.contiguous())
