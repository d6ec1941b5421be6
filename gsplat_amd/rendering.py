"""gsplat.rasterization() for the classic 3DGS path, orchestrated over the gfx950 stage ops.

Mirrors the public signature, argument meaning, output shapes, ``meta`` dict and error behaviour
of the reference (``gsplat/rendering.py:234-690``) and the stage order of its C++ orchestrator
(``gsplat/cuda/csrc/Rendering.cpp:745-1481``):

    projection (dense | packed)  ->  per-view opacities (x compensation)  ->  SH colours (+0.5, clamp)
    -> [distributed: all-to-all of projected Gaussians]  ->  append depth channel
    -> tile intersection (exact ellipse test) + sort + offsets  ->  alpha compositing
    -> expected-depth normalisation.

3DGUT: ``with_ut`` (Unscented-Transform projection through pinhole / distorted-pinhole / ortho / fisheye / f-theta cameras,
global or rolling shutter) and ``with_eval3d`` (from-world compositing: rays given or generated for each of those camera
models, hit-distance modes, normals) are built; what is NOT built is refused up front, before any kernel launches, never
approximated. External (windshield) distortion (the reference's bivariate model) is built for every one of those camera models in
both 3DGUT kernels; spinning-lidar cameras (``camera_model="lidar"``: angle-space projection and tiling, element rays, tiles of
elements composited as virtual pixel tiles) with ``with_ut=True, with_eval3d=True`` (and, like the reference's orchestrator,
the lidar's tile lists through the classic kernels with ``with_eval3d=False``).
"""
from __future__ import annotations

import math
import os
from typing import Dict, Optional, Tuple

import torch
from torch import Tensor

from ._wrapper import (
    fully_fused_projection,
    isect_offset_encode,
    isect_tiles,
    isect_tiles_begin,
    isect_tiles_finish,
    rasterize_to_pixels,
    spherical_harmonics,
)

from ._ops import isect_max_tile_len as _isect_max_tile_len

# Array-of-structures rows for the compositing kernels (see `splat` below): built, bit-identical, and NOT the default - in the
# kernel harness the forward drops from 0.203 to 0.181 ms, but inside the step the four arrays are still in cache when the
# compositing starts (forward 0.175 -> 0.171 ms, backward 0.347 -> 0.342) and writing the rows costs the SH forward 10 us
# (0.044 -> 0.054): net zero at c3, a loss with several cameras (profiles/r10_ab.md). GSPLAT_AMD_SPLAT_ROWS=1 switches it on.
_SPLAT_ROWS = os.environ.get("GSPLAT_AMD_SPLAT_ROWS", "0") not in ("0", "")
_COLOR_MODES = ("RGB", "RGB+D", "RGB+ED")
_DEPTH_MODES = ("D", "ED", "RGB+D", "RGB+ED")
_HIT_MODES = ("d", "Ed", "RGB-d", "RGB-Ed")


# GSPLAT_AMD_VIEW_OPACITIES=0 (A/B switch): the per-view / per-row opacities are built with torch again (a broadcast view, an
# index_select) and autograd reduces their gradient with its own kernels
_VIEW_OPACITIES = os.environ.get("GSPLAT_AMD_VIEW_OPACITIES", "1") != "0"


def _resolve_tile_size(tile_size: Optional[int], with_eval3d: bool = False, width: int = 0, height: int = 0) -> int:
    """None -> the path's default: 16 for the classic path; for the from-world path 16 at 1080p and above, else 8
    (reference gsplat/rendering.py:201-231)."""
    if tile_size is not None:
        return int(tile_size)
    if with_eval3d:
        return 16 if min(width, height) >= 1080 else 8
    return 16


def rasterization(
    means: Tensor,  # [..., N, 3]
    quats: Optional[Tensor],  # [..., N, 4]
    scales: Optional[Tensor],  # [..., N, 3]
    opacities: Tensor,  # [..., N]
    colors: Optional[Tensor],  # [..., (C,) N, D] or [N, K, D]; or (sh0 [N, 1, D], shN [N, K - 1, D]) with sh_degree
    viewmats: Tensor,  # [..., C, 4, 4]
    Ks: Tensor,  # [..., C, 3, 3]
    width: int,
    height: int,
    near_plane: float = 0.01,
    far_plane: float = 1e10,
    radius_clip: float = 0.0,
    eps2d: float = 0.3,
    sh_degree: Optional[int] = None,
    packed: bool = True,
    tile_size: Optional[int] = None,
    backgrounds: Optional[Tensor] = None,
    render_mode: str = "RGB",
    sparse_grad: bool = False,
    absgrad: bool = False,
    rasterize_mode: str = "classic",
    channel_chunk: int = 32,
    distributed: bool = False,
    camera_model: str = "pinhole",
    segmented: bool = False,
    covars: Optional[Tensor] = None,
    with_ut: bool = False,
    with_eval3d: bool = False,
    return_normals: bool = False,
    global_z_order: bool = True,
    rays: Optional[Tensor] = None,
    radial_coeffs: Optional[Tensor] = None,
    tangential_coeffs: Optional[Tensor] = None,
    thin_prism_coeffs: Optional[Tensor] = None,
    ftheta_coeffs=None,
    lidar_coeffs=None,
    external_distortion_coeffs=None,
    rolling_shutter=None,
    viewmats_rs: Optional[Tensor] = None,
    ut_params=None,
    extra_signals: Optional[Tensor] = None,
    extra_signals_sh_degree: Optional[int] = None,
    renderer_config=None,
    _covars_triu: bool = False,
) -> Tuple[Tensor, Tensor, Dict]:
    """Rasterize a set of 3D Gaussians (N) to a batch of image planes (C). See the reference
    docstring (``gsplat/rendering.py:292-525``) for the meaning of every argument; this
    implementation covers the classic (EWA, non-3DGUT) path."""
    if render_mode not in _COLOR_MODES + ("D", "ED") + _HIT_MODES:
        raise ValueError(f"Unsupported render_mode: {render_mode}")
    if rasterize_mode not in ("classic", "antialiased"):
        raise ValueError(f"Unsupported rasterize_mode: {rasterize_mode}")
    has_color = render_mode in _COLOR_MODES or render_mode.startswith("RGB")
    use_hit_distance = render_mode in _HIT_MODES  # depth channel = distance along the ray (from-world rasterizer only)
    has_depth = render_mode in _DEPTH_MODES or use_hit_distance
    expected_depth = render_mode in ("ED", "RGB+ED", "Ed", "RGB-Ed")
    tile_size = _resolve_tile_size(tile_size, with_eval3d, width, height)

    batch_dims = tuple(means.shape[:-2])
    nb = len(batch_dims)
    B = math.prod(batch_dims)
    N = means.shape[-2] if means.dim() >= 2 else 0
    C = viewmats.shape[-3] if viewmats.dim() >= 3 else 0
    I = B * C
    device = means.device
    _validate_rasterization_inputs(
        means, covars, quats, scales, opacities, colors, viewmats, Ks, render_mode=render_mode,
        rasterize_mode=rasterize_mode, sh_degree=sh_degree, packed=packed, sparse_grad=sparse_grad, absgrad=absgrad,
        distributed=distributed, camera_model=camera_model, with_ut=with_ut, with_eval3d=with_eval3d,
        return_normals=return_normals, global_z_order=global_z_order, rays=rays, radial_coeffs=radial_coeffs,
        tangential_coeffs=tangential_coeffs, thin_prism_coeffs=thin_prism_coeffs, lidar_coeffs=lidar_coeffs,
        external_distortion_coeffs=external_distortion_coeffs, rolling_shutter=rolling_shutter, viewmats_rs=viewmats_rs,
        extra_signals=extra_signals, extra_signals_sh_degree=extra_signals_sh_degree, backgrounds=backgrounds,
        channel_chunk=channel_chunk, covars_triu=_covars_triu)
    # what validates but belongs to paths this backend does not build (a reference build with BUILD_3DGUT=0)
    unsupported = {
        "camera_model='ftheta' / ftheta_coeffs without the UT projection (with_ut=True)":
            (camera_model == "ftheta" or ftheta_coeffs is not None) and not with_ut,
    }
    bad = [k for k, v in unsupported.items() if v]
    if bad:
        raise RuntimeError(
            "gsplat_amd builds the classic 3DGS path and 3DGUT (the unscented projection and the from-world rasterizer for "
            "pinhole / distorted-pinhole / orthographic / fisheye / f-theta / spinning-lidar cameras); "
            f"these sub-features are not built - not supported, refused rather than approximated: {', '.join(bad)}"
        )
    if camera_model not in ("pinhole", "ortho", "fisheye", "ftheta", "lidar"):
        raise ValueError(f"camera_model '{camera_model}' is not supported (pinhole / ortho / fisheye / ftheta / lidar)")
    is_lidar = camera_model == "lidar"
    if is_lidar:
        if hasattr(lidar_coeffs, "to_cpp"):  # the reference's Python parameter object: its custom-class record
            lidar_coeffs = lidar_coeffs.to_cpp()
        # with_eval3d=False is accepted like in the reference (Rendering.cpp:1399-1425): the classic compositing kernels then
        # take the lidar's tile lists as lists of tile_size x tile_size PIXEL tiles of a [n_rows, n_columns] image - that is
        # what the reference's orchestrator does too; a lidar's elements are rendered by with_eval3d=True
    if (camera_model == "ftheta") != (ftheta_coeffs is not None):
        raise ValueError("ftheta_coeffs must be given if and only if camera_model is 'ftheta'")
    # `segmented` (gsplat/rendering.py:262; IntersectTile.cu:1125-1176) only selects how the reference sorts - per image instead
    # of one global radix sort. Both give the same (image, tile, depth) order with ties in emission order, which is what the
    # per-tile sort of this backend produces, so the flag changes nothing - except that, as in the reference, the intersection
    # refuses it together with packed rows (Intersect.cpp:207-211).

    if covars is not None and _covars_triu:
        # gsplat::rasterization_3dgs receives the upper-triangular 6-vectors (gsplat/rendering.py:540-544 converts)
        quats, scales = None, None
    elif covars is not None:
        quats, scales = None, None
        ti = ([0, 0, 0, 1, 1, 2], [0, 1, 2, 1, 2, 2])
        covars = covars[..., ti[0], ti[1]]

    world_size = 1
    dist_ctx = None
    if distributed:
        from . import distributed as gdist

        dist_ctx = gdist.DistributedRasterContext.create(
            batch_dims=batch_dims, sparse_grad=sparse_grad, absgrad=absgrad, camera_model=camera_model,
            colors=colors, sh_degree=sh_degree, n_cameras=C, device=device, n_local=N)
        world_size = dist_ctx.world_size
        # Seam A: every rank projects its Gaussian shard against ALL cameras.
        viewmats_proj, Ks_proj = dist_ctx.gather_cameras(viewmats, Ks)
        C_proj = viewmats_proj.shape[-3]
    else:
        viewmats_proj, Ks_proj, C_proj = viewmats, Ks, C

    if packed:
        from ._ops import clear_row_map_cache

        clear_row_map_cache()  # the previous step's row map (and the id tensors it holds) can go
    calc_comp = rasterize_mode == "antialiased"
    view_opacities = None
    rs_type = _ROLLING_SHUTTER_GLOBAL if rolling_shutter is None else int(rolling_shutter)
    # rolling shutter (with_ut only, validated above): the projection interpolates the pose per sigma point, the from-world
    # rasterizer per pixel row / column, and the view-dependent colours use the camera offset averaged over the two ends of
    # the frame (Rendering.cpp:1057, SphericalHarmonics.cuh:40-65) - a synthetic view matrix built with differentiable tensor
    # operations, so pose gradients reach both ends
    viewmats_sh = viewmats_proj
    if viewmats_rs is not None:
        from ._ops import _sh_rs_viewmats

        viewmats_sh = _sh_rs_viewmats(viewmats_proj, viewmats_rs)
    if with_ut:
        # Unscented-Transform projection through the (possibly distorted) camera model; no gradient reaches the geometry
        # on this path (the reference runs the op under no_grad as well, Rendering.cpp:890-925)
        from ._wrapper import fully_fused_projection_with_ut

        proj = fully_fused_projection_with_ut(
            means, quats, scales, opacities, viewmats_proj, Ks_proj, width, height, eps2d=eps2d, near_plane=near_plane,
            far_plane=far_plane, radius_clip=radius_clip, calc_compensations=calc_comp, camera_model=camera_model,
            ut_params=ut_params, radial_coeffs=radial_coeffs, tangential_coeffs=tangential_coeffs,
            thin_prism_coeffs=thin_prism_coeffs, ftheta_coeffs=ftheta_coeffs, global_z_order=global_z_order,
            rolling_shutter=rs_type, viewmats_rs=viewmats_rs, external_distortion_coeffs=external_distortion_coeffs,
            lidar_coeffs=lidar_coeffs)
    elif not packed and not calc_comp and means.is_cuda and _VIEW_OPACITIES:
        # dense rows, classic mode: the per-view opacities come out of the projection's own autograd node, whose backward sums
        # their gradient over the views inside the kernel that reads the gradient rows anyway (_autograd.py)
        from ._wrapper import fully_fused_projection_view_opacities

        proj = fully_fused_projection_view_opacities(
            means, covars, quats, scales, viewmats_proj, Ks_proj, width, height, opacities, eps2d=eps2d,
            near_plane=near_plane, far_plane=far_plane, radius_clip=radius_clip, camera_model=camera_model)
        proj, view_opacities = proj[:5], proj[5]
    elif packed and not sparse_grad and not calc_comp and means.is_cuda and _VIEW_OPACITIES:
        # packed rows: the rows' opacities come out of the projection's own autograd node as well (no index_add in the backward)
        from ._wrapper import fully_fused_projection_packed_row_opacities

        proj = fully_fused_projection_packed_row_opacities(
            means, covars, quats, scales, viewmats_proj, Ks_proj, width, height, opacities, eps2d=eps2d,
            near_plane=near_plane, far_plane=far_plane, radius_clip=radius_clip, camera_model=camera_model)
        proj, view_opacities = proj[:9], proj[9]
    else:
        proj = fully_fused_projection(
            means, covars, quats, scales, viewmats_proj, Ks_proj, width, height, eps2d=eps2d, near_plane=near_plane,
            far_plane=far_plane, radius_clip=radius_clip, packed=packed, sparse_grad=sparse_grad,
            calc_compensations=calc_comp, camera_model=camera_model, opacities=opacities)

    if packed:
        batch_ids, camera_ids, gaussian_ids, indptr, radii, means2d, depths, conics, compensations = proj
        # index_select (backward = atomic index_add) instead of advanced indexing (backward = index_put, which SORTS
        # the nnz indices first: ~0.2 ms per step at 1M Gaussians)
        proj_opacities = view_opacities if view_opacities is not None else \
            opacities.reshape(-1).index_select(0, batch_ids * N + gaussian_ids if B > 1 else gaussian_ids)
        image_ids = camera_ids if B == 1 else batch_ids * C_proj + camera_ids
    else:
        radii, means2d, depths, conics, compensations = proj
        batch_ids = camera_ids = gaussian_ids = image_ids = None
        proj_opacities = view_opacities if view_opacities is not None else \
            torch.broadcast_to(opacities[..., None, :], batch_dims + (C_proj, N))
    if compensations is not None:
        proj_opacities = proj_opacities * compensations

    # Distributed, dense rows: the geometry rows for seam B are assembled HERE, before the colours exist, so that in the
    # backward pass the wait for their returning gradients sits after the SH backward (distributed.py: _AsyncExchange)
    overlap_exchange = dist_ctx is not None and dist_ctx.overlaps(packed)
    geo_payload = dist_ctx.geometry_payload(radii, means2d, depths, conics, proj_opacities) if overlap_exchange else None

    # ---- tile intersection, first half (single-process path) ----------------------------------------
    # count + scan + asynchronous host read of the intersection total are enqueued BEFORE the SH kernels, which do
    # not depend on them; by the time isect_tiles_finish() needs the number on the host the GPU is still busy.
    tile_width = math.ceil(width / float(tile_size))
    tile_height = math.ceil(height / float(tile_size))
    if is_lidar:  # the tiles of the lidar's own tiling, in angle space
        tile_width, tile_height = int(lidar_coeffs.n_bins_azimuth), int(lidar_coeffs.n_bins_elevation)
    isect_pending = None
    if dist_ctx is None and not is_lidar:
        isect_pending = isect_tiles_begin(
            means2d, radii, depths, tile_size, tile_width, tile_height, sort=True, segmented=segmented, packed=packed,
            n_images=I, image_ids=image_ids, gaussian_ids=gaussian_ids, conics=None if with_ut else conics,
            opacities=None if with_ut else proj_opacities.contiguous())  # UT: plain radius boxes (Rendering.cpp:1307-1308)

    # ---- feature channels: [..., C, N, D] or [nnz, D] ------------------------------------------
    feats = None
    splat_rows = None
    if has_color:
        splat = None
        if (_SPLAT_ROWS and sh_degree is not None and not isinstance(colors, (tuple, list)) and colors.dim() == 3
                and colors.shape[-1] == 3 and colors.dtype == torch.float32 and not has_depth and extra_signals is None
                and dist_ctx is None and not with_ut and not with_eval3d and means.is_cuda and len(batch_dims) <= 1):
            # RGB from SH coefficients, nothing else in the rows: the SH forward also writes the compositing kernels' 48-byte
            # array-of-structures row of every visible Gaussian (x, y, conic, opacity, colours) - the compositing kernels then
            # stage a list entry with three 16-byte loads from one row instead of four gathers from four arrays
            # (c3 forward 0.203 -> 0.181 ms in the kernel harness, profiles/r10_ab.md)
            n_rows = means2d.numel() // 2
            splat_rows = torch.empty((n_rows, 12), device=device, dtype=means.dtype)
            splat = (means2d.detach(), conics.detach(), proj_opacities.detach().contiguous(), splat_rows)
        feats = _project_features(colors, sh_degree, True, means, viewmats_sh, radii, batch_dims, B, C_proj, N,
                                  batch_ids, camera_ids, gaussian_ids, splat)
    n_primary = feats.shape[-1] if feats is not None else 0
    n_extra = 0
    if extra_signals is not None:
        ex = _project_features(extra_signals, extra_signals_sh_degree, False, means, viewmats_sh, radii, batch_dims,
                               B, C_proj, N, batch_ids, camera_ids, gaussian_ids)
        n_extra = ex.shape[-1]
        feats = ex if feats is None else torch.cat([feats, ex], dim=-1)

    meta = {
        "batch_ids": batch_ids, "camera_ids": camera_ids, "gaussian_ids": gaussian_ids, "radii": radii,
        "means2d": means2d, "depths": depths, "conics": conics, "opacities": proj_opacities,
    }

    # ---- Seam B: ship every projected Gaussian to the rank that owns its camera ------------------
    recv_features = None
    if overlap_exchange:
        # two messages: geometry (waited for here), feature rows (in flight until compositing needs them)
        radii, means2d, depths, conics, proj_opacities, recv_features = dist_ctx.scatter_dense_begin(geo_payload, feats)
        image_ids = gaussian_ids_r = None
        n_rows_per_image = dist_ctx.total_gaussians
    elif dist_ctx is not None:
        (radii, means2d, depths, conics, proj_opacities, feats, image_ids, gaussian_ids_r) = dist_ctx.scatter_projection(
            packed, radii, means2d, depths, conics, proj_opacities, feats, batch_ids, camera_ids, gaussian_ids)
        n_rows_per_image = dist_ctx.total_gaussians
    else:
        n_rows_per_image = N

    # ---- tile intersection (second half) -----------------------------------------------------------
    if is_lidar:  # boxes in angle space against the lidar's tiling (gsplat::intersect_tile_lidar, Rendering.cpp:1309-1320)
        from . import _ops

        tiles_per_gauss, isect_ids, flatten_ids = _ops.intersect_tile_lidar(
            lidar_coeffs, means2d.contiguous(), radii.contiguous(), depths.contiguous(), None, None, None, True, segmented)
        isect_offsets = isect_offset_encode(isect_ids, I, tile_width, tile_height)
        longest_list = 0
    else:
        if isect_pending is None:
            isect_pending = isect_tiles_begin(
                means2d, radii, depths, tile_size, tile_width, tile_height, sort=True, segmented=segmented, packed=packed,
                n_images=I, image_ids=image_ids, gaussian_ids=gaussian_ids_r, conics=conics,
                opacities=proj_opacities.contiguous())
        tiles_per_gauss, isect_ids, flatten_ids = isect_tiles_finish(isect_pending)
        if isect_pending.offsets is not None:  # the fused intersection path produces the tile offsets as a by-product
            isect_offsets = isect_pending.offsets
        else:
            isect_offsets = isect_offset_encode(isect_ids, I, tile_width, tile_height)
        # the intersection also reports its longest tile list (same host words as n_isects): long lists are composited in
        # segments (csrc/raster3d_seg.hip); 0 when the path taken does not report it
        longest_list = _isect_max_tile_len(isect_pending)
    isect_offsets = isect_offsets.reshape(batch_dims + (C, tile_height, tile_width))

    # ---- feature rows still in flight (distributed, dense): needed from here on -----------------------
    if recv_features is not None:
        feats = recv_features()

    # ---- depth channel ---------------------------------------------------------------------------
    if has_depth:
        # hit-distance modes: the channel is a placeholder, the kernel puts every sample's own hit distance there
        # (Rendering.cpp:646-648)
        d = torch.zeros_like(depths[..., None]) if use_hit_distance else depths[..., None]
        feats = d if feats is None else torch.cat([feats, d], dim=-1)
        if backgrounds is not None:
            if has_color:
                backgrounds = torch.cat([backgrounds, torch.zeros_like(backgrounds[..., :1])], dim=-1)
            else:
                backgrounds = torch.zeros(batch_dims + (C, feats.shape[-1]), device=device, dtype=means.dtype)
    assert feats is not None

    # ---- compositing (channel chunks; alphas from the first chunk) -------------------------------
    D_total = feats.shape[-1]
    render_normals = None
    if D_total > channel_chunk and not with_eval3d:  # the from-world kernel chunks channels itself
        rc, ra = [], None
        for s in range(0, D_total, channel_chunk):
            e = min(D_total, s + channel_chunk)
            bg = None if backgrounds is None else backgrounds[..., s:e].contiguous()
            c_, a_ = rasterize_to_pixels(means2d, conics, feats[..., s:e].contiguous(), proj_opacities, width, height,
                                         tile_size, isect_offsets, flatten_ids, backgrounds=bg, packed=packed,
                                         absgrad=absgrad, _longest_tile_list=longest_list)
            rc.append(c_)
            if ra is None:
                ra = a_
        render_colors, render_alphas = torch.cat(rc, dim=-1), ra
    else:
        if with_eval3d:
            # from-world compositing: every sample is the Gaussian's response along the pixel's ray (forward only so far:
            # the op refuses inputs that require gradients)
            from ._wrapper import rasterize_to_pixels_eval3d_extra

            render_colors, render_alphas, _li, _sc, render_normals = rasterize_to_pixels_eval3d_extra(
                means, quats, scales, feats.contiguous(), proj_opacities.contiguous(), viewmats, Ks, width, height,
                tile_size, isect_offsets, flatten_ids, backgrounds=backgrounds, camera_model=camera_model,
                ut_params=ut_params, rays=rays, rolling_shutter=rs_type, viewmats_rs=viewmats_rs,
                radial_coeffs=radial_coeffs, tangential_coeffs=tangential_coeffs, thin_prism_coeffs=thin_prism_coeffs,
                ftheta_coeffs=ftheta_coeffs, external_distortion_coeffs=external_distortion_coeffs, lidar_coeffs=lidar_coeffs,
                use_hit_distance=use_hit_distance, return_normals=bool(return_normals), return_last_ids=False)
        else:
            render_colors, render_alphas = rasterize_to_pixels(
                means2d, conics, feats, proj_opacities, width, height, tile_size, isect_offsets, flatten_ids,
                backgrounds=backgrounds, packed=packed, absgrad=absgrad, _longest_tile_list=longest_list,
                _splat_rows=splat_rows if feats.shape[-1] == 3 else None)

    # ---- post-process: split extra signals, normalise expected depth ------------------------------
    render_extra = None
    if n_extra > 0:
        render_extra = render_colors[..., n_primary:n_primary + n_extra]
        parts = [render_colors[..., :n_primary]]
        if has_depth:
            dch = render_colors[..., -1:]
            if expected_depth:
                dch = dch / render_alphas.clamp_min(1e-10)
            parts.append(dch)
        render_colors = torch.cat(parts, dim=-1)
    elif expected_depth:
        render_colors = torch.cat(
            [render_colors[..., :-1], render_colors[..., -1:] / render_alphas.clamp_min(1e-10)], dim=-1)

    if not packed:
        meta["batch_ids"] = meta["camera_ids"] = meta["gaussian_ids"] = None
    meta.update({
        "tile_width": tile_width, "tile_height": tile_height, "tiles_per_gauss": tiles_per_gauss,
        "isect_ids": isect_ids, "flatten_ids": flatten_ids, "isect_offsets": isect_offsets, "width": width,
        "height": height, "tile_size": tile_size, "n_batches": B, "n_cameras": C,
    })
    if extra_signals is not None:
        meta["render_extra_signals"] = render_extra
    if return_normals:
        meta["normals"] = render_normals  # gsplat/rendering.py:687-688
    return render_colors, render_alphas, meta


def _check(cond: bool, *msg) -> None:
    """TORCH_CHECK: the reference's C++ validation surfaces as RuntimeError."""
    if not cond:
        raise RuntimeError("".join(str(m) for m in msg))


_ROLLING_SHUTTER_GLOBAL = 4  # gsplat/cuda/_wrapper.py RollingShutterType.GLOBAL


def _validate_rasterization_inputs(means, covars, quats, scales, opacities, colors, viewmats, Ks, *, render_mode,
                                   rasterize_mode, sh_degree, packed, sparse_grad, absgrad, distributed, camera_model,
                                   with_ut, with_eval3d, return_normals, global_z_order, rays, radial_coeffs,
                                   tangential_coeffs, thin_prism_coeffs, lidar_coeffs, external_distortion_coeffs,
                                   rolling_shutter, viewmats_rs, extra_signals, extra_signals_sh_degree, backgrounds,
                                   channel_chunk, covars_triu) -> None:
    """Host-side restatement of the reference's input validation (``Rendering.cpp:120-480``): same conditions, same
    order, same messages, RuntimeError like ``TORCH_CHECK`` (the reference's tests match on the messages,
    ``tests/test_rasterization.py:723-800``). Two deliberate differences: tile sizes 1..16 are accepted (the reference's
    classic kernels are compiled for 4 and 16 only), and the checks run before any kernel or collective."""
    has_color = render_mode.startswith("RGB")
    hit = render_mode in _HIT_MODES
    rs_global = rolling_shutter is None or int(rolling_shutter) == _ROLLING_SHUTTER_GLOBAL
    if distributed:  # named first, so that the error is about the distributed limitation (Rendering.cpp:187-233)
        _check(means.dim() == 2, "distributed=True does not support batch dimensions")
        _check(not sparse_grad, "distributed=True does not support sparse_grad=True")
        _check(not absgrad, "distributed=True does not support absgrad=True")
        _check(not with_ut, "distributed=True does not support with_ut=True")
        _check(not with_eval3d, "distributed=True does not support with_eval3d=True")
        _check(not return_normals, "distributed=True does not support return_normals=True")
        _check(rays is None, "distributed=True does not support rays")
        _check(camera_model == "pinhole", "distributed=True only supports camera_model='pinhole'")
        _check(global_z_order, "distributed=True does not support global_z_order=False")
        _check(rs_global and viewmats_rs is None, "distributed=True does not support rolling shutter")
        _check(radial_coeffs is None and tangential_coeffs is None and thin_prism_coeffs is None
               and external_distortion_coeffs is None, "distributed=True does not support camera distortion")
        _check(lidar_coeffs is None, "distributed=True does not support lidar coefficients")
        if has_color and sh_degree is None:
            _check(colors is not None and not isinstance(colors, (tuple, list)) and colors.dim() == 2,
                   "distributed=True only supports per-Gaussian colors")
        if extra_signals is not None and extra_signals_sh_degree is None:
            _check(extra_signals.dim() == 2, "distributed=True only supports per-Gaussian extra signals")
    if rasterize_mode != "classic" and (with_ut or with_eval3d):  # gsplat/rendering.py:170-175
        raise ValueError("3DGUT rendering only supports rasterize_mode='classic'. "
                         f"Got rasterize_mode='{rasterize_mode}' with with_ut={with_ut} and with_eval3d={with_eval3d}.")
    _check(global_z_order or with_ut, "global_z_order can be false only if with_ut=True")
    _check(with_ut or camera_model != "ftheta",
           "ftheta camera is only supported via UT, please set with_ut=True in the rasterization()")
    _check((camera_model == "lidar") == (lidar_coeffs is not None),
           "Lidar coefficients must be given if and only if camera model is lidar")
    _check(camera_model != "lidar" or with_ut, "Lidar camera model requires with_ut=True")
    _check(channel_chunk > 0, "channel_chunk must be > 0")
    _check(not hit or with_eval3d, "hit-distance render modes require with_eval3d=True")
    _check(not return_normals or with_eval3d, "return_normals=True requires with_eval3d=True")
    _check(not sparse_grad or packed, "sparse_grad is only supported when packed is True")
    _check(not sparse_grad or means.dim() == 2, "sparse_grad does not support batch dimensions")
    _check(camera_model != "ortho" or (radial_coeffs is None and tangential_coeffs is None and thin_prism_coeffs is None),
           "ortho camera model does not support radial_coeffs, tangential_coeffs, or thin_prism_coeffs parameters")

    _check(means.dim() >= 2 and means.shape[-1] == 3, "means must have shape [..., N, 3], got ", list(means.shape))
    batch, N = tuple(means.shape[:-2]), means.shape[-2]
    nb = len(batch)
    _check(opacities.dim() == nb + 1, "opacities must have shape [..., N], got ", list(opacities.shape))
    _check(viewmats.dim() == nb + 3, "viewmats must have shape [..., C, 4, 4], got ", list(viewmats.shape))
    _check(Ks.dim() == nb + 3, "Ks must have shape [..., C, 3, 3], got ", list(Ks.shape))
    C = viewmats.shape[nb]
    _check(tuple(opacities.shape) == batch + (N,), "opacities must have shape [..., N], got ", list(opacities.shape))
    _check(tuple(viewmats.shape) == batch + (C, 4, 4), "viewmats must have shape [..., C, 4, 4], got ",
           list(viewmats.shape))
    _check(tuple(Ks.shape) == batch + (C, 3, 3), "Ks must have shape [..., C, 3, 3], got ", list(Ks.shape))
    if with_eval3d:
        _check(not packed, "Packed mode is not supported with Eval3D")
        _check(not sparse_grad, "Sparse grad is not supported with Eval3D")
    if with_ut:
        _check(not packed, "Packed mode is not supported with UT")
        _check(not sparse_grad, "Sparse grad is not supported with UT")
    if covars is not None:
        _check(not with_eval3d and not with_ut, "UT and Eval3D rasterization require quats and scales, not covars")
        want = batch + ((N, 6) if covars_triu else (N, 3, 3))
        _check(tuple(covars.shape) == want, "covars must have shape [..., N, 3, 3] or [..., N, 6], got ",
               list(covars.shape))
    else:
        _check(quats is not None, "covars or quats is required")
        _check(scales is not None, "covars or scales is required")
        _check(tuple(quats.shape) == batch + (N, 4), "quats must have shape [..., N, 4], got ", list(quats.shape))
        _check(tuple(scales.shape) == batch + (N, 3), "scales must have shape [..., N, 3], got ", list(scales.shape))
    if rs_global:
        _check(viewmats_rs is None, "viewmats_rs should be None for global rolling shutter")
    else:
        _check(with_ut, "Rolling shutter requires with_ut=True")
        _check(viewmats_rs is not None, "Rolling shutter requires viewmats_rs")
    _check(rays is None or with_eval3d, "Rays input is only supported with Eval3D")
    _check(radial_coeffs is None or with_ut, "Radial distortion requires with_ut=True")
    _check(tangential_coeffs is None or with_ut, "Tangential distortion requires with_ut=True")
    _check(thin_prism_coeffs is None or with_ut, "Thin-prism distortion requires with_ut=True")
    _check(external_distortion_coeffs is None or with_ut, "External distortion requires with_ut=True")
    if has_color:
        _check(colors is not None, "colors must be provided for color render modes")
        if isinstance(colors, (tuple, list)):
            # (sh0 [N, 1, D], shN [N, K - 1, D]): the trainer's parameter layout, evaluated by the split SH kernels without
            # ever forming torch.cat([sh0, shN], 1) (an extension: the reference's rasterization() takes the concatenation)
            _check(sh_degree is not None and len(colors) == 2, "colors = (sh0, shN) needs sh_degree")
            _check(not distributed and nb == 0, "colors = (sh0, shN) is not supported with batch dimensions / distributed=True")
            sh0, shN = colors
            _check(sh0.dim() == 3 and tuple(sh0.shape[:2]) == (N, 1), "sh0 must have shape [N, 1, D], got ", list(sh0.shape))
            _check(shN.dim() == 3 and shN.shape[0] == N and shN.shape[-1] == sh0.shape[-1],
                   "shN must have shape [N, K - 1, D], got ", list(shN.shape))
            _check((sh_degree + 1) ** 2 - 1 <= shN.shape[-2], "sh_degree requires more color SH coefficients than provided")
        elif sh_degree is not None:
            _check(colors.dim() == 3 and colors.shape[0] == N, "SH colors must have shape [N, K, D], got ",
                   list(colors.shape))
            _check((sh_degree + 1) ** 2 <= colors.shape[-2], "sh_degree requires more color SH coefficients than provided")
        else:
            per_gaussian = colors.dim() == nb + 2 and colors.shape[nb] == N
            per_view = colors.dim() == nb + 3 and colors.shape[nb] == C and colors.shape[nb + 1] == N
            _check(per_gaussian or per_view, "colors must have shape [..., N, D] or [..., C, N, D], got ",
                   list(colors.shape))
    _check(has_color or sh_degree is None, "sh_degree must be None when colors is None")
    if backgrounds is not None:
        _check(tuple(backgrounds.shape[:-1]) == batch + (C,), "backgrounds must have shape [..., C, D], got ",
               list(backgrounds.shape))


def _project_features(features, sh_degree, clamp, means, viewmats, radii, batch_dims, B, C, N, batch_ids, camera_ids,
                      gaussian_ids, splat=None):
    """Per-view feature rows: [..., C, N, D] (dense) or [nnz, D] (packed).
    Reference: normalize_features_layout_3dgs / maybe_evaluate_feature_sh, Rendering.cpp:577-644."""
    nb = len(batch_dims)
    packed = gaussian_ids is not None
    if sh_degree is None:
        D = features.shape[-1]
        per_view = features.dim() == nb + 3
        # packed rows: index_select on the flattened row index (backward = atomic index_add) instead of advanced indexing
        # (backward = index_put, which SORTS the nnz indices first: 0.77 ms per step at 49 M Gaussians / 7 M visible rows)
        if per_view:
            if packed:
                return features.reshape(B * C * N, D).index_select(0, (batch_ids * C + camera_ids) * N + gaussian_ids)
            return features
        if packed:
            return features.reshape(B * N, D).index_select(0, gaussian_ids if B == 1 else batch_ids * N + gaussian_ids)
        return torch.broadcast_to(features[..., None, :, :], batch_dims + (C, N, D))
    if isinstance(features, (tuple, list)):
        # split coefficients (sh0, shN): band 0 is view-independent, bands >= 1 come from the band kernels reading shN in
        # place (csrc/sh_band.hip); the sum, the + 0.5 and the clamp are three small element-wise passes over [.., D]
        from ._wrapper import spherical_harmonics_l0, spherical_harmonics_l1_plus

        sh0, shN = features
        base = spherical_harmonics_l0(sh0)  # [N, D]
        base = base.index_select(0, gaussian_ids) if packed else base[None].expand(C, -1, -1)
        valid = None if packed else (radii > 0).all(dim=-1)
        vals = base
        if sh_degree > 0:
            if packed:
                # shN stays in its [N, K - 1, D] layout, read through gaussian_ids (the public op's packed contract is the
                # reference's: rows pre-gathered to [nnz, K - 1, D])
                vals = vals + _ShBandUngathered.apply(sh_degree, means, viewmats, shN, batch_ids, camera_ids, gaussian_ids)
            else:
                vals = vals + spherical_harmonics_l1_plus(sh_degree, means, viewmats, shN, masks=valid)
        vals = vals + 0.5
        if clamp:
            vals = vals.clamp_min(0.0)
        return vals if valid is None else vals * valid[..., None]
    if clamp:
        # primary colours: SH + the `clamp_min(colors + 0.5, 0)` post-op + the radii > 0 row mask in ONE kernel each
        # way (the reference runs them as separate torch ops: Rendering.cpp:1146-1160, rendering.py:714-718)
        return _ShColors.apply(sh_degree, means, viewmats, features, None if packed else radii, batch_ids, camera_ids,
                               gaussian_ids, splat)
    if packed:
        # Every packed row is visible by construction (projection only emits radii > 0), so no mask; the
        # coefficient rows are read THROUGH gaussian_ids inside the kernel instead of materialising the
        # reference's coeffs.index({gaussian_ids}) copy (Rendering.cpp:629; 4*K*D B/row each way + a sort in
        # the index_put backward).
        vals = _ShUngathered.apply(sh_degree, means, viewmats, features, batch_ids, camera_ids, gaussian_ids)
    else:
        valid = (radii > 0).all(dim=-1)
        vals = spherical_harmonics(sh_degree, means, viewmats, features, masks=valid)
    return vals + 0.5


class _ShColors(torch.autograd.Function):
    """colors = clamp_min(spherical_harmonics(...) + 0.5, 0) on the rows with radii > 0, fused (C-ABI gsx_sh_{fwd,bwd}
    with `post` / `post_colors` / `radii`). Dense: rows [..., C, N]; packed: rows [nnz] read through gaussian_ids.
    Same values and gradients as the unfused chain (the clamp VJP passes the gradient where the output is > 0)."""

    @staticmethod
    def forward(ctx, degree, means, viewmats, coeffs, radii, batch_ids, camera_ids, gaussian_ids, splat=None):
        from ._ops import impl

        means, viewmats, coeffs = means.contiguous(), viewmats.contiguous(), coeffs.contiguous()
        # splat = (means2d, conics, opacities, rows): the kernel also writes the compositing kernels' 48-byte row of every live
        # row into `rows` (plain data for the kernels that follow; no gradient flows through it)
        colors = impl("spherical_harmonics")(degree, means, viewmats, coeffs, None, batch_ids, camera_ids, gaussian_ids,
                                             None, _gathered=False, _radii=radii, _post=True, _splat=splat)
        ctx.degree = degree
        ctx.save_for_backward(means, viewmats, coeffs, radii, batch_ids, camera_ids, gaussian_ids, colors)
        return colors

    @staticmethod
    def backward(ctx, v_colors):
        from ._ops import impl

        means, viewmats, coeffs, radii, batch_ids, camera_ids, gaussian_ids, colors = ctx.saved_tensors
        v_coeffs, v_means, v_viewmats, _ = impl("spherical_harmonics_bwd")(
            ctx.degree, means, viewmats, coeffs, None, batch_ids, camera_ids, gaussian_ids, None, v_colors,
            ctx.needs_input_grad[1], ctx.needs_input_grad[2], False, _gathered=False, _radii=radii,
            _post_colors=colors)
        return None, v_means, v_viewmats, v_coeffs, None, None, None, None, None


class _ShBandUngathered(torch.autograd.Function):
    """spherical_harmonics_l1_plus on packed rows with shN left in its [N, K - 1, D] layout (csrc/sh_band.hip reads the rows
    through gaussian_ids). Same math and gradients as spherical_harmonics_l1_plus(shN[gaussian_ids])."""

    @staticmethod
    def forward(ctx, degree, means, viewmats, shN, batch_ids, camera_ids, gaussian_ids):
        from ._ops import impl

        means, viewmats, shN = means.contiguous(), viewmats.contiguous(), shN.contiguous()
        ctx.degree = degree
        ctx.save_for_backward(means, viewmats, shN, batch_ids, camera_ids, gaussian_ids)
        return impl("spherical_harmonics_l1_plus")(degree, means, viewmats, shN, None, batch_ids, camera_ids, gaussian_ids,
                                                   None, _gathered=False)

    @staticmethod
    def backward(ctx, v_colors):
        from ._ops import impl

        means, viewmats, shN, batch_ids, camera_ids, gaussian_ids = ctx.saved_tensors
        v_shN, v_means, v_viewmats, _ = impl("spherical_harmonics_l1_plus_bwd")(
            ctx.degree, means, viewmats, shN, None, batch_ids, camera_ids, gaussian_ids, None, v_colors.contiguous(),
            ctx.needs_input_grad[1], ctx.needs_input_grad[2], False, _gathered=False)
        return None, v_means, v_viewmats, v_shN, None, None, None


class _ShUngathered(torch.autograd.Function):
    """spherical_harmonics on packed rows with coeffs left in their [N, K, D] layout (C-ABI
    gsx_sh_{fwd,bwd} with coeffs_gathered=0). Same math and gradients as
    spherical_harmonics(coeffs[gaussian_ids]) (reference _wrapper.py:553-632)."""

    @staticmethod
    def forward(ctx, degree, means, viewmats, coeffs, batch_ids, camera_ids, gaussian_ids):
        from ._ops import impl

        means, viewmats, coeffs = means.contiguous(), viewmats.contiguous(), coeffs.contiguous()
        ctx.degree = degree
        ctx.save_for_backward(means, viewmats, coeffs, batch_ids, camera_ids, gaussian_ids)
        return impl("spherical_harmonics")(degree, means, viewmats, coeffs, None, batch_ids, camera_ids,
                                           gaussian_ids, None, _gathered=False)

    @staticmethod
    def backward(ctx, v_colors):
        from ._ops import impl

        means, viewmats, coeffs, batch_ids, camera_ids, gaussian_ids = ctx.saved_tensors
        v_coeffs, v_means, v_viewmats, _ = impl("spherical_harmonics_bwd")(
            ctx.degree, means, viewmats, coeffs, None, batch_ids, camera_ids, gaussian_ids, None,
            v_colors.contiguous(), ctx.needs_input_grad[1], ctx.needs_input_grad[2], False, _gathered=False)
        return None, v_means, v_viewmats, v_coeffs, None, None, None


# ==================================================================================================
# 2DGS
# ==================================================================================================
def _depth_to_points(depths: Tensor, camtoworlds: Tensor, Ks: Tensor) -> Tensor:
    """z-depth map -> world-space points (reference depth_to_points_2dgs, Rendering.cpp:1653-1681)."""
    H, W = depths.shape[-3], depths.shape[-2]
    x, y = torch.meshgrid(torch.arange(W, device=depths.device, dtype=depths.dtype),
                          torch.arange(H, device=depths.device, dtype=depths.dtype), indexing="xy")
    fx, fy = Ks[..., 0, 0][..., None, None], Ks[..., 1, 1][..., None, None]
    cx, cy = Ks[..., 0, 2][..., None, None], Ks[..., 1, 2][..., None, None]
    dirs = torch.stack([(x - cx + 0.5) / fx, (y - cy + 0.5) / fy, torch.ones_like((x - cx) / fx)], dim=-1)
    directions = torch.einsum("...ij,...hwj->...hwi", camtoworlds[..., :3, :3], dirs)
    origins = camtoworlds[..., :3, 3]
    return origins[..., None, None, :] + depths * directions


def _depth_to_normal(depths: Tensor, camtoworlds: Tensor, Ks: Tensor) -> Tensor:
    """Surface normals from a z-depth map (reference depth_to_normal_2dgs, Rendering.cpp:1686-1702)."""
    points = _depth_to_points(depths, camtoworlds, Ks)
    dx = points[..., 2:, 1:-1, :] - points[..., :-2, 1:-1, :]
    dy = points[..., 1:-1, 2:, :] - points[..., 1:-1, :-2, :]
    normals = torch.linalg.cross(dx, dy, dim=-1)
    normals = normals / normals.pow(2).sum(-1, keepdim=True).sqrt().clamp_min(1e-12)
    return torch.nn.functional.pad(normals, (0, 0, 1, 1, 1, 1), value=0.0)


class _SurfelPost(torch.autograd.Function):
    """The per-pixel tail of rasterization_2dgs as one launch per direction (csrc/surfel_post.hip): expected-depth
    normalisation, camera->world rotation of the rendered normals, normals from the depth map. Same maths as the tensor-op
    composition below it in rasterization_2dgs (the reference's: gsplat/rendering.py:1519-1552); no gradient to the cameras,
    so the caller uses it only when viewmats / Ks do not require one."""

    @staticmethod
    def forward(ctx, colors, alphas, normals, median, viewmats, Ks, expected_depth: bool, depth_source: int):
        from ._cabi import call, ptr

        I = viewmats.numel() // 16
        H, W, D = colors.shape[-3], colors.shape[-2], colors.shape[-1]
        colors, alphas, normals = colors.contiguous(), alphas.contiguous(), normals.contiguous()
        median = median.contiguous() if depth_source == 2 else None
        viewmats, Ks = viewmats.contiguous(), Ks.contiguous()
        colors_out = torch.empty_like(colors) if expected_depth else None
        normals_world = torch.empty_like(normals)
        surf = torch.empty_like(normals) if depth_source else None
        call("gsx_surfel_post_fwd", ptr(colors), ptr(alphas), ptr(normals), ptr(median), ptr(viewmats), ptr(Ks), I, W, H, D,
             int(expected_depth), depth_source, ptr(colors_out), ptr(normals_world), ptr(surf))
        ctx.save_for_backward(colors, alphas, normals, median, viewmats, Ks)
        ctx.cfg = (I, W, H, D, bool(expected_depth), depth_source)
        # an output the loss does not use gets None in backward, not a zero image: with no cotangent for the depth-map normals
        # the backward kernel skips the whole depth-gradient stage (its LDS tiles, a third of its time)
        ctx.set_materialize_grads(False)
        unused = [colors.new_empty(0) for _ in range(2)]  # placeholders for the outputs this mode does not produce
        if not expected_depth:
            colors_out = unused[0]
            ctx.mark_non_differentiable(colors_out)
        if not depth_source:
            surf = unused[1]
            ctx.mark_non_differentiable(surf)
        return colors_out, normals_world, surf

    @staticmethod
    def backward(ctx, v_colors_out, v_normals_world, v_surf):
        from ._cabi import call, ptr

        colors, alphas, normals, median, viewmats, Ks = ctx.saved_tensors
        I, W, H, D, expected_depth, depth_source = ctx.cfg
        v_colors_out = v_colors_out.contiguous() if expected_depth and v_colors_out is not None else None
        if expected_depth and v_colors_out is None:
            v_colors_out = torch.zeros_like(colors)
        v_normals_world = torch.zeros_like(normals) if v_normals_world is None else v_normals_world.contiguous()
        v_surf = v_surf.contiguous() if depth_source and v_surf is not None else None
        v_colors, v_normals = torch.empty_like(colors), torch.empty_like(normals)
        v_alphas = torch.empty_like(alphas) if expected_depth else None
        v_median = torch.empty_like(median) if depth_source == 2 else None
        call("gsx_surfel_post_bwd", ptr(colors), ptr(alphas), ptr(normals), ptr(median), ptr(viewmats), ptr(Ks), I, W, H, D,
             int(expected_depth), depth_source, ptr(v_colors_out), ptr(v_normals_world), ptr(v_surf), ptr(v_colors),
             ptr(v_alphas), ptr(v_normals), ptr(v_median))
        return v_colors, v_alphas, v_normals, v_median, None, None, None, None


def rasterization_2dgs(
    means: Tensor,  # [..., N, 3]
    quats: Tensor,  # [..., N, 4]
    scales: Tensor,  # [..., N, 3]
    opacities: Tensor,  # [..., N]
    colors: Tensor,  # [..., (C,) N, D] or [N, K, D]
    viewmats: Tensor,  # [..., C, 4, 4]
    Ks: Tensor,  # [..., C, 3, 3]
    width: int,
    height: int,
    near_plane: float = 0.01,
    far_plane: float = 1e10,
    radius_clip: float = 0.0,
    eps2d: float = 0.3,
    sh_degree: Optional[int] = None,
    packed: bool = False,
    tile_size: int = 16,
    backgrounds: Optional[Tensor] = None,
    render_mode: str = "RGB",
    sparse_grad: bool = False,
    absgrad: bool = False,
    distloss: bool = False,
    depth_mode: str = "expected",
):
    """Rasterize 2D Gaussians (surfels). Same signature, outputs and ``meta`` as the reference
    (``gsplat/rendering.py:1358-1568``; orchestrator ``Rendering.cpp:1705-1960``): returns (render_colors,
    render_alphas, render_normals [world space], surf_normals [from depth] or None, render_distort, render_median,
    meta). Always uses the AABB tile test and never chunks channels, like the reference."""
    from ._wrapper import fully_fused_projection_2dgs, rasterize_to_pixels_2dgs

    if render_mode not in _COLOR_MODES + ("D", "ED"):
        raise ValueError(f"Unsupported render_mode for rasterization_2dgs: {render_mode}")
    if depth_mode not in ("expected", "median"):
        raise ValueError(f"Unsupported depth_mode: {depth_mode}")
    has_color = render_mode in _COLOR_MODES
    append_depth = render_mode in _DEPTH_MODES
    expected_depth = render_mode in ("ED", "RGB+ED")
    # check_rasterization_2dgs_inputs (Rendering.cpp:1588-1636): same conditions and messages, RuntimeError like TORCH_CHECK
    _check(means.dim() >= 2 and means.shape[-1] == 3, "means must have shape [..., N, 3], got ", list(means.shape))
    batch_dims = tuple(means.shape[:-2])
    nb = len(batch_dims)
    B = math.prod(batch_dims)
    N = means.shape[-2]
    _check(quats.dim() >= 2 and tuple(quats.shape[-2:]) == (N, 4), "quats must have shape [..., N, 4], got ",
           list(quats.shape))
    _check(scales.dim() >= 2 and tuple(scales.shape[-2:]) == (N, 3), "scales must have shape [..., N, 3], got ",
           list(scales.shape))
    _check(opacities.dim() >= 1 and opacities.shape[-1] == N, "opacities must have shape [..., N], got ",
           list(opacities.shape))
    _check(viewmats.dim() >= 3 and tuple(viewmats.shape[-2:]) == (4, 4), "viewmats must have shape [..., C, 4, 4], got ",
           list(viewmats.shape))
    _check(Ks.dim() >= 3 and tuple(Ks.shape[-2:]) == (3, 3), "Ks must have shape [..., C, 3, 3], got ", list(Ks.shape))
    C = viewmats.shape[-3]
    I = B * C
    if sh_degree is not None:
        _check(colors.dim() == 3 and colors.shape[0] == N, "SH coefficients must have shape [N, K, D], got ",
               list(colors.shape))
        _check((sh_degree + 1) ** 2 <= colors.shape[-2], "SH degree ", sh_degree, " too high for ", colors.shape[-2],
               " coefficient bands")
    _check(not distloss or append_depth, "distloss requires a depth render mode")
    _check(not sparse_grad or packed, "sparse_grad is only supported when packed is True")
    _check(tuple(quats.shape) == batch_dims + (N, 4) and tuple(scales.shape) == batch_dims + (N, 3)
           and tuple(opacities.shape) == batch_dims + (N,) and tuple(viewmats.shape) == batch_dims + (C, 4, 4)
           and tuple(Ks.shape) == batch_dims + (C, 3, 3), "inputs must share the batch dimensions of means ",
           list(batch_dims))

    view_opacities = None
    if not packed and means.is_cuda and _VIEW_OPACITIES:
        # dense rows: the per-view opacities come out of the projection's own autograd node (see rasterization())
        from ._wrapper import fully_fused_projection_2dgs_view_opacities

        proj = fully_fused_projection_2dgs_view_opacities(means, quats, scales, viewmats, Ks, width, height, opacities,
                                                          eps2d=eps2d, near_plane=near_plane, far_plane=far_plane,
                                                          radius_clip=radius_clip)
        proj, view_opacities = proj[:5], proj[5]
    else:
        proj = fully_fused_projection_2dgs(means, quats, scales, viewmats, Ks, width, height, eps2d=eps2d,
                                           near_plane=near_plane, far_plane=far_plane, radius_clip=radius_clip,
                                           packed=packed, sparse_grad=sparse_grad)
    if packed:
        batch_ids, camera_ids, gaussian_ids, _indptr, radii, means2d, depths, ray_transforms, normals = proj
        proj_opacities = opacities.reshape(-1).index_select(0, batch_ids * N + gaussian_ids if B > 1 else gaussian_ids)
        image_ids = batch_ids * C + camera_ids
    else:
        radii, means2d, depths, ray_transforms, normals = proj
        batch_ids = camera_ids = gaussian_ids = image_ids = None
        proj_opacities = view_opacities if view_opacities is not None else \
            torch.broadcast_to(opacities[..., None, :], batch_dims + (C, N))
    densify = torch.zeros_like(means2d).requires_grad_(True)

    # tile intersection in two halves around the SH kernels, as in rasterization(): the host round trip for the number of
    # intersections is hidden behind work that does not depend on it
    tile_width = math.ceil(width / float(tile_size))
    tile_height = math.ceil(height / float(tile_size))
    isect_pending = isect_tiles_begin(
        means2d, radii, depths, tile_size, tile_width, tile_height, sort=True, segmented=False, packed=packed,
        n_images=I, image_ids=image_ids, gaussian_ids=gaussian_ids)

    feats = None
    if has_color:
        feats = _project_features(colors, sh_degree, True, means, viewmats, radii, batch_dims, B, C, N, batch_ids,
                                  camera_ids, gaussian_ids)
    tiles_per_gauss, isect_ids, flatten_ids = isect_tiles_finish(isect_pending)
    if isect_pending.offsets is not None:  # the fused intersection path produces the tile offsets as a by-product
        isect_offsets = isect_pending.offsets
    else:
        isect_offsets = isect_offset_encode(isect_ids, I, tile_width, tile_height)
    isect_offsets = isect_offsets.reshape(batch_dims + (C, tile_height, tile_width))
    raster_bg = backgrounds
    if append_depth:
        if has_color:
            feats = torch.cat([feats, depths[..., None]], dim=-1)
            if backgrounds is not None:
                raster_bg = torch.cat([backgrounds, torch.zeros_like(backgrounds[..., :1])], dim=-1)
        else:
            feats = depths[..., None]
    assert feats is not None, "rasterization_2dgs requires at least one color or depth channel"

    render_colors, render_alphas, render_normals, render_distort, render_median = rasterize_to_pixels_2dgs(
        means2d, ray_transforms, feats, proj_opacities, normals, densify, width, height, tile_size, isect_offsets,
        flatten_ids, backgrounds=raster_bg, packed=packed, absgrad=absgrad, distloss=distloss)

    want_surf = append_depth and has_color
    surf_normals = None
    if not (viewmats.requires_grad or Ks.requires_grad) and render_colors.dtype == torch.float32:
        # one launch per direction for the whole per-pixel tail (csrc/surfel_post.hip)
        depth_source = (2 if depth_mode == "median" else 1) if want_surf else 0
        colors_out, render_normals, surf = _SurfelPost.apply(render_colors, render_alphas, render_normals, render_median,
                                                              viewmats, Ks, expected_depth, depth_source)
        if expected_depth:
            render_colors = colors_out
        if want_surf:
            surf_normals = surf.squeeze(0)  # as Rendering.cpp:1926
    else:  # cameras are being optimised: the same maths from tensor ops, so that autograd reaches viewmats / Ks
        if expected_depth:
            ed = render_colors[..., -1:] / render_alphas.clamp_min(1e-10)
            render_colors = torch.cat([render_colors[..., :-1], ed], dim=-1) if render_colors.shape[-1] > 1 else ed
        camtoworlds = torch.linalg.inv(viewmats)
        if want_surf:
            depth_for_normal = render_median if depth_mode == "median" else render_colors[..., -1:]
            surf_normals = _depth_to_normal(depth_for_normal, camtoworlds, Ks).squeeze(0)
        render_normals = torch.einsum("...ij,...hwj->...hwi", camtoworlds[..., :3, :3], render_normals)

    meta = {
        "camera_ids": camera_ids, "gaussian_ids": gaussian_ids, "radii": radii, "means2d": means2d, "depths": depths,
        "ray_transforms": ray_transforms, "opacities": proj_opacities, "normals": normals, "tile_width": tile_width,
        "tile_height": tile_height, "tiles_per_gauss": tiles_per_gauss, "isect_ids": isect_ids,
        "flatten_ids": flatten_ids, "isect_offsets": isect_offsets, "width": width, "height": height,
        "tile_size": tile_size, "n_cameras": C, "render_distort": render_distort, "gradient_2dgs": densify,
    }
    return render_colors, render_alphas, render_normals, surf_normals, render_distort, render_median, meta
