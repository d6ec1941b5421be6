"""Autograd for the torch.ops.gsplat.* stage ops.

The reference attaches autograd to its C++ ops from Python with
``torch.library.register_autograd`` (``gsplat/cuda/_wrapper.py:74-111``, one ``Register*`` class
per op). This module does the same for the ops defined in ``_ops.py``, with the same saved-tensor
and ``needs_input_grad`` contract, so that the ``*_bwd`` ops see the argument order fixed by the
reference (e.g. raster ``_wrapper.py:2078-2098``, projection ``:1022-1045``).

When this package is used as a drop-in ``gsplat.csrc`` under the reference's own Python
(INTEGRATION.md), the reference's ``_wrapper.py`` performs the registration instead and this
module is not imported.
"""
from __future__ import annotations

import torch

from . import _ops  # noqa: F401  (defines the ops)

NS = _ops.NS
_registered = False


def _bwd(name):
    """The `<name>_bwd` op body. Called directly rather than through the dispatcher: same function, but undefined
    incoming gradients can be passed as None (= zeros) instead of being materialised as zero tensors first."""
    return _ops.impl(name + "_bwd")


def _cg(v):
    """An incoming gradient made contiguous; an undefined one (None = zeros) stays None for ops that take it as such."""
    return None if v is None else v.contiguous()


def _z(v, shape, like):
    """Materialise an undefined gradient (None) as zeros only where an op needs a real tensor."""
    return like.new_zeros(shape) if v is None else v


# ---- quat_scale_to_covar_preci (reference _wrapper.py:719-760) ----------------------------------
def _qs_setup(ctx, inputs, output):
    quats, scales, compute_covar, compute_preci, triu = inputs
    ctx.triu = triu
    ctx.save_for_backward(quats, scales)


def _qs_backward(ctx, v_covars, v_precis):
    quats, scales = ctx.saved_tensors
    if v_covars is not None and v_covars.is_sparse:
        v_covars = v_covars.to_dense()
    if v_precis is not None and v_precis.is_sparse:
        v_precis = v_precis.to_dense()
    v_quats, v_scales = _bwd("quat_scale_to_covar_preci")(
        quats, scales, ctx.triu,
        None if v_covars is None else v_covars.contiguous(),
        None if v_precis is None else v_precis.contiguous(),
    )
    return v_quats, v_scales, None, None, None


# ---- spherical_harmonics (reference _wrapper.py:553-632) ----------------------------------------
def _sh_setup(ctx, inputs, output):
    (degrees_to_use, means, viewmats, coeffs, masks, batch_ids, camera_ids, gaussian_ids, viewmats_rs) = inputs
    ctx.degrees_to_use = degrees_to_use
    ctx.save_for_backward(means, viewmats, coeffs, masks, batch_ids, camera_ids, gaussian_ids, viewmats_rs)


def _sh_backward(ctx, v_colors):
    means, viewmats, coeffs, masks, batch_ids, camera_ids, gaussian_ids, viewmats_rs = ctx.saved_tensors
    v_coeffs, v_means, v_viewmats, v_viewmats_rs = _bwd("spherical_harmonics")(
        ctx.degrees_to_use, means, viewmats, coeffs, masks, batch_ids, camera_ids, gaussian_ids, viewmats_rs,
        v_colors.contiguous(), ctx.needs_input_grad[1], ctx.needs_input_grad[2],
        len(ctx.needs_input_grad) > 8 and ctx.needs_input_grad[8],
    )
    return (None, v_means, v_viewmats, v_coeffs, None, None, None, None, v_viewmats_rs)


# ---- projection, dense (reference _wrapper.py:966-1062) -----------------------------------------
def _proj_setup(ctx, inputs, output):
    (means, covars, quats, scales, _opacities, viewmats, Ks, width, height, eps2d, _near, _far, _clip, _calc,
     camera_model) = inputs
    radii, _means2d, _depths, conics, compensations = output
    ctx.width, ctx.height, ctx.eps2d, ctx.camera_model = width, height, eps2d, camera_model
    ctx.set_materialize_grads(False)  # undefined v_depths / v_compensations stay None (no zero-fill kernels)
    ctx.m2_shape = _means2d.shape
    ctx.save_for_backward(means, covars, quats, scales, viewmats, Ks, radii, conics, compensations)


def _proj_backward(ctx, v_radii, v_means2d, v_depths, v_conics, v_compensations, v_view_opacities=None):
    means, covars, quats, scales, viewmats, Ks, radii, conics, compensations = ctx.saved_tensors
    # v_means2d / v_conics are NOT made contiguous here: when they are column views of the compositing backward's AoS
    # gradient rows the op reads them in place through a row stride (_ops._row_view)
    if v_compensations is not None:
        v_compensations = v_compensations.contiguous()
    res = _bwd("projection_ewa_3dgs_fused")(
        means, covars, quats, scales, viewmats, Ks, ctx.width, ctx.height, ctx.eps2d, ctx.camera_model, radii, conics,
        compensations, _z(v_means2d, ctx.m2_shape, conics), None if v_depths is None else v_depths.contiguous(),
        _z(v_conics, conics.shape, conics), v_compensations, ctx.needs_input_grad[5], _v_view_opacities=v_view_opacities,
    )
    v_means, v_covars, v_quats, v_scales, v_viewmats = res[:5]
    v_opacities = res[5] if v_view_opacities is not None else None  # ProjectionWithViewOpacities only
    if not ctx.needs_input_grad[0]:
        v_means = None
    if not ctx.needs_input_grad[1]:
        v_covars = None
    if not ctx.needs_input_grad[2]:
        v_quats = None
    if not ctx.needs_input_grad[3]:
        v_scales = None
    return (v_means, v_covars, v_quats, v_scales, v_opacities, v_viewmats) + (None,) * 9


class ProjectionWithViewOpacities(torch.autograd.Function):
    """projection_ewa_3dgs_fused (dense rows) that also hands out the per-view opacities [..., C, N] - the broadcast view
    rasterization() builds from `opacities` (reference gsplat/rendering.py:511-520) - as one of ITS outputs, so that their
    cotangent comes back to the projection backward: the kernel sums it over the views while it reads the gradient rows it is
    a column of. Left to autograd, the strided column is copied out by a kernel that reads every row again (7.7 us at c3) and,
    with several views, reduced by another. gsplat_amd's rasterization() only; the op keeps the reference's schema."""

    @staticmethod
    def forward(ctx, *args):
        with torch._C._AutoDispatchBelowAutograd():
            out = getattr(torch.ops, NS).projection_ewa_3dgs_fused.default(*args)
        _proj_setup(ctx, args, out)
        opacities, viewmats = args[4], args[5]
        per_view = torch.broadcast_to(opacities[..., None, :], opacities.shape[:-1] + (viewmats.shape[-3], opacities.shape[-1]))
        # several views: ONE materialised copy here - the intersection and the compositing op each made their own from the
        # stride-0 view (two copies of [C, N] per step: 45 us at 4 x 4 M rows); a single view stays a view (already contiguous)
        per_view = per_view.contiguous()
        ctx.mark_non_differentiable(out[0])
        return tuple(out) + (per_view,)

    @staticmethod
    def backward(ctx, v_radii, v_means2d, v_depths, v_conics, v_compensations, v_per_view):
        if not ctx.needs_input_grad[4]:
            v_per_view = None
        return _proj_backward(ctx, v_radii, v_means2d, v_depths, v_conics, v_compensations, v_per_view)


# ---- projection, packed (reference _wrapper.py:1065-1191) ---------------------------------------
def _projp_setup(ctx, inputs, output):
    (means, covars, quats, scales, _opacities, viewmats, Ks, width, height, eps2d, _near, _far, _clip, sparse_grad,
     _calc, camera_model) = inputs
    (batch_ids, camera_ids, gaussian_ids, _indptr, _radii, _means2d, _depths, conics, compensations) = output
    ctx.width, ctx.height, ctx.eps2d, ctx.camera_model, ctx.sparse_grad = width, height, eps2d, camera_model, sparse_grad
    ctx.set_materialize_grads(False)
    ctx.m2_shape = _means2d.shape
    ctx.save_for_backward(means, covars, quats, scales, viewmats, Ks, batch_ids, camera_ids, gaussian_ids, conics,
                          compensations)


def _projp_backward(ctx, v_batch_ids, v_camera_ids, v_gaussian_ids, v_indptr, v_radii, v_means2d, v_depths, v_conics,
                    v_compensations, v_view_opacities=None):
    (means, covars, quats, scales, viewmats, Ks, batch_ids, camera_ids, gaussian_ids, conics,
     compensations) = ctx.saved_tensors
    if v_compensations is not None:
        v_compensations = v_compensations.contiguous()
    res = _bwd("projection_ewa_3dgs_packed")(
        means, covars, quats, scales, viewmats, Ks, ctx.width, ctx.height, ctx.eps2d, ctx.camera_model,
        ctx.sparse_grad, batch_ids, camera_ids, gaussian_ids, conics, compensations,
        _z(v_means2d, ctx.m2_shape, conics), None if v_depths is None else v_depths.contiguous(),
        _z(v_conics, conics.shape, conics), v_compensations, ctx.needs_input_grad[5], _v_view_opacities=v_view_opacities,
    )
    v_means, v_covars, v_quats, v_scales, v_viewmats = res[:5]
    v_opacities = res[5] if v_view_opacities is not None else None  # PackedProjectionWithViewOpacities only
    if not ctx.needs_input_grad[0]:
        v_means = None
    if not ctx.needs_input_grad[1]:
        v_covars = None
    if not ctx.needs_input_grad[2]:
        v_quats = None
    if not ctx.needs_input_grad[3]:
        v_scales = None
    return (v_means, v_covars, v_quats, v_scales, v_opacities, v_viewmats) + (None,) * 10


class PackedProjectionWithViewOpacities(torch.autograd.Function):
    """projection_ewa_3dgs_packed that also hands out the packed rows' opacities [nnz] (`opacities[gaussian_ids]` in the
    reference's rasterization(), gsplat/rendering.py:507-510) as one of ITS outputs: their cotangent - a column of the
    compositing backward's gradient rows - comes back to the projection backward, whose Gaussian-major kernel sums it per
    Gaussian while it walks the row map (no index_add, no zero fill). gsplat_amd's rasterization() only."""

    @staticmethod
    def forward(ctx, *args):
        with torch._C._AutoDispatchBelowAutograd():
            out = getattr(torch.ops, NS).projection_ewa_3dgs_packed.default(*args)
        _projp_setup(ctx, args, out)
        opacities = args[4]
        batch_ids, gaussian_ids = out[0], out[2]
        N = opacities.shape[-1]
        flat_ids = gaussian_ids if opacities.dim() == 1 else batch_ids * N + gaussian_ids
        row_opacities = opacities.reshape(-1).index_select(0, flat_ids)
        ctx.mark_non_differentiable(*[t for t in out[:5] if t is not None])
        return tuple(out) + (row_opacities,)

    @staticmethod
    def backward(ctx, v_b, v_c, v_g, v_indptr, v_radii, v_means2d, v_depths, v_conics, v_compensations, v_row_opacities):
        if not ctx.needs_input_grad[4]:
            v_row_opacities = None
        return _projp_backward(ctx, v_b, v_c, v_g, v_indptr, v_radii, v_means2d, v_depths, v_conics, v_compensations,
                               v_row_opacities)


# ---- rasterize_to_pixels (reference _wrapper.py:2010-2117) --------------------------------------
def _rast_setup(ctx, inputs, output):
    (means2d, conics, colors, opacities, backgrounds, masks, image_width, image_height, tile_size, isect_offsets,
     flatten_ids, _packed, absgrad) = inputs
    _render_colors, render_alphas, means2d_absgrad, last_ids = output
    ctx.mark_non_differentiable(last_ids, means2d_absgrad)
    ctx.width, ctx.height, ctx.tile_size, ctx.absgrad = image_width, image_height, tile_size, absgrad
    ctx.set_materialize_grads(False)  # an unused render_alphas (or render_colors) gets no zero-filled gradient
    ctx.rc_shape = _render_colors.shape
    ctx.longest = _ops.long_tile_hint_of_call()  # the backward cuts the same long lists into segments
    ctx.splat_rows = _ops.splat_rows_hint_of_call()  # rasterization()'s array-of-structures rows of these Gaussians (or None)
    ctx.save_for_backward(means2d, conics, colors, opacities, backgrounds, masks, isect_offsets, flatten_ids,
                          render_alphas, last_ids, means2d_absgrad)


def _rast_backward(ctx, v_render_colors, v_render_alphas, v_means2d_absgrad, v_last_ids):
    (means2d, conics, colors, opacities, backgrounds, masks, isect_offsets, flatten_ids, render_alphas, last_ids,
     means2d_absgrad) = ctx.saved_tensors
    hint = _ops
    hint.set_long_tile_hint(ctx.longest)  # this (autograd) thread's hint; the op body consumes it
    rows = getattr(ctx, "splat_rows", None)
    if rows is not None:
        hint.set_splat_rows_hint(rows, means2d)
    try:
        v_means2d_abs, v_means2d, v_conics, v_colors, v_opacities, v_backgrounds = _bwd("rasterize_to_pixels_3dgs")(
            means2d, conics, colors, opacities, backgrounds, masks, isect_offsets, flatten_ids, render_alphas, last_ids,
            ctx.width, ctx.height, ctx.tile_size, ctx.absgrad, _z(v_render_colors, ctx.rc_shape, render_alphas),
            None if v_render_alphas is None else v_render_alphas.contiguous(), ctx.needs_input_grad[4],
        )  # v_render_colors as it comes: the body reads pixel-linear views in place (the gradient of sum() is one float)
    finally:
        hint.set_long_tile_hint(0)
        if rows is not None:
            hint.set_splat_rows_hint(None, None)
    if ctx.absgrad and v_means2d_abs is not None:
        means2d_absgrad.copy_(v_means2d_abs)
    return (v_means2d, v_conics, v_colors, v_opacities, v_backgrounds) + (None,) * 8


# ---- rasterize_to_pixels_sparse (reference _wrapper.py:2120-2260) --------------------------------
def _rsp_setup(ctx, inputs, output):
    (means2d, conics, colors, opacities, backgrounds, masks, image_ids, image_width, image_height, tile_size, tile_width,
     tile_height, active_tiles, tile_offsets, flatten_ids, tile_pixel_mask, tile_pixel_cumsum, pixel_map, _packed,
     absgrad) = inputs
    _render_colors, render_alphas, means2d_absgrad, last_ids = output
    ctx.mark_non_differentiable(last_ids, means2d_absgrad)
    ctx.geom = (image_width, image_height, tile_size, tile_width, tile_height)
    ctx.absgrad = absgrad
    ctx.set_materialize_grads(False)
    ctx.rc_shape = _render_colors.shape
    ctx.save_for_backward(means2d, conics, colors, opacities, backgrounds, masks, image_ids, active_tiles, tile_offsets,
                          flatten_ids, tile_pixel_mask, tile_pixel_cumsum, pixel_map, render_alphas, last_ids,
                          means2d_absgrad)


def _rsp_backward(ctx, v_render_colors, v_render_alphas, v_means2d_absgrad, v_last_ids):
    (means2d, conics, colors, opacities, backgrounds, masks, image_ids, active_tiles, tile_offsets, flatten_ids,
     tile_pixel_mask, tile_pixel_cumsum, pixel_map, render_alphas, last_ids, means2d_absgrad) = ctx.saved_tensors
    v_means2d_abs, v_means2d, v_conics, v_colors, v_opacities, v_backgrounds = _bwd("rasterize_to_pixels_sparse")(
        means2d, conics, colors, opacities, backgrounds, masks, image_ids, active_tiles, tile_offsets, flatten_ids,
        tile_pixel_mask, tile_pixel_cumsum, pixel_map, render_alphas, last_ids, *ctx.geom, ctx.absgrad,
        _z(v_render_colors, ctx.rc_shape, render_alphas).contiguous(),
        None if v_render_alphas is None else v_render_alphas.contiguous(), ctx.needs_input_grad[4],
    )
    if ctx.absgrad and v_means2d_abs is not None:
        means2d_absgrad.copy_(v_means2d_abs)
    return (v_means2d, v_conics, v_colors, v_opacities, v_backgrounds) + (None,) * 15


# ---- 2DGS projection (reference _wrapper.py:2730-2915) -------------------------------------------
def _p2_setup(ctx, inputs, output):
    means, quats, scales, viewmats, Ks, width, height, _eps2d, _near, _far, _clip = inputs
    radii, _means2d, _depths, ray_transforms, _normals = output
    ctx.width, ctx.height = width, height
    ctx.save_for_backward(means, quats, scales, viewmats, Ks, radii, ray_transforms)


def _p2_backward(ctx, v_radii, v_means2d, v_depths, v_ray_transforms, v_normals, v_view_opacities=None):
    means, quats, scales, viewmats, Ks, radii, ray_transforms = ctx.saved_tensors
    res = _bwd("projection_2dgs_fused")(
        means, quats, scales, viewmats, Ks, ctx.width, ctx.height, radii, ray_transforms, v_means2d,
        v_depths, v_ray_transforms, v_normals, ctx.needs_input_grad[3],  # row / column views are read in place
        _v_view_opacities=v_view_opacities)
    if v_view_opacities is not None:  # Projection2DGSWithViewOpacities: its twelfth input is `opacities`
        return tuple(res[:4]) + (None,) * 7 + (res[4],)
    return tuple(res) + (None,) * 7


class Projection2DGSWithViewOpacities(torch.autograd.Function):
    """projection_2dgs_fused (dense rows) that also hands out the per-view opacities [..., C, N] as one of its outputs, so that
    their cotangent - a column of the 2DGS compositing backward's gradient rows - is summed over the views by the projection
    backward kernel (see ProjectionWithViewOpacities). `opacities` is the LAST input. gsplat_amd's rasterization_2dgs() only."""

    @staticmethod
    def forward(ctx, *args):
        op_args, opacities = args[:-1], args[-1]
        with torch._C._AutoDispatchBelowAutograd():
            out = getattr(torch.ops, NS).projection_2dgs_fused.default(*op_args)
        _p2_setup(ctx, op_args, out)
        viewmats = op_args[3]
        per_view = torch.broadcast_to(opacities[..., None, :], opacities.shape[:-1] + (viewmats.shape[-3], opacities.shape[-1]))
        # several views: ONE materialised copy here - the intersection and the compositing op each made their own from the
        # stride-0 view (two copies of [C, N] per step: 45 us at 4 x 4 M rows); a single view stays a view (already contiguous)
        per_view = per_view.contiguous()
        ctx.mark_non_differentiable(out[0])
        return tuple(out) + (per_view,)

    @staticmethod
    def backward(ctx, v_radii, v_means2d, v_depths, v_ray_transforms, v_normals, v_per_view):
        if v_per_view is None or not ctx.needs_input_grad[11]:
            return _p2_backward(ctx, v_radii, v_means2d, v_depths, v_ray_transforms, v_normals) + (None,)
        return _p2_backward(ctx, v_radii, v_means2d, v_depths, v_ray_transforms, v_normals, v_per_view)


def _p2p_setup(ctx, inputs, output):
    means, quats, scales, viewmats, Ks, width, height, _near, _far, _clip, sparse_grad = inputs
    batch_ids, camera_ids, gaussian_ids, _indptr, _radii, _means2d, _depths, ray_transforms, _normals = output
    ctx.width, ctx.height, ctx.sparse_grad = width, height, sparse_grad
    ctx.save_for_backward(means, quats, scales, viewmats, Ks, batch_ids, camera_ids, gaussian_ids, ray_transforms)


def _p2p_backward(ctx, v_b, v_c, v_g, v_indptr, v_radii, v_means2d, v_depths, v_ray_transforms, v_normals):
    means, quats, scales, viewmats, Ks, batch_ids, camera_ids, gaussian_ids, ray_transforms = ctx.saved_tensors
    v_means, v_quats, v_scales, v_viewmats = _bwd("projection_2dgs_packed")(
        means, quats, scales, viewmats, Ks, ctx.width, ctx.height, ctx.sparse_grad, batch_ids, camera_ids,
        gaussian_ids, ray_transforms, v_means2d, v_depths.contiguous(), v_ray_transforms, v_normals,
        ctx.needs_input_grad[3])  # row views are read in place
    return (v_means, v_quats, v_scales, v_viewmats) + (None,) * 7


# ---- 2DGS compositing (reference _wrapper.py:3004-3150) ------------------------------------------
def _r2_setup(ctx, inputs, output):
    (means2d, ray_transforms, colors, opacities, normals, densify, backgrounds, masks, image_width, image_height,
     tile_size, tile_offsets, flatten_ids, _packed, absgrad, _distloss) = inputs
    (render_colors, render_alphas, _rn, _rd, _rm, means2d_absgrad, last_ids, median_ids) = output
    ctx.mark_non_differentiable(last_ids, median_ids, means2d_absgrad)
    ctx.width, ctx.height, ctx.tile_size, ctx.absgrad = image_width, image_height, tile_size, absgrad
    ctx.set_materialize_grads(False)  # an output the loss does not use gets no zero-filled gradient (the kernels take NULL = zeros)
    ctx.rc_shape = render_colors.shape
    ctx.save_for_backward(means2d, ray_transforms, colors, opacities, normals, densify, backgrounds, masks,
                          tile_offsets, flatten_ids, render_colors, render_alphas, last_ids, median_ids,
                          means2d_absgrad)


def _r2_backward(ctx, v_render_colors, v_render_alphas, v_render_normals, v_render_distort, v_render_median,
                 v_absgrad, v_last_ids, v_median_ids):
    (means2d, ray_transforms, colors, opacities, normals, densify, backgrounds, masks, tile_offsets, flatten_ids,
     render_colors, render_alphas, last_ids, median_ids, means2d_absgrad) = ctx.saved_tensors
    (v_means2d_abs, v_means2d, v_ray_transforms, v_colors, v_opacities, v_normals, v_densify,
     v_backgrounds) = _bwd("rasterize_to_pixels_2dgs")(
        means2d, ray_transforms, colors, opacities, normals, densify, backgrounds, masks, tile_offsets, flatten_ids,
        render_colors, render_alphas, last_ids, median_ids, ctx.width, ctx.height, ctx.tile_size, ctx.absgrad,
        _z(v_render_colors, ctx.rc_shape, render_alphas).contiguous(), _cg(v_render_alphas), _cg(v_render_normals),
        _cg(v_render_distort), _cg(v_render_median), ctx.needs_input_grad[6])
    if ctx.absgrad and v_means2d_abs is not None:
        means2d_absgrad.copy_(v_means2d_abs)
    return (v_means2d, v_ray_transforms, v_colors, v_opacities, v_normals, v_densify, v_backgrounds) + (None,) * 9


# ---- split SH + proj (reference _wrapper.py:782-816 and the l0 / l1_plus Register classes) -----------------
def _l0_setup(ctx, inputs, output):
    ctx.save_for_backward(inputs[0])


def _l0_backward(ctx, v_colors):
    (sh0,) = ctx.saved_tensors
    return _bwd("spherical_harmonics_l0")(sh0, v_colors.contiguous())


def _l1_setup(ctx, inputs, output):
    (degrees_to_use, means, viewmats, shN, masks, batch_ids, camera_ids, gaussian_ids, viewmats_rs) = inputs
    ctx.degrees_to_use = degrees_to_use
    ctx.save_for_backward(means, viewmats, shN, masks, batch_ids, camera_ids, gaussian_ids, viewmats_rs)


def _l1_backward(ctx, v_colors):
    means, viewmats, shN, masks, batch_ids, camera_ids, gaussian_ids, viewmats_rs = ctx.saved_tensors
    v_shN, v_means, v_viewmats, v_rs = _bwd("spherical_harmonics_l1_plus")(
        ctx.degrees_to_use, means, viewmats, shN, masks, batch_ids, camera_ids, gaussian_ids, viewmats_rs,
        v_colors.contiguous(), ctx.needs_input_grad[1], ctx.needs_input_grad[2],
        len(ctx.needs_input_grad) > 8 and ctx.needs_input_grad[8])
    return (None, v_means, v_viewmats, v_shN, None, None, None, None, v_rs)


def _simple_setup(ctx, inputs, output):
    means, covars, Ks, width, height, camera_model = inputs
    ctx.width, ctx.height, ctx.camera_model = width, height, camera_model
    ctx.save_for_backward(means, covars, Ks)


def _simple_backward(ctx, v_means2d, v_covars2d):
    means, covars, Ks = ctx.saved_tensors
    v_means, v_covars = _bwd("projection_ewa_simple")(means, covars, Ks, ctx.width, ctx.height, ctx.camera_model,
                                                      v_means2d.contiguous(), v_covars2d.contiguous())
    return v_means, v_covars, None, None, None, None


_TABLE = {
    "spherical_harmonics_l0": (_l0_backward, _l0_setup),
    "spherical_harmonics_l1_plus": (_l1_backward, _l1_setup),
    "projection_ewa_simple": (_simple_backward, _simple_setup),
    "projection_2dgs_fused": (_p2_backward, _p2_setup),
    "projection_2dgs_packed": (_p2p_backward, _p2p_setup),
    "rasterize_to_pixels_2dgs": (_r2_backward, _r2_setup),
    "quat_scale_to_covar_preci": (_qs_backward, _qs_setup),
    "spherical_harmonics": (_sh_backward, _sh_setup),
    "projection_ewa_3dgs_fused": (_proj_backward, _proj_setup),
    "projection_ewa_3dgs_packed": (_projp_backward, _projp_setup),
    "rasterize_to_pixels_3dgs": (_rast_backward, _rast_setup),
    "rasterize_to_pixels_sparse": (_rsp_backward, _rsp_setup),
}


def register(extra=None) -> None:
    global _registered
    if _registered:
        return
    table = dict(_TABLE)
    if extra:
        table.update(extra)
    for name, (bwd, setup) in table.items():
        try:
            torch.library.register_autograd(f"{NS}::{name}", bwd, setup_context=setup)
        except RuntimeError as e:  # already registered by the reference's _wrapper.py in this process
            if "already" not in str(e).lower():
                raise
    _registered = True


register()


# ---- the same autograd without the dispatcher's Python trampoline --------------------------------
# An op called through torch.ops with autograd attached by torch.library.register_autograd passes through
# torch/_library/autograd.py (autograd_impl -> fill_defaults -> a generated Function): ~70 us of interpreter time per call,
# more than the launch of the kernel behind it (tools/host_time.py: 0.14 ms of a 0.42 ms host-bound step for two ops).
# gsplat_amd's own wrappers therefore attach the SAME setup / backward pair through a plain torch.autograd.Function and call
# the op body below the Autograd dispatch key. torch.ops.gsplat.<op> itself keeps the registration above (that is what a
# caller of the raw op - or the reference's tests - gets).
class _FastOps:
    """`torch.ops.gsplat` with direct autograd Functions for the ops in _TABLE; every other attribute passes through."""

    def __init__(self, extra=None):
        self._table = dict(_TABLE)
        if extra:
            self._table.update(extra)
        self._ns = getattr(torch.ops, NS)

    def __getattr__(self, name):
        entry = self._table.get(name)
        op = getattr(self._ns, name)
        if entry is None:
            fn = op
        else:
            bwd, setup = entry
            overload = op.default

            class _Fn(torch.autograd.Function):
                @staticmethod
                def forward(ctx, *args):
                    with torch._C._AutoDispatchBelowAutograd():
                        out = overload(*args)
                    setup(ctx, args, out)
                    return out

                @staticmethod
                def backward(ctx, *grads):
                    return bwd(ctx, *grads)

            _Fn.__name__ = _Fn.__qualname__ = f"gsplat_{name}"
            fn = _Fn.apply
        setattr(self, name, fn)  # next lookup is a plain attribute
        return fn


fast_ops = _FastOps()
