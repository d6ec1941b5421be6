"""CPU oracle for the Gaussian-rasterization hot path (TEST INFRASTRUCTURE ONLY).

Nothing under ``gsplat_amd/`` may import this module. It is used by ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg, only as the checker /
timed CPU baseline.

Two halves:

* torch-CPU (differentiable) restatements of the per-Gaussian stages — projection, spherical
  harmonics, quat/scale -> covariance — each citing the reference lines it follows. Gradients
  come from torch autograd, exactly how the reference's own tests obtain theirs
  (``tests/test_basic.py:436-504``).
* ctypes bindings to ``oracle/gsplat_oracle.c`` (plain C + OpenMP) for the integer / per-pixel
  stages — tile intersection, key packing, offsets, alpha compositing forward and backward.

Pinning status: see ``oracle/pin_against_reference.py`` — every function here is checked against
the reference's own Python implementation (``gsplat/cuda/_torch_impl.py``) where that runs on
CPU, and against ``accumulate()`` (driven through a restated nerfacc) for the compositing stage;
the resulting golden vectors live in ``tests/golden/``.
"""
from __future__ import annotations

import ctypes
import math
import os
import subprocess
from typing import Optional, Tuple

import numpy as np
import torch
from torch import Tensor

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libgsplat_oracle.so")

ALPHA_THRESHOLD = 1.0 / 255.0  # gsplat/cuda/include/Common.h:97
GAUSSIAN_EXTEND = 3.33  # Common.h:99
MAX_ALPHA = 0.99  # Common.h:105
MIN_COMPENSATION = 0.005  # Common.h:110


def build(force: bool = False) -> str:
    """Compile oracle/gsplat_oracle.c into oracle/_build/libgsplat_oracle.so (gcc, OpenMP)."""
    src = os.path.join(_HERE, "gsplat_oracle.c")
    if (
        not force
        and os.path.exists(_LIB_PATH)
        and os.path.getmtime(_LIB_PATH) >= os.path.getmtime(src)
    ):
        return _LIB_PATH
    os.makedirs(os.path.dirname(_LIB_PATH), exist_ok=True)
    cmd = [
        "gcc", "-O2", "-std=c11", "-fPIC", "-shared", "-fopenmp", "-ffp-contract=off",
        "-fno-fast-math", src, "-lm", "-o", _LIB_PATH,
    ]
    subprocess.run(cmd, check=True)
    return _LIB_PATH


_lib = None


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        _lib.gso_det_logf.restype = ctypes.c_float
        _lib.gso_det_logf.argtypes = [ctypes.c_float]
        _lib.gso_raster3d_indices.restype = ctypes.c_int64
        _lib.gso_raster2d_indices.restype = ctypes.c_int64
        _lib.gso_isect_emit.restype = ctypes.c_int
    return _lib


def set_threads(n: int) -> None:
    """Threads for the OpenMP loops of the C half and for torch-CPU (very wide hosts are slower with all cores:
    fork/join and atomic contention dominate these small kernels)."""
    lib().gso_set_threads(ctypes.c_int(int(n)))
    torch.set_num_threads(int(n))


def _p(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _np(t, dtype) -> Optional[np.ndarray]:
    if t is None:
        return None
    if isinstance(t, Tensor):
        t = t.detach().cpu().numpy()
    return np.ascontiguousarray(t, dtype=dtype)


def bits_for_count(count: int) -> int:
    """gsplat/cuda/csrc/MathUtils.h:25-35, gsplat/cuda/_torch_impl.py:42-50."""
    return (count - 1).bit_length() if count > 1 else 0


# ----------------------------------------------------------------------------------------------
# per-Gaussian stages (torch, differentiable)
# ----------------------------------------------------------------------------------------------
def quat_to_rotmat(quats: Tensor) -> Tensor:
    """Utils.cuh:228-250 / gsplat/cuda/_math.py:655-674 (wxyz, normalised inside)."""
    q = quats / quats.norm(dim=-1, keepdim=True)
    w, x, y, z = q.unbind(-1)
    R = torch.stack(
        [
            1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
            2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
            2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y),
        ],
        dim=-1,
    )
    return R.reshape(quats.shape[:-1] + (3, 3))


def _triu6(M: Tensor) -> Tensor:
    f = M.reshape(M.shape[:-2] + (9,))
    return (f[..., [0, 1, 2, 4, 5, 8]] + f[..., [0, 3, 6, 4, 7, 8]]) / 2.0


def quat_scale_to_covar_preci(quats, scales, compute_covar=True, compute_preci=True, triu=False):
    """Utils.cuh:285-311 / gsplat/cuda/_math.py:689-720."""
    R = quat_to_rotmat(quats)
    covars = precis = None
    if compute_covar:
        M = R * scales[..., None, :]
        covars = M @ M.transpose(-1, -2)
        if triu:
            covars = _triu6(covars)
    if compute_preci:
        P = R * (1.0 / scales[..., None, :])
        precis = P @ P.transpose(-1, -2)
        if triu:
            precis = _triu6(precis)
    return covars, precis


def _covar6_to_mat(c6: Tensor) -> Tensor:
    i = [0, 1, 2, 1, 3, 4, 2, 4, 5]
    return c6[..., i].reshape(c6.shape[:-1] + (3, 3))


def det_logf(x: Tensor) -> Tensor:
    """Element-wise gso_det_logf (bit-identical to the device det_logf) on a float32 tensor."""
    a = np.ascontiguousarray(x.detach().cpu().numpy(), dtype=np.float32)
    out = np.empty_like(a)
    lib().gso_det_logf_array(_p(a), _p(out), ctypes.c_int64(a.size))
    return torch.from_numpy(out)


def fully_fused_projection(
    means: Tensor,  # [B, N, 3]
    covars: Optional[Tensor],  # [B, N, 6]
    quats: Optional[Tensor],  # [B, N, 4]
    scales: Optional[Tensor],  # [B, N, 3]
    viewmats: Tensor,  # [B, C, 4, 4]
    Ks: Tensor,  # [B, C, 3, 3]
    width: int,
    height: int,
    eps2d: float = 0.3,
    near_plane: float = 0.01,
    far_plane: float = 1e10,
    radius_clip: float = 0.0,
    calc_compensations: bool = False,
    camera_model: str = "pinhole",
    opacities: Optional[Tensor] = None,  # [B, N]
):
    """The CUDA kernel's behaviour (ProjectionEWA3DGSFused.cu:38-219), including what the
    reference's torch version lacks: opacity-aware extent (:163-181), radius_clip (:188),
    det <= 0 cull (:153), <=/>= frustum test (:196-199), near/far with </> (:100).
    Math per Utils.cuh:81-148 (world->cam), :455-463 (blur), :498-728 (projections).
    Returns radii int32 [B,C,N,2], means2d, depths, conics, compensations (or None);
    culled rows hold zeros."""
    dt = means.dtype
    if covars is None:
        covars_w = quat_scale_to_covar_preci(quats, scales, True, False, False)[0]
    else:
        covars_w = _covar6_to_mat(covars)
    R = viewmats[..., :3, :3]  # [B,C,3,3]
    t = viewmats[..., :3, 3]
    means_c = torch.einsum("bcij,bnj->bcni", R, means) + t[:, :, None, :]
    covars_c = R[:, :, None] @ covars_w[:, None] @ R[:, :, None].transpose(-1, -2)  # [B,C,N,3,3]
    x, y, z = means_c.unbind(-1)
    fx = Ks[..., 0, 0, None]
    fy = Ks[..., 1, 1, None]
    cx = Ks[..., 0, 2, None]
    cy = Ks[..., 1, 2, None]
    O = torch.zeros_like(x)
    if camera_model == "pinhole":
        tan_fovx = 0.5 * width / fx
        tan_fovy = 0.5 * height / fy
        lxp = (width - cx) / fx + 0.3 * tan_fovx
        lxn = cx / fx + 0.3 * tan_fovx
        lyp = (height - cy) / fy + 0.3 * tan_fovy
        lyn = cy / fy + 0.3 * tan_fovy
        tx = z * torch.minimum(lxp, torch.maximum(-lxn, x / z))
        ty = z * torch.minimum(lyp, torch.maximum(-lyn, y / z))
        J = torch.stack([fx / z, O, -fx * tx / z**2, O, fy / z, -fy * ty / z**2], dim=-1)
        m2 = torch.stack([fx * x / z + cx, fy * y / z + cy], dim=-1)
    elif camera_model == "ortho":
        J = torch.stack([fx + O, O, O, O, fy + O, O], dim=-1)
        m2 = torch.stack([fx * x + cx, fy * y + cy], dim=-1)
    elif camera_model == "fisheye":
        eps = 1e-7
        r = (x * x + y * y) ** 0.5 + eps
        th = torch.atan2(r, z + eps)
        m2 = torch.stack([x * fx * th / r + cx, y * fy * th / r + cy], dim=-1)
        x2 = x * x + eps
        y2 = y * y
        xy = x * y
        r2 = x2 + y2
        il2 = 1.0 / (r2 + z * z)
        bb = torch.atan2(r, z) / r / r2
        aa = z * il2 / r2
        J = torch.stack(
            [fx * (x2 * aa + y2 * bb), fx * xy * (aa - bb), -fx * x * il2,
             fy * xy * (aa - bb), fy * (y2 * aa + x2 * bb), -fy * y * il2],
            dim=-1,
        )
    else:
        raise ValueError(camera_model)
    J = J.reshape(J.shape[:-1] + (2, 3))
    cov2d = J @ covars_c @ J.transpose(-1, -2)
    a_, b_, d_ = cov2d[..., 0, 0], cov2d[..., 0, 1], cov2d[..., 1, 1]
    det_orig = a_ * d_ - b_ * b_
    a_b, d_b = a_ + eps2d, d_ + eps2d
    det_blur = a_b * d_b - b_ * b_
    valid = (z >= near_plane) & (z <= far_plane) & (det_blur > 0)
    safe_det = torch.where(det_blur > 0, det_blur, torch.ones_like(det_blur))
    comp = torch.sqrt(torch.clamp(det_orig / safe_det, min=MIN_COMPENSATION**2))
    conics = torch.stack([d_b / safe_det, -b_ / safe_det, a_b / safe_det], dim=-1)
    extend = torch.full_like(z, GAUSSIAN_EXTEND)
    if opacities is not None:
        op = opacities[:, None, :].expand_as(z)
        if calc_compensations:
            op = op * comp
        valid = valid & ~(op < ALPHA_THRESHOLD)
        with torch.no_grad():
            lg = det_logf((op.detach().float() * 255.0).clamp_min(1.0)).to(dt)
        extend = torch.minimum(extend, torch.sqrt(2.0 * lg))
    with torch.no_grad():
        rx = torch.ceil(extend * torch.sqrt(a_b.clamp_min(0)))
        ry = torch.ceil(extend * torch.sqrt(d_b.clamp_min(0)))
        valid = valid & ~((rx <= radius_clip) & (ry <= radius_clip))
        inside = ~(
            (m2[..., 0] + rx <= 0) | (m2[..., 0] - rx >= width) | (m2[..., 1] + ry <= 0) | (m2[..., 1] - ry >= height)
        )
        valid = valid & inside
        radii = torch.stack([rx, ry], dim=-1) * valid[..., None]
        radii = torch.nan_to_num(radii, nan=0.0, posinf=0.0, neginf=0.0).to(torch.int32)
    vm = valid[..., None]
    zero = torch.zeros((), dtype=dt)
    means2d = torch.where(vm, m2, zero)
    depths = torch.where(valid, z, zero)
    conics = torch.where(vm, conics, zero)
    comps = torch.where(valid, comp, zero) if calc_compensations else None
    return radii, means2d, depths, conics, comps


def proj(means_c: Tensor, covars_c: Tensor, Ks: Tensor, width: int, height: int, camera_model: str = "pinhole"):
    """gsplat.proj / projection_ewa_simple (ProjectionEWASimple.cu; _torch_impl.py:53-260 _persp_proj / _ortho_proj /
    _fisheye_proj): camera-space means [..., C, N, 3], covars [..., C, N, 3, 3], Ks [..., C, 3, 3] ->
    means2d [..., C, N, 2], covars2d [..., C, N, 2, 2]. Differentiable."""
    x, y, z = means_c.unbind(-1)
    fx, fy = Ks[..., 0, 0, None], Ks[..., 1, 1, None]
    cx, cy = Ks[..., 0, 2, None], Ks[..., 1, 2, None]
    O = torch.zeros_like(x)
    if camera_model == "pinhole":
        tan_fovx, tan_fovy = 0.5 * width / fx, 0.5 * height / fy
        lxp, lxn = (width - cx) / fx + 0.3 * tan_fovx, cx / fx + 0.3 * tan_fovx
        lyp, lyn = (height - cy) / fy + 0.3 * tan_fovy, cy / fy + 0.3 * tan_fovy
        tx = z * torch.minimum(lxp, torch.maximum(-lxn, x / z))
        ty = z * torch.minimum(lyp, torch.maximum(-lyn, y / z))
        J = torch.stack([fx / z, O, -fx * tx / z**2, O, fy / z, -fy * ty / z**2], dim=-1)
        m2 = torch.stack([fx * x / z + cx, fy * y / z + cy], dim=-1)
    elif camera_model == "ortho":
        J = torch.stack([fx + O, O, O, O, fy + O, O], dim=-1)
        m2 = torch.stack([fx * x + cx, fy * y + cy], dim=-1)
    elif camera_model == "fisheye":
        eps = 1e-7
        r = (x * x + y * y) ** 0.5 + eps
        th = torch.atan2(r, z + eps)
        m2 = torch.stack([x * fx * th / r + cx, y * fy * th / r + cy], dim=-1)
        x2, y2, xy = x * x + eps, y * y, x * y
        r2 = x2 + y2
        il2 = 1.0 / (r2 + z * z)
        bb = torch.atan2(r, z) / r / r2
        aa = z * il2 / r2
        J = torch.stack([fx * (x2 * aa + y2 * bb), fx * xy * (aa - bb), -fx * x * il2,
                         fy * xy * (aa - bb), fy * (y2 * aa + x2 * bb), -fy * y * il2], dim=-1)
    else:
        raise ValueError(camera_model)
    J = J.reshape(J.shape[:-1] + (2, 3))
    return m2, J @ covars_c @ J.transpose(-1, -2)


def sh_bases(degree: int, dirs: Tensor) -> Tensor:
    """Sloan's polynomial SH basis, degree <= 4 (SphericalHarmonicsCUDA.cu:48-146 /
    gsplat/cuda/_torch_impl.py:968-1047). dirs must be unit length. Returns [..., (deg+1)^2]."""
    x, y, z = dirs.unbind(-1)
    out = [torch.full_like(x, 0.2820947917738781)]
    if degree >= 1:
        out += [-0.48860251190292 * y, 0.48860251190292 * z, -0.48860251190292 * x]
    if degree >= 2:
        z2 = z * z
        fB = -1.092548430592079 * z
        fC1 = x * x - y * y
        fS1 = 2 * x * y
        out += [0.5462742152960395 * fS1, fB * y, 0.9461746957575601 * z2 - 0.3153915652525201, fB * x,
                0.5462742152960395 * fC1]
    if degree >= 3:
        fC = -2.285228997322329 * z2 + 0.4570457994644658
        fB = 1.445305721320277 * z
        fA = -0.5900435899266435
        fC2 = x * fC1 - y * fS1
        fS2 = x * fS1 + y * fC1
        out += [fA * fS2, fB * fS1, fC * y, z * (1.865881662950577 * z2 - 1.119528997770346), fC * x, fB * fC1,
                fA * fC2]
    if degree >= 4:
        fD = z * (-4.683325804901025 * z2 + 2.007139630671868)
        fC = 3.31161143515146 * z2 - 0.47308734787878
        fB = -1.770130769779931 * z
        fA = 0.6258357354491763
        fC3 = x * fC2 - y * fS2
        fS3 = x * fS2 + y * fC2
        out += [
            fA * fS3, fB * fS2, fC * fS1, fD * y,
            1.984313483298443 * z2 * (1.865881662950577 * z2 - 1.119528997770346)
            - 1.006230589874905 * (0.9461746957575601 * z2 - 0.3153915652525201),
            fD * x, fC * fC1, fB * fC2, fA * fC3,
        ]
    return torch.stack(out, dim=-1)


def spherical_harmonics(degree: int, means: Tensor, viewmats: Tensor, coeffs: Tensor, masks: Optional[Tensor] = None):
    """gsplat.spherical_harmonics (dense): means [B,N,3], viewmats [B,C,4,4], coeffs [N,K,D],
    masks bool [B,C,N] -> colors [B,C,N,D]. dir = mean - camera centre (= mean + R^T t),
    SphericalHarmonics.cuh:39-69; evaluation _torch_impl.py:1052-1067. Masked rows are zero."""
    R = viewmats[..., :3, :3]
    t = viewmats[..., :3, 3]
    campos = -torch.einsum("bcji,bcj->bci", R, t)  # -R^T t
    dirs = means[:, None, :, :] - campos[:, :, None, :]
    dirs = dirs / dirs.norm(dim=-1, keepdim=True)
    Y = sh_bases(degree, dirs)  # [B,C,N,nb]
    nb = Y.shape[-1]
    colors = torch.einsum("bcnk,nkd->bcnd", Y, coeffs[:, :nb, :])
    if masks is not None:
        colors = colors * masks[..., None]
    return colors


def assemble_proj_features(degree: int, means: Tensor, viewmats: Tensor, coeffs: Tensor, extra: Optional[Tensor],
                           depths: Optional[Tensor], masks: Optional[Tensor], color_post: int, extra_post: int,
                           has_depth: bool, extra_has_c: bool) -> Tensor:
    """gsplat::assemble_proj_features_unpacked_fwd (kernel SphericalHarmonicsCUDA.cu:1100-1250; host
    SphericalHarmonics.cpp:572-676): rows [B,C,N, Dc+E+has_depth] = [ post(SH colours) | extra | depth ].
    post (apply_sh_post, :1017-1029): 0 none, 1 x + 0.5, 2 max(x + 0.5, 0). Masked rows get ZERO colours - the post op is
    not applied to them (:1225-1231) - while their extra / depth columns are still written; extras are shifted by 0.5
    only when extra_post == 1 (:1155); depths None with has_depth = the zero column of `depth_is_zero`.
    means [B,N,3], viewmats [B,C,4,4], coeffs [N,K,Dc], extra [B,C,N,E] or [B,N,E], depths / masks [B,C,N]."""
    sh = spherical_harmonics(degree, means, viewmats, coeffs)
    col = sh if color_post == 0 else sh + 0.5
    if color_post == 2:
        col = col.clamp_min(0)
    if masks is not None:
        col = torch.where(masks[..., None], col, torch.zeros_like(col))
    parts = [col]
    if extra is not None:
        e = extra + (0.5 if extra_post == 1 else 0.0)
        parts.append(e if extra_has_c else e[:, None].expand(col.shape[:-1] + (e.shape[-1],)))
    if has_depth:
        parts.append(torch.zeros_like(col[..., :1]) if depths is None else depths[..., None])
    return torch.cat(parts, dim=-1)


# ----------------------------------------------------------------------------------------------
# integer / per-pixel stages (C)
# ----------------------------------------------------------------------------------------------
def isect_tiles(
    means2d, radii, depths, tile_size: int, tile_width: int, tile_height: int, sort: bool = True,
    conics=None, opacities=None, image_ids=None, n_images: Optional[int] = None,
):
    """gsplat.isect_tiles (IntersectTile.cu:214-464, _torch_impl.py:356-451). Dense inputs are
    [I,N,*]; packed inputs are [nnz,*] with image_ids. Returns (tiles_per_gauss int32,
    isect_ids int64 [M], flatten_ids int32 [M]); sorted with a STABLE sort when sort=True
    (cub radix sort is stable, IntersectTile.cu:1096-1104)."""
    if _is_f64(means2d):
        return _isect_tiles_f64(means2d, radii, depths, tile_size, tile_width, tile_height, sort, conics, opacities, image_ids,
                                n_images)
    m2 = _np(means2d, np.float32)
    packed = image_ids is not None
    if packed:
        rows, n_per = m2.shape[0], 1
        I = int(n_images)
        out_shape = (rows,)
    else:
        I, n_per = int(np.prod(m2.shape[:-2])), m2.shape[-2]
        rows = I * n_per
        out_shape = m2.shape[:-1]
    m2 = m2.reshape(rows, 2)
    rd = _np(radii, np.int32).reshape(rows, 2)
    dp = _np(depths, np.float32).reshape(rows)
    cn = None if conics is None else _np(conics, np.float32).reshape(rows, 3)
    op = None if opacities is None else _np(opacities, np.float32).reshape(rows)
    ii = None if not packed else _np(image_ids, np.int64).reshape(rows)
    tpg = np.zeros(rows, dtype=np.int32)
    L = lib()
    L.gso_isect_count(_p(m2), _p(rd), _p(cn), _p(op), ctypes.c_int64(rows), ctypes.c_uint32(tile_size),
                      ctypes.c_uint32(tile_width), ctypes.c_uint32(tile_height), _p(tpg))
    cum = np.cumsum(tpg, dtype=np.int64)
    M = int(cum[-1]) if rows > 0 else 0
    ids = np.zeros(M, dtype=np.int64)
    fl = np.zeros(M, dtype=np.int32)
    rc = L.gso_isect_emit(_p(m2), _p(rd), _p(dp), _p(cn), _p(op), _p(ii), _p(cum), ctypes.c_int64(rows),
                          ctypes.c_uint32(n_per), ctypes.c_uint32(I), ctypes.c_uint32(tile_size),
                          ctypes.c_uint32(tile_width), ctypes.c_uint32(tile_height), _p(ids), _p(fl))
    if rc != 0:
        raise RuntimeError("isect key overflow: image bits + tile bits > 32")
    if sort and M > 0:
        order = np.argsort(ids.view(np.uint64), kind="stable")
        ids, fl = ids[order], fl[order]
    return (torch.from_numpy(tpg.reshape(out_shape)), torch.from_numpy(ids), torch.from_numpy(fl))


def _is_f64(x) -> bool:
    return getattr(x, "dtype", None) in (torch.float64, np.float64) or (isinstance(x, np.ndarray) and x.dtype == np.float64)


def _isect_tiles_f64(means2d, radii, depths, tile_size, tile_width, tile_height, sort, conics, opacities, image_ids, n_images):
    """float64 rows (the reference instantiates intersect_tile for double: IntersectTile.cu AT_DISPATCH_FLOATING_TYPES; its
    torch restatement _torch_impl.py:356-451 runs in the input dtype): radius boxes floor((m - r) / ts) .. ceil((m + r) / ts) in
    double, clamped to the tile grid; rows enumerated y-major; the key carries the depth narrowed to float32
    (tests/test_basic.py:1282-1287). The exact ellipse test is fp32 only."""
    if conics is not None or opacities is not None:
        raise TypeError("float64 rows: radius-box test only")
    m2 = _np(means2d, np.float64)
    packed = image_ids is not None
    if packed:
        rows, n_per, I, out_shape = m2.shape[0], 1, int(n_images), (m2.shape[0],)
    else:
        I, n_per = int(np.prod(m2.shape[:-2])), m2.shape[-2]
        rows, out_shape = I * n_per, m2.shape[:-1]
    m2 = m2.reshape(rows, 2)
    rd = _np(radii, np.int32).reshape(rows, 2)
    dp = _np(depths, np.float64).reshape(rows)
    iid = _np(image_ids, np.int64).reshape(rows) if packed else np.arange(rows, dtype=np.int64) // max(n_per, 1)
    ts = float(tile_size)
    t, r = m2 / ts, rd.astype(np.float64) / ts

    def to_int(x, hi):  # f2i_trunc_sat + clamp (csrc/isect_walk.hpp)
        x = np.where(np.isnan(x), 0.0, np.clip(x, -2.0e9, 2.0e9))
        return np.clip(x.astype(np.int64), 0, hi)

    x0, y0 = to_int(np.floor(t[:, 0] - r[:, 0]), tile_width), to_int(np.floor(t[:, 1] - r[:, 1]), tile_height)
    x1, y1 = to_int(np.ceil(t[:, 0] + r[:, 0]), tile_width), to_int(np.ceil(t[:, 1] + r[:, 1]), tile_height)
    live = (rd[:, 0] > 0) & (rd[:, 1] > 0)
    w, h = np.maximum(x1 - x0, 0), np.maximum(y1 - y0, 0)
    tpg = np.where(live, w * h, 0).astype(np.int32)
    tile_bits, image_bits = bits_for_count(tile_width * tile_height), bits_for_count(I)
    if tile_bits + image_bits > 32:
        raise RuntimeError("isect key overflow: image bits + tile bits > 32")
    M = int(tpg.sum())
    row = np.repeat(np.arange(rows, dtype=np.int64), tpg)
    k = np.arange(M, dtype=np.int64) - np.repeat(np.cumsum(tpg, dtype=np.int64) - tpg, tpg)
    wr = np.maximum(w[row], 1)
    tile = (y0[row] + k // wr) * tile_width + (x0[row] + k % wr)
    dbits = dp.astype(np.float32).view(np.uint32).astype(np.uint64)
    ids = ((iid[row].astype(np.uint64) << np.uint64(32 + tile_bits)) | (tile.astype(np.uint64) << np.uint64(32)) | dbits[row])
    fl = row.astype(np.int32)
    if sort and M > 0:
        order = np.argsort(ids, kind="stable")
        ids, fl = ids[order], fl[order]
    return (torch.from_numpy(tpg.reshape(out_shape)), torch.from_numpy(ids.view(np.int64).copy()), torch.from_numpy(fl))


def isect_offset_encode(isect_ids, n_images: int, tile_width: int, tile_height: int) -> Tensor:
    """gsplat.isect_offset_encode (IntersectTile.cu:925-988, _torch_impl.py:455-481)."""
    ids = _np(isect_ids, np.int64)
    off = np.zeros((n_images, tile_height, tile_width), dtype=np.int32)
    lib().gso_isect_offsets(_p(ids), ctypes.c_int64(ids.shape[0]), ctypes.c_uint32(n_images),
                            ctypes.c_uint32(tile_width), ctypes.c_uint32(tile_height), _p(off))
    return torch.from_numpy(off)


def _raster_common(means2d, conics, colors, opacities, isect_offsets):
    off = _np(isect_offsets, np.int32)
    I, th, tw = int(np.prod(off.shape[:-2])), off.shape[-2], off.shape[-1]
    cdim = colors.shape[-1]
    m2 = _np(means2d, np.float32).reshape(-1, 2)
    cn = _np(conics, np.float32).reshape(-1, 3)
    cl = _np(colors, np.float32).reshape(-1, cdim)
    op = _np(opacities, np.float32).reshape(-1)
    return off, I, th, tw, cdim, m2, cn, cl, op


def rasterize_to_pixels(
    means2d, conics, colors, opacities, image_width: int, image_height: int, tile_size: int, isect_offsets,
    flatten_ids, backgrounds=None, masks=None,
):
    """gsplat.rasterize_to_pixels forward (RasterizeToPixels3DGSSerialBatchFwd.cu:41-297).
    Returns render_colors [I,H,W,D], render_alphas [I,H,W,1], last_ids int32 [I,H,W]."""
    off, I, th, tw, cdim, m2, cn, cl, op = _raster_common(means2d, conics, colors, opacities, isect_offsets)
    fl = _np(flatten_ids, np.int32)
    bg = None if backgrounds is None else _np(backgrounds, np.float32).reshape(I, cdim)
    mk = None if masks is None else _np(masks, np.uint8).reshape(I, th, tw)
    rc = np.zeros((I, image_height, image_width, cdim), dtype=np.float32)
    ra = np.zeros((I, image_height, image_width, 1), dtype=np.float32)
    li = np.zeros((I, image_height, image_width), dtype=np.int32)
    lib().gso_raster3d_fwd(_p(m2), _p(cn), _p(cl), _p(op), _p(bg), _p(mk), _p(off), _p(fl), ctypes.c_uint32(I),
                           ctypes.c_uint32(fl.shape[0]), ctypes.c_uint32(cdim), ctypes.c_uint32(image_width),
                           ctypes.c_uint32(image_height), ctypes.c_uint32(tile_size), ctypes.c_uint32(tw),
                           ctypes.c_uint32(th), _p(rc), _p(ra), _p(li))
    return torch.from_numpy(rc), torch.from_numpy(ra), torch.from_numpy(li)


def rasterize_to_pixels_bwd(
    means2d, conics, colors, opacities, image_width: int, image_height: int, tile_size: int, isect_offsets,
    flatten_ids, render_alphas, last_ids, v_render_colors, v_render_alphas, backgrounds=None, masks=None,
    absgrad: bool = False, sample_f64: bool = False, sum_f32: bool = False, abs_sums: bool = False,
):
    """gsplat rasterize_to_pixels backward (RasterizeToPixels3DGSSerialBatchBwd.cu:41-320,
    Rasterization.cpp:567-577 for v_backgrounds). Returns dict of float64 numpy arrays. sample_f64: the per-sample math in
    fp64 too (gso_raster3d_bwd_f64: the value the fp32 evaluations approximate; per-element bands are measured against it);
    sum_f32: the sums in fp32 as well (gso_raster3d_bwd_f32sum: an fp32 evaluation in one of the orders the reference's atomics
    may take - the envelope of those bands); abs_sums (with sample_f64): also out["abs_terms"], the sum of |term| behind
    every gradient element."""
    off, I, th, tw, cdim, m2, cn, cl, op = _raster_common(means2d, conics, colors, opacities, isect_offsets)
    fl = _np(flatten_ids, np.int32)
    rows = m2.shape[0]
    bg = None if backgrounds is None else _np(backgrounds, np.float32).reshape(I, cdim)
    mk = None if masks is None else _np(masks, np.uint8).reshape(I, th, tw)
    ra = _np(render_alphas, np.float32).reshape(I, image_height, image_width)
    li = _np(last_ids, np.int32).reshape(I, image_height, image_width)
    vc = _np(v_render_colors, np.float32).reshape(I, image_height, image_width, cdim)
    va = _np(v_render_alphas, np.float32).reshape(I, image_height, image_width)
    v_abs = np.zeros((rows, 2), dtype=np.float64) if absgrad else None
    v_m = np.zeros((rows, 2), dtype=np.float64)
    v_cn = np.zeros((rows, 3), dtype=np.float64)
    v_cl = np.zeros((rows, cdim), dtype=np.float64)
    v_op = np.zeros((rows,), dtype=np.float64)
    assert not (sum_f32 and (sample_f64 or absgrad))
    fn = lib().gso_raster3d_bwd_f64 if sample_f64 else (lib().gso_raster3d_bwd_f32sum if sum_f32 else lib().gso_raster3d_bwd)
    abs_terms = None
    if abs_sums:
        assert sample_f64, "abs_sums is an output of the fp64 evaluation"
        abs_terms = np.zeros((rows, 6 + cdim), dtype=np.float64)
        lib().gso_set_abs_terms(_p(abs_terms))
    fn(_p(m2), _p(cn), _p(cl), _p(op), _p(bg), _p(mk), _p(off), _p(fl), _p(ra), _p(li), _p(vc),
                           _p(va), ctypes.c_uint32(I), ctypes.c_uint32(fl.shape[0]), ctypes.c_uint32(cdim),
                           ctypes.c_uint32(image_width), ctypes.c_uint32(image_height), ctypes.c_uint32(tile_size),
                           ctypes.c_uint32(tw), ctypes.c_uint32(th), ctypes.c_int64(rows), _p(v_abs), _p(v_m),
                           _p(v_cn), _p(v_cl), _p(v_op))
    out = {"v_means2d": v_m, "v_conics": v_cn, "v_colors": v_cl, "v_opacities": v_op}
    if abs_terms is not None:  # sum of |term| per gradient component: the conditioning of each sum (gso_set_abs_terms)
        out["abs_terms"] = {"v_means2d": abs_terms[:, 0:2], "v_conics": abs_terms[:, 2:5], "v_opacities": abs_terms[:, 5],
                            "v_colors": abs_terms[:, 6:]}
    if absgrad:
        out["v_means2d_abs"] = v_abs
    if backgrounds is not None:
        # v_backgrounds = sum_{h,w} v_colors * (1 - alpha)   (Rasterization.cpp:567-577)
        out["v_backgrounds"] = (vc.astype(np.float64) * (1.0 - ra.astype(np.float64))[..., None]).sum(axis=(1, 2))
    return out


def rasterize_to_indices(means2d, conics, opacities, image_width, image_height, tile_size, isect_offsets, flatten_ids):
    """(gaussian_ids, pixel_ids, image_ids) of every contributing (gaussian, pixel) pair, the input
    of the reference's torch compositor `accumulate` (RasterizeToIndices3DGSSerialBatch.cu:128-192)."""
    off = _np(isect_offsets, np.int32)
    I, th, tw = int(np.prod(off.shape[:-2])), off.shape[-2], off.shape[-1]
    m2 = _np(means2d, np.float32)
    n_per = m2.shape[-2]
    m2 = m2.reshape(-1, 2)
    cn = _np(conics, np.float32).reshape(-1, 3)
    op = _np(opacities, np.float32).reshape(-1)
    fl = _np(flatten_ids, np.int32)
    args = [_p(m2), _p(cn), _p(op), _p(off), _p(fl), ctypes.c_uint32(I), ctypes.c_uint32(fl.shape[0]),
            ctypes.c_uint32(n_per), ctypes.c_uint32(image_width), ctypes.c_uint32(image_height),
            ctypes.c_uint32(tile_size), ctypes.c_uint32(tw), ctypes.c_uint32(th)]
    n = lib().gso_raster3d_indices(*args, None, None, None)
    g = np.zeros(n, dtype=np.int64)
    p = np.zeros(n, dtype=np.int64)
    i = np.zeros(n, dtype=np.int64)
    lib().gso_raster3d_indices(*args, _p(g), _p(p), _p(i))
    return torch.from_numpy(g), torch.from_numpy(p), torch.from_numpy(i)


# ----------------------------------------------------------------------------------------------
# 2DGS (surfels)
# ----------------------------------------------------------------------------------------------
def fully_fused_projection_2dgs(means, quats, scales, viewmats, Ks, width: int, height: int, near_plane=0.01,
                                far_plane=1e10, radius_clip=0.0):
    """gsplat.fully_fused_projection_2dgs, dense (Projection2DGSFused.cu:39-339; torch restatement
    _torch_impl_2dgs.py:27-108). Differences of the CUDA kernel from the torch reference that are followed here:
    the third scale is ignored (RS = R * Rq * diag(sx, sy, 1), :167), near/far culling uses <= / >= (:155),
    `distance == 0` culls (:216), radius_clip (:236), image culling with <= / >= (:246-249).
    means [..., N, 3] ... -> radii int32 [..., C, N, 2], means2d, depths, ray_transforms [..., C, N, 3, 3] (rows
    M0, M1, M2 of K [RS0 RS1 mean_c]), normals [..., C, N, 3]. Differentiable (torch autograd = the reference's way of
    obtaining gradients for its tests, tests/test_2dgs.py)."""
    R_cw, t_cw = viewmats[..., :3, :3], viewmats[..., :3, 3]
    means_c = torch.einsum("...cij,...nj->...cni", R_cw, means) + t_cw[..., None, :]
    Rq = quat_to_rotmat(quats)
    s2 = torch.cat([scales[..., :2], torch.ones_like(scales[..., :1])], dim=-1)
    RS_wl = Rq * s2[..., None, :]
    RS_cl = torch.einsum("...cij,...njk->...cnik", R_cw, RS_wl)
    normals = RS_cl[..., 2]
    cos = -(normals * means_c).sum(-1, keepdim=True)
    normals = normals * torch.where(cos > 0, 1.0, -1.0)
    T_cl = torch.cat([RS_cl[..., :2], means_c[..., None]], dim=-1)
    Kfull = torch.zeros_like(Ks)
    Kfull[..., 0, 0], Kfull[..., 0, 2] = Ks[..., 0, 0], Ks[..., 0, 2]
    Kfull[..., 1, 1], Kfull[..., 1, 2] = Ks[..., 1, 1], Ks[..., 1, 2]
    Kfull[..., 2, 2] = 1.0  # the kernel only reads Ks[0], Ks[2], Ks[4], Ks[5] (:176)
    M = torch.einsum("...cij,...cnjk->...cnik", Kfull, T_cl)  # rows M0, M1, M2
    test = torch.tensor([1.0, 1.0, -1.0], dtype=means.dtype)
    d = (M[..., 2, :] * M[..., 2, :] * test).sum(-1, keepdim=True)
    ok = d != 0
    f = torch.where(ok, test / torch.where(ok, d, torch.ones_like(d)), torch.zeros_like(test))
    means2d = torch.stack([(f * M[..., 0, :] * M[..., 2, :]).sum(-1), (f * M[..., 1, :] * M[..., 2, :]).sum(-1)], -1)
    tmp = torch.stack([(f * M[..., 0, :] * M[..., 0, :]).sum(-1), (f * M[..., 1, :] * M[..., 1, :]).sum(-1)], -1)
    ext = torch.sqrt((means2d.detach() ** 2 - tmp.detach()).clamp_min(1e-4))
    radius = torch.ceil(GAUSSIAN_EXTEND * ext)
    depths = means_c[..., 2]
    valid = ok.squeeze(-1) & (depths.detach() > near_plane) & (depths.detach() < far_plane)
    valid = valid & ~((radius[..., 0] <= radius_clip) & (radius[..., 1] <= radius_clip))
    m = means2d.detach()
    valid = valid & (m[..., 0] + radius[..., 0] > 0) & (m[..., 0] - radius[..., 0] < width) \
        & (m[..., 1] + radius[..., 1] > 0) & (m[..., 1] - radius[..., 1] < height)
    radii = torch.where(valid[..., None], radius, torch.zeros_like(radius)).to(torch.int32)
    return radii, means2d, depths, M, normals


def _raster2d_common(means2d, ray_transforms, colors, opacities, normals, isect_offsets):
    off = _np(isect_offsets, np.int32)
    I, th, tw = int(np.prod(off.shape[:-2])), off.shape[-2], off.shape[-1]
    cdim = colors.shape[-1]
    return (off, I, th, tw, cdim, _np(means2d, np.float32).reshape(-1, 2),
            _np(ray_transforms, np.float32).reshape(-1, 9), _np(colors, np.float32).reshape(-1, cdim),
            _np(opacities, np.float32).reshape(-1), _np(normals, np.float32).reshape(-1, 3))


def rasterize_to_pixels_2dgs(means2d, ray_transforms, colors, opacities, normals, image_width, image_height, tile_size,
                             isect_offsets, flatten_ids, backgrounds=None, masks=None, distloss=False):
    """gsplat.rasterize_to_pixels_2dgs forward (RasterizeToPixels2DGSSerialBatchFwd.cu:43-465). Returns
    (render_colors, render_alphas, render_normals, render_distort, render_median, last_ids, median_ids)."""
    off, I, th, tw, cdim, m2, rt, cl, op, nr = _raster2d_common(means2d, ray_transforms, colors, opacities, normals,
                                                                isect_offsets)
    fl = _np(flatten_ids, np.int32)
    bg = None if backgrounds is None else _np(backgrounds, np.float32).reshape(I, cdim)
    mk = None if masks is None else _np(masks, np.uint8).reshape(I, th, tw)
    H, W = image_height, image_width
    rc = np.zeros((I, H, W, cdim), np.float32)
    ra = np.zeros((I, H, W, 1), np.float32)
    rn = np.zeros((I, H, W, 3), np.float32)
    rd = np.zeros((I, H, W, 1), np.float32)
    rm = np.zeros((I, H, W, 1), np.float32)
    li = np.zeros((I, H, W), np.int32)
    mi = np.zeros((I, H, W), np.int32)
    u32 = ctypes.c_uint32
    lib().gso_raster2d_fwd(_p(m2), _p(rt), _p(cl), _p(op), _p(nr), _p(bg), _p(mk), _p(off), _p(fl), u32(I),
                           u32(fl.shape[0]), u32(cdim), u32(W), u32(H), u32(tile_size), u32(tw), u32(th),
                           ctypes.c_int(int(distloss)), _p(rc), _p(ra), _p(rn), _p(rd), _p(rm), _p(li), _p(mi))
    return tuple(torch.from_numpy(x) for x in (rc, ra, rn, rd, rm, li, mi))


def rasterize_to_pixels_2dgs_bwd(means2d, ray_transforms, colors, opacities, normals, image_width, image_height,
                                 tile_size, isect_offsets, flatten_ids, render_colors, render_alphas, last_ids,
                                 median_ids, v_render_colors, v_render_alphas, v_render_normals, v_render_distort,
                                 v_render_median, backgrounds=None, masks=None, absgrad=False):
    """gsplat rasterize_to_pixels_2dgs backward (RasterizeToPixels2DGSSerialBatchBwd.cu:41-700). v_render_distort
    None = distloss off. Returns dict of float64 numpy arrays."""
    off, I, th, tw, cdim, m2, rt, cl, op, nr = _raster2d_common(means2d, ray_transforms, colors, opacities, normals,
                                                                isect_offsets)
    fl = _np(flatten_ids, np.int32)
    rows = m2.shape[0]
    H, W = image_height, image_width
    bg = None if backgrounds is None else _np(backgrounds, np.float32).reshape(I, cdim)
    mk = None if masks is None else _np(masks, np.uint8).reshape(I, th, tw)
    rc = _np(render_colors, np.float32).reshape(I, H, W, cdim)
    ra = _np(render_alphas, np.float32).reshape(I, H, W)
    li = _np(last_ids, np.int32).reshape(I, H, W)
    mi = _np(median_ids, np.int32).reshape(I, H, W)
    vc = _np(v_render_colors, np.float32).reshape(I, H, W, cdim)
    va = _np(v_render_alphas, np.float32).reshape(I, H, W)
    vn = _np(v_render_normals, np.float32).reshape(I, H, W, 3)
    vd = None if v_render_distort is None else _np(v_render_distort, np.float32).reshape(I, H, W)
    vm = _np(v_render_median, np.float32).reshape(I, H, W)
    z = lambda *s: np.zeros(s, np.float64)
    out = {"v_means2d": z(rows, 2), "v_ray_transforms": z(rows, 9), "v_colors": z(rows, cdim), "v_opacities": z(rows),
           "v_normals": z(rows, 3), "v_densify": z(rows, 2)}
    v_abs = z(rows, 2) if absgrad else None
    u32 = ctypes.c_uint32
    lib().gso_raster2d_bwd(_p(m2), _p(rt), _p(cl), _p(op), _p(nr), _p(bg), _p(mk), _p(off), _p(fl), _p(rc), _p(ra),
                           _p(li), _p(mi), _p(vc), _p(va), _p(vn), _p(vd), _p(vm), u32(I), u32(fl.shape[0]), u32(cdim),
                           u32(W), u32(H), u32(tile_size), u32(tw), u32(th), ctypes.c_int64(rows), _p(v_abs),
                           _p(out["v_means2d"]), _p(out["v_ray_transforms"]), _p(out["v_colors"]),
                           _p(out["v_opacities"]), _p(out["v_normals"]), _p(out["v_densify"]))
    if absgrad:
        out["v_means2d_abs"] = v_abs
    if backgrounds is not None:
        out["v_backgrounds"] = (vc.astype(np.float64) * (1.0 - ra.astype(np.float64))[..., None]).sum(axis=(1, 2))
    return out


def rasterize_to_indices_2dgs(means2d, ray_transforms, opacities, image_width, image_height, tile_size, isect_offsets,
                              flatten_ids):
    """(gaussian_ids, pixel_ids, image_ids) of the contributing pairs — input of the reference's accumulate_2dgs."""
    off = _np(isect_offsets, np.int32)
    I, th, tw = int(np.prod(off.shape[:-2])), off.shape[-2], off.shape[-1]
    m2 = _np(means2d, np.float32)
    n_per = m2.shape[-2]
    m2 = m2.reshape(-1, 2)
    rt = _np(ray_transforms, np.float32).reshape(-1, 9)
    op = _np(opacities, np.float32).reshape(-1)
    fl = _np(flatten_ids, np.int32)
    u32 = ctypes.c_uint32
    args = [_p(m2), _p(rt), _p(op), _p(off), _p(fl), u32(I), u32(fl.shape[0]), u32(n_per), u32(image_width),
            u32(image_height), u32(tile_size), u32(tw), u32(th)]
    n = lib().gso_raster2d_indices(*args, None, None, None)
    g, p, i = (np.zeros(n, np.int64) for _ in range(3))
    lib().gso_raster2d_indices(*args, _p(g), _p(p), _p(i))
    return torch.from_numpy(g), torch.from_numpy(p), torch.from_numpy(i)


# --------------------------------------------------------------------------------------------------
# training-step ops — torch-CPU restatements, TEST INFRASTRUCTURE ONLY
# --------------------------------------------------------------------------------------------------
def adam_step(param, grad, exp_avg, exp_avg_sq, valid, lr, b1, b2, eps):
    """gsplat::adam (gsplat/cuda/csrc/AdamCUDA.cu:34-75): masked Adam step WITHOUT bias correction. Returns new
    (param, exp_avg, exp_avg_sq); rows with valid == False are returned unchanged."""
    m = b1 * exp_avg + (1.0 - b1) * grad
    v = b2 * exp_avg_sq + (1.0 - b2) * grad * grad
    p = param - lr * m / (torch.sqrt(v) + eps)
    if valid is not None:
        sel = valid.reshape((-1,) + (1,) * (param.dim() - 1)).expand_as(param)
        p, m, v = torch.where(sel, p, param), torch.where(sel, m, exp_avg), torch.where(sel, v, exp_avg_sq)
    return p, m, v


def relocation(opacities, scales, ratios, binoms, min_opacity=0.0):
    """gsplat::relocation (gsplat/cuda/csrc/RelocationCUDA.cu:34-80), MCMC Eq. 9, evaluated in float64."""
    import math as _m

    n_max = binoms.shape[0]
    o = opacities.double()
    eps32 = float(torch.finfo(torch.float32).eps)
    new_o = (1.0 - (1.0 - o) ** (1.0 / ratios.double())).clamp(min_opacity, 1.0 - eps32)
    denom = torch.zeros_like(o)
    for idx in range(o.shape[0]):
        s = 0.0
        for i in range(1, int(ratios[idx]) + 1):
            for k in range(i):
                s += float(binoms[i - 1, k]) * ((-1.0) ** k / _m.sqrt(k + 1)) * float(new_o[idx]) ** (k + 1)
        denom[idx] = s
    coeff = o / denom
    return new_o.float(), (coeff[:, None] * scales.double()).float()


def mcmc_perturb(positions, quats, scales_log, opacities_logit, noise, noise_scale, t=0.005, k=100.0):
    """gsplat::mcmc_perturb_positions (gsplat/cuda/csrc/MCMCPerturbCUDA.cu:24-58; torch restatement
    tests/test_mcmc_perturb.py:22-45 and gsplat/strategy/ops.py:493-511)."""
    covars, _ = quat_scale_to_covar_preci(quats, torch.exp(scales_log), compute_covar=True, compute_preci=False)
    w = torch.sigmoid(-float(k) * (torch.sigmoid(opacities_logit) - float(t))) * noise_scale
    return positions + torch.einsum("bij,bj->bi", covars, noise * w[:, None])


# --------------------------------------------------------------------------------------------------
# query rasterizers — built on rasterize_to_indices; TEST INFRASTRUCTURE ONLY
# --------------------------------------------------------------------------------------------------
def raster_contributions(means2d, conics, opacities, image_width, image_height, tile_size, isect_offsets, flatten_ids):
    """Per pixel, front to back: the local Gaussian ids that contribute and their radiance weights alpha * T
    (reference RasterizeContributingCommon.cuh:28-198 walk; thresholds of RasterizeToPixels3DGSDevice.cuh:44-56).
    Returns (lists_of_ids, lists_of_weights, alphas [I,H,W]) with one Python list per pixel (small scenes only)."""
    gids, pids, iids = rasterize_to_indices(means2d, conics, opacities, image_width, image_height, tile_size, isect_offsets,
                                            flatten_ids)
    I = int(np.prod(isect_offsets.shape[:-2]))
    n_per = means2d.shape[-2]
    m2 = means2d.reshape(I, n_per, 2).double()
    cn = conics.reshape(I, n_per, 3).double()
    op = opacities.reshape(I, n_per).double()
    px = (pids % image_width).double() + 0.5
    py = (pids // image_width).double() + 0.5
    dx, dy = m2[iids, gids, 0] - px, m2[iids, gids, 1] - py
    c = cn[iids, gids]
    sigma = 0.5 * (c[:, 0] * dx * dx + c[:, 2] * dy * dy) + c[:, 1] * dx * dy
    alpha = torch.clamp(op[iids, gids] * torch.exp(-sigma), max=0.99)
    P = image_width * image_height
    key = (iids * P + pids).numpy()
    ids = [[] for _ in range(I * P)]
    wts = [[] for _ in range(I * P)]
    T = np.ones(I * P)
    a_np, g_np = alpha.numpy(), gids.numpy()
    for j in range(key.shape[0]):  # entries of one pixel are contiguous and in front-to-back order
        k = key[j]
        ids[k].append(int(g_np[j]))
        wts[k].append(float(a_np[j] * T[k]))
        T[k] *= 1.0 - a_np[j]
    return ids, wts, torch.from_numpy(1.0 - T).float().reshape(I, image_height, image_width)


def top_contributions(ids, wts, K):
    """Top-K selection rule of RasterizeTopContributingGaussianIds.cu:96-131 applied to per-pixel lists."""
    out_i = np.full((len(ids), K), -1, dtype=np.int64)
    out_w = np.zeros((len(ids), K), dtype=np.float64)
    for p, (li, lw) in enumerate(zip(ids, wts)):
        sw, sd, si = [0.0] * K, [2 ** 32 - 1] * K, [-1] * K
        for d, (g, w) in enumerate(zip(li, lw)):
            kmin = min(range(K), key=lambda k: (sw[k], k))
            if w > sw[kmin]:
                sw[kmin], sd[kmin], si[kmin] = w, d, g
        order = sorted(range(K), key=lambda k: sd[k])  # stable: unused slots (depth 2^32-1) keep their relative order
        out_i[p] = [si[k] for k in order]
        out_w[p] = [sw[k] for k in order]
    return out_i, out_w


# ---- sparse pixel sets -----------------------------------------------------------------------------------------------
def sparse_tile_layout(pixels, image_ids, n_images: int, tile_size: int, tile_width: int, tile_height: int):
    """gsplat.build_sparse_tile_layout (Intersect.cpp:696-794; key / bitmask kernels in SparseTileLayout.cu; contract in
    _wrapper.py:1433-1493), restated with python loops. Returns numpy (active_tiles int32 [AT], active_tile_mask bool
    [I, th, tw], tile_pixel_mask uint64 [AT, words], tile_pixel_cumsum int64 [AT] inclusive ([1] zero when empty),
    pixel_map int64 [P])."""
    px = _np(pixels, np.int64).reshape(-1, 2)
    im = _np(image_ids, np.int64).reshape(-1)
    P, n_tiles, words = px.shape[0], tile_width * tile_height, (tile_size * tile_size + 63) // 64
    if P == 0 or n_images == 0:
        return (np.zeros(0, np.int32), np.zeros((n_images, tile_height, tile_width), bool), np.zeros((0, words), np.uint64),
                np.zeros(1, np.int64), np.zeros(0, np.int64))
    entries = []
    for p in range(P):
        r, c = int(px[p, 0]), int(px[p, 1])
        tile = int(im[p]) * n_tiles + (r // tile_size) * tile_width + (c // tile_size)
        entries.append((tile, (r % tile_size) * tile_size + (c % tile_size), p))
    entries.sort()  # (tile id, raster position inside the tile); keys are unique -> deterministic
    pixel_map = np.array([e[2] for e in entries], np.int64)
    active = sorted({e[0] for e in entries})
    slot = {t: i for i, t in enumerate(active)}
    mask = np.zeros((len(active), words), np.uint64)
    counts = np.zeros(len(active), np.int64)
    for tile, pos, _p in entries:
        mask[slot[tile], pos >> 6] |= np.uint64(1) << np.uint64(pos & 63)
        counts[slot[tile]] += 1
    tmask = np.zeros(n_images * n_tiles, bool)
    tmask[active] = True
    return (np.array(active, np.int32), tmask.reshape(n_images, tile_height, tile_width), mask, np.cumsum(counts), pixel_map)


def isect_tiles_sparse(means2d, radii, depths, tile_mask, active_tiles, n_images: int, tile_size: int, tile_width: int,
                       tile_height: int, image_ids=None):
    """gsplat.isect_tiles_sparse (Intersect.cpp:563-690): the AABB enumeration of isect_tiles restricted to the tiles
    flagged in tile_mask, sorted by (image, tile, depth), with offsets compacted to the active tiles + a sentinel."""
    _tpg, ids, fl = isect_tiles(means2d, radii, depths, tile_size, tile_width, tile_height, sort=True, image_ids=image_ids,
                                n_images=n_images)
    ids, fl = ids.numpy(), fl.numpy()
    n_tiles, tile_bits = tile_width * tile_height, bits_for_count(tile_width * tile_height)
    hi = (ids >> 32).astype(np.int64)  # image << tile_bits | tile
    tile_of = (hi >> tile_bits) * n_tiles + (hi & ((1 << tile_bits) - 1))
    keep = _np(tile_mask, bool).reshape(-1)[tile_of]
    ids, fl, tile_of = ids[keep], fl[keep], tile_of[keep]
    act = _np(active_tiles, np.int64)
    offsets = np.concatenate([np.searchsorted(tile_of, act, side="left"), [len(fl)]]).astype(np.int32)
    return torch.from_numpy(offsets), torch.from_numpy(fl)
