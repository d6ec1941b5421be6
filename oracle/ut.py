"""CPU oracle for the Unscented-Transform projection of 3DGUT (TEST INFRASTRUCTURE ONLY; nothing under gsplat_amd/ may
import it).

Restates the behaviour of ``gsplat::projection_ut_3dgs_fused`` (reference kernel
``gsplat/cuda/csrc/ProjectionUT3DGSFused.cu``; the reference's own torch statement of it is
``gsplat/cuda/_torch_impl_ut.py:69-644`` with the camera models of ``gsplat/cuda/_torch_cameras.py``) for the camera
models built so far: perfect pinhole (``_torch_cameras.py:696-757``), OpenCV pinhole with radial / tangential / thin-prism
distortion (``:927-1086``), orthographic (``:793-848``) and OpenCV fisheye (``:1335-1697``: odd 9th-degree polynomial in
the ray angle, clamped at the angle where the polynomial stops being monotonic) and f-theta (``:1786-2056``: pixel distance
as a degree-5 polynomial of the ray angle, or the 3-step Newton inverse of the angle-of-pixel-distance polynomial; affine
(c, d, e) sensor map; principal point shifted by half a pixel), global shutter. Lidar, rolling shutter and the windshield
model are not restated yet. (The product kernel covers pinhole / OpenCV pinhole / ortho / fisheye; f-theta is oracle-only.)

Pinned: ``oracle/pin_ut_against_reference.py`` runs the reference's ``_fully_fused_projection_with_ut`` on the CPU — its
parameter records (``torch.classes.gsplat.UnscentedTransformParameters``) come from this backend's
``libgsplat_amd_torch.so`` installed as ``gsplat.csrc`` — and writes ``tests/golden/ut_ref.npz``;
``tests/test_oracle_ut.py`` checks this module against those vectors.

The algorithm, per (camera, Gaussian):
  1. seven sigma points: the mean and mean +- sqrt(3 + lambda) * scale_i * R[:, i], lambda = alpha^2 (3 + kappa) - 3
     (``_torch_impl_ut.py:111-170``);
  2. each is moved to the camera frame and projected by the camera model, which also says whether the point is valid
     (in front, distortion factor > 0.8, inside the image grown by ``in_image_margin_factor``);
  3. mean2d = sum w_m p_i, cov2d = sum w_c (p_i - mean2d)(p_i - mean2d)^T with the UT weights (``:69-108``); when
     ``require_all_sigma_points_valid`` the sums stop at the first invalid point (the kernel's early exit, ``:222-262``);
  4. blur + compensation, determinant / diagonal checks, conic = inverse, opacity-aware extent, eigenvalue-bounded radii,
     radius clip, image-bounds cull (``:470-644``). Invalid rows are zero.
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import torch
from torch import Tensor

ALPHA_THRESHOLD = 1.0 / 255.0
MIN_COMPENSATION = 0.005  # gsplat/cuda/_constants.py


def ut_weights(alpha: float, beta: float, kappa: float) -> Tuple[float, float, float, float]:
    """(centre weight of the mean, centre weight of the covariance, weight of the six others, sigma-point spread)."""
    lam = alpha * alpha * (3.0 + kappa) - 3.0
    w0 = lam / (3.0 + lam)
    return w0, w0 + (1.0 - alpha * alpha + beta), 1.0 / (2.0 * (3.0 + lam)), math.sqrt(3.0 + lam)


def _rotmat(quats: Tensor) -> Tensor:
    """Unit quaternion (w, x, y, z) -> rotation matrix; zero-length quaternions stay zero (culled by the caller)."""
    n = quats.norm(dim=-1, keepdim=True)
    q = quats / torch.where(n > 0, n, torch.ones_like(n))
    w, x, y, z = q.unbind(-1)
    return torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
        2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
        2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], dim=-1).reshape(q.shape[:-1] + (3, 3))


def _smallest_positive_root(a: Tensor, b: Tensor, c: Tensor) -> Tensor:
    """Smallest positive root of 1 + a x + b x^2 + c x^3 (inf when there is none), branch by branch like the reference
    (``_torch_cameras.py:1531-1621``): linear, quadratic, one real cubic root (Cardano), three real roots (trigonometric)."""
    inf = torch.full_like(a, float("inf"))
    lin = torch.where(a >= 0.0, inf, -1.0 / a)
    dq = a * a - 4.0 * b
    dt = torch.sqrt(dq) - a
    quad = torch.where(dt > 0.0, 2.0 / dt, inf)
    boc = b / c
    t1 = (9.0 * a * boc - 2.0 * b * boc * boc - 27.0) / c
    t2 = 3.0 * a / c - boc * boc
    dc = t1 * t1 + 4.0 * t2 * t2 * t2
    h = (torch.sqrt(dc) + t1) / 2.0
    cr = torch.sign(h) * torch.abs(h) ** (1.0 / 3.0)
    one = torch.where(cr != 0, (cr - t2 / cr - boc) / 3.0, inf)
    one = torch.where(one > 0.0, one, inf)
    th = torch.atan2(torch.sqrt(-dc), t1) / 3.0
    t3 = 2.0 * torch.sqrt(-t2)
    three = inf
    for i in (-1, 0, 1):
        r = (t3 * torch.cos(th + i * (2.0 * math.pi / 3.0)) - boc) / 3.0
        three = torch.minimum(three, torch.where(r > 0.0, r, inf))
    c0, b0 = c.abs() < 1e-10, b.abs() < 1e-10
    out = torch.where(dc < 0.0, three, inf)
    out = torch.where(dc >= 0.0, one, out)
    out = torch.where(c0, torch.where(dq >= 0.0, quad, inf), out)
    return torch.where(c0 & b0, lin, out)


def fisheye_max_angle(k: Tensor, fx: Tensor, fy: Tensor, cx: Tensor, cy: Tensor, width: int, height: int) -> Tensor:
    """Largest ray angle the OpenCV fisheye model projects (``_torch_cameras.py:1344-1521``): where the derivative of
    theta (1 + k1 theta^2 + k2 theta^4 + k3 theta^6 + k4 theta^8) first vanishes - a cubic in theta^2 when k4 = 0, Newton
    from 1.57 otherwise (20 steps, |step| < 1e-6 = converged) - and never beyond the image corner. k [..., 4]."""
    k1, k2, k3, k4 = k.unbind(-1)
    cubic = torch.sqrt(_smallest_positive_root(3.0 * k1, 5.0 * k2, 7.0 * k3))
    x = torch.full_like(k1, 1.57)
    done = torch.zeros_like(k1, dtype=torch.bool)
    for _ in range(20):
        x2 = x * x
        f = 1.0 + x2 * (3.0 * k1 + x2 * (5.0 * k2 + x2 * (7.0 * k3 + x2 * 9.0 * k4)))
        df = x * (6.0 * k1 + x2 * (20.0 * k2 + x2 * (42.0 * k3 + x2 * 72.0 * k4)))
        dx = f / df
        x = torch.where(done, x, x - dx)
        done = done | (dx.abs() < 1e-6)
    newton = torch.where(done & (x > 0.0), x, torch.full_like(x, float("inf")))
    ang = torch.where(k4.abs() < 1e-10, cubic, newton)
    rx, ry = torch.maximum(width - cx, cx), torch.maximum(height - cy, cy)
    rmax = torch.sqrt(rx * rx + ry * ry)
    return torch.minimum(ang, torch.maximum(rmax / fx, rmax / fy))


def _polyval(coeffs, x: Tensor) -> Tensor:
    """c0 + c1 x + ... by Horner (coeffs: python floats, lowest degree first)."""
    y = torch.full_like(x, float(coeffs[-1]))
    for c in reversed(coeffs[:-1]):
        y = y * x + float(c)
    return y


def project_points(p: Tensor, camera_model: str, fx: Tensor, fy: Tensor, cx: Tensor, cy: Tensor, width: int, height: int,
                   margin: float, radial: Optional[Tensor], tangential: Optional[Tensor], thin_prism: Optional[Tensor],
                   ftheta: Optional[dict] = None):
    """Camera-frame points p [..., C, M, 3] -> (pixels [..., C, M, 2], valid [..., C, M]). Per-camera parameters are
    [..., C, 1] (broadcast over M)."""
    front = p[..., 2] > 0.0
    if camera_model == "ftheta":
        ax_, ay_ = p[..., 0].abs(), p[..., 1].abs()
        big, small = torch.maximum(ax_, ay_), torch.minimum(ax_, ay_)
        ratio = torch.where(big > 0.0, small / big, torch.zeros_like(big))
        rxy = torch.where(big > 0.0, big * torch.sqrt(1.0 + ratio * ratio), torch.zeros_like(big))
        rxy = torch.where(rxy <= 0.0, torch.full_like(rxy, torch.finfo(p.dtype).eps), rxy)
        th_full = torch.atan2(rxy, p[..., 2])
        th = th_full.clamp(max=float(ftheta["max_angle"]))
        a2p, p2a = list(ftheta["angle_to_pixeldist_poly"]), list(ftheta["pixeldist_to_angle_poly"])
        if int(ftheta["reference_poly"]) == 0:  # the angle-of-distance polynomial is the calibrated one: invert it
            dist = _polyval(a2p, th)
            slope = [k * p2a[k] for k in range(1, 6)]
            done = torch.zeros_like(dist, dtype=torch.bool)
            for _ in range(3):
                step = (_polyval(p2a, dist) - th) / _polyval(slope, dist)
                dist = torch.where(done, dist, dist - step)
                done = done | (step.abs() < 1e-6)
        else:
            dist = _polyval(a2p, th)
        ix, iy = dist * p[..., 0] / rxy, dist * p[..., 1] / rxy
        c_, d_, e_ = (float(v) for v in ftheta["linear_cde"])
        px, py = c_ * ix + d_ * iy + (cx + 0.5), e_ * ix + iy + (cy + 0.5)
        inb = (px >= -width * margin) & (px < width + width * margin) & (py >= -height * margin) & (py < height + height * margin)
        return torch.stack([px, py], -1), (th_full < float(ftheta["max_angle"])) & inb  # no "in front" test for this model
    if camera_model == "fisheye":
        k = radial if radial is not None else torch.zeros(fx.shape[:-1] + (4,), dtype=p.dtype)
        amax = fisheye_max_angle(k, fx[..., 0], fy[..., 0], cx[..., 0], cy[..., 0], width, height)[..., None]
        ax_, ay_ = p[..., 0].abs(), p[..., 1].abs()
        big, small = torch.maximum(ax_, ay_), torch.minimum(ax_, ay_)
        ratio = torch.where(big > 0.0, small / big, torch.zeros_like(big))
        rxy = torch.where(big > 0.0, big * torch.sqrt(1.0 + ratio * ratio), torch.zeros_like(big))
        rxy = torch.where(rxy <= 0.0, torch.full_like(rxy, torch.finfo(p.dtype).eps), rxy)
        th_full = torch.atan2(rxy, p[..., 2])
        th = torch.minimum(th_full, amax)
        t2 = th * th
        poly = th * (1.0 + t2 * (k[..., 0:1] + t2 * (k[..., 1:2] + t2 * (k[..., 2:3] + t2 * k[..., 3:4]))))
        delta = poly / rxy
        px, py = delta * p[..., 0] * fx + cx, delta * p[..., 1] * fy + cy
        inb = (px >= -width * margin) & (px < width + width * margin) & (py >= -height * margin) & (py < height + height * margin)
        return torch.stack([px, py], -1), front & (delta > 0.0) & (th_full < amax) & inb
    if camera_model == "ortho":
        u, v = p[..., 0], p[..., 1]
        ok = front
    else:
        u, v = p[..., 0] / p[..., 2], p[..., 1] / p[..., 2]
        ok = front
        if radial is not None or tangential is not None or thin_prism is not None:
            z = torch.zeros_like(fx)
            k = [z] * 6 if radial is None else [radial[..., i:i + 1] if i < radial.shape[-1] else z for i in range(6)]
            p1, p2 = (z, z) if tangential is None else (tangential[..., 0:1], tangential[..., 1:2])
            s = [z] * 4 if thin_prism is None else [thin_prism[..., i:i + 1] for i in range(4)]
            uu, vv = u * u, v * v
            r2 = uu + vv
            a1, a2, a3 = 2.0 * u * v, r2 + 2.0 * uu, r2 + 2.0 * vv
            icd = (1.0 + r2 * (k[0] + r2 * (k[1] + r2 * k[2]))) / (1.0 + r2 * (k[3] + r2 * (k[4] + r2 * k[5])))
            du = p1 * a1 + p2 * a2 + r2 * (s[0] + r2 * s[1])
            dv = p1 * a3 + p2 * a1 + r2 * (s[2] + r2 * s[3])
            u, v = icd * u + du, icd * v + dv
            ok = ok & (icd > 0.8)  # the distorted model does NOT zero points behind the camera (_torch_cameras.py:1052-1086)
            px, py = u * fx + cx, v * fy + cy
            inb = (px >= -width * margin) & (px < width + width * margin) & (py >= -height * margin) & (py < height + height * margin)
            return torch.stack([px, py], -1), ok & inb
    px, py = u * fx + cx, v * fy + cy
    px, py = torch.where(front, px, torch.zeros_like(px)), torch.where(front, py, torch.zeros_like(py))
    inb = (px >= -width * margin) & (px < width + width * margin) & (py >= -height * margin) & (py < height + height * margin)
    return torch.stack([px, py], -1), ok & inb


def fully_fused_projection_with_ut(
    means: Tensor, quats: Tensor, scales: Tensor, opacities: Optional[Tensor], viewmats: Tensor, Ks: Tensor, width: int,
    height: int, eps2d: float = 0.3, near_plane: float = 0.01, far_plane: float = 1e10, radius_clip: float = 0.0,
    calc_compensations: bool = False, camera_model: str = "pinhole", alpha: float = 0.1, beta: float = 2.0,
    kappa: float = 0.0, in_image_margin_factor: float = 0.1, require_all_sigma_points_valid: bool = False,
    radial_coeffs: Optional[Tensor] = None, tangential_coeffs: Optional[Tensor] = None,
    thin_prism_coeffs: Optional[Tensor] = None, ftheta: Optional[dict] = None,
):
    """means [..., N, 3], quats [..., N, 4], scales [..., N, 3], opacities [..., N] or None, viewmats [..., C, 4, 4],
    Ks [..., C, 3, 3] -> radii int32 [..., C, N, 2], means2d [..., C, N, 2], depths [..., C, N], conics [..., C, N, 3],
    compensations [..., C, N] or None."""
    if camera_model not in ("pinhole", "ortho", "fisheye", "ftheta"):
        raise NotImplementedError(f"oracle.ut: camera model '{camera_model}' is not restated yet")
    dt = means.dtype
    w_m0, w_c0, w_i, spread = ut_weights(alpha, beta, kappa)
    R = _rotmat(quats)  # [..., N, 3, 3], columns = principal axes
    axes = (spread * R * scales[..., None, :]).transpose(-1, -2)  # rows = the three offsets
    sigma = torch.cat([means[..., None, :], means[..., None, :] + axes, means[..., None, :] - axes], dim=-2)  # [..., N, 7, 3]

    Rc, tc = viewmats[..., :3, :3], viewmats[..., :3, 3]
    cam = torch.einsum("...cij,...nkj->...cnki", Rc, sigma) + tc[..., None, None, :]  # [..., C, N, 7, 3]
    lead = cam.shape[:-3]
    N = means.shape[-2]
    par = lambda t: t[..., None]  # noqa: E731  [..., C] -> [..., C, 1]
    pts, ok = project_points(cam.reshape(lead + (N * 7, 3)), camera_model, par(Ks[..., 0, 0]), par(Ks[..., 1, 1]),
                             par(Ks[..., 0, 2]), par(Ks[..., 1, 2]), width, height, in_image_margin_factor,
                             radial_coeffs, tangential_coeffs, thin_prism_coeffs, ftheta)
    pts, ok = pts.reshape(lead + (N, 7, 2)), ok.reshape(lead + (N, 7))

    wm = torch.tensor([w_m0] + [w_i] * 6, dtype=dt)
    wc = torch.tensor([w_c0] + [w_i] * 6, dtype=dt)
    if require_all_sigma_points_valid:
        upto = torch.cumprod(ok.to(dt), dim=-1)  # 1 until the first invalid point
        valid = upto[..., -1] > 0
        wm, wc = wm * upto, wc * upto
    else:
        valid = ok.any(dim=-1)
    mean2d = (wm[..., None] * pts).sum(dim=-2)
    d = pts - mean2d[..., None, :]
    cxx, cxy, cyy = (wc * d[..., 0] * d[..., 0]).sum(-1), (wc * d[..., 0] * d[..., 1]).sum(-1), (wc * d[..., 1] * d[..., 1]).sum(-1)

    centre = cam[..., 0, :]  # sigma point 0 is the mean
    z = centre[..., 2]
    eps = torch.finfo(dt).eps
    alive = ((quats * quats).sum(-1) > eps) & (scales > eps).all(-1)  # [..., N]
    valid = valid & (z >= near_plane) & (z <= far_plane) & alive[..., None, :]

    det0 = cxx * cyy - cxy * cxy
    cxx, cyy = cxx + eps2d, cyy + eps2d
    det = cxx * cyy - cxy * cxy
    comp = torch.sqrt(torch.clamp(det0 / det, min=MIN_COMPENSATION * MIN_COMPENSATION))
    valid = valid & (det > 0.0) & (cxx > 0.0) & (cyy > 0.0)

    ixx, iyy = cxx + 1e-6, cyy + 1e-6  # the reference inverts cov + 1e-6 I (_torch_impl_ut.py:526-528)
    idet = ixx * iyy - cxy * cxy
    conics = torch.stack([iyy / idet, -cxy / idet, ixx / idet], dim=-1)

    extend = torch.full_like(det, 3.33)
    if opacities is not None:
        op = opacities[..., None, :] * comp
        valid = valid & (op >= ALPHA_THRESHOLD)
        extend = torch.minimum(extend, torch.sqrt(2.0 * torch.log(torch.clamp(op / ALPHA_THRESHOLD, min=1.0))))
    b = 0.5 * (cxx + cyy)
    lam_max = b + torch.sqrt(torch.clamp(b * b - det, min=0.01))
    r_eig = extend * torch.sqrt(lam_max.clamp(min=0.0))
    rx = torch.ceil(torch.minimum(extend * torch.sqrt(cxx.clamp(min=0.0)), r_eig))
    ry = torch.ceil(torch.minimum(extend * torch.sqrt(cyy.clamp(min=0.0)), r_eig))
    valid = valid & (torch.maximum(rx, ry) > radius_clip)
    valid = valid & (mean2d[..., 0] + rx > 0) & (mean2d[..., 0] - rx < width) & (mean2d[..., 1] + ry > 0) & (mean2d[..., 1] - ry < height)

    zero = torch.zeros_like(z)
    radii = torch.where(valid[..., None], torch.stack([rx, ry], -1), zero[..., None]).to(torch.int32)
    means2d = torch.where(valid[..., None], mean2d, zero[..., None])
    depths = torch.where(valid, z, zero)
    conics = torch.where(valid[..., None], conics, zero[..., None])
    comps = torch.where(valid, comp, zero) if calc_compensations else None
    return radii, means2d, depths, conics, comps
