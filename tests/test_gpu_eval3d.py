"""GPU: from-world (eval3d) compositing, forward (gsx_raster_world_fwd / gsplat::rasterize_to_pixels_from_world_3dgs)
against the golden vectors of the reference's torch implementation and against the CPU oracle on a larger scene."""
import math
import os

import numpy as np
import pytest
import torch

from _util import assert_close_ratio, make_scene

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = "cuda"


@pytest.fixture(scope="module")
def G():
    if not torch.cuda.is_available():
        pytest.skip("needs a ROCm GPU")
    import gsplat_amd
    from gsplat_amd import _ops  # noqa: F401  (defines torch.ops.gsplat.*; the package itself loads lazily)

    return gsplat_amd


@pytest.mark.parametrize("name", ["a", "b"])
def test_eval3d_forward_matches_reference_outputs(G, name):
    gold = dict(np.load(os.path.join(ROOT, "tests", "golden", "eval3d_ref.npz")))
    N, C, W, H, ts = (int(v) for v in gold[f"{name}.shape"])
    t = lambda k: torch.from_numpy(gold[f"{name}.{k}"]).to(DEV)  # noqa: E731
    bg = t("backgrounds") if f"{name}.backgrounds" in gold else None
    out = torch.ops.gsplat.rasterize_to_pixels_from_world_3dgs(
        t("means"), t("quats"), t("scales"), t("colors"), t("opacities"), bg, None, W, H, ts, t("viewmats"), None, t("Ks"), 0,
        torch.classes.gsplat.UnscentedTransformParameters(), 4, None, None, None, None,
        torch.classes.gsplat.FThetaCameraDistortionParameters(), None, None, t("isect_offsets"), t("flatten_ids"), False,
        False, False, 0, True)
    ren, alp, last = out[0].cpu(), out[1].cpu(), out[2].cpu()
    assert out[3] is None and out[4] is None
    # fp32 evaluation orders differ by ~1e-4 in alpha when 1 / scale is large (see oracle/eval3d.py); decision flips at the
    # 1/255 and 1e-4 thresholds are allowed on a handful of pixels, like for the classic compositing
    assert_close_ratio(ren, torch.from_numpy(gold[f"{name}.ref.render"]), 2e-3, 2e-4, max_bad_ratio=2e-3, name="render")
    assert_close_ratio(alp, torch.from_numpy(gold[f"{name}.ref.alpha"]), 2e-3, 2e-4, max_bad_ratio=2e-3, name="alpha")
    assert float((last == torch.from_numpy(gold[f"{name}.ref.last_ids"])).float().mean()) > 0.995
    # the wrapper with generated pinhole rays gives the same images as explicit rays
    rc2, ra2 = G.rasterize_to_pixels_eval3d(t("means"), t("quats"), t("scales"), t("colors"), t("opacities"), t("viewmats"),
                                            t("Ks"), W, H, ts, t("isect_offsets"), t("flatten_ids"), backgrounds=bg,
                                            rays=t("rays"))
    assert float((rc2.cpu() - ren).abs().max()) < 1e-4 and float((ra2.cpu() - alp).abs().max()) < 1e-4


def test_eval3d_forward_larger_scene_vs_oracle(G):
    """2000 Gaussians, 2 cameras, 128 x 96, tile lists from the product's own projection + intersection."""
    from oracle import eval3d as E

    sc, W, H = make_scene(N=2000, C=2, width=128, height=96, seed=12, scale_range=(0.05, 0.2))
    a = {k: v.to(DEV) for k, v in sc.items()}
    ts = 16
    tw, th = math.ceil(W / ts), math.ceil(H / ts)
    radii, means2d, depths, conics, _ = G.fully_fused_projection(a["means"], None, a["quats"], a["scales"], a["viewmats"],
                                                                 a["Ks"], W, H)
    _, isect_ids, flatten_ids = G.isect_tiles(means2d, radii * 2, depths, ts, tw, th)
    offsets = G.isect_offset_encode(isect_ids, 2, tw, th)
    colors = a["colors"][None].expand(2, -1, -1).contiguous()
    opac = a["opacities"][None].expand(2, -1).contiguous()
    bg = torch.tensor([[0.1, 0.2, 0.3], [0.5, 0.4, 0.3]], device=DEV)
    rc, ra = G.rasterize_to_pixels_eval3d(a["means"], a["quats"], a["scales"], colors, opac, a["viewmats"], a["Ks"], W, H,
                                          ts, offsets, flatten_ids, backgrounds=bg)
    rays = E.pinhole_rays(sc["viewmats"], sc["Ks"], W, H)
    ref_c, ref_a, _ = E.rasterize_to_pixels_eval3d(sc["means"], sc["quats"], sc["scales"], colors.cpu(), opac.cpu(), rays, W, H,
                                                   ts, offsets.cpu(), flatten_ids.cpu(), backgrounds=bg.cpu())
    assert float(ref_a.mean()) > 0.05  # the scene is not empty
    assert_close_ratio(rc.cpu(), ref_c, 2e-3, 2e-4, max_bad_ratio=2e-3, name="render")
    assert_close_ratio(ra.cpu(), ref_a, 2e-3, 2e-4, max_bad_ratio=2e-3, name="alpha")
    # the dispatcher op is differentiable too (autograd lives inside the op, like the reference's C++ autograd function)
    m = a["means"].clone().requires_grad_(True)
    out = torch.ops.gsplat.rasterize_to_pixels_from_world_3dgs(
        m, a["quats"], a["scales"], colors, opac, None, None, W, H, ts,
        a["viewmats"], None, a["Ks"], 0, torch.classes.gsplat.UnscentedTransformParameters(), 4, None, None, None, None,
        torch.classes.gsplat.FThetaCameraDistortionParameters(), None, None, offsets, flatten_ids, False, False, False,
        0, False)
    assert out[2] is None and out[0].requires_grad
    out[0].sum().backward()
    assert torch.isfinite(m.grad).all() and float(m.grad.abs().max()) > 0


def test_rasterization_with_eval3d_forward(G):
    """rasterization(with_eval3d=True) under no_grad: UT or EWA projection for the tile lists, from-world compositing for the
    image; close to the classic render (the two footprint models differ slightly), and loud when gradients are requested."""
    sc, W, H = make_scene(N=3000, C=2, width=144, height=96, seed=8, scale_range=(0.05, 0.15))
    a = {k: v.to(DEV) for k, v in sc.items()}
    args = (a["means"], a["quats"], a["scales"], a["opacities"], a["colors"], a["viewmats"], a["Ks"], W, H)
    with torch.no_grad():
        rc0, ra0, _ = G.rasterization(*args, packed=False)
        rc1, ra1, meta = G.rasterization(*args, packed=False, with_eval3d=True, with_ut=True, render_mode="RGB+ED")
        rc2, ra2, _ = G.rasterization(*args, packed=False, with_eval3d=True)
    assert meta["tile_size"] == 8 and rc1.shape == (2, H, W, 4) and torch.isfinite(rc1).all()
    assert float((rc1[..., :3] - rc0).abs().mean()) < 4e-2 and float((ra1 - ra0).abs().mean()) < 4e-2
    assert float((rc2 - rc0).abs().mean()) < 4e-2
    # training through the from-world path: gradients reach every leaf and are close to the classic path's
    names = ("means", "quats", "scales", "opacities", "colors")
    grads = {}
    for kw in (dict(), dict(with_eval3d=True)):
        leaves = {k: a[k].clone().requires_grad_(True) for k in names}
        rc, ra, _ = G.rasterization(*[leaves[k] for k in names], a["viewmats"], a["Ks"], W, H, packed=False, **kw)
        (rc.sum() + ra.sum()).backward()
        grads[bool(kw)] = {k: leaves[k].grad for k in names}
    for k in names:
        g0, g1 = grads[False][k].double().flatten(), grads[True][k].double().flatten()
        assert torch.isfinite(g1).all()
        cos = float(g0 @ g1 / (g0.norm() * g1.norm() + 1e-30))
        assert cos > 0.8, (k, cos)  # two footprint models (EWA vs along-ray response): close, not equal


@pytest.mark.parametrize("name", ["a", "b"])
def test_eval3d_backward_matches_reference_gradients(G, name):
    """gsx_raster_world_bwd against the gradients the reference's autograd gives (tests/golden/eval3d_ref.npz)."""
    gold = dict(np.load(os.path.join(ROOT, "tests", "golden", "eval3d_ref.npz")))
    N, C, W, H, ts = (int(v) for v in gold[f"{name}.shape"])
    t = lambda k: torch.from_numpy(gold[f"{name}.{k}"]).to(DEV)  # noqa: E731
    leaves = {k: t(k).clone().requires_grad_(True) for k in ("means", "quats", "scales", "colors", "opacities")}
    bg = t("backgrounds") if f"{name}.backgrounds" in gold else None
    ren, alp = G.rasterize_to_pixels_eval3d(leaves["means"], leaves["quats"], leaves["scales"], leaves["colors"],
                                            leaves["opacities"], t("viewmats"), t("Ks"), W, H, ts, t("isect_offsets"),
                                            t("flatten_ids"), backgrounds=bg, rays=t("rays"))
    ((ren * t("v_render")).sum() + (alp * t("v_alpha")).sum()).backward()
    from _util import assert_grad_close

    for k, leaf in leaves.items():
        assert_grad_close(leaf.grad.cpu(), torch.from_numpy(gold[f"{name}.ref.v_{k}"]), rel=5e-3, max_bad_ratio=1e-3, name=f"v_{k}")


@pytest.mark.parametrize("name", ["hit", "normals", "both"])
def test_eval3d_hit_distance_normals_and_sample_counts_match_reference(G, name):
    """gsx_raster_world_{fwd,bwd}_ex against the outputs AND autograd gradients of the reference's torch statement with
    use_hit_distance / return_normals (tests/golden/eval3d_extras_ref.npz, oracle/pin_eval3d_extras_against_reference.py): the
    last colour channel is every sample's hit distance |scale * d' hit_t|, the normals are the Gaussians' third axes turned to
    face the ray; the hit distance carries gradients to means, quaternions and - directly - scales, the normals to quaternions."""
    gold = dict(np.load(os.path.join(ROOT, "tests", "golden", "eval3d_extras_ref.npz")))
    N, C, W, H, ts, D, hit, nrm = (int(v) for v in gold[f"{name}.shape"])
    t = lambda k: torch.from_numpy(gold[f"{name}.{k}"]).to(DEV)  # noqa: E731
    leaves = {k: t(k).clone().requires_grad_(True) for k in ("means", "quats", "scales", "colors", "opacities")}
    bg = t("backgrounds").requires_grad_(True) if f"{name}.backgrounds" in gold else None
    rays = t("rays").requires_grad_(True)  # the rays and the backgrounds are differentiable inputs of the reference's op too
    ren, alp, last, cnt, nor = G.rasterize_to_pixels_eval3d_extra(
        leaves["means"], leaves["quats"], leaves["scales"], leaves["colors"], leaves["opacities"], t("viewmats"), t("Ks"), W, H, ts,
        t("isect_offsets"), t("flatten_ids"), backgrounds=bg, rays=rays, return_sample_counts=True,
        use_hit_distance=bool(hit), return_normals=bool(nrm))
    from _util import assert_close_ratio, assert_grad_close

    assert_close_ratio(ren.detach().cpu(), torch.from_numpy(gold[f"{name}.ref.render"]), 1e-4, 2e-5, max_bad_ratio=1e-3, name="render")
    assert_close_ratio(alp.detach().cpu(), torch.from_numpy(gold[f"{name}.ref.alpha"]), 1e-4, 2e-5, max_bad_ratio=1e-3, name="alpha")
    same = (cnt.cpu() == torch.from_numpy(gold[f"{name}.ref.sample_counts"])).float().mean()
    assert float(same) > 0.995, float(same)  # a sample within rounding of the 1/255 or 1e-4 test moves a pixel's count by one
    assert bool(((cnt > 0) == (last >= 0)).all())
    loss = (ren * t("v_render")).sum() + (alp * t("v_alpha")).sum()
    if nrm:
        assert_close_ratio(nor.detach().cpu(), torch.from_numpy(gold[f"{name}.ref.normals"]), 1e-4, 2e-5, max_bad_ratio=1e-3, name="normals")
        loss = loss + (nor * t("v_normals")).sum()
    else:
        assert nor is None
    loss.backward()
    for k, leaf in leaves.items():
        assert_grad_close(leaf.grad.cpu(), torch.from_numpy(gold[f"{name}.ref.v_{k}"]), rel=5e-3, max_bad_ratio=2e-3, name=f"{name} v_{k}")
    assert_grad_close(rays.grad.cpu(), torch.from_numpy(gold[f"{name}.ref.v_rays"]), rel=5e-3, max_bad_ratio=5e-3, name=f"{name} v_rays")
    if bg is not None:
        assert_grad_close(bg.grad.cpu(), torch.from_numpy(gold[f"{name}.ref.v_backgrounds"]), rel=2e-3, name=f"{name} v_backgrounds")


_RAY_CASES = ["pinhole_global", "pinhole_top_bottom", "pinhole_left_right", "pinhole_bottom_top", "pinhole_right_left",
              "opencv_global", "opencv_radial4_top_bottom", "opencv_strong", "ortho_global", "ortho_left_right",
              "fisheye_global", "fisheye_k4_bottom_top", "fisheye_plain", "ftheta_forward_global", "ftheta_backward_right_left"]


@pytest.mark.parametrize("name", _RAY_CASES)
def test_camera_rays_match_reference_camera_models(G, name):
    """gsx_camera_rays (the ray of every pixel centre when the from-world rasterizer is given no `rays`) against the reference's
    torch statement of its camera models - perfect and OpenCV-distorted pinhole (Newton undistortion), orthographic, OpenCV
    fisheye (Newton on the odd polynomial), f-theta (either calibrated polynomial) - under a global shutter and the four rolling
    ones (tests/golden/camera_rays_ref.npz, oracle/pin_camera_rays_against_reference.py)."""
    from gsplat_amd import _ops

    gold = dict(np.load(os.path.join(ROOT, "tests", "golden", "camera_rays_ref.npz")))
    rs, model, W, H = (int(v) for v in gold[f"{name}.meta"])
    t = lambda k: torch.from_numpy(gold[f"{name}.{k}"]).float().to(DEV) if f"{name}.{k}" in gold else None  # noqa: E731
    ft = None
    if f"{name}.ftheta" in gold:
        r = [float(v) for v in gold[f"{name}.ftheta"]]
        ft = torch.classes.gsplat.FThetaCameraDistortionParameters(int(r[0]), r[1:7], r[7:13], r[13], r[14:17])
    rays = _ops.camera_pixel_rays(t("viewmats"), t("viewmats_rs"), t("Ks"), W, H, model, rs, t("radial_coeffs"),
                                  t("tangential_coeffs"), t("thin_prism_coeffs"), ft).cpu()
    ref = torch.from_numpy(gold[f"{name}.ref.rays"])
    assert rays.shape == ref.shape
    # origins in scene units (a few units from the origin), directions unit vectors: fp32 rounding of two different evaluation
    # orders (the Newton iterations end at their 1e-6 step test)
    torch.testing.assert_close(rays[..., :3], ref[..., :3], rtol=0, atol=5e-6)
    torch.testing.assert_close(rays[..., 3:], ref[..., 3:], rtol=0, atol=5e-6)


def test_rasterization_eval3d_generates_rays_for_distorted_cameras(G):
    """rasterization(with_ut=True, with_eval3d=True) without `rays` through a distorted pinhole and a fisheye camera: the
    generated rays (gsx_camera_rays) render the same image as the same rays passed explicitly, and a distortion that is zero
    renders what the perfect pinhole renders."""
    from gsplat_amd import _ops

    sc, W, H = make_scene(N=2000, C=2, width=96, height=64, seed=21)
    a = {k: v.to(DEV) for k, v in sc.items()}
    args = (a["means"], a["quats"], a["scales"], a["opacities"], a["colors"], a["viewmats"], a["Ks"], W, H)
    rad = torch.tensor([0.08, -0.02, 0.004, 0.0, 0.0, 0.0], device=DEV).repeat(2, 1)
    for kw, model in ((dict(radial_coeffs=rad), 0), (dict(camera_model="fisheye", radial_coeffs=rad[:, :4] * 0.3), 2)):
        rc, ra, _ = G.rasterization(*args, packed=False, with_ut=True, with_eval3d=True, **kw)
        rays = _ops.camera_pixel_rays(a["viewmats"], None, a["Ks"], W, H, model, 4, kw["radial_coeffs"])
        rc2, ra2, _ = G.rasterization(*args, packed=False, with_ut=True, with_eval3d=True, rays=rays, **kw)
        assert torch.equal(rc, rc2) and torch.equal(ra, ra2)
        assert float(ra.max()) > 0.5 and bool(torch.isfinite(rc).all())
    rc0, ra0, _ = G.rasterization(*args, packed=False, with_ut=True, with_eval3d=True)
    rcz, raz, _ = G.rasterization(*args, packed=False, with_ut=True, with_eval3d=True, radial_coeffs=torch.zeros_like(rad))
    torch.testing.assert_close(rcz, rc0, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(raz, ra0, rtol=1e-4, atol=1e-4)


def _windshield(h, v, hi=None, vi=None):
    p = torch.classes.gsplat.BivariateWindshieldModelParameters()
    ident_h, ident_v = [0.0, 1.0, 0.0], [0.0, 0.0, 1.0]
    p.horizontal_poly, p.vertical_poly = torch.tensor(h, dtype=torch.float32), torch.tensor(v, dtype=torch.float32)
    p.horizontal_poly_inverse = torch.tensor(ident_h if hi is None else hi, dtype=torch.float32)
    p.vertical_poly_inverse = torch.tensor(ident_v if vi is None else vi, dtype=torch.float32)
    return p


@pytest.mark.parametrize("order", [0, 1, 2, 3, 4, 5])
def test_external_distortion_matches_reference_statement(G, order):
    """gsplat::eval_bivariate_poly / gsplat::distort_camera_rays (gsx_eval_bivariate_poly, gsx_distort_camera_rays) against the
    reference's Python statement of the bivariate windshield model for every polynomial order - the functions its own
    tests/test_external_distortion.py holds its CUDA kernels to (tests/golden/external_distortion_ref.npz)."""
    gold = dict(np.load(os.path.join(ROOT, "tests", "golden", "external_distortion_ref.npz")))
    h, v = gold[f"o{order}.h"].tolist(), gold[f"o{order}.v"].tolist()
    x, y = torch.from_numpy(gold[f"o{order}.x"]).to(DEV), torch.from_numpy(gold[f"o{order}.y"]).to(DEV)
    got = torch.ops.gsplat.eval_bivariate_poly(x, y, torch.tensor(h, dtype=torch.float32, device=DEV), order).cpu().double()
    torch.testing.assert_close(got, torch.from_numpy(gold[f"o{order}.ref.poly"]), rtol=1e-5, atol=1e-6)
    rays = torch.from_numpy(gold[f"o{order}.rays"]).to(DEV)
    hp, vp = torch.tensor(h, dtype=torch.float32, device=DEV), torch.tensor(v, dtype=torch.float32, device=DEV)
    ident = torch.tensor([0.0, 1.0, 0.0], device=DEV), torch.tensor([0.0, 0.0, 1.0], device=DEV)
    out = torch.ops.gsplat.distort_camera_rays(rays, hp, vp, ident[0], ident[1], 1, False).cpu().double()
    torch.testing.assert_close(out, torch.from_numpy(gold[f"o{order}.ref.distorted"]), rtol=0, atol=5e-6)
    # `inverse` applies the other pair: with the pairs swapped it is the same computation
    out_inv = torch.ops.gsplat.distort_camera_rays(rays, ident[0], ident[1], hp, vp, 1, True).cpu().double()
    assert torch.equal(out, out_inv)


def test_external_distortion_in_the_3dgut_kernels(G):
    """The windshield model inside the two 3DGUT kernels: (a) the identity polynomials change nothing (the reference's
    test_identity_distortion_matches_no_distortion / test_ortho_identity_distortion_matches_no_distortion); (b) a real distortion
    with its exact inverse pair keeps projection and ray generation consistent - the generated ray of the pixel a point projects
    to passes through the point (pinhole, fisheye); (c) rasterization(external_distortion_coeffs=...) renders."""
    from gsplat_amd import _ops

    sc, W, H = make_scene(N=1500, C=2, width=96, height=64, seed=23)
    a = {k: v.to(DEV) for k, v in sc.items()}
    args = (a["means"], a["quats"], a["scales"], a["opacities"], a["colors"], a["viewmats"], a["Ks"], W, H)
    ident = _windshield([0.0, 1.0, 0.0], [0.0, 0.0, 1.0])
    for kw in (dict(), dict(camera_model="ortho")):
        Ks = a["Ks"].clone()
        if kw:
            Ks[:, 0, 0] = Ks[:, 1, 1] = 20.0
        base = list(args)
        base[6] = Ks
        rc0, ra0, _ = G.rasterization(*base, packed=False, with_ut=True, with_eval3d=True, **kw)
        rc1, ra1, _ = G.rasterization(*base, packed=False, with_ut=True, with_eval3d=True, external_distortion_coeffs=ident, **kw)
        torch.testing.assert_close(rc1, rc0, rtol=1e-4, atol=1e-4)
        torch.testing.assert_close(ra1, ra0, rtol=1e-4, atol=1e-4)
    # (b) phi' = 1.05 phi + 0.01, theta' = 0.97 theta - 0.02 and its exact inverse
    fwd = _windshield([0.01, 1.05, 0.0], [-0.02, 0.0, 0.97], [-0.01 / 1.05, 1 / 1.05, 0.0], [0.02 / 0.97, 0.0, 1 / 0.97])
    for model, name in ((0, "pinhole"), (2, "fisheye")):
        radii, m2, depths, _, _ = G.fully_fused_projection_with_ut(
            a["means"], a["quats"], a["scales"] * 0.05, None, a["viewmats"], a["Ks"], W, H, camera_model=name,
            external_distortion_coeffs=fwd)
        rays = _ops.camera_pixel_rays(a["viewmats"], None, a["Ks"], W, H, model, 4, None, None, None, None, fwd)
        vis = (radii > 0).all(-1)
        assert int(vis.sum()) > 200
        cam, idx = torch.where(vis)
        px = m2[cam, idx]
        ix, iy = px[:, 0].floor().long().clamp(0, W - 1), px[:, 1].floor().long().clamp(0, H - 1)
        inside = (px[:, 0] >= 0) & (px[:, 0] < W) & (px[:, 1] >= 0) & (px[:, 1] < H)
        r = rays[cam, iy, ix]
        to_pt = torch.nn.functional.normalize(a["means"][idx] - r[:, :3], dim=-1)
        ang = torch.acos((to_pt * r[:, 3:]).sum(-1).clamp(-1, 1))[inside]
        # the point lies within the pixel the projection names: the angle to the pixel CENTRE's ray is under a pixel's span
        pixel_span = 1.5 / float(a["Ks"][0, 0, 0])
        assert float(ang.max()) < pixel_span, (name, float(ang.max()), pixel_span)
    rc, ra, _ = G.rasterization(*args, packed=False, with_ut=True, with_eval3d=True, external_distortion_coeffs=fwd)
    assert bool(torch.isfinite(rc).all()) and float(ra.max()) > 0.5
