"""GPU: Unscented-Transform projection (gsx_project_ut_fwd / gsplat::projection_ut_3dgs_fused) against the golden vectors
of the reference's torch implementation and against the CPU oracle, and rasterization(with_ut=True) on top of it."""
import os

import numpy as np
import pytest
import torch

from _util import make_scene
from test_oracle_ut import CASES, FTHETA, check_against_reference, ut_case

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = "cuda"


@pytest.fixture(scope="module")
def G():
    if not torch.cuda.is_available():
        pytest.skip("needs a ROCm GPU")
    import gsplat_amd

    return gsplat_amd


@pytest.fixture(scope="module")
def gold():
    return dict(np.load(os.path.join(ROOT, "tests", "golden", "ut_ref.npz")))


@pytest.mark.parametrize("name", CASES + FTHETA)
def test_ut_projection_matches_reference_outputs(G, gold, name):
    kw, ut, dist, use_op, (N, C, W, H) = ut_case(name)
    if "ftheta" in kw:  # the f-theta parameter record, as the reference's Python builds it (gsplat/rendering.py:576-580)
        kw = dict(kw, ftheta_coeffs=torch.classes.gsplat.FThetaCameraDistortionParameters(**kw["ftheta"]))
        del kw["ftheta"]
    sc = {k: torch.from_numpy(gold[f"{name}.{k}"]).to(DEV) for k in ("means", "quats", "scales", "opacities", "viewmats", "Ks")}
    cam = {k + "_coeffs": (None if v is None else torch.tensor(v, device=DEV).repeat(C, 1)) for k, v in dist.items()}
    got = G.fully_fused_projection_with_ut(
        sc["means"], sc["quats"], sc["scales"], sc["opacities"] if use_op else None, sc["viewmats"], sc["Ks"], W, H,
        ut_params=torch.classes.gsplat.UnscentedTransformParameters(**ut), **cam, **kw)
    assert got[0].dtype == torch.int32 and got[0].shape == (C, N, 2) and got[3].shape == (C, N, 3)
    got = tuple(None if t is None else t.cpu() for t in got)
    # the reference's CUDA-vs-torch tolerances for this op (tests/test_basic.py:838-960): validity mismatches < 0.1 %,
    # radii atol 2, means2d atol 5e-2; the data allow tighter bounds
    check_against_reference(got, gold, name, radii_atol=1, means_atol=2e-2, conic_rel=3e-2, max_flips=2)


def test_ut_projection_larger_scene_vs_oracle(G):
    """100k Gaussians, 3 distorted cameras: kernel vs CPU oracle (same tolerances, a few boundary rows may flip)."""
    from oracle import ut as O

    sc, W, H = make_scene(N=100_000, C=3, width=320, height=200, seed=21)
    rad = torch.tensor([[-0.12, 0.03, 0.0, 0.0, 0.0, 0.0]]).repeat(3, 1)
    tan = torch.tensor([[0.002, -0.001]]).repeat(3, 1)
    ref = O.fully_fused_projection_with_ut(sc["means"], sc["quats"], sc["scales"], sc["opacities"], sc["viewmats"], sc["Ks"],
                                           W, H, calc_compensations=True, radial_coeffs=rad, tangential_coeffs=tan)
    got = G.fully_fused_projection_with_ut(
        sc["means"].to(DEV), sc["quats"].to(DEV), sc["scales"].to(DEV), sc["opacities"].to(DEV), sc["viewmats"].to(DEV),
        sc["Ks"].to(DEV), W, H, calc_compensations=True, radial_coeffs=rad.to(DEV), tangential_coeffs=tan.to(DEV))
    got = [t.cpu() for t in got]
    vr, vg = (ref[0] > 0).all(-1), (got[0] > 0).all(-1)
    assert float((vr != vg).float().mean()) < 1e-3
    both = vr & vg
    assert both.sum() > 50_000
    assert int((ref[0] - got[0]).abs()[both].max()) <= 2
    assert float(((ref[0] - got[0]).abs()[both] > 0).float().mean()) < 5e-3
    assert float((ref[1] - got[1]).abs()[both].max()) < 5e-2
    torch.testing.assert_close(got[2][both], ref[2][both], rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(got[4][both], ref[4][both], rtol=5e-3, atol=1e-4)


def test_rasterization_with_ut(G):
    """with_ut=True (classic compositing on UT-projected Gaussians): close to the EWA render for an undistorted pinhole,
    zero distortion coefficients change nothing, colours / opacities still receive gradients, packed rows are refused."""
    sc, W, H = make_scene(N=4000, C=2, width=160, height=112, seed=6)
    a = {k: v.to(DEV) for k, v in sc.items()}
    args = (a["means"], a["quats"], a["scales"], a["opacities"], a["colors"], a["viewmats"], a["Ks"], W, H)
    rc0, ra0, _ = G.rasterization(*args, packed=False)
    rc1, ra1, meta = G.rasterization(*args, packed=False, with_ut=True)
    assert rc1.shape == rc0.shape and meta["radii"].shape == (2, 4000, 2)
    assert float((rc1 - rc0).abs().mean()) < 2e-2 and float((ra1 - ra0).abs().mean()) < 2e-2
    rc2, ra2, _ = G.rasterization(*args, packed=False, with_ut=True, radial_coeffs=torch.zeros(2, 6, device=DEV))
    assert float((rc2 - rc1).abs().max()) < 1e-5
    rc3, _, _ = G.rasterization(*args, packed=False, with_ut=True,
                                radial_coeffs=torch.tensor([[-0.2, 0.05, 0, 0, 0, 0.0]], device=DEV).repeat(2, 1))
    assert float((rc3 - rc1).abs().mean()) > 1e-3  # the distortion moves things
    colors = a["colors"].clone().requires_grad_(True)
    opac = a["opacities"].clone().requires_grad_(True)
    rc4, ra4, _ = G.rasterization(a["means"], a["quats"], a["scales"], opac, colors, a["viewmats"], a["Ks"], W, H,
                                  packed=False, with_ut=True)
    (rc4.sum() + ra4.sum()).backward()
    assert float(colors.grad.abs().sum()) > 0 and float(opac.grad.abs().sum()) > 0
    with pytest.raises(RuntimeError, match="Packed mode is not supported with UT"):
        G.rasterization(*args, packed=True, with_ut=True)
    # f-theta: an (almost) equidistant lens r = f theta with the pinhole's focal length renders the image centre like the
    # pinhole does (theta ~ tan theta there); the record is required, and the model needs the UT projection
    fx = float(a["Ks"][0, 0, 0])
    ft = torch.classes.gsplat.FThetaCameraDistortionParameters(
        reference_poly=1, pixeldist_to_angle_poly=[0.0, 1.0 / fx, 0.0, 0.0, 0.0, 0.0],
        angle_to_pixeldist_poly=[0.0, fx, 0.0, 0.0, 0.0, 0.0], max_angle=1.3, linear_cde=[1.0, 0.0, 0.0])
    t1, ta1, _ = G.rasterization(*args, packed=False, camera_model="ftheta", with_ut=True, ftheta_coeffs=ft)
    assert torch.isfinite(t1).all() and float(ta1.mean()) > 0.05
    cy, cx2 = H // 2, W // 2
    centre = (slice(None), slice(cy - 8, cy + 8), slice(cx2 - 8, cx2 + 8))
    assert float((t1[centre] - rc1[centre]).abs().mean()) < 3e-2
    with pytest.raises(ValueError, match="ftheta_coeffs must be given"):
        G.rasterization(*args, packed=False, camera_model="ftheta", with_ut=True)
    with pytest.raises(RuntimeError, match="only supported via UT"):
        G.rasterization(*args, packed=False, camera_model="ftheta", ftheta_coeffs=ft)
    # fisheye: UT render vs the EWA fisheye render, and with distortion coefficients
    f0, fa0, _ = G.rasterization(*args, packed=False, camera_model="fisheye")
    f1, fa1, _ = G.rasterization(*args, packed=False, camera_model="fisheye", with_ut=True)
    assert float((f1 - f0).abs().mean()) < 2e-2 and float((fa1 - fa0).abs().mean()) < 2e-2
    f2, _, _ = G.rasterization(*args, packed=False, camera_model="fisheye", with_ut=True,
                               radial_coeffs=torch.tensor([[-0.05, 0.01, 0.0, 0.0]], device=DEV).repeat(2, 1))
    assert torch.isfinite(f2).all() and float((f2 - f1).abs().mean()) > 1e-4


def _rs_cases():
    import importlib.util

    spec = importlib.util.spec_from_file_location("pin_ut_rs", os.path.join(ROOT, "oracle", "pin_ut_rs_against_reference.py"))
    pin = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(pin)
    return pin


@pytest.mark.parametrize("name", ["rs_top_bottom_pinhole", "rs_left_right_pinhole_all_valid", "rs_bottom_top_opencv",
                                  "rs_right_left_fisheye", "rs_top_bottom_ortho", "rs_left_right_ftheta", "rs_top_bottom_distance",
                                  "global_distance_pinhole", "global_distance_ftheta"])
def test_ut_projection_rolling_shutter_and_distance_depth_match_reference_outputs(G, name):
    """gsx_project_ut_rs_fwd against the outputs of the reference's own torch statement with a ROLLING shutter (every shutter
    direction, every camera model) and with global_z_order=False (tests/golden/ut_rs_ref.npz, written by
    oracle/pin_ut_rs_against_reference.py). The read-out time is floor(pixel row or column) / (size - 1): a sigma point that lands
    within rounding of a row boundary takes the neighbouring row's pose in one of the two evaluations and moves by the motion of
    one row, so a small share of the rows may differ by more than the smooth tolerance (the reference's own CUDA-vs-torch test
    budgets the rolling modes the same way, tests/test_basic.py:898-1010)."""
    pin = _rs_cases()
    gold = dict(np.load(os.path.join(ROOT, "tests", "golden", "ut_rs_ref.npz")))
    seed, rs, gz, kw = pin.CASES[name]
    kw, ut, dist = pin.split(kw)
    N, C, W, H = pin.N, pin.C, pin.W, pin.H
    if "ftheta" in kw:
        kw = dict(kw, ftheta_coeffs=torch.classes.gsplat.FThetaCameraDistortionParameters(**kw["ftheta"]))
        del kw["ftheta"]
    sc = {k: torch.from_numpy(gold[f"{name}.{k}"]).to(DEV) for k in ("means", "quats", "scales", "opacities", "viewmats", "Ks")}
    vm1 = torch.from_numpy(gold[f"{name}.viewmats_rs"]).to(DEV) if rs != 4 else None
    cam = {k + "_coeffs": (None if v is None else torch.tensor(v, device=DEV).repeat(C, 1)) for k, v in dist.items()}
    got = G.fully_fused_projection_with_ut(
        sc["means"], sc["quats"], sc["scales"], sc["opacities"], sc["viewmats"], sc["Ks"], W, H,
        ut_params=torch.classes.gsplat.UnscentedTransformParameters(**ut), rolling_shutter=rs, viewmats_rs=vm1,
        global_z_order=gz, **cam, **kw)
    got = [None if t is None else t.cpu() for t in got]
    ref = {k: torch.from_numpy(gold[f"{name}.ref.{k}"]) for k in ("radii", "means2d", "depths", "conics")}
    vr, vg = (ref["radii"] > 0).all(-1), (got[0] > 0).all(-1)
    rows = vr.numel()
    flips = int((vr != vg).sum())
    both = vr & vg
    d_mean = (got[1] - ref["means2d"]).abs().amax(-1)[both]
    d_rad = (got[0] - ref["radii"]).abs().amax(-1)[both]
    d_depth = (got[2] - ref["depths"]).abs()[both] / (ref["depths"].abs()[both] + 1e-6)
    d_con = ((got[3] - ref["conics"]).abs() / (ref["conics"].abs() + 1e-3)).amax(-1)[both]
    smooth = (d_mean <= 5e-2) & (d_rad <= 1) & (d_con <= 5e-2)
    rough = int((~smooth).sum())
    print(f"{name}: visible {int(vr.sum())}/{rows}, validity flips {flips}, rows beyond the smooth tolerance {rough}, "
          f"max |d means2d| {float(d_mean.max()):.3e} (median {float(d_mean.median()):.1e}), max rel d depth {float(d_depth.max()):.2e}")
    assert flips <= max(2, rows // 200), name              # <= 0.5 % of the rows change validity
    assert float(d_depth.max()) < 1e-4, name                 # the depth uses the mid-frame pose: smooth
    budget = 0 if rs == 4 else max(2, int(both.sum()) // 50)  # rolling: <= 2 % of the rows sit on a read-out boundary
    assert rough <= budget, (name, rough, budget)
    if got[4] is not None and f"{name}.ref.compensations" in gold:
        ok = torch.zeros_like(both)
        ok[both] = smooth
        assert float((got[4] - torch.from_numpy(gold[f"{name}.ref.compensations"])).abs()[ok].max()) < 5e-3, name
