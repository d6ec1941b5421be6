"""CPU: oracle/ut.py (Unscented-Transform projection) against the golden vectors produced by the reference's own torch
implementation (oracle/pin_ut_against_reference.py -> tests/golden/ut_ref.npz). Tolerances are the reference's
CUDA-vs-torch tolerances for this op (tests/test_basic.py:838-960, global shutter), tightened where the data allow."""
import os

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = ["pinhole", "pinhole_all_valid", "pinhole_comp_clip", "pinhole_no_opacity", "opencv_full", "opencv_radial4",
         "opencv_strong", "ortho", "fisheye_plain", "fisheye_k", "fisheye_k4", "fisheye_tight"]
FTHETA = ["ftheta_forward", "ftheta_inverse"]  # f-theta cameras: parameters travel as a record (ftheta=dict(...) in the case table)


@pytest.fixture(scope="module")
def gold():
    return dict(np.load(os.path.join(ROOT, "tests", "golden", "ut_ref.npz")))


def ut_case(name):
    """Keyword arguments of a golden case (same table as the pin script)."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("pin_ut", os.path.join(ROOT, "oracle", "pin_ut_against_reference.py"))
    pin = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(pin)
    kw, ut, dist, use_op = pin.split(pin.CASES[name][1])
    return kw, ut, dist, use_op, (pin.N, pin.C, pin.W, pin.H)


def check_against_reference(got, gold, name, radii_atol=1, means_atol=2e-2, conic_rel=3e-2, max_flips=1):
    ref = {k: torch.from_numpy(gold[f"{name}.ref.{k}"]) for k in ("radii", "means2d", "depths", "conics")}
    vr, vg = (ref["radii"] > 0).all(-1), (got[0] > 0).all(-1)
    assert int((vr != vg).sum()) <= max_flips, f"{name}: visibility differs on {int((vr != vg).sum())} rows"
    both = vr & vg
    assert both.sum() > 100
    assert int((ref["radii"] - got[0]).abs()[both].max()) <= radii_atol
    assert float((ref["means2d"] - got[1]).abs()[both].max()) < means_atol
    torch.testing.assert_close(got[2][both], ref["depths"][both], rtol=1e-5, atol=1e-5)
    assert float(((ref["conics"] - got[3]).abs() / (ref["conics"].abs() + 1e-3))[both].max()) < conic_rel
    # rows only one side keeps are zero on the other: invalid rows are written as zeros
    assert float(got[1][~vg].abs().max()) == 0.0 and float(got[3][~vg].abs().max()) == 0.0
    if f"{name}.ref.compensations" in gold:
        rc = torch.from_numpy(gold[f"{name}.ref.compensations"])
        torch.testing.assert_close(got[4][both], rc[both], rtol=2e-3, atol=1e-4)
    else:
        assert got[4] is None


@pytest.mark.parametrize("name", CASES + FTHETA)
def test_ut_oracle_matches_reference(gold, name):
    from oracle import ut as O

    kw, ut, dist, use_op, (N, C, W, H) = ut_case(name)
    sc = {k: torch.from_numpy(gold[f"{name}.{k}"]) for k in ("means", "quats", "scales", "opacities", "viewmats", "Ks")}
    cam = {k + "_coeffs": (None if v is None else torch.tensor(v).repeat(C, 1)) for k, v in dist.items()}
    got = O.fully_fused_projection_with_ut(sc["means"], sc["quats"], sc["scales"], sc["opacities"] if use_op else None,
                                           sc["viewmats"], sc["Ks"], W, H, **ut, **cam, **kw)
    check_against_reference(got, gold, name)


def test_ut_weights_sum_to_one():
    from oracle import ut as O

    for a, b, k in ((0.1, 2.0, 0.0), (0.5, 1.0, 0.5), (1.0, 0.0, 0.0)):
        w0, c0, wi, spread = O.ut_weights(a, b, k)
        assert abs(w0 + 6 * wi - 1.0) < 1e-9 and spread > 0
        assert abs(c0 - w0 - (1 - a * a + b)) < 1e-12


def test_fisheye_angle_limit_matches_reference(gold):
    """Largest projected ray angle of the OpenCV fisheye model (the reference camera's `max_angle`,
    _torch_cameras.py:1344-1521) for 69 coefficient sets covering every branch: the oracle's restatement AND the
    product's (gsplat_amd._ops.fisheye_max_angle, host-side tensor code that feeds gsx_project_ut_fwd)."""
    from gsplat_amd import _ops
    from oracle import ut as O

    k, Ks = torch.from_numpy(gold["fisheye_limit.k"]), torch.from_numpy(gold["fisheye_limit.Ks"])
    ref = torch.from_numpy(gold["fisheye_limit.ref.max_angle"])
    fin = torch.isfinite(ref)
    for got in (O.fisheye_max_angle(k, Ks[:, 0, 0], Ks[:, 1, 1], Ks[:, 0, 2], Ks[:, 1, 2], 2000, 1600),
                _ops.fisheye_max_angle(k, Ks, 2000, 1600)):
        assert bool((torch.isfinite(got) == fin).all())
        torch.testing.assert_close(got[fin], ref[fin], rtol=1e-5, atol=1e-6)
    assert len(set(ref.tolist())) > 30  # the table really exercises different limits
