"""CPU: oracle/eval3d.py (from-world compositing of 3DGUT) against the golden vectors of the reference's own torch
implementation (oracle/pin_eval3d_against_reference.py -> tests/golden/eval3d_ref.npz): images, last sample ids and the
gradients of all five inputs. No product kernel exists for this stage yet; this pins the checker it will be built against."""
import os

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def gold():
    return dict(np.load(os.path.join(ROOT, "tests", "golden", "eval3d_ref.npz")))


@pytest.mark.parametrize("name", ["a", "b"])
def test_eval3d_oracle_matches_reference(gold, name):
    from oracle import eval3d as E

    N, C, W, H, ts = (int(v) for v in gold[f"{name}.shape"])
    t = lambda k: torch.from_numpy(gold[f"{name}.{k}"])  # noqa: E731
    leaves = {k: t(k).clone().requires_grad_(True) for k in ("means", "quats", "scales", "opacities", "colors")}
    bg = t("backgrounds") if f"{name}.backgrounds" in gold else None
    ren, alp, last = E.rasterize_to_pixels_eval3d(leaves["means"], leaves["quats"], leaves["scales"], leaves["colors"],
                                                  leaves["opacities"], t("rays"), W, H, ts, t("isect_offsets"),
                                                  t("flatten_ids"), backgrounds=bg)
    assert ren.shape == (C, H, W, 3) and alp.shape == (C, H, W, 1) and last.dtype == torch.int32
    torch.testing.assert_close(ren, t("ref.render"), rtol=1e-5, atol=2e-5)
    torch.testing.assert_close(alp, t("ref.alpha"), rtol=1e-5, atol=2e-5)
    assert float((last == t("ref.last_ids")).float().mean()) > 0.999
    ((ren * t("v_render")).sum() + (alp * t("v_alpha")).sum()).backward()
    for k, leaf in leaves.items():
        ref = t(f"ref.v_{k}")
        assert float((leaf.grad - ref).abs().max()) <= 2e-4 * float(ref.abs().max()), k


def test_pinhole_rays_are_unit_and_hit_the_pixel_centres(gold):
    from oracle import eval3d as E

    N, C, W, H, ts = (int(v) for v in gold["a.shape"])
    viewmats, Ks = torch.from_numpy(gold["a.viewmats"]), torch.from_numpy(gold["a.Ks"])
    rays = E.pinhole_rays(viewmats, Ks, W, H)
    torch.testing.assert_close(rays, torch.from_numpy(gold["a.rays"]))
    torch.testing.assert_close(rays[..., 3:].norm(dim=-1), torch.ones(C, H, W), rtol=0, atol=1e-6)
    # a point one unit along the ray of pixel (x, y) projects back to (x + 0.5, y + 0.5)
    p = rays[..., :3] + rays[..., 3:]
    pc = torch.einsum("cij,chwj->chwi", viewmats[:, :3, :3], p) + viewmats[:, None, None, :3, 3]
    u = pc[..., 0] / pc[..., 2] * Ks[:, 0, 0, None, None] + Ks[:, 0, 2, None, None]
    v = pc[..., 1] / pc[..., 2] * Ks[:, 1, 1, None, None] + Ks[:, 1, 2, None, None]
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    assert float((u - (xs + 0.5)).abs().max()) < 1e-3 and float((v - (ys + 0.5)).abs().max()) < 1e-3
