"""CPU: BASELINE.json configs[0] ("c1": 10k random Gaussians, 256 x 256 pinhole, SH degree 0 - the reference's own
CPU-runnable case). The oracle pipeline renders the scene and is compared with the fixture that
oracle/pin_c1_against_reference.py made from the REFERENCE's CPU path (_torch_impl projection + SH + accumulate): 4096
sampled pixels and the per-tile means of the reference image."""
import importlib.util
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c1_oracle_pipeline_matches_reference_cpu_path():
    from oracle.pipeline import rasterization_cpu

    spec = importlib.util.spec_from_file_location("pin_c1", os.path.join(ROOT, "oracle", "pin_c1_against_reference.py"))
    # only the scene generator is needed; the module's top-level imports are reference-free
    pin = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(pin)
    sc, W, H = pin.c1_scene()
    gold = np.load(os.path.join(ROOT, "tests", "golden", "c1_ref.npz"))
    out = rasterization_cpu(sc["means"], sc["quats"], sc["scales"], sc["opacities"], sc["colors"], sc["viewmats"], sc["Ks"],
                            W, H, sh_degree=0, want_grads=False)
    img = torch.cat([out["render_colors"], out["render_alphas"]], -1)
    assert img.shape == (1, 256, 256, 4)
    got = img.reshape(W * H, 4)[torch.from_numpy(gold["pixel_ids"])]
    torch.testing.assert_close(got, torch.from_numpy(gold["pixels"]), rtol=1e-4, atol=5e-5)
    tiles = img.reshape(16, 16, 16, 16, 4).mean(dim=(1, 3))
    torch.testing.assert_close(tiles, torch.from_numpy(gold["tile_means"]), rtol=1e-4, atol=2e-5)
    assert int(gold["n_visible"]) == 10_000 and 0.0 <= float(img[..., 3].min()) and float(img[..., 3].max()) <= 1.0
