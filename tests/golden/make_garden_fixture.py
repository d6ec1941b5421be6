#!/usr/bin/env python
"""Writes tests/golden/garden_scene.npz: the INPUT scene of BASELINE.json config c2 (assets/test_garden.npz, cropped to
[-2,2]^3 like gsplat/_helper.py:50-75 load_test_data(scene_grid=1), cameras rescaled to 1920x1080 like
profiling/main.py:100-102). Only the data that cannot be regenerated travels (means f32, colours u8, cameras); the random
per-Gaussian attributes are drawn in the test from CPU generators with fixed seeds, as load_test_data does (:92-101).
Run where the reference checkout exists:  python tests/golden/make_garden_fixture.py [--ref /root/reference]"""
import argparse
import os

import numpy as np

ap = argparse.ArgumentParser()
ap.add_argument("--ref", default="/root/reference")
args = ap.parse_args()
data = np.load(os.path.join(args.ref, "assets", "test_garden.npz"))
means, colors = data["means3d"].astype(np.float32), data["colors"]
sel = ((means >= -2) & (means <= 2)).all(-1)
means, colors = means[sel], colors[sel].astype(np.uint8)
W0, H0 = int(data["width"]), int(data["height"])
Ks = data["Ks"].astype(np.float32).copy()
Ks[:, 0, :] *= 1920 / W0
Ks[:, 1, :] *= 1080 / H0
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "garden_scene.npz")
np.savez_compressed(out, means=means, colors_u8=colors, viewmats=data["viewmats"].astype(np.float32), Ks=Ks,
                    wh=np.array([1920, 1080]))
print(out, means.shape, os.path.getsize(out) / 1e6, "MB")
