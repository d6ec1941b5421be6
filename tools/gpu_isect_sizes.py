#!/usr/bin/env python
"""Intersection paths across input sizes: the Gaussian-major fused path against the tile-owner-major binned path (forced), and
what the automatic rule picks, on the uniform synthetic scene at several (Gaussians, cameras)."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gpu_isect_check as G
import bench
dev = torch.device("cuda", 0)
for N, C in ((60_000, 1), (120_000, 1), (250_000, 1), (600_000, 1), (1_000_000, 1), (2_000_000, 1), (250_000, 4), (1_000_000, 4)):
    sc, W, H = bench.make_workload(N, dev, n_cameras=C)
    rad, m2, d, con, op = G.project(sc, W, H)
    tw, th = (W + 15) // 16, (H + 15) // 16
    row = {"N": N, "C": C}
    for name, env in (("auto", {}), ("legacy", {"GSX_ISECT_PATH": "legacy"}), ("binned", {"GSX_ISECT_PATH": "binned"})):
        for k in ("GSX_ISECT_PATH",):
            os.environ.pop(k, None)
        os.environ.update(env)
        r = G.timed(m2, rad, d, con, op, C, tw, th)
        row[name] = {"sum_ms": r["sum_ms"], "path": "binned" if any(k.startswith("binned") for k in r) else "fused", "M": r["M"]}
    os.environ.pop("GSX_ISECT_PATH", None)
    print(json.dumps(row), flush=True)


# clustered scenes (the rule's thresholds were read off uniform scenes): the screen-space extent of the cloud shrunk towards
# the image centre (x0.6 / x0.3: the same rows on 36 % / 9 % of the screen), and a mixture - a fifth of the rows in a cluster
# of a tenth of the extent over a uniform background. "binned" is FORCED here: no skew test, no way back.
def family(N, C, kind):
    sc, W, H = bench.make_workload(N, dev, n_cameras=C)
    if kind.startswith("x"):
        sc["means"][:, :2] *= float(kind[1:])
    else:  # mixture
        n = N // 5
        sc["means"][:n, :2] *= 0.1
    return sc, W, H


for N, C, kind in ((250_000, 1, "x0.6"), (250_000, 1, "x0.3"), (1_000_000, 1, "x0.6"), (1_000_000, 1, "x0.3"), (1_000_000, 1, "mix"),
                   (250_000, 1, "mix"), (1_000_000, 4, "x0.6")):
    sc, W, H = family(N, C, kind)
    rad, m2, d, con, op = G.project(sc, W, H)
    tw, th = (W + 15) // 16, (H + 15) // 16
    row = {"N": N, "C": C, "scene": kind}
    for name, env in (("auto", {}), ("auto_again", {}), ("legacy", {"GSX_ISECT_PATH": "legacy"}), ("binned", {"GSX_ISECT_PATH": "binned"})):
        os.environ.pop("GSX_ISECT_PATH", None)
        os.environ.update(env)
        r = G.timed(m2, rad, d, con, op, C, tw, th)
        row[name] = {"sum_ms": r["sum_ms"], "wall_ms": r["wall_ms"], "path": "binned" if any(k.startswith("binned_emit") for k in r) else "fused", "M": r["M"]}
    os.environ.pop("GSX_ISECT_PATH", None)
    print(json.dumps(row), flush=True)


# packed rows of the c3 scene cut at a far plane: few rows, each of them a near (large) Gaussian
import time
import gsplat_amd
from gsplat_amd import _cabi
from gsplat_amd._wrapper import isect_tiles_begin, isect_tiles_finish
sc, W, H = bench.make_workload(1_000_000, dev, n_cameras=1)
tw, th = (W + 15) // 16, (H + 15) // 16
for far in (3.0, 6.0, 10.0):
    with torch.no_grad():
        radii, means2d, depths, conics, _ = gsplat_amd.fully_fused_projection(
            sc["means"], None, sc["quats"], sc["scales"], sc["viewmats"], sc["Ks"], W, H, far_plane=far, opacities=sc["opacities"])
    vis = (radii[0] > 0).all(-1)
    gi = torch.where(vis)[0]
    args = (means2d[0][vis].contiguous(), radii[0][vis].contiguous(), depths[0][vis].contiguous())
    kw = dict(packed=True, n_images=1, image_ids=torch.zeros_like(gi), gaussian_ids=gi, conics=conics[0][vis].contiguous(),
              opacities=sc["opacities"][vis].contiguous())
    row = {"packed_rows": int(gi.numel()), "far_plane": far}
    for name, env in (("auto", {}), ("legacy", {"GSX_ISECT_PATH": "legacy"}), ("binned", {"GSX_ISECT_PATH": "binned"})):
        os.environ.pop("GSX_ISECT_PATH", None)
        os.environ.update(env)
        run = lambda: isect_tiles_finish(isect_tiles_begin(*args, 16, tw, th, **kw))  # noqa: E731
        for _ in range(3):
            out = run()
        torch.cuda.synchronize()
        _cabi.profile_begin()
        for _ in range(10):
            out = run()
        torch.cuda.synchronize()
        prof = _cabi.profile_end()
        row[name] = {"sum_ms": round(sum(sum(v) / len(v) for v in prof.values()), 4),
                     "path": "binned" if any("binned" in k for k in prof) else "fused", "M": int(out[1].numel())}
    os.environ.pop("GSX_ISECT_PATH", None)
    print(json.dumps(row), flush=True)
