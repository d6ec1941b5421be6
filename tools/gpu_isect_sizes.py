#!/usr/bin/env python
"""Intersection paths across input sizes: the Gaussian-major fused path against the tile-owner-major binned path (forced), and
what the automatic rule picks, on the uniform synthetic scene at several (Gaussians, cameras)."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gpu_isect_check as G
import bench
dev = torch.device("cuda", 0)
for N, C in ((250_000, 1), (600_000, 1), (1_000_000, 1), (2_000_000, 1), (250_000, 4), (1_000_000, 4)):
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
