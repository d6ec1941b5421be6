#!/usr/bin/env python
"""BASELINE.json configs[0] ("c1": 10 k random Gaussians, 256 x 256, SH degree 0) timed through the REFERENCE's own CPU path
(gsplat/cuda/_torch_impl.py: _fully_fused_projection, _spherical_harmonics, accumulate + torch autograd for the backward; the
tile / sample lists, CUDA-only in the reference, come from the pinned oracle stages as in oracle/pin_c1_against_reference.py)
next to the repo's own CPU port (oracle/pipeline.py) on the same scene and host. Needs the reference's PYTHON files only:
a checkout (`--ref /root/reference`, the build container) or the archive `__graft_entry__.build()` stages,
oracle/_ref/reference_py.zip, which travels to the GPU box - bench.py runs this script there in a subprocess and reports the
result as `cpu_baseline.c1_reference` (measured in that run, on that host's cores).
    python tools/time_c1_reference_cpu.py [--ref /root/reference | oracle/_ref/reference_py.zip] [--reps 3] [--out file]"""
import argparse
import json
import os
import platform
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "c1_reference_cpu.json"))
    ap.add_argument("--where", default="build container")
    args = ap.parse_args()
    sys.path.insert(0, args.ref)
    sys.dont_write_bytecode = True
    from pin_against_reference import install_nerfacc_stub
    from pin_c1_against_reference import c1_scene

    install_nerfacc_stub()
    import gsplat  # noqa: F401
    from gsplat.cuda import _torch_impl as R
    from gsplat.cuda._math import _quat_scale_to_covar_preci
    from oracle import oracle as O
    from oracle.pipeline import rasterization_cpu

    sc, W, H = c1_scene()
    ts, tw, th = 16, 16, 16
    names = ("means", "quats", "scales", "opacities", "colors")

    def reference_step():
        lv = {k: sc[k].clone().requires_grad_(True) for k in names}
        covars, _ = _quat_scale_to_covar_preci(lv["quats"], lv["scales"], True, False, triu=False)
        radii, means2d, depths, conics, _ = R._fully_fused_projection(lv["means"], covars, sc["viewmats"], sc["Ks"], W, H)
        dirs = lv["means"][None] - torch.linalg.inv(sc["viewmats"])[:, None, :3, 3]
        cols = torch.clamp_min(R._spherical_harmonics(0, dirs, lv["colors"]) + 0.5, 0.0)
        op = lv["opacities"][None]
        with torch.no_grad():
            rad_i = radii.to(torch.int32)
            _, ids, fl = O.isect_tiles(means2d.detach(), rad_i, depths.detach(), ts, tw, th)
            off = O.isect_offset_encode(ids, 1, tw, th)
            g_ids, p_ids, i_ids = O.rasterize_to_indices(means2d.detach(), conics.detach(), op.detach(), W, H, ts, off, fl)
        t0 = time.perf_counter()
        rc, ra = R.accumulate(means2d, conics, op.contiguous(), cols.contiguous(), g_ids, p_ids, i_ids, W, H)
        t_acc_fwd = time.perf_counter() - t0
        rc.sum().backward()
        return t_acc_fwd, int(fl.numel())

    def port_step():
        g = torch.ones(1, H, W, 3)
        rasterization_cpu(sc["means"], sc["quats"], sc["scales"], sc["opacities"], sc["colors"], sc["viewmats"], sc["Ks"], W, H,
                          sh_degree=0, v_render_colors=g, v_render_alphas=torch.zeros(1, H, W, 1))

    def timed(fn):
        fn()
        ts_ = []
        for _ in range(args.reps):
            t0 = time.perf_counter()
            fn()
            ts_.append(time.perf_counter() - t0)
        return min(ts_)

    n_isects = reference_step()[1]
    t_ref, t_port = timed(reference_step), timed(port_step)
    pix = W * H
    rec = {
        "config": "c1: 10 k random Gaussians, 256 x 256 pinhole, SH degree 0, fwd+bwd (loss = sum of colours)",
        "n_isects": n_isects,
        "reference_cpu": {"s_per_step": round(t_ref, 4), "mpixels_per_s": round(pix / t_ref / 1e6, 4),
                          "path": "gsplat/cuda/_torch_impl.py _fully_fused_projection + _spherical_harmonics + accumulate "
                                  "(restated nerfacc) + torch autograd; tile / sample lists from the pinned oracle stages "
                                  "(included in the time)"},
        "port_cpu": {"s_per_step": round(t_port, 4), "mpixels_per_s": round(pix / t_port / 1e6, 4),
                     "path": "oracle/pipeline.py (OpenMP C compositing + torch-CPU projection / SH)"},
        "host": {"machine": platform.machine(), "cores": os.cpu_count(), "torch_threads": torch.get_num_threads(),
                 "where": args.where, "reference_from": os.path.basename(args.ref.rstrip("/"))},
        "reps": args.reps,
    }
    if args.out != "-":
        json.dump(rec, open(args.out, "w"), indent=1)
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
