#!/usr/bin/env python
"""Binned tile intersection (csrc/isect_binned.hip) against the Gaussian-major fused path (csrc/isect_fused.hip, which the
test-suite pins bit for bit against the C oracle): every output must be IDENTICAL - tiles_per_gauss, isect_ids, flatten_ids,
offsets - on scenes chosen to hit every branch (ellipse / box test, tile sizes, several images, packed rows, depth ties,
giant Gaussians = oversized tiles and the workspace-overflow retry, sparse tile masks). Then times both on c3 / c4 for the
bin shapes and LDS capacities the library can be switched to (GSX_ISECT_BIN, GSX_ISECT_CAP).
usage: gpu_isect_check.py [check] [bench] [c4]"""
import json, math, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import bench, gsplat_amd
from gsplat_amd import _cabi
from gsplat_amd._wrapper import isect_tiles_begin, isect_tiles_finish

dev = torch.device("cuda", 0)


def isect(m2, rad, d, ts, tw, th, legacy, sort_int=False, **kw):
    os.environ["GSX_ISECT_PATH"] = "legacy" if legacy else "binned"
    if sort_int:
        os.environ["GSX_ISECT_SORT"] = "int"
    try:
        tpg, ids, fl = isect_tiles_finish(isect_tiles_begin(m2, rad, d, ts, tw, th, **kw))
        I = kw.get("n_images") or (math.prod(m2.shape[:-2]) if m2.dim() > 2 else 1)
        off = gsplat_amd.isect_offset_encode(ids, I, tw, th)
    finally:
        os.environ.pop("GSX_ISECT_PATH", None)
        os.environ.pop("GSX_ISECT_SORT", None)
    return tpg, ids, fl, off


def compare(name, m2, rad, d, ts, W, H, **kw):
    tw, th = math.ceil(W / ts), math.ceil(H / ts)
    a = isect(m2, rad, d, ts, tw, th, False, **kw)
    b = isect(m2, rad, d, ts, tw, th, True, sort_int=True, **kw)  # round 3's integer network: the baseline
    ok = all(torch.equal(x, y) for x, y in zip(a, b))
    for other in (isect(m2, rad, d, ts, tw, th, True, **kw), isect(m2, rad, d, ts, tw, th, False, sort_int=True, **kw)):
        ok &= all(torch.equal(x, y) for x, y in zip(other, b))
    print(f"{'OK  ' if ok else 'FAIL'} {name}: M={a[1].numel()} rows={rad.numel() // 2} tiles={tw}x{th}", flush=True)
    if not ok:
        for nm, x, y in zip(("tpg", "ids", "flat", "off"), a, b):
            if x.shape != y.shape:
                print("   ", nm, "shape", tuple(x.shape), tuple(y.shape))
            elif not torch.equal(x, y):
                bad = (x != y).flatten().nonzero().flatten()
                print("   ", nm, "differs at", bad.numel(), "first", bad[:5].tolist(), x.flatten()[bad[:5]].tolist(), y.flatten()[bad[:5]].tolist())
    return ok


def project(sc, W, H):
    with torch.no_grad():
        radii, means2d, depths, conics, _ = gsplat_amd.fully_fused_projection(
            sc["means"], None, sc["quats"], sc["scales"], sc["viewmats"], sc["Ks"], W, H, opacities=sc["opacities"])
    C = sc["viewmats"].shape[0]
    opac = sc["opacities"][None].expand(C, -1).contiguous()
    return radii, means2d, depths, conics, opac


def check():
    from _util import make_scene
    ok = True
    for (N, C, W, H, seed) in ((20000, 3, 320, 200, 6), (200000, 1, 1920, 1080, 1), (3000, 2, 160, 112, 0)):
        sc, W, H = make_scene(N=N, C=C, width=W, height=H, seed=seed)
        sc = {k: v.to(dev) for k, v in sc.items()}
        rad, m2, d, con, op = project(sc, W, H)
        for ts in (16, 8, 4):
            if W * H // (ts * ts) > 30000:
                continue
            ok &= compare(f"ellipse N={N} C={C} ts={ts}", m2, rad, d, ts, W, H, conics=con, opacities=op)
            ok &= compare(f"box     N={N} C={C} ts={ts}", m2, rad, d, ts, W, H)
        # packed rows of one image
        vis = (rad[0] > 0).all(-1)
        gi = torch.where(vis)[0]
        ok &= compare(f"packed  N={N}", m2[0][vis], rad[0][vis], d[0][vis], 16, W, H, packed=True, n_images=1,
                      image_ids=torch.zeros_like(gi), gaussian_ids=gi, conics=con[0][vis], opacities=op[0][vis])
        # depth ties: a handful of distinct depths
        dq = (d * 2).round() / 2 + 0.25
        ok &= compare(f"ties    N={N}", m2, rad, dq, 16, W, H, conics=con, opacities=op)
        # all depths equal
        ok &= compare(f"flat    N={N}", m2, rad, torch.ones_like(d), 16, W, H)
        # depths the f64 network cannot order (negative, zero, denormal, inf, NaN): those lists take the integer network
        dodd = d.clone()
        r = torch.rand_like(d)
        for lo, v in ((0.00, -2.0), (0.01, 0.0), (0.02, 1e-42), (0.03, float("inf")), (0.04, float("nan"))):
            dodd[(r >= lo) & (r < lo + 0.01)] = v
        ok &= compare(f"odd     N={N}", m2, rad, dodd, 16, W, H, conics=con, opacities=op)
    # giant Gaussians: more (row, bin) entries than the workspace holds -> the GSX_ISECT_RETRY path
    for (N, scale, W, H) in ((30000, 0.5, 640, 360), (5000, 1.5, 1920, 1080), (60000, 0.2, 1920, 1080)):
        sc, W, H = make_scene(N=N, C=1, width=W, height=H, seed=3)
        sc = {k: v.to(dev) for k, v in sc.items()}
        sc["scales"] = sc["scales"] * 0 + scale
        rad, m2, d, con, op = project(sc, W, H)
        ok &= compare(f"giant   N={N} s={scale} ellipse", m2, rad, d, 16, W, H, conics=con, opacities=op)
        ok &= compare(f"giant   N={N} s={scale} box", m2, rad, d, 16, W, H)
    # clusters: tiles longer than the LDS sort of kernel F (work-list sort: in LDS up to 9216 entries, through HBM beyond)
    for (N, shrink) in ((60000, 0.02), (60000, 0.12), (300000, 0.3)):
        sc, W, H = make_scene(N=N, C=2, width=1280, height=720, seed=4)
        sc = {k: v.to(dev) for k, v in sc.items()}
        sc["means"][:, :2] *= shrink
        rad, m2, d, con, op = project(sc, W, H)
        ok &= compare(f"cluster N={N} x{shrink} ellipse", m2, rad, d, 16, W, H, conics=con, opacities=op)
        ok &= compare(f"cluster N={N} x{shrink} ties", m2, rad, (d * 4).round() / 4 + 0.5, 16, W, H, conics=con, opacities=op)
    # c3 / c4 at full size
    for (n, c) in ((1_000_000, 1), (1_000_000, 4)):
        sc, W, H = bench.make_workload(n, dev, n_cameras=c)
        rad, m2, d, con, op = project(sc, W, H)
        ok &= compare(f"bench scene N={n} C={c}", m2, rad, d, 16, W, H, conics=con, opacities=op)
    print("ISECT CHECK", "PASSED" if ok else "FAILED", flush=True)
    return ok


def timed(m2, rad, d, con, op, C, tw, th, n=10):
    def run():
        return isect_tiles_finish(isect_tiles_begin(m2, rad, d, 16, tw, th, n_images=C, conics=con, opacities=op))
    for _ in range(3):
        out = run()
    torch.cuda.synchronize()
    _cabi.profile_begin()
    t0 = time.perf_counter()
    for _ in range(n):
        out = run()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / n * 1e3
    prof = _cabi.profile_end()
    rec = {k.replace("gsx_isect_", ""): round(sum(v) / len(v), 4) for k, v in sorted(prof.items())}
    rec["sum_ms"] = round(sum(rec.values()), 4)
    rec["wall_ms"] = round(wall, 4)
    rec["M"] = int(out[1].numel())
    return rec


def bench_all(c4):
    sc, W, H = bench.make_workload(4_000_000 if c4 else 1_000_000, dev, n_cameras=4 if c4 else 1)
    rad, m2, d, con, op = project(sc, W, H)
    C = sc["viewmats"].shape[0]
    tw, th = (W + 15) // 16, (H + 15) // 16
    variants = [("auto", {}), ("auto-intsort", {"GSX_ISECT_SORT": "int"}), ("legacy", {"GSX_ISECT_PATH": "legacy"}),
                ("legacy-intsort", {"GSX_ISECT_PATH": "legacy", "GSX_ISECT_SORT": "int"})]
    for b in ("4x4", "4x2", "2x2", "8x2", "2x4"):
        variants.append((f"bin{b}", {"GSX_ISECT_BIN": b, "GSX_ISECT_PATH": "binned"}))
    for name, env in variants:
        for k in ("GSX_ISECT_PATH", "GSX_ISECT_BIN", "GSX_ISECT_SORT"):
            os.environ.pop(k, None)
        os.environ.update(env)
        print(json.dumps({"scene": "c4" if c4 else "c3", "variant": name, **timed(m2, rad, d, con, op, C, tw, th)}), flush=True)
    for k in ("GSX_ISECT_PATH", "GSX_ISECT_BIN", "GSX_ISECT_SORT"):
        os.environ.pop(k, None)


def bench_one(c4):
    """The auto path only (for rocprofv3 per-kernel traces and the GSX_ISECT_DBG ablations) + the list-length distribution."""
    sc, W, H = bench.make_workload(4_000_000 if c4 else 1_000_000, dev, n_cameras=4 if c4 else 1)
    rad, m2, d, con, op = project(sc, W, H)
    C = sc["viewmats"].shape[0]
    tw, th = (W + 15) // 16, (H + 15) // 16
    rec = timed(m2, rad, d, con, op, C, tw, th)
    tpg, ids, fl = isect_tiles_finish(isect_tiles_begin(m2, rad, d, 16, tw, th, n_images=C, conics=con, opacities=op))
    off = gsplat_amd.isect_offset_encode(ids, C, tw, th).flatten()
    n = torch.diff(torch.cat([off, torch.tensor([fl.numel()], device=dev, dtype=off.dtype)])).float()
    q = lambda t, p: int(torch.quantile(t, p).item())  # noqa: E731
    rec["tile_list"] = {"mean": round(float(n.mean()), 1), "p50": q(n, 0.5), "p90": q(n, 0.9), "p99": q(n, 0.99), "max": int(n.max())}
    bins = n.view(C, th, tw)[:, : th // 2 * 2, : tw // 4 * 4].reshape(C, th // 2, 2, tw // 4, 4).sum((2, 4)).flatten()
    rec["bin_4x2"] = {"mean": round(float(bins.mean()), 1), "p50": q(bins, 0.5), "p90": q(bins, 0.9), "p99": q(bins, 0.99), "max": int(bins.max())}
    t = tpg.flatten()[tpg.flatten() > 0].float()
    rec["tiles_per_row"] = {"mean": round(float(t.mean()), 2), "p50": q(t, 0.5), "p90": q(t, 0.9), "p99": q(t, 0.99), "max": int(t.max()),
                            "rows": int(t.numel())}
    print(json.dumps({"scene": "c4" if c4 else "c3", "dbg": os.environ.get("GSX_ISECT_DBG", "0"), **rec}), flush=True)


if __name__ == "__main__":
    args = sys.argv[1:] or ["check", "bench"]
    rc = 0
    if "benchone" in args:
        bench_one("c4" in args)
        sys.exit(0)
    if "check" in args:
        rc = 0 if check() else 1
    if "bench" in args:
        bench_all("c4" in args)
    sys.exit(rc)
