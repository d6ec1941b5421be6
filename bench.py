#!/usr/bin/env python
"""bench.py — headline benchmark of the hot path: gsplat_amd.rasterization() forward + backward.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N>1 launched by torch.distributed.run,
one rank per GPU over RCCL. A step = one fwd+bwd pass of rasterization() over one batch of synthetic input
already resident in HBM. Rank 0 prints ONE JSON line.

Workload at N=1 = BASELINE.json configs[2] ("c3"): 1 M synthetic Gaussians, one 1920x1080 pinhole camera,
SH degree 3, 16x16 tiles, fwd+bwd, loss = render_colors.sum() (reference harness profiling/main.py:132-149).
The same line carries two sub-records timed with the same recipe: "c5" (configs[4]: rasterization_2dgs, 1 M surfels,
RGB+ED + normals + distortion) and "c4_single_gpu" (the per-rank work of configs[3] on ONE GPU: 4 M Gaussians,
4 x 1080p cameras batched) - the N=1 point of the multi-GPU curve below.

Workload at N>1 = BASELINE.json configs[3] ("c4"): a 4 M-Gaussian scene, Gaussian-sharded the way the reference trains
multi-GPU (stride shard [rank::N], examples/simple_trainer.py:326-328; 500 k per rank at N=8), FOUR 1080p cameras per
rank: every rank projects its 4M/N Gaussians against all 4N cameras (16 M (camera, Gaussian) pairs per rank at any N),
the rows travel to the rank that owns the camera (all-gather cameras + all-to-all, gsplat_amd/distributed.py), and
every rank composites its four cameras over the whole 4 M-Gaussian scene. Per-GPU work is therefore fixed and the job
renders 4N images per step: weak scaling, ideal throughput = N x the "c4_single_gpu" value of the N=1 line.
`--workload c3shard` keeps the older shape (the 1 M c3 scene sharded, one camera per rank).

metric = Mpixels/s fwd+bwd = (images * H * W * steps) / wall time, whole job.
roofline  = dominant kernel (the compositing backward) priced against HBM: algorithmic bytes per launch
             / its mean launch duration measured live with HIP events on the launch stream;
            `valu` prices the two compositing kernels against the fp32 vector peak with counted (pixel, Gaussian) pairs.
cpu_baseline = the CPU oracle pipeline (oracle/pipeline.py, OpenMP C + torch-CPU) on the same workload, rank 0, N=1.
"""
import argparse
import json
import math
import os
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WIDTH, HEIGHT, TILE = 1920, 1080, 16
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy ceiling)
FP32_PEAK_TFLOPS = 157.3
N_SIMDS = 256 * 4  # 256 CUs x 4 SIMDs
NAMES = ("means", "quats", "scales", "opacities", "colors")


def make_workload(n_gaussians: int, device, n_cameras: int = 1, seed: int = 0, rank: int = 0, world: int = 1):
    """means uniform in a frustum-filling box z in [1, 20]; log-scales ~ N(log 0.01, 0.5)
    (clipped); opacities U(0.05, 0.95); SH degree-3 coefficients N(0, 0.3) (+0.5 DC); cameras 1920x1080, f=1200.
    `n_gaussians` is the size of the WHOLE scene (same seed on every rank); with world > 1 this rank keeps the stride
    shard [rank::world] and gets its own `n_cameras` cameras (small yaw / shift per camera so that views differ)."""
    g = torch.Generator().manual_seed(seed)
    fx = 1200.0
    z = torch.rand(n_gaussians, generator=g) * 19.0 + 1.0
    x = (torch.rand(n_gaussians, generator=g) - 0.5) * (WIDTH / fx) * z * 1.05
    y = (torch.rand(n_gaussians, generator=g) - 0.5) * (HEIGHT / fx) * z * 1.05
    means = torch.stack([x, y, z], -1)
    quats = torch.nn.functional.normalize(torch.randn(n_gaussians, 4, generator=g), dim=-1)
    log_s = math.log(0.01) + 0.5 * torch.randn(n_gaussians, 3, generator=g)
    scales = torch.exp(log_s.clamp(math.log(0.002), math.log(0.05)))
    opacities = torch.rand(n_gaussians, generator=g) * 0.9 + 0.05
    colors = torch.randn(n_gaussians, 16, 3, generator=g) * 0.3
    colors[:, 0, :] += 0.5
    viewmats = torch.eye(4).repeat(n_cameras, 1, 1)
    for c in range(n_cameras):  # small yaw / shift per camera so views differ
        ang = 0.02 * (c + n_cameras * rank)
        viewmats[c, 0, 0] = math.cos(ang); viewmats[c, 0, 2] = math.sin(ang)
        viewmats[c, 2, 0] = -math.sin(ang); viewmats[c, 2, 2] = math.cos(ang)
        viewmats[c, 0, 3] = 0.05 * (c + n_cameras * rank)
    Ks = torch.tensor([[fx, 0, WIDTH / 2], [0, fx, HEIGHT / 2], [0, 0, 1.0]]).repeat(n_cameras, 1, 1)
    if world > 1:
        means, quats, scales, opacities, colors = (t[rank::world] for t in (means, quats, scales, opacities, colors))
    sc = dict(means=means, quats=quats, scales=scales, opacities=opacities, colors=colors, viewmats=viewmats, Ks=Ks)
    return {k: v.to(device).contiguous() for k, v in sc.items()}, WIDTH, HEIGHT


def raster_source_hash():
    """Fingerprint of the compositing kernels' sources: the PMC traffic in profiles/pmc_traffic.json is only quoted while it
    belongs to THESE sources (the GPU box has no git history to compare commits with)."""
    import hashlib

    h = hashlib.sha256()
    for f in ("raster3d_fwd.hip", "raster3d_bwd.hip", "tile_order.hip", "raster3d.hpp", "common.hpp"):
        with open(os.path.join(ROOT, "gsplat_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def stage_of(entry: str) -> str:
    """C-ABI entries that are one stage of the step table: the per-tile compositing backward is gsx_raster3d_bwd / _ws (with a
    tile-order workspace) / _fill (zero-fills its rows itself) / _seg (long lists in slices); the dense projection backward is
    gsx_project_ewa_bwd / _opac (also reduces the per-view opacity cotangent)."""
    if entry.endswith("_rows") and any(n in entry for n in ("raster3d_bwd", "raster3d_fwd", "sh_fwd")):
        entry = entry[: -len("_rows")]  # the same launch with the array-of-structures rows beside the four arrays
    for tail in ("_ws", "_fill", "_seg", "_opac"):
        if entry.endswith(tail) and any(n in entry for n in ("raster3d_bwd", "raster3d_fwd", "raster2d_bwd", "project_ewa_bwd")):
            return entry[: -len(tail)]
    return entry


def algorithmic_bytes(M, V, P, T, D):
    """SURVEY.md §8(d), fp32, compulsory traffic only."""
    fwd = (28 + 4 * D) * M + 4 * T + (4 * D + 8) * P
    bwd = (28 + 4 * D) * M + (4 * D + 12) * P + 2 * (4 * D + 24) * V
    return fwd, bwd


def _git_head():
    """Commit of this tree: git when there is a checkout, else the stamp __graft_entry__.build() leaves next to the library
    (the GPU box receives a snapshot without .git)."""
    if os.environ.get("GSX_COMMIT"):
        return os.environ["GSX_COMMIT"]
    try:
        out = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True, timeout=5)
        if out.returncode == 0 and out.stdout.strip():
            return out.stdout.strip()
    except Exception:
        pass
    try:
        return json.load(open(os.path.join(ROOT, "gsplat_amd", "csrc", "build_info.json"))).get("commit")
    except Exception:
        return None


def timed(step_fn, steps, warmup, barrier, profile_only=None):
    """W untimed warm-up steps, then exactly K timed steps between barrier + synchronize pairs. Returns
    (seconds, last meta, {entry point: [ms per call]} for the entry points in `profile_only`)."""
    from gsplat_amd import _cabi

    meta = None
    for _ in range(warmup):
        meta = step_fn()
    barrier()
    if os.environ.get("GSPLAT_BENCH_NO_PROFILE"):  # debugging aid: no HIP events inside the timed region
        profile_only = None
    if profile_only:
        _cabi.profile_begin(only=profile_only)
    import gc

    gc_was_on = gc.isenabled()
    gc.disable()  # no cyclic-GC sweep inside the timed steps (a full sweep is 40-60 ms once torch.distributed is imported)
    t0 = time.perf_counter()
    per = []
    for _ in range(steps):
        ts = time.perf_counter()
        meta = step_fn()
        per.append(time.perf_counter() - ts)
    t_enq = time.perf_counter() - t0
    barrier()
    elapsed = time.perf_counter() - t0
    if os.environ.get("GSPLAT_BENCH_DEBUG"):  # host-side enqueue time vs. wall time of the timed steps
        print(f"[bench] enqueue {t_enq / steps * 1e3:.3f} ms/step, wall {elapsed / steps * 1e3:.3f} ms/step; per step: "
              + " ".join(f"{x * 1e3:.1f}" for x in per), file=sys.stderr)
    prof = _cabi.profile_end() if profile_only else {}
    if gc_was_on:
        gc.enable()
    return elapsed, meta, prof


def smi_state_under_load(step_fn, device_index: int, sample: bool, fixed_steps: int = 0):
    """Clock / power state of the GPU WHILE the timed workload runs: `rocm-smi --showclocks --showpower --showmaxpower
    --showperflevel --json` is started and steps are issued until it returns (an idle GPU reports its parked clocks: 94 MHz).
    Lets a reader separate box-to-box spread (different power cap or clock) from run-to-run noise. None if rocm-smi is absent.
    `fixed_steps` > 0 (N > 1): EVERY rank runs exactly that many steps - a step contains collectives, so the ranks must not
    decide for themselves how many to run - and only the sampling rank (`sample`) starts rocm-smi."""
    proc = None
    if sample:
        try:
            proc = subprocess.Popen(["rocm-smi", "-d", str(device_index), "--showclocks", "--showpower", "--showmaxpower",
                                     "--showperflevel", "--json"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
        except Exception:
            proc = None
    if fixed_steps > 0:
        for _ in range(fixed_steps):
            step_fn()
    elif proc is not None:
        t0 = time.perf_counter()
        while proc.poll() is None and time.perf_counter() - t0 < 20.0:
            step_fn()
    torch.cuda.synchronize()
    if proc is None:
        return None
    try:
        out, _ = proc.communicate(timeout=5)
        card = next(iter(json.loads(out).values()))
    except Exception:
        return None
    keep = {}
    for k, v in card.items():
        kl = k.lower()
        if any(w in kl for w in ("sclk", "mclk", "fclk", "power", "performance level")):
            keep[k] = v
    return keep or None


def pair_stats(meta, n_images, W, H):
    """Work counters of the compositing pass (C-ABI gsx_raster3d_pair_stats: instrumentation, replays the forward walk)."""
    from gsplat_amd._cabi import call, ptr

    stats = torch.zeros(8, dtype=torch.int64, device=meta["means2d"].device)
    tw, th = math.ceil(W / TILE), math.ceil(H / TILE)
    call("gsx_raster3d_pair_stats", ptr(meta["means2d"].contiguous()), ptr(meta["conics"].contiguous()),
         ptr(meta["opacities"].contiguous()), ptr(meta["isect_offsets"].contiguous()), ptr(meta["flatten_ids"]),
         n_images, meta["flatten_ids"].numel(), W, H, TILE, tw, th, ptr(stats))
    return [int(v) for v in stats.tolist()]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--gaussians", type=int, default=None, help="Gaussians per rank (overrides the workload's default)")
    ap.add_argument("--cameras", type=int, default=None, help="cameras per rank (overrides the workload's default)")
    ap.add_argument("--workload", choices=("auto", "c3", "c4", "c3shard"), default="auto",
                    help="auto: c3 at N=1, c4 (4 M Gaussians sharded, 4 cameras per rank) at N>1; c3shard: the 1 M c3 scene "
                         "sharded with one camera per rank")
    ap.add_argument("--packed", action="store_true",
                    help="time packed=True as the headline (default: packed=False, the faster layout when nearly every "
                         "Gaussian is visible, as in c3; the other layout is timed too and reported in 'other_layout')")
    ap.add_argument("--dense", action="store_true", help=argparse.SUPPRESS)  # old flag, now the default
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the c5 and c4_single_gpu sub-records")
    ap.add_argument("--force-distributed", action="store_true",
                    help="run the distributed=True code path in a 1-rank RCCL group (measures the seams' overhead)")
    ap.add_argument("--windows", type=int, default=6,
                    help="extra repeats of the K-step timed window after the headline one (reported as windows_ms, value_median, "
                         "value_best); 0 switches them off")
    ap.add_argument("--lean", action="store_true",
                    help="only warmup + timed steps (no stage table, no sub-records, no CPU baseline): for rocprofv3 runs")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: launch the N ranks ourselves, one per GPU, the way gsplat/distributed.py:319-375
        # (cli) spawns its workers - here by re-executing under torch.distributed.run on the loopback address
        import socket

        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
        sys.stdout.flush()
        os.execv(sys.executable, cmd)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1 or args.force_distributed
    # The ONE JSON line goes to the real stdout; everything else written to fd 1 by libraries (RCCL prints a version
    # banner to stdout when the first communicator is created) is sent to stderr.
    json_out = os.fdopen(os.dup(1), "w")
    sys.stdout.flush()
    os.dup2(2, 1)
    assert torch.cuda.is_available(), "bench.py needs a ROCm GPU (there is no CPU fallback for the product path)"
    # GSPLAT_BENCH_REHEARSAL=1: N ranks share GPU 0 and talk over gloo (RCCL refuses two ranks per device). It exists to
    # run the N > 1 code path - sharding, both seams, barriers, max over ranks, the one JSON line - on a box with one GPU;
    # the line is tagged "rehearsal": true and its timings mean nothing (messages cross the host).
    rehearsal = os.environ.get("GSPLAT_BENCH_REHEARSAL") == "1" and world > 1
    if rehearsal:
        local_rank = 0
        os.environ["GSPLAT_AMD_ALLOW_NON_NCCL"] = "1"
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if distributed:
        import torch.distributed as dist

        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if world == 1:  # --force-distributed without a launcher
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if "MASTER_PORT" not in os.environ:
                import socket

                with socket.socket() as sk:
                    sk.bind(("127.0.0.1", 0))
                    os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if rehearsal:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=device)
    n_gpus = world
    assert args.gpus == n_gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    import gsplat_amd
    from gsplat_amd import _cabi

    workload = args.workload
    if workload == "auto":
        workload = "c4" if world > 1 else "c3"
    if workload == "c4":
        n_total, n_cams = 4_000_000, 4
    else:
        n_total, n_cams = 1_000_000, 1
    if args.gaussians:
        n_total = args.gaussians * world
    if args.cameras:
        n_cams = args.cameras
    sc, W, H = make_workload(n_total, device, n_cameras=n_cams, rank=rank, world=world)
    n_local = sc["means"].shape[0]
    leaves = {k: sc[k].clone().requires_grad_(True) for k in NAMES}
    # dense rows by default, also at N>1: nearly every (camera, Gaussian) pair is visible in this scene, and dense
    # rows need no id columns in the all-to-all (48 B/row instead of 64 B/visible row) and no compaction pass
    packed = bool(args.packed)

    def make_step(leaves, sc, packed=packed, distributed=distributed):
        def step():
            for t in leaves.values():
                t.grad = None
            rc, ra, meta = gsplat_amd.rasterization(
                leaves["means"], leaves["quats"], leaves["scales"], leaves["opacities"], leaves["colors"], sc["viewmats"],
                sc["Ks"], W, H, sh_degree=3, packed=packed, tile_size=TILE, distributed=distributed)
            rc.sum().backward()
            return meta
        return step

    step = make_step(leaves, sc)

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    # Python's cyclic GC: a generation-2 sweep walks every tracked object (~170 k once torch.distributed is imported:
    # 40-60 ms measured on the MI355X box) and lands in the middle of a step every few dozen steps. Collect once and
    # move the survivors to the permanent generation, as a long-running trainer would; later sweeps only see objects
    # created by the steps themselves. The sweep runs after the FIRST warm-up step (which creates the long-lived objects),
    # not between the warm-up and the timed window: those 40-60 ms of host time leave the GPU idle, and the steps right after
    # an idle gap run ~4 % slower (r4e / r4f: first window 1.004 / 1.064 ms, its six repeats 0.963 / 1.033 ms).
    import gc

    for i in range(args.warmup):
        meta = step()
        if i == 0:
            gc.collect()
            gc.freeze()
    if args.warmup == 0:
        gc.collect()
        gc.freeze()
    # HIP events (on the launch stream) around the two compositing launches only: the dominant kernels are timed live
    # inside the timed region without the bookkeeping of ~40 event pairs per step perturbing it.
    # The warm-up's last `meta` pins one set of per-step buffers (sorted intersection lists: 0.7 GB at c4). Left alive next to
    # the timed loop's own previous-step `meta` it forces a THIRD set in the second timed step: fresh hipMallocs of that size
    # stall the host for ~45 ms, once (seen as a 55 ms second step on a fresh box). Steady state holds two sets.
    meta = None
    # The headline window carries NO instrumentation (an event pair around each of the two compositing launches costs the step
    # ~50 us: r4d measured 1.05 ms per step with them, 0.996 ms without). The dominant kernels are timed live with HIP events
    # (on the launch stream) in the FIRST REPEAT of the same window, which is reported but kept out of value_median / value_best.
    raster_entries = ("gsx_raster3d_fwd", "gsx_raster3d_bwd", "gsx_raster3d_bwd_ws", "gsx_raster3d_bwd_fill", "gsx_raster3d_fwd_seg",
                      "gsx_raster3d_bwd_seg", "gsx_raster3d_bwd_seg_reuse", "gsx_raster3d_fwd_rows", "gsx_raster3d_bwd_fill_rows")
    elapsed, meta, prof = timed(step, args.steps, 0, barrier, profile_only=raster_entries if args.lean else None)

    def max_over_ranks(x: float) -> float:
        if not distributed:
            return x
        t = torch.tensor([x], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    # The headline stays the window the driver specified (the K steps above). The SAME window is then repeated: a 20-step
    # window is ~20 ms, a 3 % kernel change is invisible in one of them next to box-to-box spread - the repeats give run noise
    # (min / median / max) on THIS box, the SMI state below says which box it was.
    n_windows = 0 if args.lean else args.windows
    exchange_bytes = None
    if distributed:
        from gsplat_amd import distributed as _gd
    windows_s = [max_over_ranks(elapsed)]
    if args.lean:  # profiling runs: one window only, instrumented
        instrumented_window_s = windows_s[0]
    else:
        e_i, _, prof = timed(step, args.steps, 0, barrier, profile_only=raster_entries)
        instrumented_window_s = max_over_ranks(e_i)
    merged = {}  # gsx_raster3d_bwd / _ws / _fill / _seg are one stage: their launch times are pooled, never overwritten
    for k, v in prof.items():
        merged.setdefault(stage_of(k), []).extend(v)
    prof = merged
    if distributed:
        _gd.reset_exchange_stats()
    for _ in range(n_windows):
        e_w, _, _ = timed(step, args.steps, 0, barrier)
        windows_s.append(max_over_ranks(e_w))
    if distributed and n_windows:
        exchange_bytes = _gd.EXCHANGE_STATS["bytes_to_peers"] / (n_windows * args.steps)
    smi = None
    if not args.lean:
        smi = smi_state_under_load(step, local_rank, sample=(rank == 0), fixed_steps=(max(args.steps, 20) if distributed else 0))
    if distributed:
        dist.barrier()
    # per-stage table: a few extra (untimed) steps with an event pair around every C-ABI call
    n_stage = 0 if args.lean else min(5, args.steps)
    _cabi.profile_begin()
    for _ in range(n_stage):
        step()
    stage_prof = _cabi.profile_end()
    # the other row layout (packed <-> dense), same workload, same timing recipe (not the headline)
    other = None
    extras_allowed = rank == 0 and not distributed and not args.lean and not args.no_extra and workload == "c3"
    if not distributed and not args.lean:
        t_other, _, _ = timed(make_step(leaves, sc, packed=not packed), args.steps, min(3, args.warmup), barrier)
        t_other /= args.steps
        other = {"packed": not packed, "ms_per_step": round(t_other * 1e3, 4),
                 "value": round(n_cams * W * H / t_other / 1e6, 2)}
        # Packed rows exist for scenes of which a camera sees a fraction (the reference profiles its 49 M / 107 M-Gaussian scenes
        # packed): the same scene with the far plane pulled in to z = 6 (of [1, 20]: about a quarter of the Gaussians survive),
        # both layouts, same recipe - the workload on which the packed layout is supposed to win.
        if extras_allowed:
            def make_near_step(pk, far):
                def near_step():
                    for t in leaves.values():
                        t.grad = None
                    rc, ra, meta = gsplat_amd.rasterization(
                        leaves["means"], leaves["quats"], leaves["scales"], leaves["opacities"], leaves["colors"], sc["viewmats"],
                        sc["Ks"], W, H, sh_degree=3, packed=pk, tile_size=TILE, far_plane=far)
                    rc.sum().backward()
                    return meta
                return near_step

            # two depths of cut: about a quarter and about a twelfth of the Gaussians visible (where does the packed layout
            # start to pay for its second projection pass and its row-count round trip?)
            for key, far in (("low_visibility", 6.0), ("very_low_visibility", 3.0)):
                low = {}
                for pk in (False, True):
                    t_l, m_l, _ = timed(make_near_step(pk, far), args.steps, 3, barrier)
                    low["packed" if pk else "dense"] = round(t_l / args.steps * 1e3, 4)
                    if pk:
                        low["visible_fraction"] = round(m_l["gaussian_ids"].numel() / float(n_local * n_cams), 4)
                other[key] = dict(low, workload=f"c3 scene, far_plane = {far:g} (ms per step, both layouts)")
    elapsed = windows_s[0]  # max over ranks of the headline window

    images = n_cams * n_gpus
    pixels = images * W * H
    mpix_s = pixels * args.steps / elapsed / 1e6

    # ---- roofline of the dominant kernel (rank 0's launches) -------------------------------------
    M = int(meta["isect_ids"].numel())
    # rows entering compositing on this rank (distributed: the rows received for this rank's cameras)
    V = int((meta["radii"] > 0).all(-1).sum().item()) if not distributed else int(torch.unique(meta["flatten_ids"]).numel())
    P_local = n_cams * W * H
    T_local = n_cams * math.ceil(W / TILE) * math.ceil(H / TILE)
    D = 3
    b_fwd, b_bwd = algorithmic_bytes(M, V, P_local, T_local, D)
    mean_ms = {k: sum(v) / len(v) for k, v in prof.items()}
    per_step_ms = {}
    for k, v in stage_prof.items():
        per_step_ms[stage_of(k)] = per_step_ms.get(stage_of(k), 0.0) + sum(v) / max(n_stage, 1)
    t_fwd = mean_ms.get("gsx_raster3d_fwd", float("nan"))
    t_bwd = mean_ms.get("gsx_raster3d_bwd", float("nan"))
    dom, dom_bytes, dom_ms = ("raster3d_bwd", b_bwd, t_bwd) if not (t_fwd > t_bwd) else ("raster3d_fwd", b_fwd, t_fwd)
    achieved = dom_bytes / (dom_ms * 1e-3) / 1e9
    # HBM bytes per launch from the rocprofv3 PMC passes (tools/gpu_profile.sh -> profiles/pmc_traffic.json); the file
    # names the commit it was measured at, so a stale number is visible as such
    traffic, traffic_src, traffic_raw = None, None, None
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tpath):
        try:
            pmc = json.load(open(tpath))
            # the backward runs as variant W (or T) for <= 4 channels (csrc/raster3d_bwd.hip): counters are filed under the kernel's name
            pmc_dom = pmc.get(dom + "_w") or pmc.get(dom + "_t") or pmc.get(dom) or {}
            # FETCH_SIZE counts a wide coalesced streaming read at half its bytes on gfx950 (MI355X_MICROARCH.md, HBM section:
            # 128-byte requests tallied at 64 B); other access widths and WRITE_SIZE are uncalibrated. The compositing kernels
            # gather 8 / 12 / 4-byte pieces of rows through flatten_ids, so neither reading is exact: `traffic` quotes the
            # CORRECTED figure (fetch x 2 + write: the upper bound), `traffic_raw` the counters as read.
            traffic = pmc_dom.get("bytes_if_fetch_x2", pmc_dom.get("bytes"))
            traffic_raw = {"bytes": pmc_dom.get("bytes"), "fetch_bytes": pmc_dom.get("fetch_bytes"),
                           "write_bytes": pmc_dom.get("write_bytes")}
            meta_pmc = pmc.get("_meta") or {}
            traffic_src = {"file": "profiles/pmc_traffic.json", "measured_at_commit": meta_pmc.get("commit"),
                           "bench_commit": _git_head(), "kernel_sources": raster_source_hash()}
            if meta_pmc.get("kernel_sources") != traffic_src["kernel_sources"]:
                # counters of OTHER kernel sources are not this run's traffic: report none rather than a stale number
                traffic_src["stale"] = f"measured on kernel sources {meta_pmc.get('kernel_sources')}: re-run tools/gpu_profile.sh"
                traffic = None
            elif not (workload == "c3" and not distributed and n_local == 1_000_000 and n_cams == 1 and not packed):
                # the counters were taken on the default c3 scene: they say nothing about another size / layout
                traffic_src["not_applicable"] = "counters belong to the default c3 workload (1 M Gaussians, one camera, dense rows)"
                traffic = None
        except Exception:
            traffic = None
    # Instruction-issue view of the dominant kernel: its VALU wave-instructions per launch (SQ_INSTS_VALU of the same PMC
    # passes) over launch time and the chip's 1024 SIMDs, against the best VALU issue rate tools/issue_rate.hip measured on
    # this GPU (profiles/issue_rate.json: v_add_f32, 8 waves per SIMD) - the roof the compositing kernels are priced by.
    valu_issue = None
    try:
        rate = json.load(open(os.path.join(ROOT, "profiles", "issue_rate.json")))["summary"]
        sq = pmc_dom.get("sq") if traffic is not None else None
        if sq and sq.get("SQ_INSTS_VALU") and dom_ms == dom_ms:
            per_simd_ns = sq["SQ_INSTS_VALU"] / N_SIMDS / (dom_ms * 1e6)
            valu_issue = {"valu_instructions_per_launch": int(sq["SQ_INSTS_VALU"]),
                          "salu_instructions_per_launch": int(sq.get("SQ_INSTS_SALU", 0)) or None,
                          "lds_instructions_per_launch": int(sq.get("SQ_INSTS_LDS", 0)) or None,
                          "achieved_instr_per_ns_per_simd": round(per_simd_ns, 4),
                          "peak_instr_per_ns_per_simd": rate["valu_peak_instr_per_ns_per_simd"],
                          "peak_source": "profiles/issue_rate.json (tools/issue_rate.hip on MI355X)",
                          "frac": round(per_simd_ns / rate["valu_peak_instr_per_ns_per_simd"], 4)}
            # The same count priced by instruction class (rates of profiles/issue_rate.json): fma, mul / add and
            # transcendental are counted by the hardware; everything else - compares, selects, min / max, moves, integer,
            # DPP - is priced at the mean of the full-rate (v_mov, v_and, v_add_u32) and half-rate (v_cmp, v_cndmask,
            # v_min, VOP3 integer, DPP) classes.
            cls = rate.get("by_class_instr_per_ns_per_simd", {})
            if all(k in sq for k in ("SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_ADD_F32",
                                     "SQ_INSTS_VALU_TRANS_F32")) and cls:
                fma, mul, add, trans = (sq["SQ_INSTS_VALU_FMA_F32"], sq["SQ_INSTS_VALU_MUL_F32"], sq["SQ_INSTS_VALU_ADD_F32"],
                                        sq["SQ_INSTS_VALU_TRANS_F32"])
                rest = max(sq["SQ_INSTS_VALU"] - fma - mul - add - trans, 0.0)
                r_fma, r_full = cls["v_fma_f32 (VOP3)"], cls["VOP2 fp32 / int add, mul"]
                r_trans, r_half = cls["v_exp_f32 / v_rcp_f32 / v_permlane32_swap"], cls["v_cmp_e64 -> sgpr pair"]
                r_other = 2.0 / (1.0 / r_full + 1.0 / r_half)
                ns = (fma / r_fma + (mul + add) / r_full + trans / r_trans + rest / r_other) / N_SIMDS
                valu_issue["by_class"] = {"fma": int(fma), "mul_add": int(mul + add), "transcendental": int(trans),
                                          "other": int(rest), "issue_ns_per_simd": round(ns, 1),
                                          "frac_of_launch": round(ns / (dom_ms * 1e6), 4)}
    except Exception:
        valu_issue = None
    # vector-ALU view of the same two kernels: counted (pixel, Gaussian) pairs x (14 + 2 D) flop / time
    valu = None
    if not distributed:
        ps = pair_stats(meta, n_cams, W, H)
        fl = 14 + 2 * D
        valu = {
            "pairs_reference_walk": ps[0], "lane_evaluations": ps[1], "lane_evaluations_open_pixels": ps[2],
            "contributing_pairs": ps[3], "lane_evaluations_of_empty_wave_pairs": ps[4], "flop_per_pair": fl, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
            # the reference kernel's work (every list entry up to saturation, per pixel) done in this kernel's time ...
            "fwd_reference_work_TFLOPs": round(ps[0] * fl / (t_fwd * 1e-3) / 1e12, 2) if t_fwd == t_fwd else None,
            # ... and what the vector ALU really executed (after wave-level culling), forward only
            "fwd_executed_TFLOPs": round(ps[1] * fl / (t_fwd * 1e-3) / 1e12, 2) if t_fwd == t_fwd else None,
        }
        if t_fwd == t_fwd:
            valu["fwd_frac_of_peak_reference_work"] = round(valu["fwd_reference_work_TFLOPs"] / FP32_PEAK_TFLOPS, 4)
            valu["fwd_frac_of_peak_executed"] = round(valu["fwd_executed_TFLOPs"] / FP32_PEAK_TFLOPS, 4)
    def hbm_view(nbytes, ms):
        if not (ms == ms and ms > 0):
            return None
        gbs = nbytes / (ms * 1e-3) / 1e9
        return {"algorithmic_bytes_per_launch": int(nbytes), "launch_ms": round(ms, 4), "achieved": round(gbs, 2),
                "frac": round(gbs / HBM_PEAK_GBS, 5)}

    roofline = {
        "bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_raw": traffic_raw if traffic is not None else None,
        "traffic_basis": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; quoted = 2 x FETCH_SIZE + WRITE_SIZE (gfx950 "
                         "correction for wide reads applied to all of the fetch: an upper bound), raw counters beside it",
        "traffic_source": traffic_src,
        "algorithmic_bytes_per_launch": int(dom_bytes), "launch_ms": round(dom_ms, 4),
        "n_isects": M, "rows": V, "pixels_per_launch": P_local,
        # the forward launch and the pair the north star quotes (rasterize_to_pixels forward + backward), same recipe
        "fwd": hbm_view(b_fwd, t_fwd), "bwd": hbm_view(b_bwd, t_bwd), "fwd_plus_bwd": hbm_view(b_fwd + b_bwd, t_fwd + t_bwd),
        "valu": valu,
        "valu_issue_frac": valu_issue["frac"] if valu_issue else None, "valu_issue": valu_issue,
        "note": "compositing is bound by instruction issue (scalar + vector) and LDS, not by HBM (SURVEY.md 8(d)); see DESIGN.md",
    }

    # HBM-bound stages: algorithmic bytes / measured stage time, against the same 8 TB/s peak
    K_sh, N_rows = 16, n_local * (n_cams * n_gpus if distributed else n_cams)
    stage_bytes = {
        "gsx_project_ewa_fwd": 44 * n_local + 32 * V + 8 * N_rows,
        "gsx_project_ewa_bwd": 88 * V + 40 * n_local,
        "gsx_sh_fwd": (4 * K_sh * D + 12) * V + 4 * D * V,
        "gsx_sh_bwd": (4 * K_sh * D + 12 + 4 * D) * V + 4 * K_sh * D * n_local,
        # fused intersection: count reads 36 B per live row and writes the per-row tile counts; emit
        # reads 44 B per live row, writes and re-reads the 8-byte (depth, row) pairs, and writes 12 B of key + value each
        "gsx_isect_fused_count": 36 * V + 4 * N_rows,
        "gsx_isect_fused_emit_sort": 44 * V + 28 * M,
        # tile-owner-major path (csrc/isect_binned.hip), the same algorithmic bytes: what one pass has to read and write
        "gsx_isect_binned_count": 36 * V + 4 * N_rows,
        "gsx_isect_binned_emit_sort": 44 * V + 28 * M,
        # compositing (not HBM-bound: listed so that every stage of the step has its line)
        "gsx_raster3d_fwd": b_fwd,
        "gsx_raster3d_bwd": b_bwd,
    }
    stage_roofline = {}
    for k, nbytes in stage_bytes.items():
        if per_step_ms.get(k):
            gbs = nbytes / (per_step_ms[k] * 1e-3) / 1e9
            stage_roofline[k.replace("gsx_", "")] = {"achieved_GBps": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 4)}

    if workload == "c3" and not distributed:
        wl = "c3: 1M synthetic Gaussians, 1x1920x1080, SH deg 3, 16x16 tiles, fwd+bwd"
    else:
        wl = (f"{workload}: {n_total} synthetic Gaussians stride-sharded over {n_gpus} rank(s) ({n_local} per rank), "
              f"{n_cams}x1920x1080 camera(s) per rank ({n_cams * n_gpus} images per step), SH deg 3, 16x16 tiles, fwd+bwd"
              + (", distributed=True (all-gather cameras + all-to-all projected rows)" if distributed else ""))
    result = {
        "metric": "Mpixels/s fwd+bwd @1M Gaussians/1080p" if workload != "c4" else
                  "Mpixels/s fwd+bwd @4M Gaussians sharded / 4x1080p cameras per GPU",
        "value": round(mpix_s, 2), "unit": "Mpixels/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": wl, "gaussians_per_gpu": n_local, "cameras_per_gpu": n_cams, "packed": packed,
                   "parallelism": f"gaussian-sharded x{n_gpus}" if distributed else "single"},
        "roofline": roofline,
        **({"rehearsal": True, "rehearsal_note": "all ranks on ONE GPU over gloo: code-path check, timings meaningless"}
           if rehearsal else {}),
        # the driver-specified window first, then its repeats: ms per step of each, and the throughput of the median / best window
        "windows_ms": [round(w / args.steps * 1e3, 4) for w in windows_s],
        "instrumented_window_ms": round(instrumented_window_s / args.steps * 1e3, 4),  # HIP events around the compositing launches
        "value_median": round(pixels * args.steps / sorted(windows_s)[len(windows_s) // 2] / 1e6, 2),
        "value_best": round(pixels * args.steps / min(windows_s) / 1e6, 2),
        "gpu_state_under_load": smi,
        "raster_launch_ms": {"fwd": round(t_fwd, 4) if t_fwd == t_fwd else None,
                             "bwd": round(t_bwd, 4) if t_bwd == t_bwd else None},  # HIP events inside the timed region
        "stage_ms_per_step": {k.replace("gsx_", ""): round(v, 4) for k, v in sorted(per_step_ms.items())},
        "stage_roofline_hbm": stage_roofline,
    }
    if other is not None:
        result["other_layout"] = other
    if exchange_bytes is not None:
        # rank 0's row exchanges (forward messages + reverse exchange of the gradients), bytes handed to OTHER ranks per step
        result["a2a_bytes_per_rank"] = int(exchange_bytes)

    extras = rank == 0 and not distributed and not args.lean and not args.no_extra and workload == "c3"
    # ---- c5 (BASELINE.json configs[4]): 2DGS, same scene, RGB+ED + normals + distortion, same timing recipe -------------
    if extras:
        l5 = {k: sc[k].clone().requires_grad_(True) for k in NAMES}

        def step5():
            for t in l5.values():
                t.grad = None
            rc, ra, rn, sn, rd, rm, m5 = gsplat_amd.rasterization_2dgs(
                l5["means"], l5["quats"], l5["scales"], l5["opacities"], l5["colors"], sc["viewmats"], sc["Ks"], W, H,
                sh_degree=3, packed=False, render_mode="RGB+ED", distloss=True)
            (rc.sum() + rn.sum() + rd.sum()).backward()
            return m5

        steps5 = max(3, args.steps // 2)
        t5, m5, _ = timed(step5, steps5, 3, barrier)  # the clean window gives ms per step, an instrumented repeat the launches
        _, _, p5 = timed(step5, steps5, 0, barrier, profile_only=("gsx_raster2d_fwd", "gsx_raster2d_bwd", "gsx_raster2d_bwd_ws",
                                                                 "gsx_raster2d_bwd_fill"))
        p5 = {stage_of(k): v for k, v in p5.items()}
        M5, D5 = int(m5["isect_ids"].numel()), 4
        V5 = int((m5["radii"] > 0).all(-1).sum().item())
        ms5 = {k.replace("gsx_", ""): round(sum(v) / len(v), 4) for k, v in p5.items()}
        # algorithmic bytes of the 2DGS compositing (52 + 4 D bytes staged per intersection; pixels:
        # colours + alpha + normals + distortion + median (+ ids); rows: 17 + D gradient floats written once, zeroed once)
        b5f = (52 + 4 * D5) * M5 + (4 * D5 + 4 + 12 + 4 + 4 + 8) * W * H
        b5b = (52 + 4 * D5) * M5 + (4 * D5 + 4 + 12 + 4 + 4 + 8 + 4 * D5 + 4 + 12 + 4) * W * H + 2 * 4 * (17 + D5) * V5
        t5b = ms5.get("raster2d_bwd")
        result["c5"] = {
            "workload": "c5: rasterization_2dgs, 1M surfels, 1x1920x1080, SH deg 3, RGB+ED + normals + distortion, fwd+bwd",
            "value": round(W * H * steps5 / t5 / 1e6, 2), "unit": "Mpixels/s", "ms_per_step": round(t5 / steps5 * 1e3, 4),
            "steps": steps5, "n_isects": M5, "raster_launch_ms": ms5,
            "roofline": {"bound": "hbm", "kernel": "raster2d_bwd", "algorithmic_bytes_per_launch": int(b5b),
                         "achieved": round(b5b / (t5b * 1e-3) / 1e9, 2) if t5b else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(b5b / (t5b * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if t5b else None,
                         "fwd_algorithmic_bytes_per_launch": int(b5f)},
        }
        del l5, m5
    # ---- c4 on one GPU: the per-rank work of BASELINE.json configs[3] (4 M Gaussians, 4 cameras batched) ----------------
    # At N > 1 (c4) rank 0 measures the same thing after the timed region - the whole scene, its cameras, no exchange - while
    # the other ranks wait at a barrier: the line then carries its own single-GPU reference (the per-GPU work of this very
    # workload without the exchange) next to `value`; the N = 1 line's `value` is the c3 headline workload, not this one.
    scale_ref = distributed and world > 1 and workload == "c4" and not args.lean and not args.no_extra
    if extras or (scale_ref and rank == 0):
        def c4_single_gpu_record():
            torch.cuda.empty_cache()
            n4, c4n = (4_000_000, 4) if extras else (n_total, n_cams)
            sc4, _, _ = make_workload(n4, device, n_cameras=c4n)
            l4 = {k: sc4[k].clone().requires_grad_(True) for k in NAMES}
            steps4 = max(3, args.steps // 4)
            step4 = make_step(l4, sc4, packed=False, distributed=False)
            t4, m4, _ = timed(step4, steps4, 2, torch.cuda.synchronize)
            _, _, p4 = timed(step4, steps4, 0, torch.cuda.synchronize, profile_only=raster_entries)
            p4m = {}
            for k, v in p4.items():
                p4m.setdefault(stage_of(k), []).extend(v)
            p4 = p4m
            return {
                "workload": f"c4 per-rank work on one GPU: {n4} synthetic Gaussians, {c4n}x1920x1080 cameras batched, SH deg 3, "
                            "fwd+bwd, no exchange",
                "value": round(c4n * W * H * steps4 / t4 / 1e6, 2), "unit": "Mpixels/s",
                "ms_per_step": round(t4 / steps4 * 1e3, 4), "steps": steps4, "n_isects": int(m4["isect_ids"].numel()),
                "raster_launch_ms": {k.replace("gsx_", ""): round(sum(v) / len(v), 4) for k, v in p4.items()},
                "note": "N=1 point of the `--gpus N` curve (same per-GPU work at every N; ideal value at N GPUs = N x this)"
                        + ("; measured on rank 0 of this run, after the timed region" if scale_ref else ""),
            }

        try:
            result["c4_single_gpu"] = c4_single_gpu_record()
        except Exception as e:  # at N > 1 an extra must not strand the other ranks at the barrier below
            if not scale_ref:
                raise
            result["c4_single_gpu"] = {"error": f"{type(e).__name__}: {e}"}
        torch.cuda.empty_cache()
        if scale_ref and result["c4_single_gpu"].get("value"):
            # same workload on both sides of the ratio: the per-GPU work of this run on ONE GPU without the exchange
            result["speedup_vs_1gpu"] = round(result["value"] / result["c4_single_gpu"]["value"], 3)
            result["efficiency"] = round(result["speedup_vs_1gpu"] / n_gpus, 3)
    if scale_ref:
        dist.barrier()
    # ---- c2: the garden scene of BASELINE.json configs[1], by the reference's own profiling recipe ------------------------
    # (profiling/main.py:43-52, 132-149 through tools/bench_reference_profile.py: assets/test_garden.npz cropped and tiled
    # 5 x 5 = 2.8 M Gaussians, 1080p, 3 channels without SH, radius_clip 3; forward and backward timed apart). Unlike the
    # synthetic c3 scene its tile lists are skewed (mean ~390 entries, longest ~8800): it is the real-scene counterweight
    # to the headline number.
    if extras:
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import bench_reference_profile as _brp

            torch.cuda.empty_cache()
            g2 = _brp.run(1, 3, 5, False, 10, device, stages=True, quiet=True)
            t2 = 1.0 / g2["fps_fwd"] + 1.0 / g2["fps_bwd"]
            result["c2_garden"] = {
                "workload": "c2: garden (tests/golden/garden_scene.npz) x 25 = %d Gaussians, 1x1920x1080, 3 channels, "
                            "radius_clip 3, fwd and bwd timed apart (the reference's profiling/main.py recipe)" % g2["n_gaussians"],
                "value": g2["mpix_s_fwd_bwd"], "unit": "Mpixels/s", "ms_fwd_plus_bwd": round(t2 * 1e3, 4),
                "fps_fwd": g2["fps_fwd"], "fps_bwd": g2["fps_bwd"], "n_isects": g2["n_isects"],
                "raster_launch_ms": {"raster3d_fwd": g2["stages"]["fwd_ms"].get("raster3d_fwd"),
                                     "raster3d_bwd": g2["stages"]["bwd_ms"].get("raster3d_bwd_seg", g2["stages"]["bwd_ms"].get("raster3d_bwd"))},
                "stages_ms": g2["stages"],
                "published_titan_rtx_fps_fwd_bwd": g2["published_titan_rtx_fps_fwd_bwd"],
            }
        except Exception as e:
            result["c2_garden"] = {"error": f"{type(e).__name__}: {e}"}
        torch.cuda.empty_cache()
        # ---- the reference's published WIDE-CHANNEL profile row (profile.rst: "32 ch, batch 1", scene_grid 1): 32 colour
        # channels through the matrix-core compositing kernels (csrc/raster3d_{fwd,bwd}_m.hip), same harness as c2
        try:
            g3 = _brp.run(1, 32, 1, False, 10, device, stages=True, quiet=True)
            f_ms, b_ms = g3["stages"]["fwd_ms"].get("raster3d_fwd"), g3["stages"]["bwd_ms"].get("raster3d_bwd")
            M3, P3, V3 = g3["n_isects"], 1920 * 1080, g3["n_gaussians"]  # rows: every Gaussian of the one image (upper bound of the visible ones)
            by_b = (28 + 4 * 32) * M3 + (4 * 32 + 12) * P3 + 2 * (4 * 32 + 24) * V3  # SURVEY.md 8(d) backward bytes at D = 32
            result["garden_32ch"] = {
                "workload": "garden x 1 = %d Gaussians, 1x1920x1080, 32 colour channels, fwd and bwd timed apart (the reference's "
                            "profiling/main.py recipe, its published '32 ch' row)" % g3["n_gaussians"],
                "fps_fwd": g3["fps_fwd"], "fps_bwd": g3["fps_bwd"], "n_isects": M3,
                "raster_launch_ms": {"raster3d_fwd": f_ms, "raster3d_bwd": b_ms},
                "bwd_hbm": {"algorithmic_bytes_per_launch": int(by_b), "frac": round(by_b / (b_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)} if b_ms else None,
                "published_titan_rtx_fps_fwd_bwd": g3["published_titan_rtx_fps_fwd_bwd"],
            }
        except Exception as e:
            result["garden_32ch"] = {"error": f"{type(e).__name__}: {e}"}
        torch.cuda.empty_cache()
    # ---- the training step around the rasterizer: tools/train_step_bench.py ----------------
    if extras:
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import train_step_bench

            torch.cuda.empty_cache()
            result["train_step"] = train_step_bench.run(steps=100, device=device)
            torch.cuda.empty_cache()
            # what the SSIM term of the reference's loss costs: the same step with L1 alone
            l1_only = train_step_bench.run(steps=100, device=device, ssim_lambda=0.0)
            result["train_step"]["l1_only_ms_per_step"] = l1_only["ms_per_step"]
            torch.cuda.empty_cache()
            # the same step with the coefficients handed over as they are stored, (sh0, shN): the split SH kernels read and
            # write them in place - no torch.cat each way (an extension of rasterization()'s `colors` argument)
            split = train_step_bench.run(steps=100, device=device, split_sh=True)
            result["train_step"]["split_sh"] = {k: split[k] for k in ("ms_per_step", "steps_per_s", "refinement_step_ms",
                                                                        "gaussians_after", "final_loss")}
        except Exception as e:
            result["train_step"] = {"error": f"{type(e).__name__}: {e}"}
        torch.cuda.empty_cache()

    # ---- CPU baseline (rank 0, N=1): the oracle pipeline, one fwd+bwd step of the SAME workload -------------
    # Bounded by construction: one step of c3 is ~10-30 s on <= 32 host threads (more threads are slower for these
    # OpenMP/torch-CPU loops: fork/join + atomic contention), so the default bench run stays within a few minutes.
    if rank == 0 and not distributed and not args.no_cpu_baseline and not args.lean and workload == "c3":
        from oracle import oracle as _oracle
        from oracle.pipeline import rasterization_cpu

        cores = max(1, min(os.cpu_count() or 1, 32))
        _oracle.set_threads(cores)
        cs = {k: v.detach().cpu() for k, v in sc.items()}
        ref = rasterization_cpu(cs["means"], cs["quats"], cs["scales"], cs["opacities"], cs["colors"], cs["viewmats"],
                                cs["Ks"], W, H, sh_degree=3, render_mode="RGB")
        t_cpu = ref["t_fwd"] + ref["t_bwd"]
        result["cpu_baseline"] = {
            "value": round(n_cams * W * H / t_cpu / 1e6, 4), "unit": "Mpixels/s", "cores": cores, "kind": "port",
            "sample": "1 full fwd+bwd step of the same 1M-Gaussian/1080p workload (oracle/pipeline.py: OpenMP C "
                      f"compositing/intersection + torch-CPU projection/SH) on {cores} host threads",
            "t_fwd_s": round(ref["t_fwd"], 3), "t_bwd_s": round(ref["t_bwd"], 3), "n_isects": ref["n_isects"],
        }
        # the reference's OWN CPU path (gsplat/cuda/_torch_impl.py) can only run configs[0] (c1): timed HERE, on this host's
        # cores, from the archive of the reference's Python files that build() stages and that travels with the snapshot
        # (oracle/_ref/reference_py.zip; a subprocess, so that the reference's `gsplat` package never enters this process).
        # Without the archive the committed record of the build container is quoted, labelled as such.
        ref_zip = os.path.join(ROOT, "oracle", "_ref", "reference_py.zip")
        c1 = None
        if os.path.exists(ref_zip):
            try:
                out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "time_c1_reference_cpu.py"), "--ref", ref_zip,
                                      "--out", "-", "--reps", "3", "--where", "this run's host"], capture_output=True, text=True,
                                     timeout=180)
                c1 = json.loads(out.stdout.strip().splitlines()[-1])
                c1["source"] = "measured in this run (tools/time_c1_reference_cpu.py on oracle/_ref/reference_py.zip)"
            except Exception as e:
                c1 = None
                result["cpu_baseline"]["c1_reference_error"] = f"{type(e).__name__}: {e}"[:300]
        c1p = os.path.join(ROOT, "profiles", "c1_reference_cpu.json")
        if c1 is None and os.path.exists(c1p):
            try:
                c1 = json.load(open(c1p))
                c1["source"] = "profiles/c1_reference_cpu.json (not measured in this run)"
            except Exception:
                c1 = None
        if c1 is not None:
            result["cpu_baseline"]["c1_reference"] = {
                "config": c1["config"], "reference_mpixels_per_s": c1["reference_cpu"]["mpixels_per_s"],
                "reference_s_per_step": c1["reference_cpu"]["s_per_step"], "port_mpixels_per_s": c1["port_cpu"]["mpixels_per_s"],
                "cores": c1["host"].get("torch_threads"), "host": c1["host"], "kind": "reference", "source": c1["source"]}
    if rank == 0:
        print(json.dumps(result), file=json_out, flush=True)
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
