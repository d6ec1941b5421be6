#!/usr/bin/env python
"""Summarise a tools/gpu_profile.sh output directory: per-kernel time from rocprofv3 --kernel-trace --stats and
per-launch PMC counters from the separate --pmc passes. Prints text; with --json also writes pmc_traffic.json
(HBM bytes per launch of the compositing kernels, FETCH_SIZE doubled as MI355X_MICROARCH.md 'HBM' prescribes
for wide coalesced reads -- both raw and corrected numbers are reported)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def short(name: str) -> str:
    name = name.replace("void ", "")
    for k in ("gsx::", "at::native::", "(anonymous namespace)::"):
        name = name.replace(k, "")
    return name[:90]


def main():
    out = sys.argv[1]
    write_json = len(sys.argv) > 2 and sys.argv[2] == "--json"
    stats = glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True)
    print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
    for path in stats:
        rows = list(csv.DictReader(open(path)))
        tot = sum(float(r.get("TotalDurationNs", 0) or 0) for r in rows)
        print(f"# {os.path.relpath(path, out)}  total kernel time {tot / 1e6:.3f} ms")
        print(f"{'kernel':92s} {'calls':>6s} {'avg_us':>10s} {'total_ms':>10s} {'%':>6s}")
        for r in sorted(rows, key=lambda r: -float(r.get("TotalDurationNs", 0) or 0))[:40]:
            print(f"{short(r['Name']):92s} {r['Calls']:>6s} {float(r['AverageNs']) / 1e3:10.2f} "
                  f"{float(r['TotalDurationNs']) / 1e6:10.3f} {float(r['Percentage']):6.2f}")
    traffic = {}
    for d in sorted(glob.glob(os.path.join(out, "pmc_*"))):
        if not os.path.isdir(d):
            continue
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            acc = defaultdict(lambda: defaultdict(list))
            for r in csv.DictReader(open(path)):
                acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
            print(f"\n== PMC {os.path.basename(d)} (mean per launch) ==")
            for k, cs in sorted(acc.items(), key=lambda kv: -sum(sum(v) for v in kv[1].values())):
                if "gsx" not in k:
                    continue
                line = ", ".join(f"{c}={sum(v) / len(v):.4g} (n={len(v)})" for c, v in cs.items())
                print(f"{short(k):70s} {line}")
                for c, v in cs.items():
                    traffic.setdefault(short(k), {})[c] = sum(v) / len(v)
    if write_json:
        json.dump(traffic, open(os.path.join(out, "pmc_counters.json"), "w"), indent=1)
        # HBM bytes per launch for bench.py's roofline.traffic: (FETCH_SIZE + WRITE_SIZE) KiB -> bytes. The compositing
        # kernels read through 4..16-byte gathers, for which MI355X_MICROARCH.md gives no FETCH_SIZE correction (the 2x
        # applies to wide 16 B/lane coalesced streams only), so the raw counter is used and the 2x-corrected read side
        # is recorded next to it as an upper bound.
        hbm = {}
        for k, cs in traffic.items():
            if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
                name = k.split("<")[0].split("(")[0].replace("_kernel", "")
                hbm[name] = {"bytes": int((cs["FETCH_SIZE"] + cs["WRITE_SIZE"]) * 1024),
                             "fetch_bytes": int(cs["FETCH_SIZE"] * 1024), "write_bytes": int(cs["WRITE_SIZE"] * 1024),
                             "bytes_if_fetch_x2": int((2 * cs["FETCH_SIZE"] + cs["WRITE_SIZE"]) * 1024)}
                # instruction counters of the same launches (bench.py: roofline.valu_issue_frac)
                sq = {c: cs[c] for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVES", "SQ_WAVE_CYCLES",
                                         "SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_LDS_BANK_CONFLICT", "SQ_INSTS_VALU_FMA_F32",
                                         "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_TRANS_F32",
                                         "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR") if c in cs}
                if sq:
                    hbm[name]["sq"] = sq
        import subprocess

        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        try:
            commit = subprocess.run(["git", "-C", root, "rev-parse", "--short", "HEAD"], capture_output=True, text=True,
                                    timeout=5).stdout.strip() or os.environ.get("GSX_COMMIT")
        except Exception:
            commit = os.environ.get("GSX_COMMIT")
        # the GPU box has no .git: tools/gpu_profile.sh is told the commit through GSX_COMMIT
        sys.path.insert(0, root)
        import bench

        hbm["_meta"] = {"commit": commit or os.environ.get("GSX_COMMIT"), "tag": os.path.basename(os.path.normpath(out)),
                        "workload": "bench.py --lean (c3)", "kernel_sources": bench.raster_source_hash()}
        json.dump(hbm, open(os.path.join(out, "pmc_traffic.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
