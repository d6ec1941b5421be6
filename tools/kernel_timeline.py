#!/usr/bin/env python
"""Kernel timeline of ONE step from a rocprofv3 --kernel-trace csv (the span between the last two launches of an anchor
kernel): start, duration and gap to the previous kernel, in microseconds. usage: kernel_timeline.py <dir> [anchor substring]"""
import csv, glob, sys
d = sys.argv[1]
anchor = sys.argv[2] if len(sys.argv) > 2 else "project_fwd_kernel"
path = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if anchor in r["Kernel_Name"]]
a, b = idx[-2], idx[-1]
t0, prev, busy = int(rows[a]["Start_Timestamp"]), None, 0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    gap = 0 if prev is None else s - prev
    busy += e - s
    name = r["Kernel_Name"].replace("void ", "").replace("gsx::", "").replace("at::native::", "")[:84]
    print(f"{s / 1e3:9.1f} {(e - s) / 1e3:8.1f} gap {gap / 1e3:6.1f}  {name}")
    prev = e
span = int(rows[b]["Start_Timestamp"]) - t0
print(f"step span {span / 1e3:.1f} us, kernels busy {busy / 1e3:.1f} us ({100.0 * busy / span:.1f} %)")
