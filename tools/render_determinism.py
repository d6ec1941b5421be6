"""Diagnostic (run on the GPU box): is the forward + backward of one fixed scene bit-reproducible inside a process - before,
after and WHILE four other threads render other scenes on their own streams? Prints the number of mismatching repeats per phase
(images, intersection keys, flatten ids, projected means). Written in round 4 after one flaky cross-process comparison in the
suite (one process had rendered a different image of the same scene than a fresh one; not reproduced since: 0 mismatches of 180)."""
import os
import sys
import threading

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gsplat_amd as G  # noqa: E402
from _util import make_scene  # noqa: E402

sc, W, H = make_scene(N=6000, C=2, width=208, height=144, seed=6000 % 97, sh_degree=3)
d = {k: v.cuda() for k, v in sc.items()}
g = torch.Generator().manual_seed(7)
bg = torch.rand(2, 3, generator=g).cuda()


def render(tag):
    leaves = {k: d[k].clone().requires_grad_(True) for k in ("means", "quats", "scales", "opacities", "colors")}
    rc, ra, meta = G.rasterization(leaves["means"], leaves["quats"], leaves["scales"], leaves["opacities"], leaves["colors"],
                                   d["viewmats"], d["Ks"], W, H, sh_degree=3, packed=False, backgrounds=bg)
    rc.sum().backward()
    return rc.detach().clone(), meta["isect_ids"].clone(), meta["flatten_ids"].clone(), meta["means2d"].detach().clone()


base = render("base")
bad = 0
for i in range(60):
    out = render(i)
    same = [torch.equal(a, b) for a, b in zip(out, base)]
    if not all(same):
        bad += 1
        print("phase 1 mismatch at", i, same, float((out[0] - base[0]).abs().max()))
print("phase 1:", bad, "mismatches of 60")


def worker(idx):
    s2, W2, H2 = make_scene(N=3000 + 2000 * idx, C=1, width=160 + 16 * idx, height=112, seed=30 + idx)
    dd = {k: v.cuda() for k, v in s2.items()}
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for _ in range(25):
            lv = {k: dd[k].clone().requires_grad_(True) for k in ("means", "quats", "scales", "opacities", "colors")}
            rc, ra, meta = G.rasterization(lv["means"], lv["quats"], lv["scales"], lv["opacities"], lv["colors"], dd["viewmats"],
                                           dd["Ks"], W2, H2)
            rc.sum().backward()
    st.synchronize()


ts = [threading.Thread(target=worker, args=(i,)) for i in range(4)]
[t.start() for t in ts]
[t.join() for t in ts]
bad = 0
for i in range(60):
    out = render(i)
    same = [torch.equal(a, b) for a, b in zip(out, base)]
    if not all(same):
        bad += 1
        print("phase 2 mismatch at", i, same, float((out[0] - base[0]).abs().max()))
print("phase 2 (after the multi-stream threads):", bad, "mismatches of 60")
# threads running WHILE the main thread renders
ts = [threading.Thread(target=worker, args=(i,)) for i in range(4)]
[t.start() for t in ts]
bad = 0
for i in range(60):
    out = render(i)
    same = [torch.equal(a, b) for a, b in zip(out, base)]
    if not all(same):
        bad += 1
        print("phase 3 mismatch at", i, same, float((out[0] - base[0]).abs().max()))
[t.join() for t in ts]
print("phase 3 (threads running concurrently):", bad, "mismatches of 60")
