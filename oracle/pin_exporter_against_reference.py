#!/usr/bin/env python
"""Golden vectors for gsplat_amd/exporter.py: runs the REFERENCE's gsplat/exporter.py (CPU tensors) on seeded models and
stores inputs + output bytes in tests/golden/exporter_ref.npz; also checks gsplat_amd.exporter against them on the spot.
Run only where the reference checkout exists:  python oracle/pin_exporter_against_reference.py [--ref /root/reference]"""
import argparse
import hashlib
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
from pin_against_reference import install_nerfacc_stub  # noqa: E402


def model(N, K, seed, bad_rows=True):
    g = torch.Generator().manual_seed(seed)
    m = dict(means=torch.randn(N, 3, generator=g) * 2.0, scales=torch.randn(N, 3, generator=g) * 0.7 - 3.0,
             quats=torch.randn(N, 4, generator=g), opacities=torch.randn(N, generator=g) * 3.0,
             sh0=torch.randn(N, 1, 3, generator=g) * 0.8, shN=torch.randn(N, K, 3, generator=g) * 0.5)
    if bad_rows and N > 40:
        m["means"][7, 1] = float("nan")
        m["scales"][19, 0] = float("inf")
        m["opacities"][23] = -12.0  # below the compressed format's opacity threshold
        m["opacities"][31] = float("-inf")
        m["scales"][5] = 30.0  # beyond the +-20 clamp of the compressed scale bounds
    return m


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(os.path.dirname(HERE), "tests", "golden", "exporter_ref.npz"))
    args = ap.parse_args()
    sys.path.insert(0, args.ref)
    sys.dont_write_bytecode = True
    install_nerfacc_stub()
    from gsplat import exporter as R
    from gsplat_amd import exporter as E

    gold = {}
    #        name  N     K   keep bytes?
    cases = [("a", 700, 15, True), ("b", 513, 3, True), ("c", 256, 0, True), ("d", 5000, 8, False)]
    for name, N, K, keep in cases:
        m = model(N, K, seed=N + K)
        for k, v in m.items():
            gold[f"{name}_{k}"] = v.numpy()
        for fmt in ("ply", "splat", "ply_compressed"):
            ref = R.export_splats(**{k: v.clone() for k, v in m.items()}, format=fmt)
            got = E.export_splats(**{k: v.clone() for k, v in m.items()}, format=fmt)
            same = ref == got
            print(f"case {name} N={N} K={K} {fmt:15s}: {len(ref):9d} bytes  sha256 {hashlib.sha256(ref).hexdigest()[:16]}  "
                  f"gsplat_amd identical: {same}")
            assert same, (name, fmt)
            gold[f"{name}_{fmt}_sha256"] = np.frombuffer(hashlib.sha256(ref).digest(), dtype=np.uint8)
            gold[f"{name}_{fmt}_len"] = np.array([len(ref)])
            if keep:
                gold[f"{name}_{fmt}_bytes"] = np.frombuffer(ref, dtype=np.uint8)
    np.savez_compressed(args.out, **gold)
    print(args.out, os.path.getsize(args.out) / 1e6, "MB")


if __name__ == "__main__":
    main()
