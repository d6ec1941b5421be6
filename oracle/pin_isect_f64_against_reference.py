#!/usr/bin/env python
"""Golden vectors for the float64 instantiation of intersect_tile: runs the REFERENCE's torch restatement
(gsplat/cuda/_torch_impl.py:356-481 _isect_tiles / _isect_offset_encode, which computes in the dtype of its inputs; the
reference's own test of the double kernel, tests/test_basic.py:1268-1316, compares against exactly this) on seeded float64
rows, checks oracle/oracle.py's float64 branch against it bit for bit, and stores inputs + outputs in
tests/golden/isect_f64_ref.npz. Run only where the reference checkout exists:
    python oracle/pin_isect_f64_against_reference.py [--ref /root/reference]"""
import argparse
import math
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
from pin_against_reference import install_nerfacc_stub  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(os.path.dirname(HERE), "tests", "golden", "isect_f64_ref.npz"))
    args = ap.parse_args()
    sys.path.insert(0, args.ref)
    sys.dont_write_bytecode = True
    install_nerfacc_stub()
    from gsplat.cuda._torch_impl import _isect_offset_encode, _isect_tiles
    from oracle import oracle as O

    torch.manual_seed(42)
    gold = {}
    #            C   N   width height tile
    cases = {"a": (3, 300, 40, 60, 16), "b": (2, 200, 200, 120, 16), "c": (1, 150, 64, 64, 8)}
    for name, (C, N, width, height, ts) in cases.items():
        m = torch.randn(C, N, 2, dtype=torch.float64) * width
        r = torch.randint(0, width, (C, N, 2), dtype=torch.int32)
        d = torch.rand(C, N, dtype=torch.float64)
        tw, th = math.ceil(width / ts), math.ceil(height / ts)
        tpg, ids, fl = _isect_tiles(m, r, d, ts, tw, th)
        off = _isect_offset_encode(ids, C, tw, th)
        o_tpg, o_ids, o_fl = O.isect_tiles(m, r, d, ts, tw, th)
        assert torch.equal(tpg.int().reshape(o_tpg.shape), o_tpg) and torch.equal(ids, o_ids) and torch.equal(fl.int(), o_fl), name
        assert torch.equal(O.isect_offset_encode(o_ids, C, tw, th).reshape(-1), off.reshape(-1).int()), name
        print(f"case {name}: {ids.numel()} intersections, oracle float64 == reference torch restatement (bit for bit)")
        for k, v in dict(means2d=m, radii=r, depths=d, tpg=tpg.int(), ids=ids, flat=fl.int(), offsets=off.int(),
                         cfg=torch.tensor([C, N, width, height, ts, tw, th])).items():
            gold[f"{name}_{k}"] = v.numpy()
    np.savez_compressed(args.out, **gold)
    print(f"ISECT F64 PINNED; wrote {args.out} ({os.path.getsize(args.out)} bytes)")


if __name__ == "__main__":
    main()
