#!/usr/bin/env python
"""Golden vectors for gsplat_amd/compression: runs the REFERENCE's gsplat/compression/png_compression.py per-field codecs on
a seeded model and stores (a) the reference-written directory, flattened into tests/golden/png_compression_ref.npz (file
bytes + inputs + the reference's own decompression of them), and checks on the spot that (b) gsplat_amd decompresses the
reference's files to the same tensors, (c) the reference decompresses gsplat_amd's files to the same tensors.

The reference imports three packages this image does not have. `imageio` is only its PNG file I/O: this script registers a
stand-in `imageio.v2` with imwrite / imread implemented by PIL (a different, independent PNG implementation from
gsplat_amd/compression/_png.py, which is the point). `torchpq` (K-means) and `plas` (sorting) are not replaceable: the
K-means COMPRESSION and the sort are therefore not pinned (parity unpinned for those two pieces; the K-means file format is
pinned through the reference's _decompress_kmeans, which needs neither).
Run only where the reference checkout exists:  python oracle/pin_png_compression_against_reference.py [--ref /root/reference]"""
import argparse
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
from pin_against_reference import install_nerfacc_stub  # noqa: E402


def install_imageio_stub():
    from PIL import Image

    v2 = types.ModuleType("imageio.v2")
    v2.imwrite = lambda path, img: Image.fromarray(np.asarray(img)).save(path, format="PNG", optimize=True)
    v2.imread = lambda path: np.array(Image.open(path))
    pkg = types.ModuleType("imageio")
    pkg.v2 = v2
    sys.modules["imageio"], sys.modules["imageio.v2"] = pkg, v2


def model(side, K, seed):
    g = torch.Generator().manual_seed(seed)
    N = side * side
    return dict(means=torch.randn(N, 3, generator=g) * 4.0, scales=torch.randn(N, 3, generator=g) * 0.7 - 3.0,
                quats=torch.nn.functional.normalize(torch.randn(N, 4, generator=g), dim=-1),
                opacities=torch.randn(N, generator=g) * 3.0, sh0=torch.randn(N, 1, 3, generator=g) * 0.8,
                shN=torch.randn(N, K, 3, generator=g) * 0.2, extra=torch.randn(N, 2, generator=g))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(os.path.dirname(HERE), "tests", "golden", "png_compression_ref.npz"))
    args = ap.parse_args()
    sys.path.insert(0, args.ref)
    sys.dont_write_bytecode = True
    install_nerfacc_stub()
    install_imageio_stub()
    from gsplat.compression import png_compression as R
    from gsplat.utils import log_transform
    from gsplat_amd.compression import png_compression as E

    side, K = 24, 8
    m = model(side, K, seed=11)
    ref_codec = {"means": (R._compress_png_16bit, R._decompress_png_16bit), "scales": (R._compress_png, R._decompress_png),
                 "quats": (R._compress_png, R._decompress_png), "opacities": (R._compress_png, R._decompress_png),
                 "sh0": (R._compress_png, R._decompress_png), "extra": (R._compress_npz, R._decompress_npz)}
    gold = {f"in_{k}": v.numpy() for k, v in m.items()}
    with tempfile.TemporaryDirectory() as ref_dir, tempfile.TemporaryDirectory() as our_dir:
        pre = dict(m)
        pre["means"] = log_transform(m["means"])
        meta = {}
        for name, (comp, _) in ref_codec.items():
            meta[name] = comp(ref_dir, name, pre[name], n_sidelen=side, verbose=False)
            meta[name]["shape"] = list(meta[name]["shape"])
        # (b) the reference's files through gsplat_amd's decoders
        for name, (_, decomp) in ref_codec.items():
            want = decomp(ref_dir, name, meta[name])
            got = E._CODECS.get(name, E._Npz).decompress(ref_dir, name, meta[name])
            assert got.shape == want.shape and got.dtype == want.dtype, name
            err = float((got.double() - want.double()).abs().max())
            print(f"reference files -> gsplat_amd decode   {name:10s} max |diff| {err:.3e}")
            assert err == 0.0, name
            gold[f"ref_out_{name}"] = want.numpy()
        # (c) gsplat_amd's files through the reference's decoders
        our_meta = {}
        for name in ref_codec:
            our_meta[name] = E._CODECS.get(name, E._Npz).compress(our_dir, name, pre[name], side=side, verbose=False)
        for name, (_, decomp) in ref_codec.items():
            assert our_meta[name].get("mins") == meta[name].get("mins") and our_meta[name].get("maxs") == meta[name].get("maxs"), name
            want = decomp(ref_dir, name, meta[name])
            got = decomp(our_dir, name, our_meta[name])
            err = float((got.double() - want.double()).abs().max())
            print(f"gsplat_amd files -> reference decode   {name:10s} max |diff| {err:.3e}")
            assert err == 0.0, name
        # K-means container: gsplat_amd compresses, the reference's _decompress_kmeans (no third-party code) decodes
        km = E._KMeans.compress(our_dir, "shN", m["shN"], n_clusters=64, verbose=False)
        want = R._decompress_kmeans(our_dir, "shN", km)
        got = E._KMeans.decompress(our_dir, "shN", km)
        err = float((got.double() - want.double()).abs().max())
        print(f"gsplat_amd shN.npz -> reference decode  max |diff| {err:.3e}")
        assert err == 0.0
        for fn in sorted(os.listdir(ref_dir)):
            gold["file_" + fn] = np.frombuffer(open(os.path.join(ref_dir, fn), "rb").read(), dtype=np.uint8)
        gold["meta_json"] = np.frombuffer(__import__("json").dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(args.out, **gold)
    print(f"PNG COMPRESSION PINNED (PNG / NPZ codecs both ways; K-means container); wrote {args.out} "
          f"({os.path.getsize(args.out)} bytes)")


if __name__ == "__main__":
    main()
