"""PngCompression: quantise the splat parameters onto square 8/16-bit images stored as PNG, cluster the higher SH bands.

Same class, fields, directory layout and ``meta.json`` as the reference (gsplat/compression/png_compression.py:30-147): a
directory written here decompresses with the reference and the other way round. Per field:

    means       log-transform, 16 bits per sample: ``means_l.png`` (low byte) + ``means_u.png`` (high byte)   (:193-262)
    scales, quats (normalised), opacities, sh0     8 bits per sample, ``<name>.png``                              (:150-190)
    shN         K-means (L1 distance, 2^16 clusters), centroids quantised to 6 bits: ``shN.npz``                 (:306-438)
    anything else   ``<name>.npz`` (numpy's compressed container)                                                 (:276-303)

What this module does not take from the reference are its three third-party dependencies, none of which is in this image:
``imageio`` -> the PNG codec in ``_png.py``; ``torchpq.clustering.KMeans`` -> ``kmeans_l1`` below (Lloyd's iteration with
L1 assignment, chunked); ``plas`` -> see ``sort.py``.
"""
from __future__ import annotations

import json
import os
from dataclasses import dataclass
from typing import Any, Dict, Tuple

import numpy as np
import torch
from torch import Tensor

from . import _png
from .sort import sort_splats


def log_transform(x: Tensor) -> Tensor:
    """sign(x) log(1 + |x|) (reference gsplat/utils.py:151)."""
    return torch.sign(x) * torch.log1p(torch.abs(x))


def inverse_log_transform(y: Tensor) -> Tensor:
    return torch.sign(y) * torch.expm1(torch.abs(y))


def _dtype_name(t: Tensor) -> str:
    return str(t.dtype).split(".")[1]


def _empty_meta(params: Tensor) -> Dict[str, Any]:
    return {"shape": list(params.shape), "dtype": _dtype_name(params)}


def _is_empty(meta: Dict[str, Any]) -> bool:
    return not np.all(meta["shape"])


def _quantise_grid(params: Tensor, side: int, bits: int) -> Tuple[np.ndarray, Tensor, Tensor]:
    """[N, ...] -> integer image [side, side, C] with per-channel range normalisation (constant channels map to 0)."""
    grid = params.detach().reshape(side, side, -1).float()
    mins, maxs = grid.amin(dim=(0, 1)), grid.amax(dim=(0, 1))
    span = (maxs - mins)
    norm = torch.where(span > 0, (grid - mins) / span.clamp_min(1e-38), torch.zeros_like(grid))
    img = (norm.cpu().numpy() * (2**bits - 1)).round()
    return img, mins, maxs


def _dequantise(img: np.ndarray, meta: Dict[str, Any], bits: int) -> Tensor:
    norm = torch.from_numpy(img.astype(np.float64) / (2**bits - 1))
    # float32 bounds, float64 samples: the arithmetic (and so every decoded bit) of the reference's decoders (:183-189)
    mins, maxs = torch.tensor(meta["mins"]), torch.tensor(meta["maxs"])
    if norm.dim() == 2:
        norm = norm[..., None]
    return (norm * (maxs - mins) + mins).reshape(meta["shape"]).to(getattr(torch, meta["dtype"]))


class _Png8:
    """8 bits per sample, one PNG."""

    @staticmethod
    def compress(out_dir: str, name: str, params: Tensor, side: int, **_) -> Dict[str, Any]:
        if params.numel() == 0:
            return _empty_meta(params)
        img, mins, maxs = _quantise_grid(params, side, 8)
        img = img.astype(np.uint8)
        _png.write(os.path.join(out_dir, f"{name}.png"), img[..., 0] if img.shape[-1] == 1 else img)
        return {**_empty_meta(params), "mins": mins.tolist(), "maxs": maxs.tolist()}

    @staticmethod
    def decompress(in_dir: str, name: str, meta: Dict[str, Any]) -> Tensor:
        if _is_empty(meta):
            return torch.zeros(meta["shape"], dtype=getattr(torch, meta["dtype"]))
        return _dequantise(_png.read(os.path.join(in_dir, f"{name}.png")), meta, 8)


class _Png16:
    """16 bits per sample as two 8-bit PNGs: low bytes and high bytes."""

    @staticmethod
    def compress(out_dir: str, name: str, params: Tensor, side: int, **_) -> Dict[str, Any]:
        if params.numel() == 0:
            return _empty_meta(params)
        img, mins, maxs = _quantise_grid(params, side, 16)
        img = img.astype(np.uint16)
        for suffix, plane in (("l", img & 0xFF), ("u", img >> 8)):
            plane = plane.astype(np.uint8)
            _png.write(os.path.join(out_dir, f"{name}_{suffix}.png"), plane[..., 0] if plane.shape[-1] == 1 else plane)
        return {**_empty_meta(params), "mins": mins.tolist(), "maxs": maxs.tolist()}

    @staticmethod
    def decompress(in_dir: str, name: str, meta: Dict[str, Any]) -> Tensor:
        if _is_empty(meta):
            return torch.zeros(meta["shape"], dtype=getattr(torch, meta["dtype"]))
        lo = _png.read(os.path.join(in_dir, f"{name}_l.png")).astype(np.uint16)
        hi = _png.read(os.path.join(in_dir, f"{name}_u.png")).astype(np.uint16)
        return _dequantise((hi << 8) + lo, meta, 16)


class _Npz:
    @staticmethod
    def compress(out_dir: str, name: str, params: Tensor, **_) -> Dict[str, Any]:
        path = os.path.join(out_dir, f"{name}.npz")
        os.makedirs(os.path.dirname(path), exist_ok=True)
        np.savez_compressed(path, arr=params.detach().cpu().numpy())
        return _empty_meta(params)

    @staticmethod
    def decompress(in_dir: str, name: str, meta: Dict[str, Any]) -> Tensor:
        arr = np.load(os.path.join(in_dir, f"{name}.npz"))["arr"]
        return torch.from_numpy(arr).reshape(meta["shape"]).to(getattr(torch, meta["dtype"]))


def kmeans_l1(x: Tensor, n_clusters: int, n_iters: int = 10, seed: int = 0, max_chunk_elems: int = 1 << 27,
              verbose: bool = False) -> Tuple[Tensor, Tensor]:
    """Lloyd's iteration with Manhattan-distance assignment (what the reference asks of torchpq's KMeans,
    ``distance="manhattan"``; png_compression.py:348-351). x [N, D] -> (centroids [K, D], labels int64 [N]),
    K = min(n_clusters, N). Initial centroids: K distinct rows drawn with `seed`; a cluster that loses all its members
    keeps its centroid. The assignment is evaluated in row chunks of at most `max_chunk_elems` / (K D) rows."""
    n, d = x.shape
    k = min(n_clusters, n)
    gen = torch.Generator(device="cpu").manual_seed(seed)
    centroids = x[torch.randperm(n, generator=gen)[:k].to(x.device)].clone()
    labels = torch.zeros(n, dtype=torch.int64, device=x.device)
    chunk = max(1, min(n, max_chunk_elems // max(1, k * d)))
    for it in range(max(1, n_iters)):
        for lo in range(0, n, chunk):
            labels[lo:lo + chunk] = torch.cdist(x[lo:lo + chunk], centroids, p=1).argmin(dim=1)
        sums = torch.zeros_like(centroids).index_add_(0, labels, x)
        counts = torch.zeros(k, dtype=x.dtype, device=x.device).index_add_(0, labels, torch.ones(n, dtype=x.dtype, device=x.device))
        new = torch.where(counts[:, None] > 0, sums / counts[:, None].clamp_min(1), centroids)
        shift = float((new - centroids).abs().max())
        centroids = new
        if verbose:
            print(f"kmeans_l1: iteration {it + 1}/{n_iters}, largest centroid move {shift:.3e}")
        if shift == 0.0:
            break
    for lo in range(0, n, chunk):  # labels of the final centroids
        labels[lo:lo + chunk] = torch.cdist(x[lo:lo + chunk], centroids, p=1).argmin(dim=1)
    return centroids, labels


class _KMeans:
    """Cluster the rows, keep 16-bit labels and centroids quantised to `quantization` bits."""

    @staticmethod
    def compress(out_dir: str, name: str, params: Tensor, n_clusters: int = 65536, quantization: int = 6,
                 eps: float = 1e-6, verbose: bool = True, n_iters: int = 10, **_) -> Dict[str, Any]:
        if params.numel() == 0:
            return _empty_meta(params)
        if n_clusters > 65536:
            raise ValueError("labels are stored as uint16: at most 65536 clusters")
        centroids, labels = kmeans_l1(params.detach().reshape(params.shape[0], -1).float(), n_clusters, n_iters=n_iters,
                                      verbose=verbose)
        mins, maxs = centroids.min() + eps, centroids.max()
        span = (maxs - mins)
        norm = ((centroids - mins) / span) if float(span) > 0 else torch.zeros_like(centroids)
        quant = (norm.clamp(0, 1).cpu().numpy() * (2**quantization - 1)).round().astype(np.uint8)
        np.savez_compressed(os.path.join(out_dir, f"{name}.npz"), centroids=quant,
                            labels=labels.cpu().numpy().astype(np.uint16))
        return {**_empty_meta(params), "mins": mins.tolist(), "maxs": maxs.tolist(), "quantization": quantization}

    @staticmethod
    def decompress(in_dir: str, name: str, meta: Dict[str, Any]) -> Tensor:
        if _is_empty(meta):
            return torch.zeros(meta["shape"], dtype=getattr(torch, meta["dtype"]))
        blob = np.load(os.path.join(in_dir, f"{name}.npz"))
        centroids = torch.from_numpy(blob["centroids"] / (2 ** meta["quantization"] - 1))
        mins, maxs = torch.tensor(meta["mins"]), torch.tensor(meta["maxs"])
        centroids = centroids * (maxs - mins) + mins
        labels = torch.from_numpy(blob["labels"].astype(np.int64))
        return centroids[labels].reshape(meta["shape"]).to(getattr(torch, meta["dtype"]))


_CODECS = {"means": _Png16, "scales": _Png8, "quats": _Png8, "opacities": _Png8, "sh0": _Png8, "shN": _KMeans}


@dataclass
class PngCompression:
    """Compress / decompress a dictionary of (pre-activation) splat parameters - "means", "scales", "quats", "opacities",
    "sh0", "shN", plus any extra fields (stored as .npz) - into a directory (reference PngCompression, :30-147).

    The number of splats must be a perfect square to fill the images: otherwise the lowest-opacity splats are dropped (and
    a warning printed), like the reference. ``use_sort`` orders the splats for compressibility first (``sort.py``)."""

    use_sort: bool = True
    verbose: bool = True

    def compress(self, compress_dir: str, splats: Dict[str, Tensor]) -> None:
        os.makedirs(compress_dir, exist_ok=True)
        splats = dict(splats)  # the caller's dictionary keeps its tensors
        splats["means"] = log_transform(splats["means"])
        splats["quats"] = torch.nn.functional.normalize(splats["quats"], dim=-1)
        n = len(splats["means"])
        side = int(n**0.5)
        surplus = n - side * side
        if surplus:
            keep = torch.argsort(splats["opacities"], descending=True)[:-surplus]
            splats = {k: v[keep] for k, v in splats.items()}
            print(f"Warning: Number of Gaussians was not square. Removed {surplus} Gaussians.")
        if self.use_sort:
            splats = sort_splats(splats, verbose=self.verbose)
        meta = {}
        for name, params in splats.items():
            meta[name] = _CODECS.get(name, _Npz).compress(compress_dir, name, params, side=side, verbose=self.verbose)
        with open(os.path.join(compress_dir, "meta.json"), "w") as f:
            json.dump(meta, f)

    def decompress(self, compress_dir: str) -> Dict[str, Tensor]:
        with open(os.path.join(compress_dir, "meta.json")) as f:
            meta = json.load(f)
        splats = {name: _CODECS.get(name, _Npz).decompress(compress_dir, name, m) for name, m in meta.items()}
        splats["means"] = inverse_log_transform(splats["means"])
        return splats
