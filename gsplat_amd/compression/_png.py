"""A small, dependency-free PNG codec (zlib + numpy) for 8-bit images with 1-4 channels.

The reference's PngCompression writes and reads its parameter images through ``imageio`` (gsplat/compression/
png_compression.py:163-190, 231-262), a third-party package that is not part of this image. The files are plain PNGs, so
this module produces and consumes exactly that format: the writer emits greyscale / grey+alpha / RGB / RGBA, 8 bits per
sample, non-interlaced, with a per-row choice among the None / Sub / Up filters (the ones whose inverse is a vector operation);
the reader accepts every non-interlaced 8-bit PNG of those colour types with any of the five filters (PNG spec, section 9),
i.e. also the files imageio / PIL write, so compressed directories are interchangeable with the reference's in both directions.
"""
from __future__ import annotations

import struct
import zlib

import numpy as np

_SIGNATURE = b"\x89PNG\r\n\x1a\n"
_COLOR_TYPE = {1: 0, 2: 4, 3: 2, 4: 6}  # channels -> PNG colour type
_CHANNELS = {v: k for k, v in _COLOR_TYPE.items()}


def _chunk(tag: bytes, payload: bytes) -> bytes:
    return struct.pack(">I", len(payload)) + tag + payload + struct.pack(">I", zlib.crc32(tag + payload) & 0xFFFFFFFF)


def _row_cost(residual: np.ndarray) -> np.ndarray:
    """The usual heuristic for choosing a row's filter: sum of the residual bytes read as signed values."""
    r = residual.astype(np.int16)
    return np.minimum(r, 256 - r).sum(axis=1)


def encode(img: np.ndarray, level: int = 9) -> bytes:
    """uint8 [H, W] or [H, W, C] (C in 1..4) -> PNG file bytes."""
    if img.dtype != np.uint8:
        raise TypeError(f"PNG encoder takes uint8 samples, got {img.dtype}")
    if img.ndim == 2:
        img = img[:, :, None]
    if img.ndim != 3 or img.shape[2] not in _COLOR_TYPE:
        raise ValueError(f"PNG encoder takes [H, W] or [H, W, 1..4] arrays, got shape {tuple(img.shape)}")
    h, w, c = img.shape
    raw = np.ascontiguousarray(img).reshape(h, w * c)
    left = np.zeros_like(raw)
    left[:, c:] = raw[:, :-c]
    up = np.zeros_like(raw)
    up[1:] = raw[:-1]
    candidates = np.stack([raw, raw - left, raw - up])  # filter types 0, 1, 2 (uint8 arithmetic wraps modulo 256)
    choice = np.argmin(np.stack([_row_cost(x) for x in candidates]), axis=0).astype(np.uint8)
    rows = np.empty((h, 1 + w * c), dtype=np.uint8)
    rows[:, 0] = choice
    rows[:, 1:] = candidates[choice, np.arange(h)]
    header = struct.pack(">IIBBBBB", w, h, 8, _COLOR_TYPE[c], 0, 0, 0)
    return _SIGNATURE + _chunk(b"IHDR", header) + _chunk(b"IDAT", zlib.compress(rows.tobytes(), level)) + _chunk(b"IEND", b"")


def _paeth_row(filt: np.ndarray, prev: np.ndarray, c: int) -> np.ndarray:
    out = np.zeros_like(filt)
    f, p, o = filt.astype(np.int32), prev.astype(np.int32), out.astype(np.int32)
    for i in range(filt.shape[0]):
        a = o[i - c] if i >= c else 0
        b = p[i]
        d = p[i - c] if i >= c else 0
        est = a + b - d
        pa, pb, pd = abs(est - a), abs(est - b), abs(est - d)
        pred = a if (pa <= pb and pa <= pd) else (b if pb <= pd else d)
        o[i] = (f[i] + pred) & 0xFF
    return o.astype(np.uint8)


def _average_row(filt: np.ndarray, prev: np.ndarray, c: int) -> np.ndarray:
    f, p = filt.astype(np.int32), prev.astype(np.int32)
    o = np.zeros_like(f)
    for i in range(filt.shape[0]):
        a = o[i - c] if i >= c else 0
        o[i] = (f[i] + ((a + p[i]) >> 1)) & 0xFF
    return o.astype(np.uint8)


def decode(data: bytes) -> np.ndarray:
    """PNG file bytes -> uint8 [H, W] (one channel) or [H, W, C]."""
    if data[:8] != _SIGNATURE:
        raise ValueError("not a PNG file")
    pos, idat, header = 8, [], None
    while pos < len(data):
        (length,), tag = struct.unpack(">I", data[pos:pos + 4]), data[pos + 4:pos + 8]
        payload = data[pos + 8:pos + 8 + length]
        (crc,) = struct.unpack(">I", data[pos + 8 + length:pos + 12 + length])
        if zlib.crc32(tag + payload) & 0xFFFFFFFF != crc:
            raise ValueError(f"PNG chunk {tag!r}: CRC mismatch")
        if tag == b"IHDR":
            header = struct.unpack(">IIBBBBB", payload)
        elif tag == b"IDAT":
            idat.append(payload)
        elif tag == b"IEND":
            break
        pos += 12 + length
    if header is None:
        raise ValueError("PNG without IHDR")
    w, h, depth, ctype, _comp, _filt, interlace = header
    if depth != 8 or ctype not in _CHANNELS or interlace != 0:
        raise ValueError(f"unsupported PNG (bit depth {depth}, colour type {ctype}, interlace {interlace}): "
                         "8-bit non-interlaced grey / grey+alpha / RGB / RGBA only")
    c = _CHANNELS[ctype]
    rows = np.frombuffer(zlib.decompress(b"".join(idat)), dtype=np.uint8).reshape(h, 1 + w * c)
    out = np.zeros((h, w * c), dtype=np.uint8)
    zero = np.zeros(w * c, dtype=np.uint8)
    for y in range(h):
        ftype, filt = int(rows[y, 0]), rows[y, 1:]
        prev = out[y - 1] if y else zero
        if ftype == 0:
            out[y] = filt
        elif ftype == 1:  # Sub: running sum along the row, per channel, modulo 256
            out[y] = np.cumsum(filt.reshape(w, c), axis=0, dtype=np.uint8).reshape(-1)
        elif ftype == 2:
            out[y] = filt + prev
        elif ftype == 3:
            out[y] = _average_row(filt, prev, c)
        elif ftype == 4:
            out[y] = _paeth_row(filt, prev, c)
        else:
            raise ValueError(f"PNG row {y}: unknown filter type {ftype}")
    return out.reshape(h, w) if c == 1 else out.reshape(h, w, c)


def write(path: str, img: np.ndarray) -> None:
    with open(path, "wb") as f:
        f.write(encode(img))


def read(path: str) -> np.ndarray:
    with open(path, "rb") as f:
        return decode(f.read())
