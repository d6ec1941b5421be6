"""Drop-in for the reference's compiled extension module ``gsplat.csrc``.

``gsplat/cuda/_backend.py:29-31`` does ``from gsplat import csrc as _C`` before it ever looks for
nvcc. Dropping a one-line ``gsplat/csrc.py`` that re-exports this module makes
the reference's own Python (``_wrapper.py``, ``rendering.py`` stage functions, tests) run on the
MI355X kernels: importing this module defines ``torch.ops.gsplat.*`` for the classic 3DGS ops and
provides the attributes the reference reads from ``_C`` (``ext.cpp:58-98``).
"""
from __future__ import annotations

import enum

from . import _ops  # noqa: F401  (TORCH_LIBRARY(gsplat) equivalent: schemas + CUDA-key impls)


class CameraModelType(enum.IntEnum):  # gsplat/cuda/include/Common.h:75-82
    PINHOLE = 0
    ORTHO = 1
    FISHEYE = 2
    FTHETA = 3
    LIDAR = 4


class RendererConfig(enum.IntEnum):  # ext.cpp:66-77
    MIXED_BATCH = 0
    PARALLEL_BATCH = 1


def build_config() -> dict:
    has_2dgs = "rasterize_to_pixels_2dgs" in _ops.SCHEMAS
    # 3dgut: UT projection + from-world compositing fwd / bwd for pinhole / distorted pinhole / ortho / fisheye cameras with
    # a global shutter; f-theta, lidar, rolling shutter, hit distances and normals are refused by the ops themselves
    has_3dgut = _ops.COMPOSITE_UNAVAILABLE is None and "rasterize_to_pixels_from_world_3dgs" in _ops.CLASS_SCHEMAS
    return {"3dgs": True, "2dgs": has_2dgs, "3dgut": has_3dgut, "adam": "adam" in _ops.SCHEMAS,
            "reloc": "relocation" in _ops.SCHEMAS, "losses": False, "camera_wrappers": False}


def null() -> None:  # ext.cpp:82
    return None
