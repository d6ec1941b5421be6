"""On-disk compression of trained splats (reference gsplat/compression/)."""
from .png_compression import PngCompression

__all__ = ["PngCompression"]
