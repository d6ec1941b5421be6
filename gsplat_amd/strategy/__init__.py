"""Densification strategies of the training loop that calls the rasterizer (reference ``gsplat/strategy``;
SURVEY.md section 8(f) rank 1): ``DefaultStrategy`` (3DGS / AbsGS heuristics) and ``MCMCStrategy`` (3DGS as MCMC)."""
from .base import Strategy
from .default import DefaultStrategy
from .mcmc import MCMCStrategy

__all__ = ["Strategy", "DefaultStrategy", "MCMCStrategy"]
