"""MI355X-native drop-in for the GaussianSplatting.jl rasterizer hot path.

Only what the path needs: `csrc/` (HIP kernels + the C ABI in include/gsr.h) and
the host-side mirror of the reference's `GaussianRasterizer` / `rasterize` /
`∇rasterize` / `fused_ssim` interface.  Import via `gsr_pkg.load()`.
"""
from . import synthetic  # noqa: F401
