"""MI355X-native drop-in for the GaussianSplatting.jl rasterizer hot path.

Only what the path needs: `csrc/` (HIP kernels + the C ABI in include/gsr.h) and
the host-side mirror of the reference's `GaussianRasterizer` / `rasterize` /
`∇rasterize` / `fused_ssim` interface.  Import via `gsr_pkg.load()`.
"""
from . import synthetic  # noqa: F401
from . import ply  # noqa: F401
from . import checkpoint  # noqa: F401
from . import _lib  # noqa: F401
from .camera import Camera  # noqa: F401


def __getattr__(name):
    # torch-dependent modules are imported lazily
    if name in ("rasterizer", "fused_ssim", "distributed", "optim", "densification"):
        import importlib
        return importlib.import_module(f"{__name__}.{name}")
    if name in ("GaussianRasterizer", "rasterize", "grad_rasterize", "n_color_features"):
        from . import rasterizer
        return getattr(rasterizer, name)
    raise AttributeError(name)
