"""Host mirror of the fields `rasterize` reads from the reference's `Camera`
(src/camera.jl:2-16, 37-46; intrinsics: focal in pixels, principal normalised to
[0,1], resolution) — plain data, no GL projection matrices."""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np


@dataclass
class Camera:
    width: int
    height: int
    focal: tuple
    principal: tuple = (0.5, 0.5)
    R: np.ndarray = field(default_factory=lambda: np.eye(3, dtype=np.float32))  # world->camera, row-major
    t: np.ndarray = field(default_factory=lambda: np.zeros(3, dtype=np.float32))

    @classmethod
    def simple(cls, fx: float, fy: float, width: int, height: int):
        """Camera(; fx, fy, width, height) — camera.jl:37-46"""
        return cls(width, height, (fx, fy))

    @property
    def camera_center(self) -> np.ndarray:
        """c2w[1:3,4] (camera.jl:28): -R' t"""
        return (-np.asarray(self.R, np.float64).T @ np.asarray(self.t, np.float64)).astype(np.float32)

    @property
    def resolution(self):
        return (self.width, self.height)
