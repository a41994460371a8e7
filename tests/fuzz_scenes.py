"""The seeded scene families of the fuzz campaign (tools/fuzz_parity.py; results in profiles/r03/fuzz_parity.txt), as data
builders shared by the test suite (test_gpu_parity.test_randomised_sweep_vs_oracle, test_gpu_fuzz_regressions) and the tool.
Every builder is a pure function of its case number: the recorded failing cases stay reproducible.
No product / oracle compute here: numpy + the plain Camera dataclass."""
from dataclasses import dataclass, field

import numpy as np

from oracle.oracle import Camera


@dataclass
class FuzzScene:
    means: np.ndarray
    shs: np.ndarray
    opac: np.ndarray
    scales: np.ndarray      # activated
    rots: np.ndarray
    cam: Camera
    deg: int
    mode: str
    bg: tuple
    rng: np.random.Generator   # positioned where the original test drew its cotangent from
    pose: bool = False
    note: dict = field(default_factory=dict)

    @property
    def rotations(self):
        return self.rots

    @property
    def channels(self):
        return {"rgb": 3, "rgbd": 5, "rgbdn": 8}[self.mode]

    def cotangent(self):
        W, H = self.cam.width, self.cam.height
        return self.rng.standard_normal((H, W, self.channels)).astype(np.float32)


def sweep_scene(pkg, case):
    """test_randomised_sweep_vs_oracle: modes, SH degrees, ragged resolutions, views, footprint sizes, opacity ranges."""
    rng = np.random.default_rng(9000 + case)
    mode = ["rgb", "rgbd", "rgbdn"][case % 3]
    deg = int(rng.integers(0, 4))
    W, H = int(rng.integers(20, 140)), int(rng.integers(20, 110))
    n = int(rng.integers(1, 1500))
    s = pkg.synthetic.make_scene(n, W, H, deg, 9100 + case, sigma_px=float(rng.uniform(1.5, 9.0)),
                                 K=16 if case % 4 == 0 else None)
    opac = (s.opacities * rng.uniform(0.05, 1.0)).astype(np.float32) if case % 2 else s.opacities
    R, t = pkg.synthetic.view_pose(int(rng.integers(0, 8)))
    cam = Camera(W, H, s.focal, R=R, t=t, principal=(float(rng.uniform(0.4, 0.6)), float(rng.uniform(0.4, 0.6))))
    bg = tuple(float(x) for x in rng.uniform(0, 1, 3))
    return FuzzScene(s.means, s.shs, opac, s.scales, s.rotations, cam, deg, mode, bg, rng)


def deep_scene(pkg, case):
    """Dense scenes on 1..9 tiles: lists of 300 .. 40 000 instances per tile (every sort tier, the tier launches of both
    compositing kernels, bins that overflow into compact mode)."""
    rng = np.random.default_rng(77000 + case)
    mode = ["rgb", "rgbd", "rgbdn"][case % 3]
    deg = int(rng.integers(0, 3))
    W, H = int(rng.integers(16, 49)), int(rng.integers(16, 49))
    target = int(np.exp(rng.uniform(np.log(300), np.log(40000))))          # instances in the deepest tile, roughly
    n = int(target * rng.uniform(1.0, 1.6))
    s = pkg.synthetic.make_scene(n, W, H, deg, 77100 + case)
    spread = float(rng.uniform(0.02, 0.3))
    means = np.stack([rng.uniform(-spread, spread, n), rng.uniform(-spread, spread, n), rng.uniform(2, 8, n)], 1).astype(np.float32)
    cam = Camera(W, H, s.focal)
    opac = np.full(n, float(rng.uniform(0.004, 0.05)), np.float32) * rng.uniform(0.5, 1.5, n).astype(np.float32)
    scales = s.scales * float(rng.uniform(1.0, 3.0))
    bg = tuple(float(x) for x in rng.uniform(0, 1, 3))
    return FuzzScene(means, s.shs, opac, scales, s.rotations, cam, deg, mode, bg, rng)


def edge_scene(pkg, case):
    """Hostile inputs: fx != fy, arbitrary poses, cutting near / far planes, opacities 0 / 1 / 0.995 / 1/255, sub-pixel /
    screen-filling / negative / needle (27 : 1) scales, means behind the camera or far off-screen, clamping SH."""
    rng = np.random.default_rng(55000 + case)
    mode = ["rgb", "rgbd", "rgbdn"][case % 3]
    deg = int(rng.integers(0, 4))
    W, H = int(rng.integers(17, 150)), int(rng.integers(17, 120))
    n = int(rng.integers(1, 1200))
    s = pkg.synthetic.make_scene(n, W, H, deg, 55100 + case, sigma_px=float(rng.uniform(1.0, 12.0)))
    means, scales, opac, shs = s.means.copy(), s.scales.copy(), s.opacities.copy(), s.shs.copy()
    k = lambda frac: rng.random(n) < frac  # noqa: E731
    opac = rng.uniform(0.0, 1.0, n).astype(np.float32)
    opac[k(0.05)] = 0.0; opac[k(0.05)] = 1.0; opac[k(0.05)] = np.float32(0.995); opac[k(0.03)] = np.float32(1.0 / 255.0)
    scales[k(0.08)] *= 0.02                     # sub-pixel: radius <= 3 -> culled
    scales[k(0.03)] *= 6.0                      # screen-filling
    scales[k(0.02)] *= -1.0                     # negative (activated) scales: only their squares matter
    m = k(0.08); scales[m, 0] *= 4.0; scales[m, 1] *= 0.15   # needles, ~27 : 1
    means[k(0.04), 2] *= -1.0                   # behind the camera
    means[k(0.03), 0] += 40.0                   # far off-screen
    shs[k(0.1)] *= 8.0                          # clamping colours
    fx = float(s.focal[0]) * float(rng.uniform(0.6, 1.6)); fy = float(s.focal[1]) * float(rng.uniform(0.6, 1.6))
    ang = rng.uniform(-0.35, 0.35, 3)
    cx, sx_, cy, sy_, cz, sz_ = np.cos(ang[0]), np.sin(ang[0]), np.cos(ang[1]), np.sin(ang[1]), np.cos(ang[2]), np.sin(ang[2])
    Rx = np.array([[1, 0, 0], [0, cx, -sx_], [0, sx_, cx]]); Ry = np.array([[cy, 0, sy_], [0, 1, 0], [-sy_, 0, cy]])
    Rz = np.array([[cz, -sz_, 0], [sz_, cz, 0], [0, 0, 1]])
    Rm = (Rz @ Ry @ Rx).astype(np.float32)
    t = rng.uniform(-0.5, 0.5, 3).astype(np.float32)
    near, far = (0.2, 1000.0) if case % 4 else (float(rng.uniform(1.0, 4.0)), float(rng.uniform(6.0, 11.0)))
    cam = Camera(W, H, (np.float32(fx), np.float32(fy)), R=Rm, t=t, near_plane=near, far_plane=far,
                 principal=(float(rng.uniform(0.3, 0.7)), float(rng.uniform(0.3, 0.7))))
    bg = tuple(float(x) for x in rng.uniform(0, 1, 3))
    return FuzzScene(means, shs, opac, scales, s.rotations, cam, deg, mode, bg, rng, pose=bool(case & 2), note={"needles": m})


def needle_scene(pkg, seed, ratio=(12.0, 0.02)):
    """150 Gaussians, 15 % of them stretched to 600 : 1 needles (the conditioning study of round 3)."""
    rng = np.random.default_rng(seed)
    W, H, n, deg, mode = 64, 48, 150, 1, "rgb"
    s = pkg.synthetic.make_scene(n, W, H, deg, 500 + seed, sigma_px=4.0)
    scales = s.scales.copy()
    m = rng.random(n) < 0.15
    scales[m, 0] *= ratio[0]; scales[m, 1] *= ratio[1]
    R, t = pkg.synthetic.view_pose(2)
    cam = Camera(W, H, s.focal, R=R, t=t)
    return FuzzScene(s.means, s.shs, s.opacities, scales, s.rotations, cam, deg, mode, (0.3, 0.1, 0.6), rng, note={"needles": m})
