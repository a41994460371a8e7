"""Shared plumbing of the -m gpu parity tests: run the HIP path through the C ABI
(via the host mirror) on the same inputs as the oracle."""
import numpy as np
import torch


def dev(a, dtype=torch.float32):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype).cuda().contiguous()


class HipRun:
    def __init__(self, pkg, means, shs, opac, scales, rots, cam, sh_degree, background=(0, 0, 0), mode="rgb",
                 want_covis=False, want_uncert=False, pose_dev=False, exact_tile_cull=False):
        R = pkg.rasterizer
        self.pkg, self.cam, self.deg, self.bg, self.mode = pkg, cam, sh_degree, tuple(float(b) for b in background), mode
        self.camera = pkg.Camera(cam.width, cam.height, tuple(cam.focal), tuple(cam.principal), np.asarray(cam.R),
                                 np.asarray(cam.t))
        self.rast = R.GaussianRasterizer(cam.width, cam.height, mode=mode, near_plane=cam.near_plane,
                                         far_plane=cam.far_plane, exact_tile_cull=exact_tile_cull)
        self.t = [dev(means), dev(shs), dev(np.asarray(opac).reshape(-1, 1)), dev(scales), dev(rots)]
        n = means.shape[0]
        self.covis = torch.zeros(n, dtype=torch.uint8, device="cuda") if want_covis else None
        self.unc = torch.zeros(cam.height, cam.width, device="cuda") if want_uncert else None
        self.Rd = self.td = None
        if pose_dev:
            self.Rd = dev(np.asarray(cam.R, np.float32).T)  # column-major (3,3)
            self.td = dev(np.asarray(cam.t, np.float32))

    def forward(self):
        img = self.rast.forward_raw(*self.t, self.camera, self.deg, self.bg, self.Rd, self.td, self.covis, self.unc)
        torch.cuda.synchronize()
        return img

    def backward(self, vpixels):
        out = self.rast.backward_raw(dev(vpixels), *self.t, self.camera, self.deg, self.bg, self.Rd, self.td)
        torch.cuda.synchronize()
        return out


def rel_l2(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)


def frac_bad(a, b, rtol, atol):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float((np.abs(a - b) > rtol * np.abs(b) + atol).mean())
