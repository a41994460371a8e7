"""Independent float64 torch restatement of the path's *primal* maths, used to
pin the oracle's hand-derived adjoints through autograd —
the analogue of the reference's FiniteDifferences checks (runtests.jl:95-306)
with exact derivatives instead of a 5-point stencil.

Written from the behavioural spec, not from the oracle.
"""
import math

import torch

DT = torch.float64

SH0 = 0.28209479177387814
SH1 = 0.4886025119029199
SH2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
SH3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
       1.445305721320277, -0.5900435899266435]


def quat2rot(q):
    q = q / q.norm(dim=-1, keepdim=True)
    w, x, y, z = q.unbind(-1)
    return torch.stack([
        torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)], -1),
        torch.stack([2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)], -1),
        torch.stack([2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], -1)], -2)


def quat_scale_to_cov(q, s):
    M = quat2rot(q) * s.unsqueeze(-2)
    return M @ M.transpose(-1, -2)


def perspective_projection(mean, Sigma, focal, res, principal):
    """mean (...,3), Sigma (...,3,3) camera space -> Sigma2 (...,2,2), mean2d (...,2)"""
    focal = torch.as_tensor(focal, dtype=DT)
    res = torch.as_tensor(res, dtype=DT)
    pp = torch.as_tensor(principal, dtype=DT) * res
    tan_fov = 0.5 * res / focal
    lim = (res - pp) / focal + 0.3 * tan_fov
    lim_neg = pp / focal + 0.3 * tan_fov
    z = mean[..., 2:3]
    xy = mean[..., :2]
    m2 = focal * xy / z + pp
    txy = z * torch.minimum(lim, torch.maximum(-lim_neg, xy / z))
    zero = torch.zeros_like(z[..., 0])
    J = torch.stack([
        torch.stack([focal[0] / z[..., 0], zero, -focal[0] * txy[..., 0] / z[..., 0] ** 2], -1),
        torch.stack([zero, focal[1] / z[..., 0], -focal[1] * txy[..., 1] / z[..., 0] ** 2], -1)], -2)
    return J @ Sigma @ J.transpose(-1, -2), m2


def sh_basis(d, degree):
    x, y, z = d.unbind(-1)
    b = [torch.full_like(x, SH0)]
    if degree > 0:
        b += [-SH1 * y, SH1 * z, -SH1 * x]
    if degree > 1:
        b += [SH2[0] * x * y, SH2[1] * y * z, SH2[2] * (2 * z * z - x * x - y * y), SH2[3] * x * z,
              SH2[4] * (x * x - y * y)]
    if degree > 2:
        b += [SH3[0] * y * (3 * x * x - y * y), SH3[1] * x * y * z, SH3[2] * y * (4 * z * z - x * x - y * y),
              SH3[3] * z * (2 * z * z - 3 * x * x - 3 * y * y), SH3[4] * x * (4 * z * z - x * x - y * y),
              SH3[5] * z * (x * x - y * y), SH3[6] * x * (x * x - 3 * y * y)]
    return torch.stack(b, -1)


def render_dense(means, shs, opac, scales, rots, cam, sh_degree, background, mode, values_sorted, ranges, radii,
                 R_w2c=None, t_w2c=None):
    """Full differentiable float64 model.  Discrete structure (which Gaussian is in
    which tile, in which order; who is culled) is taken from `values_sorted`,
    `ranges`, `radii`; the per-pixel skip/stop decisions are re-derived in float64.
    Returns image (H,W,C)."""
    W, H = cam.width, cam.height
    R = torch.as_tensor(cam.R, dtype=DT) if R_w2c is None else R_w2c
    t = torch.as_tensor(cam.t, dtype=DT) if t_w2c is None else t_w2c
    pc = means @ R.T + t
    Sig = quat_scale_to_cov(rots, scales)
    Sc = R @ Sig @ R.T
    S2, m2 = perspective_projection(pc, Sc, cam.focal, (W, H), cam.principal)
    S2 = S2 + cam.blur_eps * torch.eye(2, dtype=DT)
    conic = torch.linalg.inv(S2)
    center = torch.as_tensor(cam.camera_center, dtype=DT)
    d = means - center
    d = d / d.norm(dim=-1, keepdim=True)
    basis = sh_basis(d, sh_degree)  # (N,nb)
    nb = basis.shape[-1]
    col = (basis.unsqueeze(-1) * shs[:, :nb, :]).sum(1) + 0.5 + 1.1920929e-7
    rgb = torch.clamp(col, min=0.0)
    feats = [rgb]
    C = {"rgb": 3, "rgbd": 5, "rgbdn": 8}[mode]
    if C > 3:
        feats += [pc[:, 2:3], torch.ones_like(pc[:, 2:3])]
    if C > 5:
        Rg = quat2rot(rots)
        k = torch.argmin(scales.detach(), dim=1)  # first minimum on ties == reference's <= chain
        ax = Rg[torch.arange(Rg.shape[0]), :, k]
        # projection.jl:227-229: the pose gradient deliberately does not see the normals
        ncam = ax @ R.detach().T
        sign = torch.where((ncam * pc).sum(-1).detach() > 0, -1.0, 1.0)
        feats += [ncam * sign.unsqueeze(-1)]
    feat = torch.cat(feats, 1)
    bg = torch.zeros(C, dtype=DT)
    bg[:3] = torch.as_tensor(background, dtype=DT)
    gx_n, gy_n = (W + 15) // 16, (H + 15) // 16
    img = torch.zeros(H, W, C, dtype=DT)
    vals = torch.as_tensor(values_sorted.astype("int64"))
    for gy in range(gy_n):
        for gx in range(gx_n):
            r0, r1 = int(ranges[gy * gx_n + gx, 0]), int(ranges[gy * gx_n + gx, 1])
            ys = torch.arange(gy * 16, min(gy * 16 + 16, H))
            xs = torch.arange(gx * 16, min(gx * 16 + 16, W))
            py, px = torch.meshgrid(ys, xs, indexing="ij")
            P = px.numel()
            if r1 <= r0:
                img[py, px] = bg.expand(P, C).reshape(py.shape + (C,))
                continue
            ids = vals[r0:r1]
            dx = m2[ids, 0].unsqueeze(0) - px.reshape(-1, 1).to(DT)
            dy = m2[ids, 1].unsqueeze(0) - py.reshape(-1, 1).to(DT)
            a, b, c = conic[ids, 0, 0], conic[ids, 1, 0], conic[ids, 1, 1]
            sigma = b * dx * dy + 0.5 * (a * dx * dx + c * dy * dy)
            araw = opac[ids].unsqueeze(0) * torch.exp(-sigma)
            alpha = araw + (torch.clamp(araw, max=0.99) - araw).detach()  # clamp is not differentiated
            keep = ((sigma >= 0) & (alpha >= 1.0 / 255.0)).detach()
            al = torch.where(keep, alpha, torch.zeros_like(alpha))
            Tn = torch.cumprod(1 - al, dim=1)
            stop = ((Tn < 1e-4) & keep).detach()
            dead = (torch.cumsum(stop.to(torch.int64), 1) > 0)
            al = torch.where(dead, torch.zeros_like(al), al)
            Tincl = torch.cumprod(1 - al, dim=1)
            Texcl = torch.cat([torch.ones(P, 1, dtype=DT), Tincl[:, :-1]], 1)
            w = al * Texcl
            out = w @ feat[ids] + Tincl[:, -1:] * bg
            img[py, px] = out.reshape(py.shape + (C,))
    return img
