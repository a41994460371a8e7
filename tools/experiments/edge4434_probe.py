"""Fuzz edge case 4434 (final campaign of round 4): the exact-cull image differs from the reference-lists image.  Which pixels,
which tile, and does the per-lane walk (GSR_HIP_LIB=tools/bin/libgsr_noflat.so) agree?   python tools/experiments/edge4434_probe.py [case]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [R, os.path.join(R, "tests"), os.path.join(R, "tools")]
os.chdir(R)
import numpy as np, torch
import fuzz_parity as F
case = int(sys.argv[1]) if len(sys.argv) > 1 else 4434
fs = F.fuzz_scenes.edge_scene(F.pkg, case)
pkg, orc = F.pkg, F.orc
print("case", case, "n", fs.means.shape[0], "res", fs.cam.width, fs.cam.height, "mode", fs.mode, "deg", fs.deg, "lib", os.environ.get("GSR_HIP_LIB"))
st = orc.forward(fs.means, fs.shs, fs.opac, fs.scales, fs.rots, fs.cam, fs.deg, background=fs.bg, mode=fs.mode)
ref = F.HipRun(pkg, fs.means, fs.shs, fs.opac, fs.scales, fs.rots, fs.cam, fs.deg, fs.bg, fs.mode, pose_dev=fs.pose)
img = ref.forward().clone()
cul = F.HipRun(pkg, fs.means, fs.shs, fs.opac, fs.scales, fs.rots, fs.cam, fs.deg, fs.bg, fs.mode, exact_tile_cull=True)
imc = cul.forward().clone()
d = (img != imc).reshape(fs.cam.height, fs.cam.width, -1).any(-1).cpu().numpy()
print("differing pixels", int(d.sum()), "max abs diff", float((img - imc).abs().max()))
o = torch.as_tensor(st.image).cuda().reshape(img.shape)
print("vs oracle: ref-lists max", float((img - o).abs().max()), "cull max", float((imc - o).abs().max()))
ys, xs = np.nonzero(d)
if len(ys):
    tiles = sorted(set(zip((ys // 16).tolist(), (xs // 16).tolist())))
    print("tiles (ty, tx)", tiles[:20], "n_tiles", len(tiles))
    print("n_contrib ref/cull at first px", int(ref.rast.n_contrib.reshape(fs.cam.height, fs.cam.width)[ys[0], xs[0]]), int(cul.rast.n_contrib.reshape(fs.cam.height, fs.cam.width)[ys[0], xs[0]]))
print("rendered ref", ref.rast.stats.n_rendered, "cull", cul.rast.stats.n_rendered, "oracle", st.n_rendered)
