"""Round-4 fuzz campaign, edge case 2223: the one failure outside the five gradient tensors (camera-pose gradient, 1e-4).  Prints
both sides and how ill-conditioned the sum is.  python tools/experiments/pose_case_edge2223.py"""
import sys, os, numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [R, os.path.join(R, "tests"), os.path.join(R, "tools")]
os.chdir(R)
import fuzz_parity as F
fs = F.fuzz_scenes.edge_scene(F.pkg, 2223)
orc, pkg, T = F.orc, F.pkg, F.T
st = orc.forward(fs.means, fs.shs, fs.opac, fs.scales, fs.rots, fs.cam, fs.deg, background=fs.bg, mode=fs.mode)
run = F.HipRun(pkg, fs.means, fs.shs, fs.opac, fs.scales, fs.rots, fs.cam, fs.deg, fs.bg, fs.mode, pose_dev=fs.pose)
run.forward(); vp = fs.cotangent()
g = orc.backward(st, vp, fs.means, fs.shs, fs.opac, fs.scales, fs.rots, fs.cam, fs.deg, background=fs.bg, pose_grad=True)
out = run.backward(vp)
vR, vt = T._compare_backward(g, out, st.radii > 0)
print("pose", fs.pose, "mode", fs.mode, "n", fs.means.shape[0], "rendered", st.n_rendered)
print("vR rel_l2", T.rel_l2(vR.reshape(-1), g.vR), "vt rel_l2", T.rel_l2(vt, g.vt))
print("vR hip", np.asarray(vR).reshape(-1)); print("vR orc", np.asarray(g.vR).reshape(-1))
print("vt hip", np.asarray(vt)); print("vt orc", np.asarray(g.vt))
# conditioning: the pose gradient is a sum over Gaussians of terms built from vmeans (camera-space): compare ||sum|| with sum ||.||
vm = np.asarray(g.vmeans, np.float64)
print("||sum vmeans||", np.linalg.norm(vm.sum(0)), "sum ||vmeans_i||", np.linalg.norm(vm, axis=1).sum())
