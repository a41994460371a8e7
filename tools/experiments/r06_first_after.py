"""Round 6: where does the first step after a densification spend its extra time?  Per-phase wall clock (synchronised) of the
training harness's step, for plain steps and for the first step after each densification round."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import gsr_pkg, torch
import train_harness as TH
pkg = gsr_pkg.load()
p = TH.Protocol(densify_grad_threshold=4e-5)
h = TH.Harness(pkg, p)
sync = torch.cuda.synchronize
rows = []
for step in range(1, 1301):
    first_after = (step - 1) >= 500 and (step - 1) % 100 == 0
    probe = first_after or (step > 500 and step % 100 == 50)
    if not probe:
        h.step(); continue
    # the harness's step, phase by phase
    sync(); t = [time.perf_counter()]
    h.step_no += 1
    h.opts["points"].lr = TH.lr_points(p, h.extent, h.step_no)
    if h.step_no % p.sh_ramp_interval == 0 and h.sh_degree < p.max_sh_degree: h.sh_degree += 1
    v = TH.view_of_step(p, h.step_no); cam = h.cams[v]
    if h.act is None: h.prologue()
    sync(); t.append(time.perf_counter())
    shs, oa, sa = h.act
    g0 = h.rast.stats.scratch_regrowths
    img = h.rast.forward_raw(h.gs.points, shs, oa, sa, h.gs.rotations, cam, h.sh_degree, h.bg)
    sync(); t.append(time.perf_counter())
    st = h.rast.stats
    loss, vp = pkg.fused_ssim.l1_ssim_loss(h.rast, img, h.targets[v], p.lambda_dssim)
    h.losses.append(loss)
    sync(); t.append(time.perf_counter())
    pkg.optim.fused_backward_tail_step(h.rast, vp, h.opts, h.raw(), shs, oa, sa, cam, h.sh_degree, h.bg, forward_generation=int(st.generation), color_cotangent=True)
    sync(); t.append(time.perf_counter())
    h.post_train_step(h.step_no)
    sync(); t.append(time.perf_counter())
    d = [round(1e3 * (b - a), 3) for a, b in zip(t, t[1:])]
    rows.append(dict(step=step, first_after=first_after, n=len(h.gs), prologue=d[0], forward=d[1], loss=d[2], backward=d[3], post=d[4],
                     scratch_regrown=int(st.scratch_regrowths) - int(g0), bins_cap=int(st.bin_capacity), binning=int(st.compact_binning)))
for r in rows: print(json.dumps(r))
