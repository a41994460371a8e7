"""Round 6: is the GPU ever idle inside a training step of the protocol?  Runs the harness for STEPS steps (so that the last ones are
plain steps of one densification round) — to be run under `rocprofv3 --kernel-trace`, then tools/gap_report.py on the trace."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import gsr_pkg, torch
import train_harness as TH
pkg = gsr_pkg.load()
p = TH.Protocol(densify_grad_threshold=4e-5)
h = TH.Harness(pkg, p)
for _ in range(int(sys.argv[1])):
    h.step()
torch.cuda.synchronize()
print("steps", h.step_no, "N", len(h.gs), "last view", h.history[-1])
