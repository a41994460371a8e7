"""Round 6: the held fused launch leaves the GPU idle between tile_scan and the tier sorts (22-34 us per view by the kernel trace).
Where does that time go — the host seeing the totals, the host's work before the launch, or the launch itself?  Reads a rocprofv3
--kernel-trace --hip-trace output directory (CSV) and prints, for the last steps: scan end -> next hipLaunchKernel call begin (host
reaction), that call's duration, call end -> kernel start (launch latency)."""
import csv, glob, re, sys
d = sys.argv[1]
kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
ht = glob.glob(d + "/**/*hip_api_trace.csv", recursive=True)[0]
K = []
for r in csv.DictReader(open(kt)):
    name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("<")[0].split("(")[0].split()[-1]
    K.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, int(r.get("Correlation_Id", 0) or 0)))
K.sort()
A = []
for r in csv.DictReader(open(ht)):
    A.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Function"], int(r.get("Correlation_Id", 0) or 0)))
A.sort()
launch_by_corr = {a[3]: a for a in A if "Launch" in a[2]}
scans = [i for i, k in enumerate(K) if k[2] == "tile_scan_kernel"]
rows = []
for i in scans[-40:]:
    if i + 1 >= len(K):
        continue
    nxt = K[i + 1]
    la = launch_by_corr.get(nxt[3])
    if la is None:
        continue
    # host API calls between the scan's end and the launch call
    between = [a for a in A if K[i][1] - 2000 <= a[0] <= la[0]]
    rows.append((nxt[2], (la[0] - K[i][1]) / 1e3, (la[1] - la[0]) / 1e3, (nxt[0] - la[1]) / 1e3, (nxt[0] - K[i][1]) / 1e3,
                 [(a[2], round((a[1] - a[0]) / 1e3, 1)) for a in between][-8:]))
for r in rows[-12:]:
    print(f"next kernel {r[0]:28s} scan end -> launch call {r[1]:7.1f} us, call {r[2]:5.1f} us, call end -> kernel start {r[3]:6.1f} us, total gap {r[4]:6.1f} us; host calls before it: {r[5]}")
