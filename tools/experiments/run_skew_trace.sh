cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/skewtrace; rm -rf $O; mkdir -p $O
for k in hot:8000 hot:32000; do
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/prof_${k/:/_} -o bench -- python3 bench.py --in-process --no-cpu-baseline --no-other-lists --no-loss --steps 5 --warmup 2 --skew $k > $O/log_${k/:/_}.txt 2>&1
python3 - $O/prof_${k/:/_} <<'PY'
import csv,glob,re,sys
k=glob.glob(sys.argv[1]+"/*kernel_trace.csv")[0]
rows=[]
for r in csv.DictReader(open(k)):
    n=re.sub(r"\(anonymous namespace\)::","",r["Kernel_Name"])
    short=n.split("(")[0][-60:]
    rows.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),short,r.get("Workgroup_Size_X") or r.get("Workgroup_Size"),r.get("Grid_Size_X") or r.get("Grid_Size")))
rows.sort()
last=[r for r in rows if "composite_bwd" in r[2]][-4:]
t0=last[0][0]
for s,e,n,wg,gs in last: print(n, "wg",wg,"grid",gs,"start %.1f end %.1f us"%((s-t0)/1000,(e-t0)/1000))
PY
done
