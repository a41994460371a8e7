cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/gap2; rm -rf $O; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --hip-runtime-trace --output-format csv -d $O/prof -o bench -- python3 bench.py --in-process --no-cpu-baseline --no-other-lists --steps 12 --warmup 3 > $O/prof.log 2>&1
ls $O/prof
python3 - <<'PY'
import csv, glob, re
O="gpurun_out/gap2/prof"
kt=glob.glob(O+"/*kernel_trace.csv")[0]
ht=glob.glob(O+"/*hip_api_trace.csv")[0]
K=[]
for d in csv.DictReader(open(kt)):
    name=re.sub(r"\(anonymous namespace\)::","",d["Kernel_Name"]).split("<")[0].split("(")[0].split()[-1]
    K.append((int(d["Start_Timestamp"]),int(d["End_Timestamp"]),name,int(d["Correlation_Id"])))
K.sort()
A={}
for d in csv.DictReader(open(ht)):
    A[int(d["Correlation_Id"])]=(d["Function"],int(d["Start_Timestamp"]),int(d["End_Timestamp"]))
idx=[i for i,k in enumerate(K) if "preprocess_kernel" in k[2]]
a,b=idx[-3],idx[-2]
t0=K[a][0]
for s,e,n,c in K[a:b+1]:
    f=A.get(c)
    print(f"{n:32s} kernel {(s-t0)/1000:9.1f} .. {(e-t0)/1000:9.1f} us | launch API {f[0] if f else '?':24s} {(f[1]-t0)/1000 if f else 0:9.1f} .. {(f[2]-t0)/1000 if f else 0:9.1f}")
PY
