#!/bin/bash
# On the GPU box: tools/fetch_calib.sh  -> gpurun_out/fetch_calib/{times.txt,fetch.txt}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/fetch_calib; rm -rf $O; mkdir -p $O
tools/bin/fetch_calib > $O/times.txt 2>&1
for c in FETCH_SIZE "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_REQ_sum TCC_MISS_sum"; do
  t=$(echo $c | tr ' ' '_')
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/$t -o p -- tools/bin/fetch_calib > $O/$t.log 2>&1
done
python3 - $O <<'PY' > $O/fetch.txt
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f, newline="")):
        k = r["Kernel_Name"].split("(")[0]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in acc:
    print(k, {c: sum(v) / len(v) for c, v in acc[k].items()})
PY
cat $O/times.txt $O/fetch.txt
find $O -name "*.csv" -delete
