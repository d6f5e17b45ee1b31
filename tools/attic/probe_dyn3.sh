#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/dyn3; timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/dyn3 -o k -- python tools/probe_dynamic_batch.py ${1:-4} ${2:-2} > gpurun_out/dyn3.log 2>&1
tail -3 gpurun_out/dyn3.log
python - <<PY
import csv
rows=[r for r in csv.DictReader(open("gpurun_out/dyn3/k_kernel_trace.csv"))]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
dyn=[r for r in rows if "loudnorm_dynamic" in r["Kernel_Name"]]
t0=int(dyn[0]["Start_Timestamp"])
for r in dyn: print("dyn %9.3f %9.3f q%s" % ((int(r["Start_Timestamp"])-t0)/1e9, (int(r["End_Timestamp"])-t0)/1e9, r["Queue_Id"]))
# what else ran while the batch's dynamic kernels ran: count kernels per 0.2 s bucket after the second dyn kernel starts
s1=int(dyn[1]["Start_Timestamp"]); e1=int(dyn[-1]["End_Timestamp"])
import collections
b=collections.Counter()
for r in rows:
    s=int(r["Start_Timestamp"])
    if s1<=s<=e1 and "loudnorm_dynamic" not in r["Kernel_Name"]: b[int((s-s1)/2e8)]+=1
print("other kernels per 0.2 s bucket:", [b[i] for i in range(int((e1-s1)/2e8)+1)])
PY
