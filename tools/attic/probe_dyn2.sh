#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/dyn2; timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/dyn2 -o k -- python tools/probe_dynamic_concurrency.py > gpurun_out/dyn2.log 2>&1
python - <<PY
import csv
rows=[r for r in csv.DictReader(open("gpurun_out/dyn2/k_kernel_trace.csv")) if "loudnorm_dynamic" in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
t0=int(rows[0]["Start_Timestamp"])
for r in rows: print("%9.3f %9.3f q%s" % ((int(r["Start_Timestamp"])-t0)/1e9, (int(r["End_Timestamp"])-t0)/1e9, r["Queue_Id"]))
PY
