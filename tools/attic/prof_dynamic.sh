#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/dynp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/dynp -o k -- python tools/ab_dynamic_wg.py ${1:-120} ${2:-2.5} > gpurun_out/dynp.log 2>&1
python - <<PY
import csv
rows=[r for r in csv.DictReader(open("gpurun_out/dynp/k_kernel_trace.csv")) if "loudnorm_dynamic" in r["Kernel_Name"]]
for r in rows: print("%-40s %9.3f ms" % (r["Kernel_Name"][:40], (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6))
PY
