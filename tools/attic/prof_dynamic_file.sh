#!/bin/bash
# kernel times of one 10-minute file that takes the dynamic-loudnorm fallback (file to file)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/dynf; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/dynf -o k -- python tools/probe_dynamic_batch.py 1 10 > gpurun_out/dynf.log 2>&1
grep -v "^[EW]2026" gpurun_out/dynf.log | tail -2
python - <<PY
import csv
rows=list(csv.DictReader(open("gpurun_out/dynf/k_kernel_stats.csv")))
rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
for r in rows[:12]: print("%-70s calls %4s total %9.3f ms" % (r["Name"][:70], r["Calls"], float(r["TotalDurationNs"])/1e6))
PY
