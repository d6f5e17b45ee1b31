#!/bin/bash
# which HIP API calls of the worker threads take seconds while a dynamic-loudnorm kernel of another worker runs?
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/dyn4; timeout 600 rocprofv3 --hip-runtime-trace --output-format csv -d gpurun_out/dyn4 -o k -- python tools/probe_dynamic_batch.py ${1:-6} ${2:-5} > gpurun_out/dyn4.log 2>&1
grep -v "^[EW]2026" gpurun_out/dyn4.log | tail -2
python - <<PY
import csv, glob, collections
f = glob.glob("gpurun_out/dyn4/*hip_api_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
print(len(rows), "api calls; columns", list(rows[0].keys()))
long = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Function"], r["Thread_Id"]) for r in rows]
long.sort(reverse=True)
for d, fn, t in long[:40]: print("%8.3f s  %-34s thread %s" % (d / 1e9, fn, t))
PY
