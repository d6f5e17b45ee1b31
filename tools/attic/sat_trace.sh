#!/bin/bash
# GPU busy share while a batch of ten-minute files runs with K workers (kernel trace of tools/sat_trace.py)
K=${1:-4}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/satt; timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/satt -o k -- python tools/sat_trace.py $K > gpurun_out/satt.log 2>&1
grep BATCH gpurun_out/satt.log
python3 - <<PY
import csv, re
w = float(re.search(r"BATCH ([0-9.]+)", open("gpurun_out/satt.log").read()).group(1))
ev = []
for r in csv.DictReader(open("gpurun_out/satt/k_kernel_trace.csv")):
    if "at::native" in r["Kernel_Name"]: continue
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
t1 = max(e[1] for e in ev); lo = t1 - int(w * 1e9)
pts = []; ksum = 0
for s, e, n in ev:
    if e < lo: continue
    s = max(s, lo); pts.append((s, 1)); pts.append((e, -1)); ksum += e - s
pts.sort(); busy = 0; depth = 0; last = lo; hist = {}
for t, dd in pts:
    if depth > 0: busy += t - last
    hist[min(depth, 6)] = hist.get(min(depth, 6), 0) + (t - last); depth += dd; last = t
tot = t1 - lo
print("window %.0f ms: GPU busy (>= 1 kernel) %.1f %%, summed kernel time %.0f ms; kernels in flight -> share of time:" % (tot / 1e6, 100.0 * busy / tot, ksum / 1e6), {k: round(100.0 * v / tot, 1) for k, v in sorted(hist.items())})
PY
