#!/bin/bash
# GPU busy fraction and per-queue occupancy while K contexts process 10-minute files (bench.py --minutes 10 --in-flight K), from a kernel trace
K=${1:-3}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/tli; timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tli -o k -- python bench.py --minutes 10 --steps 6 --warmup 2 --cpu-sample 0 --e2e 0 --in-flight $K > /tmp/tli.log 2>&1
python3 - <<PY
import csv, glob
ev = []
for f in glob.glob("/tmp/tli/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "at::native" in r["Kernel_Name"]: continue
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-40:]))
ev.sort()
import json
line = [l for l in open("/tmp/tli.log") if l.startswith("{")][-1]
d = json.loads(line); sat = d.get("saturation") or {"files_in_flight_per_gpu": 1, "ms_per_file": d["ms_per_step"]}
print("single: %.2f ms per file; in flight %d: %.2f ms per file" % (d["ms_per_step"], sat["files_in_flight_per_gpu"], sat["ms_per_file"]))
# the in-flight phase is the end of the trace: steps x K files
t1 = max(e[1] for e in ev)
lo = t1 - int(sat["ms_per_file"] * sat["files_in_flight_per_gpu"] * d["steps"] * 1e6 * 0.9)
pts = []
for s, e, n in ev:
    if e < lo: continue
    pts.append((max(s, lo), 1)); pts.append((e, -1))
pts.sort()
busy = 0; depth = 0; last = lo; hist = {}
for t, d in pts:
    if depth > 0: busy += t - last
    hist[depth] = hist.get(depth, 0) + (t - last)
    depth += d; last = t
tot = t1 - lo
print("window %.1f ms: GPU busy (>=1 kernel) %.1f %%; concurrency histogram (kernels in flight: share of time):" % (tot / 1e6, 100.0 * busy / tot))
print({k: round(100.0 * v / tot, 1) for k, v in sorted(hist.items())})
PY
