#!/bin/bash
# Kernel trace of the saturation leg itself (configs[3] on one GPU): GPU busy share, kernels in flight, summed kernel time per kernel.
# usage: tools/sat_timeline.sh [streams] [in_flight] [md5] [files]      (on the GPU box; writes gpurun_out/satl_*.txt)
S=${1:-1}; K=${2:-8}; M=${3:-0}; NF=${4:-32}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/satl; timeout 900 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/satl -o k -- python tools/sat_streams.py $NF $M $S $K > gpurun_out/satl.log 2>&1
grep "^streams" gpurun_out/satl.log
python3 - <<PY
import csv, re, glob, collections
line = [l for l in open("gpurun_out/satl.log") if l.startswith("streams")][-1]
ms = [float(v) for v in re.findall(r"([0-9.]+) ms/file \(", line)]
w = ms[-1] * $NF * 1e6          # the last repetition's wall in ns
ev = []
for f in glob.glob("gpurun_out/satl/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "at::native" in n or "rocprim" in n: continue
        n = re.sub(r"^void ", "", n).replace("(anonymous namespace)::", "").split("(")[0]
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n))
t1 = max(e[1] for e in ev); lo = t1 - int(w)
pts = []; ksum = collections.Counter(); kcnt = collections.Counter()
for s, e, n in ev:
    if e < lo: continue
    s = max(s, lo); pts.append((s, 1)); pts.append((e, -1)); ksum[n] += e - s; kcnt[n] += 1
pts.sort(); busy = 0; depth = 0; last = lo; hist = collections.Counter()
for t, dd in pts:
    if depth > 0: busy += t - last
    hist[min(depth, 8)] += t - last; depth += dd; last = t
tot = t1 - lo
print("last repetition: window %.0f ms (%d files): GPU busy (>= 1 kernel) %.1f %%, summed kernel time %.1f ms per file, %d launches per file" % (
    tot / 1e6, $NF, 100.0 * busy / tot, sum(ksum.values()) / 1e6 / $NF, sum(kcnt.values()) / $NF))
print("kernels in flight -> share of time:", {k: round(100.0 * v / tot, 1) for k, v in sorted(hist.items())})
print("summed kernel time per file (ms), launches per file:")
for n, v in ksum.most_common(40): print("  %8.3f  %5.1f  %s" % (v / 1e6 / $NF, kcnt[n] / $NF, n[:90]))
PY
