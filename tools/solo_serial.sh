#!/bin/bash
# uncontended per-kernel durations of a 10-minute and a 60-minute file (one stream), side by side: which kernels do not shrink with the file
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for m in 10 60; do
  rm -rf gpurun_out/solo$m; timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/solo$m -o k -- python tools/solo_serial.py $m 3 > gpurun_out/solo$m.log 2>&1
  grep SOLO gpurun_out/solo$m.log
done
python3 - <<'PY'
import csv, glob, re, collections
def load(m):
    d = collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/solo{m}/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            if "at::native" in n or "rocprim" in n: continue
            n = re.sub(r"^void ", "", n).replace("(anonymous namespace)::", "").split("(")[0]
            wg = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
            grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
            d[n].append(((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, grid // max(1, wg), wg, int(r["LDS_Block_Size"])))
    return d
a, b = load(10), load(60)
steps = 4.0
print("%-46s %8s %8s %6s | %9s %5s %7s" % ("kernel (ms per file, summed over its launches)", "10 min", "60 min", "x6/60", "wgs(10)", "wg", "LDS"))
rows = []
for n in a:
    ta = sum(v[0] for v in a[n]) / steps; tb = sum(v[0] for v in b.get(n, [])) / steps
    rows.append((ta, tb, n, max(v[1] for v in a[n]), a[n][0][2], max(v[3] for v in a[n])))
rows.sort(reverse=True)
for ta, tb, n, wgs, wg, lds in rows[:60]:
    print("%-46s %8.3f %8.3f %6.2f | %9d %5d %7d" % (n[:46], ta, tb, (6 * ta / tb) if tb else 0, wgs, wg, lds))
print("total: 10 min %.2f ms, 60 min %.2f ms (%.2f per ten minutes)" % (sum(r[0] for r in rows), sum(r[1] for r in rows), sum(r[1] for r in rows) / 6))
PY
