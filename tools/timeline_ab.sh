#!/bin/bash
# Kernel timeline of the last step of tools/ab_builds.py's child (the A/B build, JT_<KEY> variables from the environment):
#   JT_NLM_OLD=1 bash tools/timeline_ab.sh            (same output as tools/timeline.sh)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/tlab; JT_AB_CHILD=1 timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tlab -o k -- python tools/ab_builds.py --child 2 60 > gpurun_out/tlab.log 2>&1
python - <<PY
import csv, glob, re
rows = []
for f in glob.glob("gpurun_out/tlab/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "at::native" in n or "rocprim" in n: continue
        n = re.sub(r"^void ", "", n).replace("(anonymous namespace)::", "").split("(")[0]
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], n))
rows.sort()
starts = [i for i, r in enumerate(rows) if r[3].startswith("k_frame_stats")]
t0 = rows[starts[-1]][0]
for s, e, q, n in rows:
    if s < t0 - 1000000: continue
    print("%9.3f %8.3f  q%-3s %s" % ((s - t0) / 1e6, (e - s) / 1e6, q, n[:60]))
PY
