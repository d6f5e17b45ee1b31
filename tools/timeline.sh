#!/bin/bash
# Kernel timeline of the last bench step (rocprofv3 --kernel-trace): start offset, duration, queue, name
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/tl; timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -o k -- python bench.py --steps 2 --warmup 1 --cpu-sample 0 --e2e 0 --saturation 0 "$@" > gpurun_out/tl.log 2>&1
python - <<EOF
import csv, glob, re
rows = []
for f in glob.glob("gpurun_out/tl/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "at::native" in n or "rocprim" in n: continue
        n = re.sub(r"^void ", "", n).replace("(anonymous namespace)::", "").split("(")[0]
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], n))
rows.sort()
# last step: find the last k_frame_stats launch (start of pass 1)
starts = [i for i, r in enumerate(rows) if r[3].startswith("k_frame_stats")]
i0 = starts[-1]
t0 = rows[i0][0]
for s, e, q, n in rows[i0:]:
    if "flac" in n or "fd::" in n or "fl::" in n: break
    print("%9.3f %8.3f  q%-3s %s" % ((s - t0) / 1e6, (e - s) / 1e6, q, n[:60]))
EOF
