#!/bin/bash
# FETCH_SIZE / WRITE_SIZE per known 2 GiB sweep, per access width -> gpurun_out/fetch_calib.txt (copy to profiles/)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/fc_f gpurun_out/fc_w
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/fc_f -o k -- tools/ubench/fetch_calib.bin > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/fc_w -o k -- tools/ubench/fetch_calib.bin > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
B = 2 << 30
for d, key in (("gpurun_out/fc_f", "FETCH_SIZE"), ("gpurun_out/fc_w", "WRITE_SIZE")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != key: continue
            n = r["Kernel_Name"].split("(")[0]
            print("%-12s %-44s %8.3f GiB reported for 2 GiB moved: reported / moved = %.3f" % (key, n[:44], float(r["Counter_Value"]) * 1024 / (1 << 30), float(r["Counter_Value"]) * 1024 / B))
PY
