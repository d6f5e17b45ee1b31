#!/bin/bash
# Deterministic cost proxy for adeclick (issue-bound): VALU / SALU / LDS instructions per window of the level-0 kernel (PMC pass)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/pmc_dk; timeout 500 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --output-format csv -d gpurun_out/pmc_dk -o k -- python tools/bench_declick.py 1 > gpurun_out/pmc_dk.log 2>&1
python - <<EOF
import csv, glob
for f in glob.glob("gpurun_out/pmc_dk/*counter_collection.csv"):
    acc = {}
    for r in csv.DictReader(open(f)):
        for tag in ("k_adeclick<512", "k_adeclick<1024"):
            if tag in r["Kernel_Name"]:
                acc.setdefault((tag, r["Counter_Name"]), []).append(float(r["Counter_Value"]))
    nwin = 158760000 / 1212.0
    for (tag, k), v in sorted(acc.items()):
        print(tag, k, "%.0f" % max(v), "per window %.1f" % (max(v) / nwin))
EOF
