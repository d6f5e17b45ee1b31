#!/bin/bash
# SQ counters of the anlmdn kernel alone (one launch set of tools/nlm_time.py); two PMC passes, kernel-trace only
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/pmc_nlm
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d gpurun_out/pmc_nlm/a -o k -- python tools/nlm_time.py 20 > gpurun_out/pmc_nlm_a.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_nlm/b -o k -- python tools/nlm_time.py 20 > gpurun_out/pmc_nlm_b.log 2>&1
python - <<PY
import csv, glob
for d in ("a", "b"):
    acc = {}
    for f in glob.glob("gpurun_out/pmc_nlm/%s/*counter_collection.csv" % d):
        for r in csv.DictReader(open(f)):
            if "anlmdn" not in r["Kernel_Name"]: continue
            acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    for k, v in acc.items(): print("%-24s %14.0f  (calls %d)" % (k, max(v), len(v)))
PY
