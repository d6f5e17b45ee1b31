#!/bin/bash
# SQ counters of the true-peak sweeps (k_upsample32<..., 0, 4> at 48 kHz, <..., 0, 1> at 44.1 kHz) inside one pipeline run
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/pmc_tp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SMEM --output-format csv -d gpurun_out/pmc_tp/a -o k -- python tools/bench_declick.py 1 > gpurun_out/pmc_tp_a.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_tp/b -o k -- python tools/bench_declick.py 1 > gpurun_out/pmc_tp_b.log 2>&1
python - <<PY
import csv, glob
for kern in ("k_upsample32<float, double, double, 0, 4>", "k_upsample32<float, double, double, 0, 1>", "k_upsample32_stream8"):
    print("==", kern)
    for d in ("a", "b"):
        acc = {}
        for f in glob.glob("gpurun_out/pmc_tp/%s/*counter_collection.csv" % d):
            for r in csv.DictReader(open(f)):
                if kern not in r["Kernel_Name"]: continue
                acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        for k, v in acc.items(): print("  %-24s %16.0f  (largest of %d launches)" % (k, max(v), len(v)))
PY
