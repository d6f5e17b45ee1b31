#!/bin/bash
# SQ / LDS counters of the adeclick kernels inside the bench workload (two PMC passes, kernel-trace only)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/pmc_dk
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR --output-format csv -d gpurun_out/pmc_dk/a -o k -- python tools/bench_declick.py 1 > gpurun_out/pmc_dk_a.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_dk/b -o k -- python tools/bench_declick.py 1 > gpurun_out/pmc_dk_b.log 2>&1
python - <<PY
import csv, glob
for kern in ("k_adeclick_fast<512", "k_dk_solve<32", "k_dk_solve<64"):
    print("==", kern)
    for d in ("a", "b"):
        acc = {}
        for f in glob.glob("gpurun_out/pmc_dk/%s/*counter_collection.csv" % d):
            for r in csv.DictReader(open(f)):
                if kern not in r["Kernel_Name"]: continue
                acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        for k, v in acc.items(): print("%-24s %14.0f  (calls %d)" % (k, max(v), len(v)))
PY
