#!/bin/bash
# Per-kernel issue/wait picture of one bench step (SQ counters, one PMC pass): where a kernel is instruction-issue bound vs parked
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/pmc_issue; timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d gpurun_out/pmc_issue -o k -- python bench.py --steps 1 --warmup 1 --cpu-sample 0 --e2e 0 --saturation 0 > gpurun_out/pmc_issue.log 2>&1
python - <<EOF
import csv, glob, re
acc = {}
for f in glob.glob("gpurun_out/pmc_issue/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "at::native" in n or "rocclr" in n or "rocprim" in n: continue
        n = re.sub(r"^void ", "", n).replace("(anonymous namespace)::", "")
        n = n.split("(")[0]
        d = acc.setdefault(n, {})
        d[r["Counter_Name"]] = max(d.get(r["Counter_Name"], 0.0), float(r["Counter_Value"]))
print("%-44s %9s %7s %7s %7s %9s %9s %9s" % ("kernel (largest launch)", "waveMcyc", "active", "wait", "stall", "VALU(M)", "SALU(M)", "LDS(M)"))
for n, d in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    wc = d.get("SQ_WAVE_CYCLES", 0)
    if wc < 5e7: continue
    print("%-44s %9.0f %6.0f%% %6.0f%% %6.0f%% %9.1f %9.1f %9.1f" % (n[:44], wc / 1e6, 100 * d["SQ_ACTIVE_INST_ANY"] / wc, 100 * d["SQ_WAIT_ANY"] / wc,
          100 * d["SQ_WAIT_INST_ANY"] / wc, d["SQ_INSTS_VALU"] / 1e6, d["SQ_INSTS_SALU"] / 1e6, d["SQ_INSTS_LDS"] / 1e6))
EOF
