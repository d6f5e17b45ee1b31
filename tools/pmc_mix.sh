#!/bin/bash
# Instruction MIX of the long kernels of one bench step (SQ counters, one PMC pass, kernel-trace only): how many of a kernel's VALU
# wave-instructions are f64 (3.6-3.9 ns per SIMD on this part, profiles/r02_gfx950_op_costs.txt) and how many are 32-bit (1.1-1.9 ns),
# MFMA ops, LDS and scalar instructions -- and from them the ISSUE-TIME FLOOR of the kernel: the time its own instruction stream needs
# on 1024 SIMDs / 256 LDS pipes when nothing ever waits.  What DESIGN.md section 4a's ceilings are computed from.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/pmc_mix
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_LDS SQ_INSTS_SALU --output-format csv -d gpurun_out/pmc_mix -o k -- python bench.py --steps 1 --warmup 1 --cpu-sample 0 --e2e 0 --saturation 0 > gpurun_out/pmc_mix.log 2>&1
python - "$1" <<'PY'
import csv, glob, re, sys
acc = {}
for f in glob.glob("gpurun_out/pmc_mix/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "at::native" in n or "rocclr" in n or "rocprim" in n: continue
        n = re.sub(r"^void ", "", n).replace("(anonymous namespace)::", "").split("(")[0]
        d = acc.setdefault(n, {})
        d[r["Counter_Name"]] = max(d.get(r["Counter_Name"], 0.0), float(r["Counter_Value"]))
# measured durations of the same kernels (avg per launch, ms) from the round's kernel stats, if given
dur = {}
if len(sys.argv) > 1 and sys.argv[1]:
    for r in csv.DictReader(open(sys.argv[1])):
        n = re.sub(r"^void ", "", r["Name"]).replace("(anonymous namespace)::", "").split("(")[0]
        dur[n] = float(r["AverageNs"]) / 1e6
C64, C32, CLDS, CSALU = 3.7, 1.4, 4 / 2.0, 1.0      # ns per wave-instruction per SIMD (f64 / 32-bit VALU at 4 waves per SIMD), LDS: 4 clk of a 2 GHz pipe per b64 access, SALU 1 ns
print("%-40s %8s %8s %8s %7s %8s %8s | %8s %8s %8s" % ("kernel (largest launch)", "VALU(M)", "f64(M)", "MFMAop", "f64 %", "LDS(M)", "SALU(M)", "issue ms", "LDS ms", "launch"))
for n, d in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU", 0)):
    v = d.get("SQ_INSTS_VALU", 0)
    if v < 4e7: continue
    f64 = sum(d.get(k, 0) for k in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64"))
    issue = (f64 * C64 + max(0.0, v - f64) * C32) / 1024 / 1e6
    lds = d.get("SQ_INSTS_LDS", 0) * CLDS / 256 / 1e6
    print("%-40s %8.1f %8.1f %8.1f %6.0f%% %8.1f %8.1f | %8.2f %8.2f %8s" % (n[:40], v / 1e6, f64 / 1e6, d.get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0) / 1e6, 100 * f64 / v if v else 0,
          d.get("SQ_INSTS_LDS", 0) / 1e6, d.get("SQ_INSTS_SALU", 0) / 1e6, issue, lds, ("%.2f" % dur[n]) if n in dur else "-"))
print("issue ms = (f64 instructions x %.1f ns + other VALU x %.1f ns) / 1024 SIMDs; LDS ms = LDS instructions x 4 clk at 2 GHz / 256 CUs (a b64 access; b128 costs twice that)." % (C64, C32))
PY
