#!/bin/bash
# A/B of the per-XCD window heads of the adeclick front launches: kernel times and HBM fetch / write bytes with and without
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in xcd noxcd; do
  [ $v = noxcd ] && export JT_DK_NO_XCD=1 || unset JT_DK_NO_XCD
  rm -rf gpurun_out/xcd_$v
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/xcd_$v/t -o k -- python tools/bench_declick.py 4 > gpurun_out/xcd_$v.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/xcd_$v/f -o k -- python tools/bench_declick.py 1 >> gpurun_out/xcd_$v.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/xcd_$v/w -o k -- python tools/bench_declick.py 1 >> gpurun_out/xcd_$v.log 2>&1
  echo "== $v"
  python tools/kstats.py $(ls gpurun_out/xcd_$v/t/*kernel_stats.csv | head -1) 5 | grep -E "declick|dk_solve|dk_lev"
  python - <<PY
import csv, glob
for d, nm in (("f", "FETCH_SIZE"), ("w", "WRITE_SIZE")):
    acc = {}
    for f in glob.glob("gpurun_out/xcd_$v/%s/*counter_collection.csv" % d):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "declick" not in k and "dk_" not in k: continue
            acc.setdefault(k[:60], []).append(float(r["Counter_Value"]))
    for k, vals in sorted(acc.items()): print("%-10s %-62s max %12.0f KB  calls %d" % (nm, k, max(vals), len(vals)))
PY
done
