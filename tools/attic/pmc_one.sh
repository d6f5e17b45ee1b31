#!/bin/bash
# PMC passes for one command (run on the GPU box): WRITE_SIZE, FETCH_SIZE, then the SQ issue counters; prints the rows of kernels matching $1
# usage: tools/pmc_one.sh <kernel-regex> <command...>
pat=$1; shift
cd /tmp && export TMPDIR=/tmp
for set in "WRITE_SIZE" "FETCH_SIZE" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR"; do
  rm -rf /tmp/pmc1; timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pmc1 -o k -- "$@" > /tmp/pmc1.log 2>&1
  python3 - "$pat" <<PY
import csv, glob, re, sys
acc = {}
for f in glob.glob("/tmp/pmc1/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if not re.search(sys.argv[1], n): continue
        n = re.sub(r"^void ", "", n).split("(")[0]
        d = acc.setdefault(n, {})
        d[r["Counter_Name"]] = max(d.get(r["Counter_Name"], 0.0), float(r["Counter_Value"]))
for n, d in acc.items(): print(n[:60], {k: round(v / 1e6, 2) for k, v in d.items()}, "(millions; *_SIZE in KB -> GB)")
PY
done
