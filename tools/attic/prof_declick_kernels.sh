#!/bin/bash
# per-kernel times of the adeclick launches inside Pass 4 (rocprofv3 --kernel-trace --stats over tools/bench_declick.py)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/dkprof; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/dkprof -o k -- python tools/bench_declick.py 4 > gpurun_out/dkprof.log 2>&1
f=$(ls gpurun_out/dkprof/*kernel_stats.csv 2>/dev/null | head -1)
[ -z "$f" ] && { tail -5 gpurun_out/dkprof.log; exit 1; }
python tools/kstats.py "$f" 5 | grep -E "declick|dk_solve|total"
