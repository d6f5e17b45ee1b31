#!/bin/bash
# per-kernel ms per step of the bench command (rocprofv3 --kernel-trace --stats); optional grep pattern
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/kprof; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kprof -o k -- python bench.py --steps 4 --warmup 1 --cpu-sample 0 --e2e 0 --saturation 0 > gpurun_out/kprof.log 2>&1
python tools/kstats.py gpurun_out/kprof/k_kernel_stats.csv 5 | grep -E "${1:-.}"
