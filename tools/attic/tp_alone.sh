#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/tpa; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/tpa -o k -- python tools/tp_alone.py > gpurun_out/tpa.log 2>&1
python tools/kstats.py gpurun_out/tpa/k_kernel_stats.csv 3 | head -8
