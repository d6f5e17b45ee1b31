#!/bin/bash
# Per-phase wall-clock shares of k_adeclick on the bench workload (run on the GPU box): a -DJT_DK_PROFILE build, then the default one back
cd "$GRAFT_REPO_ROOT"
touch jivetalking_amd/csrc/k_declick.hip
make -s -C jivetalking_amd/csrc EXTRA="-DJT_DK_PROFILE" 2>&1 | grep -E " error"
JT_DK_PROFILE=1 python tools/bench_declick.py 2 2>&1 | grep -E "adeclick|declick_ms" | tail -4
touch jivetalking_amd/csrc/k_declick.hip
make -s -C jivetalking_amd/csrc 2>&1 | grep -E " error"
python tools/bench_declick.py 5 2>/dev/null | tail -1
