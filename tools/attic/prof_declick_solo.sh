#!/bin/bash
# per-phase clocks of k_adeclick_fast with fewer than one wave per SIMD (JT_DK_WAVES=3): the phases' own latency, without contention
cd "$GRAFT_REPO_ROOT"
touch jivetalking_amd/csrc/k_declick.hip
make -s -C jivetalking_amd/csrc EXTRA="-DJT_DK_PROFILE" 2>&1 | grep -E " error"
for w in 3 10; do echo "waves/CU=$w"; JT_DK_WAVES=$w JT_DK_PROFILE=1 python tools/bench_declick.py 1 2>&1 | grep -E "phase clocks|declick_ms" | tail -2; done
touch jivetalking_amd/csrc/k_declick.hip
make -s -C jivetalking_amd/csrc 2>&1 | grep -E " error"
