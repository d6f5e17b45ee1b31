#!/bin/bash
# A/B helper (run on the GPU box): adeclick phase clocks (JT_DK_PROFILE + JT_DK_SPLIT0 build) for builds selected by EXTRA macros
cd "$GRAFT_REPO_ROOT"
for flag in "$@"; do
  [ "$flag" = "-" ] && flag=""
  touch jivetalking_amd/csrc/k_declick.hip
  make -s -C jivetalking_amd/csrc EXTRA="-DJT_DK_PROFILE -DJT_DK_SPLIT0 $flag" 2>&1 | grep -E " error"
  echo "EXTRA='$flag'"; JT_DK_PROFILE=1 python tools/bench_declick.py 2 2>&1 | grep -E "phase clocks|declick_ms|adeclick split" | tail -3 | sed 's/.*phase clocks (JT_DK_PROFILE build)://'
done
touch jivetalking_amd/csrc/k_declick.hip; make -s -C jivetalking_amd/csrc 2>&1 | grep -E " error"
