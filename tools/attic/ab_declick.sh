#!/bin/bash
# A/B helper (run on the GPU box): adeclick launch time for builds selected by EXTRA macros ("-" = default build)
cd "$GRAFT_REPO_ROOT"
for flag in "$@"; do
  [ "$flag" = "-" ] && flag=""
  touch jivetalking_amd/csrc/k_declick.hip
  make -s -C jivetalking_amd/csrc EXTRA="$flag" 2>&1 | grep -E " error"
  echo -n "EXTRA='$flag' "; python tools/bench_declick.py 10 2>/dev/null | tail -1
done
