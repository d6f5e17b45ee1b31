#!/bin/bash
# A/B helper (run on the GPU box): rebuild one source with EXTRA macros ("-" = default) and print the bench line's pass times
cd "$GRAFT_REPO_ROOT"
src=$1; shift
for flag in "$@"; do
  [ "$flag" = "-" ] && flag=""
  touch jivetalking_amd/csrc/$src
  make -s -C jivetalking_amd/csrc EXTRA="$flag" 2>&1 | grep -E " error"
  python bench.py --cpu-sample 0 --steps 6 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('EXTRA=$flag', d['ms_per_step'], d['pass_ms'])"
done
