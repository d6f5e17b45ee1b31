#!/bin/bash
# phase clocks of the workgroup dynamic-loudnorm kernel (a JT_LN_PROFILE build of k_loudnorm.hip, restored afterwards)
cd "$GRAFT_REPO_ROOT/jivetalking_amd/csrc"
rm -f build/k_loudnorm.o; make -s EXTRA="-DJT_LN_PROFILE" >/dev/null 2>&1 || { echo build failed; exit 1; }
(cd "$GRAFT_REPO_ROOT" && timeout 300 python tools/dyn_fallback_time.py 2>&1 | grep -E "^batch|^per-peak|loudnorm dynamic" | tail -8)
rm -f build/k_loudnorm.o; make -s >/dev/null 2>&1
