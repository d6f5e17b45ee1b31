#!/bin/bash
# phase clocks of the workgroup dynamic-loudnorm kernel (a JT_LN_PROFILE build of k_loudnorm.hip, restored afterwards)
cd "$GRAFT_REPO_ROOT/jivetalking_amd/csrc"
rm -f build/k_loudnorm.o; make -s EXTRA="-DJT_LN_PROFILE" >/dev/null 2>&1 || { echo build failed; exit 1; }
(cd "$GRAFT_REPO_ROOT" && timeout 300 python tools/ab_dynamic_wg.py ${1:-120} ${2:-2.5} 2>&1 | grep -E "^wg|loudnorm dynamic")
rm -f build/k_loudnorm.o; make -s >/dev/null 2>&1
