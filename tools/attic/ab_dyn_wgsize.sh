#!/bin/bash
# workgroup size of the dynamic-loudnorm kernel: build k_loudnorm.hip with JT_LN_WG = 256 / 512 / 1024 into copies of the library and time each
cd "$GRAFT_REPO_ROOT/jivetalking_amd/csrc"
for wg in 256 512 1024; do
  rm -f build/k_loudnorm.o; make -s EXTRA="-DJT_LN_WG=$wg" >/dev/null 2>&1 || { echo build failed; exit 1; }
  echo "== WG $wg"; (cd "$GRAFT_REPO_ROOT" && timeout 300 python tools/ab_dynamic_wg.py 120 2.5 2>&1 | grep "^wg")
done
rm -f build/k_loudnorm.o; make -s >/dev/null 2>&1
