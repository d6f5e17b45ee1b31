cd "$GRAFT_REPO_ROOT"
touch jivetalking_amd/csrc/k_declick.hip
make -s -C jivetalking_amd/csrc EXTRA="-DJT_DK_PROFILE -DJT_DK_SPLIT0" 2>&1 | grep -E " error"
JT_DK_PROFILE=1 python tools/bench_declick.py 2 2>&1 | grep -E "adeclick|declick_ms" | tail -2
