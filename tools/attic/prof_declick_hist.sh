cd "$GRAFT_REPO_ROOT"
touch jivetalking_amd/csrc/k_declick.hip
make -s -C jivetalking_amd/csrc EXTRA="-DJT_DK_PROFILE -DJT_DK_HIST" 2>&1 | grep -E " error"
JT_DK_PROFILE=1 python tools/bench_declick.py 1 2>&1 | grep -E "bw|F/16" | tail -3
