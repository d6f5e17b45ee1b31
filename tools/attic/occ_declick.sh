cd "$GRAFT_REPO_ROOT"
for w in 3 5 8 10; do echo -n "waves/CU=$w "; JT_DK_WAVES=$w python tools/bench_declick.py 5 2>/dev/null | tail -1; done
