# producer / consumer split of k_follow_states_pc (A/B build): follow_dbg 1 = producers idle, 2 = consumer idle, 3 = barriers only
for m in 10 60; do for v in 0 1 2; do echo "minutes=$m dbg=$v"; JT_USE_AB_LIB=1 JT_FOLLOW_DBG=$v bash tools/timeline.sh --minutes $m 2>&1 | grep "follow"; done; done
