#!/bin/bash
# The round's committed profiles (run on the GPU box; results under gpurun_out/rprof, copy what is judged into profiles/):
#   kernel stats of the bench command, the two HBM-traffic PMC passes (FETCH_SIZE / WRITE_SIZE, separate runs, kernel-trace only),
#   the issue / wait picture, one step's timeline.  usage: tools/profile_round.sh <tag> [git hash]
tag="${1:-r03}"; git="${2:-}"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/rprof; rm -rf $O; mkdir -p $O
BENCH="python bench.py --steps 4 --warmup 1 --cpu-sample 0 --e2e 0 --saturation 0"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o k -- $BENCH > $O/bench_stats.log 2>&1
cp $O/stats/k_kernel_stats.csv $O/${tag}_kernel_stats_60min.csv
python tools/kstats.py $O/stats/k_kernel_stats.csv 5 > $O/${tag}_kernel_ms_per_step.txt
B1="python bench.py --steps 1 --warmup 1 --cpu-sample 0 --e2e 0 --saturation 0"
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o k -- $B1 > $O/fetch.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o k -- $B1 > $O/write.log 2>&1
cp $O/fetch/k_counter_collection.csv $O/${tag}_pmc_fetch_size.csv; cp $O/write/k_counter_collection.csv $O/${tag}_pmc_write_size.csv
python tools/pmc_summary.py $O/${tag}_pmc_fetch_size.csv $O/${tag}_pmc_write_size.csv $O/${tag}_pmc_traffic.json "$git"
bash tools/pmc_issue.sh > $O/${tag}_pmc_issue.txt 2>&1
bash tools/pmc_mix.sh $O/${tag}_kernel_stats_60min.csv > $O/${tag}_pmc_mix.txt 2>&1
bash tools/timeline.sh > $O/${tag}_timeline_one_step.txt 2>&1
python bench.py > $O/${tag}_bench_60min.json 2> $O/bench.err
ls -la $O | head -30; tail -3 $O/${tag}_kernel_ms_per_step.txt; head -c 600 $O/${tag}_bench_60min.json
