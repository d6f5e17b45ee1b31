#!/bin/bash
# Round artefacts on the GPU box: bench line, rocprofv3 kernel stats, the two PMC passes.  Usage: tools/profile_round.sh r01
set -u
tag=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$tag; mkdir -p $out
python bench.py > $out/bench.json 2> $out/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o k -- python bench.py --steps 3 --warmup 1 --cpu-sample 0 --e2e 0 > $out/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch -o k -- python bench.py --steps 1 --warmup 1 --cpu-sample 0 --e2e 0 > $out/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/write -o k -- python bench.py --steps 1 --warmup 1 --cpu-sample 0 --e2e 0 > $out/write.log 2>&1
cp $(ls $out/stats/*kernel_stats.csv | head -1) $out/kernel_stats.csv
cp $(ls $out/fetch/*counter_collection.csv | head -1) $out/pmc_fetch_size.csv
cp $(ls $out/write/*counter_collection.csv | head -1) $out/pmc_write_size.csv
python tools/pmc_summary.py $out/pmc_fetch_size.csv $out/pmc_write_size.csv $out/pmc_traffic.json
rm -rf $out/stats $out/fetch $out/write
tail -c 600 $out/bench.json; python tools/kstats.py $out/kernel_stats.csv 4 | tail -30
