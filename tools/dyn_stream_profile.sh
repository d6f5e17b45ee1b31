#!/bin/bash
# rocprofv3 kernel stats of the dynamic-loudnorm fallback file (tools/dyn_fallback_time.py, stream path only): where its Pass 4 goes
cd /tmp && export TMPDIR=/tmp
out="$GRAFT_REPO_ROOT/gpurun_out/dynprof"; rm -rf "$out"; mkdir -p "$out"
PYTHONPATH="$GRAFT_REPO_ROOT" JT_DYN_DIAG=1 JT_DYN_MODES=stream rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o dyn -- python "$GRAFT_REPO_ROOT/tools/dyn_fallback_time.py" > "$out/run.log" 2>&1
grep -E "^stream|stream path|delivered" "$out/run.log"
f=$(find "$out" -name "*kernel_stats.csv" | head -1)
python "$GRAFT_REPO_ROOT/tools/kstats.py" "$f" 5 | sort -t's' -k3 | head -60
