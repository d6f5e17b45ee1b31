cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/tps; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/tps -o k -- python tools/tp_solo.py > gpurun_out/tps.log 2>&1
grep -E "k_tp_stream|k_upsample32|k_kw1" gpurun_out/tps/k_kernel_stats.csv | cut -c1-200
