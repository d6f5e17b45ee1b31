cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/ans; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ans -o k -- python tools/analysis_solo.py > gpurun_out/ans.log 2>&1
tail -2 gpurun_out/ans.log
cut -d, -f1-4 gpurun_out/ans/k_kernel_stats.csv | sed 's/(.*)//' | head -30
