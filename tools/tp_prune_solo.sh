cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/tpp; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/tpp -o k -- python tools/tp_prune_solo.py > gpurun_out/tpp.log 2>&1
cat gpurun_out/tpp.log | tail -12
grep -E "k_tp_|k_upsample32|k_kw1" gpurun_out/tpp/k_kernel_stats.csv | cut -c1-160
python - <<PY
import csv
rows=[r for r in csv.DictReader(open("gpurun_out/tpp/k_kernel_trace.csv")) if "k_tp_" in r["Kernel_Name"] or "k_upsample32" in r["Kernel_Name"]]
for r in rows: print(r["Kernel_Name"][:40], (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6, r["Grid_Size"] if "Grid_Size" in r else "")
PY
