#!/bin/bash
# Are several files in flight on one GPU bound by the HIP runtime's per-process locks?  N processes x K files in flight each, same GPU.
cd "$GRAFT_REPO_ROOT"
for cfg in "1 6" "2 3" "3 2" "6 1"; do
  set -- $cfg; np=$1; k=$2
  rm -f gpurun_out/mp_*.json
  t0=$(date +%s.%N)
  for i in $(seq 1 $np); do
    python bench.py --minutes 10 --in-flight $k --steps 20 --warmup 2 --e2e 0 --saturation 0 --cpu-sample 0 > gpurun_out/mp_$i.json 2>/dev/null &
  done
  wait
  python - "$np" "$k" <<PY
import json, sys, glob
np_, k = int(sys.argv[1]), int(sys.argv[2])
tot = 0.0
for f in sorted(glob.glob("gpurun_out/mp_*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    s = d.get("resident_in_flight")
    ms = s["ms_per_file"] if s else d["ms_per_step"]
    tot += 1.0 / ms
print(f"{np_} processes x {k} in flight: {1.0 / tot:.2f} ms per 10-min file overall ({600.0 * tot * 1e3:.0f} xRT)")
PY
done
