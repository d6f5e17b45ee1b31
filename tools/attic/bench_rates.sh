#!/bin/bash
# bench.py at other input rates (run on the GPU box): 88.2 kHz runs; rates below 41 kHz are refused as the reference refuses them (lowpass above Nyquist)
for r in 88200 32000 22050; do
  timeout -k 5 200 python bench.py --rate $r --steps 3 --warmup 1 --cpu-sample 0 --e2e 0 > /tmp/r_$r.txt 2>&1
  python - $r <<PY
import json, sys
l = open("/tmp/r_%s.txt" % sys.argv[1]).read().strip().split("\n")[-1]
try:
    d = json.loads(l); print(sys.argv[1], d["ms_per_step"], d["value"], d["pass_ms"])
except Exception: print(sys.argv[1], l[-300:])
PY
done
