"""Summarise a rocprofv3 kernel_stats.csv: per-step time of our kernels (skips the torch generator kernels of bench.py)."""
import csv, sys
path, steps = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
rows = list(csv.DictReader(open(path)))
tot = 0.0
for r in rows:
    nm = r["Name"]
    if "at::native" in nm or "rocclr" in nm:
        continue
    ms = float(r["TotalDurationNs"]) / 1e6 / steps
    tot += ms
    if ms >= 0.05:
        print(f"{nm[:72]:72s} calls/step {int(r['Calls'])/steps:5.1f}  ms/step {ms:7.3f}  avg {float(r['AverageNs'])/1e6:7.3f}")
print(f"total of our kernels per step: {tot:.2f} ms")
