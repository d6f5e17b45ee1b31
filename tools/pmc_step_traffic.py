"""HBM counter traffic of ONE step of the bench (our kernels between two consecutive k_frame_stats launches, the second step of the run):
per kernel and in total, as reported (FETCH_SIZE + WRITE_SIZE) and corrected (2 x FETCH_SIZE + WRITE_SIZE).
The correction is CALIBRATED, not assumed (tools/fetch_calib.sh -> profiles/r05_fetch_calib.txt, 2 GiB sweeps on gfx950 / rocprofv3 of
ROCm 7.2): every coalesced read width this library uses -- 2, 4, 8 and 16 bytes per lane -- reports exactly 0.500 of the bytes moved,
WRITE_SIZE reports 1.000 at every width.  Every kernel here reads PCM as 64-lane-coalesced rows, so the factor
applies to all of them; for an UNcoalesced reader (a lane walking its own chunk straight from global memory: 6.2 x over-fetch in the
calibration) the doubling of the over-fetched lines is not calibrated, and no kernel of the step reads that way.
usage: pmc_step_traffic.py profiles/rNN_pmc_fetch_size.csv profiles/rNN_pmc_write_size.csv"""
import csv, re, sys, collections


def step_rows(path, key):
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == key]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("k_frame_stats")]
    lo, hi = marks[1], marks[2] if len(marks) > 2 else len(rows)
    return rows[lo:hi]


def short(n):
    n = re.sub(r"^void ", "", n).replace("(anonymous namespace)::", "")
    return n.split("(")[0]


tot = collections.defaultdict(lambda: [0.0, 0.0])
for path, key, col in ((sys.argv[1], "FETCH_SIZE", 0), (sys.argv[2], "WRITE_SIZE", 1)):
    for r in step_rows(path, key):
        n = r["Kernel_Name"]
        if "at::native" in n or "rocprim" in n or "flac" in n: continue      # (the FLAC legs are outside the step)
        tot[short(n)][col] += float(r["Counter_Value"]) * 1024.0
rows = sorted(tot.items(), key=lambda kv: -(2 * kv[1][0] + kv[1][1]))
S = sum(2 * f + w for _, (f, w) in rows)
for n, (f, w) in rows[:25]:
    print("%-50s fetch x2 %7.2f GB  write %7.2f GB" % (n[:50], 2 * f / 1e9, w / 1e9))
R = sum(f + w for _, (f, w) in rows)
print("one step, all of our kernels: %.1f GB corrected (2 x FETCH_SIZE + WRITE_SIZE; calibration: profiles/r05_fetch_calib.txt), %.1f GB as reported" % (S / 1e9, R / 1e9))
