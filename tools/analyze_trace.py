"""per-dispatch durations of the training kernels along a run, from a rocprofv3 kernel trace (tools/gpu_r02_trainprof.sh)"""
import csv, sys
import numpy as np
rows = list(csv.DictReader(open(sys.argv[1])))
names = sys.argv[2].split(",")
seq = {n: [] for n in names}
for r in rows:
    for n in names:
        if n in r["Kernel_Name"]:
            seq[n].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
for n in names:
    seq[n].sort()
    d = np.array([x[1] for x in seq[n]]) / 1e3
    if not d.size:
        continue
    d = d[d.size // 2:]  # the second (timed) training
    print(n, d.size, "pct5/25/50/75/95", np.percentile(d, [5, 25, 50, 75, 95]).round(1), "mean", d.mean().round(2))
    for lo in (0, 100, 1000, 3000, 5000, 7000):
        if lo < d.size:
            print("   steps", lo, d[lo:lo + 12].round(1))
