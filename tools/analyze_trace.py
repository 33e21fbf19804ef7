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
starts = {}
for n in names:
    seq[n].sort()
    d = np.array([x[1] for x in seq[n]]) / 1e3
    if not d.size:
        continue
    live = d[d > 2.5]  # launches past the end of a run return at once
    print(n, d.size, "live", live.size, "pct5/25/50/75/95/99", np.percentile(live, [5, 25, 50, 75, 95, 99]).round(1), "mean", live.mean().round(2),
          "sum ms", (live.sum() / 1e3).round(2))
    order = np.sort(live)[::-1]
    print("   top-50 sum ms", (order[:50].sum() / 1e3).round(2), "top-500", (order[:500].sum() / 1e3).round(2), "max", order[:5].round(0))
    for lo in (0, 100, 1000, 3000, 5000, 7000, d.size // 2, d.size // 2 + 100, d.size // 2 + 3000, d.size // 2 + 7000):
        if lo < d.size:
            print("   launch", lo, d[lo:lo + 12].round(1))
# gaps between consecutive launches of the two kernels
allk = sorted((s, s + int(dur)) for n in names for s, dur in seq[n])
gaps = np.array([allk[i + 1][0] - allk[i][1] for i in range(len(allk) - 1)]) / 1e3
gaps = gaps[(gaps >= 0) & (gaps < 50)]
print("gap between launches: pct50/95", np.percentile(gaps, [50, 95]).round(2), "mean", gaps.mean().round(2))
