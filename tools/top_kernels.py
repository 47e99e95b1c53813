#!/usr/bin/env python3
"""Print the top-N rows of a rocprofv3 kernel_stats.csv with shortened kernel names: top_kernels.py <dir-or-csv> [N]."""
import csv, glob, os, sys
path = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 15
if os.path.isdir(path):
    path = sorted(glob.glob(os.path.join(path, "**", "*kernel_stats.csv"), recursive=True))[0]
rows = list(csv.DictReader(open(path)))
total = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"{path}: {len(rows)} kernels, total {total / 1e6:.2f} ms")
for r in rows[:n]:
    print(f'{float(r["TotalDurationNs"]) / 1e6:9.2f} ms {r["Percentage"]:>6}%  x{r["Calls"]:>5}  avg {float(r["AverageNs"]) / 1e3:9.1f} us  {r["Name"][:70]}')
