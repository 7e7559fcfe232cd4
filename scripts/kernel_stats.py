"""Top kernels of a rocprofv3 --kernel-trace --stats run:  python scripts/kernel_stats.py <dir> [n]"""
import csv
import glob
import sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total GPU time {tot / 1e6:.1f} ms")
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 14]:
    name = r["Name"].replace("void ", "")[:58]
    print(f"{name:58s} calls {int(r['Calls']):6d}  avg {float(r['AverageNs']) / 1e3:8.1f} us  {float(r['Percentage']):5.1f} %")
