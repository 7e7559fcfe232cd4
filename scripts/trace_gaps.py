"""Timeline of the step kernels from a rocprofv3 --kernel-trace csv: start offsets, durations and the gaps between
consecutive kernels (usage: trace_gaps.py <kernel_trace.csv> [first_row] [rows] [name,name,...])."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
if len(sys.argv) > 4:                      # keep only kernels whose name contains one of these comma-separated words
    words = sys.argv[4].split(",")
    rows = [r for r in rows if any(w in r["Kernel_Name"] for w in words)]
first = int(sys.argv[2]) if len(sys.argv) > 2 else len(rows) // 2
cnt = int(sys.argv[3]) if len(sys.argv) > 3 else 24
prev_end = None
t0 = int(rows[first]["Start_Timestamp"])
for r in rows[first:first + cnt]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = "" if prev_end is None else f"gap {(s - prev_end) / 1e3:7.1f}"
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  {gap:12s} {r['Kernel_Name'][:60]}")
    prev_end = e
