"""Timeline of the pipelined step from a rocprofv3 kernel trace (``*_kernel_trace.csv``): every kernel of a window in
the middle of the timed region with its start (us, relative), duration, queue, and the idle gap of the device before it.

    rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o kt -- python bench.py --steps 100 --no-cpu-baseline --no-flow-bench
    python scripts/step_timeline.py /tmp/kt [n_rows] [where]     (where: fraction of the accept launches at which the window
                                                                  starts, default 0.2 = inside bench.py's timed region)
"""
import csv
import glob
import sys

d = sys.argv[1]
rows_out = int(sys.argv[2]) if len(sys.argv) > 2 else 60
path = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(path)))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:44], r.get("Queue_Id", "?"))
             for r in rows))
# a window of the steady state
acc = [i for i, e in enumerate(ev) if e[2].startswith("accept_kernel")]
where = float(sys.argv[3]) if len(sys.argv) > 3 else 0.2
i0 = acc[int(len(acc) * where)]
t0 = ev[i0][0]
busy_until = ev[i0][0]
print(f"{'start_us':>9} {'dur_us':>7} {'gap_us':>7}  queue  kernel")
for s, e, name, q in ev[i0:i0 + rows_out]:
    gap = (s - busy_until) / 1e3
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f} {gap if gap > 0 else 0:7.1f}  {q:>5}  {name}")
    busy_until = max(busy_until, e)
# whole steady-state window: device busy fraction
w = ev[i0:i0 + 40 * rows_out]
span = w[-1][1] - w[0][0]
busy, cur_s, cur_e = 0, w[0][0], w[0][1]
for s, e, _, _ in w[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(f"window {span / 1e3:.0f} us, device busy {100.0 * busy / span:.1f} %")
