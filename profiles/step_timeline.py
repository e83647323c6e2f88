"""Launch-ordered timeline of one steady-state step from a rocprofv3 --kernel-trace CSV (best with `bench.py --no-pipeline`:
one stream, dispatch order = launch order): start offset, duration, gap to the end of the previous kernel, stream / queue,
kernel.  usage: python profiles/step_timeline.py <kernel_trace.csv> > profiles/rNN_step_timeline_serial.csv"""
import csv
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import kinds  # noqa: E402

rows = list(csv.DictReader(open(sys.argv[1])))
step = kinds.steady_step(rows, key=lambda r: int(r["Start_Timestamp"]))
t0 = int(step[0]["Start_Timestamp"])
w = csv.writer(sys.stdout)
w.writerow(["start_us", "dur_us", "gap_us", "queue", "kernel"])
prev_end = t0
busy = 0
for r in step:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    w.writerow(["%.2f" % ((s - t0) / 1e3), "%.2f" % ((e - s) / 1e3), "%.2f" % ((s - prev_end) / 1e3), r.get("Queue_Id", ""),
                r["Kernel_Name"].split("(")[0][:80]])
    busy += e - s
    prev_end = max(prev_end, e)
w.writerow(["# %d kernels, wall %.1f us, kernel time %.1f us, gaps %.1f us" %
            (len(step), (prev_end - t0) / 1e3, busy / 1e3, (prev_end - t0 - busy) / 1e3)])
