"""Per-step kernel breakdown from a rocprofv3 --kernel-trace CSV (one steady-state step = the span between
two consecutive smos::tta_argmax launches).  usage: python profiles/step_breakdown.py <kernel_trace.csv> [title]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "tta_argmax" in r["Kernel_Name"]]
# the interval of median duration among the steps of the timed region (the first ones are warm-up, the last ones border on
# what the bench runs after its timed region)
spans = sorted(((int(rows[idx[k + 1]]["End_Timestamp"]) - int(rows[idx[k]]["End_Timestamp"]), k) for k in range(len(idx) // 3, len(idx) - 2)))
k = spans[len(spans) // 2][1]
a, b = idx[k], idx[k + 1]
step = rows[a + 1:b + 1]
t0, t1 = int(step[0]["Start_Timestamp"]), int(step[-1]["End_Timestamp"])
if len(sys.argv) > 2:
    print("# " + sys.argv[2])
print("# one steady-state step: wall %.3f ms, %d kernels" % ((t1 - t0) / 1e6, len(step)))
agg = collections.defaultdict(lambda: [0, 0])
for r in step:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    agg[r["Kernel_Name"][:120]][0] += 1
    agg[r["Kernel_Name"][:120]][1] += d
print("# kernel time summed over all streams %.3f ms" % (sum(v[1] for v in agg.values()) / 1e6))
# time during which at least one / at least two kernels run (the two pipeline stages overlap)
ev = sorted([(int(r["Start_Timestamp"]), 1) for r in step] + [(int(r["End_Timestamp"]), -1) for r in step])
depth, last, any_busy, two_busy = 0, t0, 0, 0
for ts, d in ev:
    if depth >= 1:
        any_busy += ts - last
    if depth >= 2:
        two_busy += ts - last
    depth += d
    last = ts
print("# at least one kernel running %.3f ms (idle %.3f ms), at least two %.3f ms  -- whether rocprofv3 --kernel-trace serialises the dispatches of the two HIP streams differs from box to box: with \"at least two\" near 0 the wall time is the SUM of the kernels and the untraced step is shorter" %
      (any_busy / 1e6, (t1 - t0 - any_busy) / 1e6, two_busy / 1e6))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%9.1f us  x%3d  %s" % (v[1] / 1e3, v[0], k))
