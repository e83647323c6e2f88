"""MFMA-busy per kernel from one rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE) over bench.py.

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d <dir> -- python3 bench.py --steps 3 --warmup 2 --frames 3 --cpu-scans 0 --no-pipeline
    python profiles/mfma_busy_summary.py <counter_collection.csv>

SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over all SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs
(MI355X_MICROARCH.md), so busy = MFMA_BUSY / (GUI_ACTIVE / 8 * 1024 SIMDs).  Kernels shorter than ~0.3 ms read high on
GUI_ACTIVE (the guide's DVFS note), i.e. their busy fraction is a lower bound.

With a third argument (the labels.json of the same run) the second table reconciles the counter with the FLOP rate, launch by
launch ("58 % busy vs 68 % of peak"): an fp32 MFMA delivers 64 FLOP per cycle and SIMD, so a launch that issues F
matrix FLOPs keeps the pipes busy for F / 64 cycles (summed over SIMDs) -- `expected`; `counter` is what
SQ_VALU_MFMA_BUSY_CYCLES reports for the launch (ratio ~1: the counter's unit is cycles summed over SIMDs);
`gui clock` = GUI_ACTIVE / 8 / launch duration, the shader clock the denominator implies (the chip does not run above 2.4 GHz:
anything higher is GUI_ACTIVE counting cycles the launch's kernel was not running, which is what pulls `busy (counter)` below
the FLOP rate on launches of 20 .. 200 us); `busy @2.4 GHz` = expected / (1024 SIMDs x duration x 2.4 GHz) is the fraction of
the fp32 matrix peak the launch sustained in this (profiled, serial) run -- the number the roofline's `frac` corresponds to."""
import collections
import csv
import sys

rows_in = list(csv.DictReader(open(sys.argv[1])))
# one steady-state step = the dispatches between the last two smos::tta_argmax launches (MIOpen's search kernels of the
# warm-up would otherwise dominate the table)
marks = sorted({int(r["Dispatch_Id"]) for r in rows_in if "tta_argmax" in r["Kernel_Name"]})
lo, hi = (marks[-2], marks[-1]) if len(marks) >= 2 else (-1, 1 << 62)
acc = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
for r in rows_in:
    if not (lo < int(r["Dispatch_Id"]) <= hi):
        continue
    k = r["Kernel_Name"]
    k = k[:k.index("(")] if "(" in k and not k.startswith("_Z") else k[:90]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        calls[k] += 1
rows = []
for k, c in acc.items():
    gui, mf = c.get("GRBM_GUI_ACTIVE", 0.0), c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    if gui <= 0:
        continue
    rows.append((gui, mf / (gui / 8.0 * 1024.0), calls[k], k))
tot = sum(r[0] for r in rows)
print("%-7s %-9s %-6s %s" % ("share", "MFMAbusy", "calls", "kernel"))
for gui, busy, n, k in sorted(rows, reverse=True)[:28]:
    print("%5.1f %%  %6.1f %%  %5d  %s" % (100 * gui / tot, 100 * busy, n, k))

if len(sys.argv) > 2:
    import json
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import kinds
    import bench
    labels = json.load(open(sys.argv[2]))
    by_disp = collections.defaultdict(dict)
    for r in rows_in:
        d = by_disp[int(r["Dispatch_Id"])]
        d["Kernel_Name"] = r["Kernel_Name"]
        d[r["Counter_Name"]] = float(r["Counter_Value"])
        if "Start_Timestamp" in r and r["Start_Timestamp"]:
            d["dur_ns"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    disp = [dict(v, Dispatch_Id=k) for k, v in by_disp.items()]
    step = kinds.steady_step(disp, key=lambda r: r["Dispatch_Id"])
    pairs, problems = kinds.match(step, labels)
    agg = collections.OrderedDict()
    for label, ds in pairs:
        if label is None:
            continue
        fl = bench.executed_launch_flops(label)
        if not fl:
            continue
        a = agg.setdefault(label, [0, 0.0, 0.0, 0.0, fl])
        a[0] += 1
        a[1] += sum(d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) for d in ds)
        a[2] += sum(d.get("GRBM_GUI_ACTIVE", 0.0) for d in ds)
        a[3] += sum(d.get("dur_ns", 0) for d in ds)
    print()
    print("%-52s %3s %11s %11s %6s %9s %9s %9s" % ("label", "n", "expected", "counter", "ratio", "gui clock", "busy(ctr)", "busy@2.4"))
    for label, (n, mf, gui, dur, fl) in sorted(agg.items(), key=lambda kv: -kv[1][3]):
        exp = fl / 64.0 * n
        clock = gui / 8.0 / dur if dur else float("nan")            # cycles per ns = GHz
        print("%-52s %3d %11.3e %11.3e %6.2f %7.2f GHz %8.1f%% %8.1f%%" % (
            label[:52], n, exp / n, mf / n, mf / exp if exp else 0.0, clock, 100 * mf / (gui / 8.0 * 1024.0) if gui else 0.0,
            100 * exp / (1024.0 * dur * 2.4) if dur else 0.0))
