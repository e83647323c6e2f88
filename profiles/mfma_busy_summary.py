"""MFMA-busy per kernel from one rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE) over bench.py.

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d <dir> -- python3 bench.py --steps 3 --warmup 2 --frames 3 --cpu-scans 0 --no-pipeline
    python profiles/mfma_busy_summary.py <counter_collection.csv>

SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over all SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs
(MI355X_MICROARCH.md), so busy = MFMA_BUSY / (GUI_ACTIVE / 8 * 1024 SIMDs).  Kernels shorter than ~0.3 ms read high on
GUI_ACTIVE (the guide's DVFS note), i.e. their busy fraction is a lower bound."""
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
