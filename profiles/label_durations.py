"""Per-LABEL kernel durations of one steady-state step (the kernel-stats CSVs aggregate by kernel
template, so the bench line's dominant label could not be read off them).

    rocprofv3 --kernel-trace --output-format csv -d <dir> -- python3 bench.py --steps 20 --warmup 6 --cpu-scans 0 --no-raw \
        --no-pipeline --label-log labels.json            (one stream: dispatch order = launch order)
    python profiles/label_durations.py <kernel_trace.csv> labels.json [bench_line.json] > profiles/rNN_label_durations.csv

Columns: label (or the kernel name for launches without one), kernel family (the kernel template the label ran on), launches
per step, mean us per launch, algorithmic FLOPs and bytes per launch (bench.algorithmic_flops / algorithmic_bytes), executed
FLOPs (Winograd forms issue 4/9 resp. 2/3 of the direct count), bound, achieved, frac of the roof (157.3 TFLOP/s fp32 MFMA /
8 TB/s HBM; MFMA-bound rows: `frac` on EXECUTED FLOPs, `alg_frac` on the direct-form count).  Below the label rows: one
`family:` row per kernel family = what bench.py's `roofline` object reports for the dominant one (must agree within 5 %).
"""
import collections
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import kinds  # noqa: E402

import bench  # noqa: E402

rows = list(csv.DictReader(open(sys.argv[1])))
labels = json.load(open(sys.argv[2]))
ctx = {}
if len(sys.argv) > 3:
    try:
        line = json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])
        if line.get("stem_rows_per_launch") is not None:
            ctx["stem_rows"] = line["stem_rows_per_launch"]
        if line.get("stem_class_rows"):
            ctx["stem_class_rows"] = line["stem_class_rows"]
        if line.get("live_fraction") is not None:
            ctx["live_fraction"] = line["live_fraction"]
    except (OSError, ValueError, IndexError):
        pass
step = kinds.steady_step(rows, key=lambda r: int(r["Start_Timestamp"]))
pairs, problems = kinds.match(step, labels)
for p in problems:
    sys.stderr.write("label_durations: %s -- left unlabelled\n" % p)
def family_of(label, kernel_name):
    for frag in ("conv_wino1d", "conv_wino", "conv_igemm", "conv_rows"):
        if frag in kernel_name:
            return frag
    return label.split("[", 1)[0] if label else kernel_name.split("(")[0].split("<")[0][:60]


agg = collections.OrderedDict()
for label, disp in pairs:
    name = label if label else disp[0]["Kernel_Name"].split("(")[0][:90]
    d = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in disp) / 1e3
    a = agg.setdefault(name, [0, 0.0, label is not None, family_of(label, disp[0]["Kernel_Name"])])
    a[0] += 1
    a[1] += d
w = csv.writer(sys.stdout)
w.writerow(["label", "family", "launches_per_step", "mean_us", "total_us", "alg_flops", "exec_flops", "alg_bytes", "bound", "achieved",
            "unit", "frac", "alg_frac"])
total = 0.0
fams = collections.OrderedDict()
for name, (n, us, labelled, fam) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    total += us
    mean = us / n
    fl = by = ex = 0
    if labelled:
        try:
            fl, by = bench.algorithmic_flops(name, ctx), bench.algorithmic_bytes(name, ctx)
            ex = bench.executed_launch_flops(name, ctx=ctx, family=fam)
        except (ValueError, KeyError, IndexError):
            pass
    f = fams.setdefault(fam, [0, 0.0, 0.0, 0.0, 0.0, labelled])
    f[0] += n
    f[1] += us
    f[2] += n * fl
    f[3] += n * ex
    f[4] += n * by
    t_m, t_h = ex / (bench.FP32_PEAK_TFLOPS * 1e12), by / (bench.HBM_PEAK_GBS * 1e9)
    if ex and t_m > t_h:
        ach = ex / (mean * 1e-6) / 1e12
        w.writerow([name, fam, n, "%.2f" % mean, "%.1f" % us, fl, ex, by, "mfma", "%.1f" % ach, "TFLOP/s", "%.4f" % (ach / bench.FP32_PEAK_TFLOPS),
                    "%.4f" % (fl / (mean * 1e-6) / 1e12 / bench.FP32_PEAK_TFLOPS)])
    elif by:
        ach = by / (mean * 1e-6) / 1e9
        w.writerow([name, fam, n, "%.2f" % mean, "%.1f" % us, fl, ex, by, "hbm", "%.1f" % ach, "GB/s", "%.4f" % (ach / bench.HBM_PEAK_GBS), ""])
    else:
        w.writerow([name, fam, n, "%.2f" % mean, "%.1f" % us, "", "", "", "", "", "", "", ""])
w.writerow(["# kernels of the step: %d, summed %.1f us" % (len(step), total)])
w.writerow(["# per kernel family (launches, us, FLOPs and bytes are sums over the step; frac as bench.py's roofline object computes it)"])
for fam, (n, us, fl, ex, by, labelled) in sorted(fams.items(), key=lambda kv: -kv[1][1]):
    t_m, t_h = ex / (bench.FP32_PEAK_TFLOPS * 1e12), by / (bench.HBM_PEAK_GBS * 1e9)
    if ex and t_m > t_h:
        ach = ex / (us * 1e-6) / 1e12
        w.writerow(["family:" + fam, fam, n, "%.2f" % (us / n), "%.1f" % us, int(fl), int(ex), int(by), "mfma", "%.1f" % ach, "TFLOP/s",
                    "%.4f" % (ach / bench.FP32_PEAK_TFLOPS), "%.4f" % (fl / (us * 1e-6) / 1e12 / bench.FP32_PEAK_TFLOPS)])
    elif by:
        ach = by / (us * 1e-6) / 1e9
        w.writerow(["family:" + fam, fam, n, "%.2f" % (us / n), "%.1f" % us, int(fl), int(ex), int(by), "hbm", "%.1f" % ach, "GB/s",
                    "%.4f" % (ach / bench.HBM_PEAK_GBS), ""])
    else:
        w.writerow(["family:" + fam, fam, n, "%.2f" % (us / n), "%.1f" % us, "", "", "", "", "", "", "", ""])
