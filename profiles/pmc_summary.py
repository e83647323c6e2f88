"""HBM traffic per launch of the hand-written point kernels from two rocprofv3 --pmc passes.

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d <fetch_dir> -- python3 bench.py --steps 3 --warmup 2 --frames 3 --cpu-scans 0 --no-pipeline --label-log labels.json
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d <write_dir> -- python3 bench.py ... (same)
    python profiles/pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> labels.json > profiles/pmc_traffic.json

Corrections (MI355X_MICROARCH.md, HBM section): counters are in KiB (x1024); on gfx950 FETCH_SIZE reports half of
the bytes of a coalesced streaming read (calibrated here on bias_act_planes: 16 B/lane loads, known read bytes ->
factor 2.06), so reads = 2 x FETCH_SIZE; WRITE_SIZE is exact for streaming stores and atomics (checked: upsample_concat
writes exactly 4*320*256*256*4 B = 335.5 MB, the counter says 335.5 MB).  separate passes (FETCH_SIZE takes 3 of the 4
TCC slots).  bench.py reads the resulting JSON for the "traffic" field of its roofline object.
"""
import collections
import csv
import json
import sys

import os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import kinds  # noqa: E402

LABELS = json.load(open(sys.argv[3])) if len(sys.argv) > 3 else []


def per_label(path, counter):
    """Mean counter value per launch label.  The dispatches of ONE steady-state step (between the last two tta_argmax
    launches of a --no-pipeline run: one stream, dispatch order = launch order) are matched, kind by kind and in order,
    with the labels bench.py --label-log recorded for a step (profiles/kinds.py); a label that covers several dispatches
    gets their sum."""
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    step = kinds.steady_step(rows, key=lambda r: int(r["Dispatch_Id"]))
    pairs, problems = kinds.match(step, LABELS)
    for p in problems:
        sys.stderr.write("pmc_summary: %s -- skipped\n" % p)
    out = collections.defaultdict(list)
    for label, disp in pairs:
        if label is not None:
            out[label].append(sum(float(r["Counter_Value"]) for r in disp) * 1024.0)
    return {k: sum(v) / len(v) for k, v in out.items()}


fetch = per_label(sys.argv[1], "FETCH_SIZE")
write = per_label(sys.argv[2], "WRITE_SIZE")
res = {}
for label in sorted(set(list(fetch) + list(write))):
    f = fetch.get(label, 0.0)
    w = write.get(label, 0.0)
    res[label] = {"fetch_size_raw_bytes": round(f), "read_bytes_corrected_x2": round(2 * f), "write_bytes": round(w),
                  "traffic_bytes": round(2 * f + w)}
json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, bench.py --steps 3 --warmup 2 --frames 3",
           "per_launch": res}, sys.stdout, indent=1)
