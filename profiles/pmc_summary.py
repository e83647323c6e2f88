"""HBM traffic per launch of the hand-written point kernels from two rocprofv3 --pmc passes.

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d <fetch_dir> -- python3 bench.py --steps 3 --warmup 2 --frames 3 --cpu-scans 0 --no-pipeline
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d <write_dir> -- python3 bench.py ... (same)
    python profiles/pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> > profiles/pmc_traffic.json

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

GS_ORDER = ["gather_scatter_cl[4x32x256x256->160000->32x1024]", "gather_scatter_cl[4x32x32x1024->160000->256x256]",
            "gather_scatter_cl[4x64x128x128->160000->16x512]", "gather_scatter_cl[4x64x16x512->160000->128x128]",
            "gather_scatter_cl[4x64x256x256->160000->0x0]"]


def per_label(path, counter):
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    out = collections.defaultdict(list)
    gs = 0
    for r in rows:
        k, v = r["Kernel_Name"], float(r["Counter_Value"]) * 1024.0
        if "pointnet_scatter" in k:
            out["pointnet_scatter[4x3x160000->512x512]:kernel"].append(v)
        elif "FillFunctor" in k and r["Grid_Size"] == "50331648":
            out["pointnet_scatter[4x3x160000->512x512]:zero_fill"].append(v)
        elif "stem_zero_rows" in k:                       # the zero fill of the compact row table (inside the span)
            out["pointnet_scatter[4x3x160000->512x512]:zero_fill"].append(v)
        elif "point_head" in k:
            out["point_head[4x160000]:kernel"].append(v)
        elif "stem_gemm" in k:
            out["stem_gemm[4x512x512x192]:kernel"].append(v)
        elif "stem_epilogue" in k:
            out["stem_epilogue[4x512x512x192]:kernel"].append(v)
        elif "upconv_ypass" in k:
            out["upconv_ypass[4x256x256x128]:kernel"].append(v)
        elif "gather_scatter_cl" in k:
            out[GS_ORDER[gs % 5]].append(v)
            gs += 1
    return {k: sum(v) / len(v) for k, v in out.items()}


fetch = per_label(sys.argv[1], "FETCH_SIZE")
write = per_label(sys.argv[2], "WRITE_SIZE")
res = {}
for label in sorted(set(k.split(":")[0] for k in list(fetch) + list(write))):
    f = sum(v for k, v in fetch.items() if k.split(":")[0] == label)
    w = sum(v for k, v in write.items() if k.split(":")[0] == label)
    res[label] = {"fetch_size_raw_bytes": round(f), "read_bytes_corrected_x2": round(2 * f), "write_bytes": round(w),
                  "traffic_bytes": round(2 * f + w)}
json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, bench.py --steps 3 --warmup 2 --frames 3",
           "per_launch": res}, sys.stdout, indent=1)
