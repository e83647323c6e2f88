"""Dev: headline fields of the JSON line in a bench.py log:  python tools/bench_line.py gpurun_out/b.log"""
import json,sys
l=[x for x in open(sys.argv[1]) if x.startswith("{")][-1]
d=json.loads(l); print(sys.argv[1], d["value"], d["ms_per_step"], d.get("host_enqueue_ms_per_step"), d.get("ms_per_step_second_half"))
