#!/bin/bash
# Round profile: kernel-trace stats of the default bench, the two PMC passes for the HBM traffic of the labelled kernels,
# and the MFMA-busy pass.  usage (on the GPU box): bash tools/profile_round.sh <tag>
# (--pmc passes carry --kernel-trace only: the pool refuses counter collection combined with other trace domains)
set -e
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t --output-format csv -- python3 -u $R/bench.py --steps 20 --warmup 6 --cpu-scans 0 --train-steps 0 --no-raw > $OUT/bench_trace.log 2>&1
echo "trace done"
# the same with the two pipeline streams folded into one: per-kernel durations without the other stream's kernels on the CUs
# (whether the tracer serialises the two streams by itself differs from box to box)
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $OUT/trace_serial -o t --output-format csv -- python3 -u $R/bench.py --steps 20 --warmup 6 --cpu-scans 0 --train-steps 0 --no-raw --no-pipeline --label-log $OUT/labels_serial.json > $OUT/bench_trace_serial.log 2>&1
echo "serial trace done"
PMCARGS="--steps 3 --warmup 2 --frames 3 --cpu-scans 0 --train-steps 0 --no-raw --no-pipeline"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o f -- python3 -u $R/bench.py $PMCARGS --label-log $OUT/labels.json > $OUT/bench_fetch.log 2>&1
echo "fetch done"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o w -- python3 -u $R/bench.py $PMCARGS > $OUT/bench_write.log 2>&1
echo "write done"
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/mfma -o m -- python3 -u $R/bench.py $PMCARGS > $OUT/bench_mfma.log 2>&1
echo "mfma done"
F=$(find $OUT/fetch -name "*counter_collection.csv" | head -1); W=$(find $OUT/write -name "*counter_collection.csv" | head -1)
python3 $R/profiles/pmc_summary.py $F $W $OUT/labels.json > $OUT/pmc_traffic.json
M=$(find $OUT/mfma -name "*counter_collection.csv" | head -1)
python3 $R/profiles/mfma_busy_summary.py $M $OUT/labels.json > $OUT/mfma_busy.txt
S=$(find $OUT/trace -name "*kernel_stats.csv" | head -1); T=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
cp $S $OUT/kernel_stats.csv
python3 $R/profiles/step_breakdown.py $T > $OUT/step_breakdown.txt 2>&1 || true
S2=$(find $OUT/trace_serial -name "*kernel_stats.csv" | head -1); T2=$(find $OUT/trace_serial -name "*kernel_trace.csv" | head -1)
cp $S2 $OUT/kernel_stats_serial.csv
python3 $R/profiles/step_breakdown.py $T2 > $OUT/step_breakdown_serial.txt 2>&1 || true
python3 $R/profiles/step_timeline.py $T2 > $OUT/step_timeline_serial.csv 2>&1 || true
python3 $R/profiles/step_timeline.py $T > $OUT/step_timeline.csv 2>&1 || true
# per-LABEL durations of the serial trace (launch order = dispatch order), with the roofline fraction of every labelled launch
grep "^{" $OUT/bench_trace_serial.log | tail -1 > $OUT/bench_traced_serial.json.log || true
python3 $R/profiles/label_durations.py $T2 $OUT/labels_serial.json $OUT/bench_traced_serial.json.log > $OUT/label_durations.csv 2> $OUT/label_durations.err || true
grep "^{" $OUT/bench_trace.log | tail -1 > $OUT/bench_traced.json.log || true
rm -rf $OUT/trace $OUT/trace_serial $OUT/fetch $OUT/write $OUT/mfma
ls -la $OUT
