# Dev: memory-path counters of the own conv on two layers (32-channel 7x3 at mt 1, conv_2 at mt 2); one rocprofv3 --pmc pass
# per counter group (kernel trace only beside it).  gpurun -- 'bash tools/pmc_conv.sh'
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_conv
mkdir -p $O
export SMOS_UBENCH_NO_LIB=1
# at most three counters of one hardware block per pass ("Request exceeds the capabilities of the hardware" otherwise)
PASSES=("TA_TA_BUSY TA_BUFFER_READ_WAVEFRONTS GRBM_GUI_ACTIVE" "TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES"
        "TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ TCP_PENDING_STALL_CYCLES"
        "TCP_TCR_TCP_STALL_CYCLES TCP_READ_TAGCONFLICT_STALL_CYCLES TCP_TCP_TA_DATA_STALL_CYCLES"
        "TCP_TCC_READ_REQ_LATENCY TCP_TCP_LATENCY" "TCC_HIT TCC_MISS TCC_REQ"
        "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_WAVE_CYCLES")
for layer in ${LAYERS:-7x3}; do
  i=0
  for g in "${PASSES[@]}"; do
    i=$((i+1))
    timeout -k 10 150 rocprofv3 --pmc $g --kernel-trace --output-format csv -d $O/${layer}_g$i -- python3 $R/tools/ubench_conv.py $layer > $O/${layer}_g$i.log 2>&1 || echo "pass $layer g$i failed"
    echo "$layer g$i done"
  done
done
python3 - <<'PY'
import csv, glob, os, collections
root = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_conv"
for d in sorted(glob.glob(root + "/*_g*")):
    if not os.path.isdir(d): continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "conv_igemm" not in k: continue
            k = k.split("(")[0].replace("void smos::", "")
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
    for k in acc:
        print(os.path.basename(d), k, " ".join("%s=%.4g" % (c, v / n[(k, c)]) for c, v in sorted(acc[k].items())))
PY
