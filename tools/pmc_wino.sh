# Dev: where a conv_wino wave's cycles go (SQ counters) on conv_2 (128 -> 64 @256^2, mb 2): one rocprofv3 --pmc pass per counter
# group, kernel trace only beside it.  gpurun -- 'bash tools/pmc_wino.sh'
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_wino
mkdir -p $O
PASSES=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"
        "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS"
        "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" "SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVES SQ_INST_LEVEL_LDS")
i=0
for g in "${PASSES[@]}"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --pmc $g --kernel-trace --output-format csv -d $O/g$i -- python3 $R/tools/ubench_wino.py conv_2 > $O/g$i.log 2>&1 || echo "pass g$i failed"
  echo "g$i done"
done
python3 - <<'PY'
import csv, glob, os, collections
root = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_wino"
out = open(root + "/summary.txt", "w")
for d in sorted(glob.glob(root + "/g*")):
    if not os.path.isdir(d): continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "conv_wino" not in k: continue
            k = k.split("(")[0].replace("void smos::", "")
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
    for k in acc:
        line = "%s %s %s" % (os.path.basename(d), k, " ".join("%s=%.5g" % (c, v / n[(k, c)]) for c, v in sorted(acc[k].items())))
        print(line); out.write(line + "\n")
PY
