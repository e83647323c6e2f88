set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r2c
for n in 2 3 4; do echo "== blocks/CU $n"; SMOS_CONV_BLOCKS_PER_CU=$n python $R/tools/ubench_conv.py "3x3" ; done > $R/gpurun_out/r2c/percu.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/r2c/pmc1 -- python $R/tools/ubench_conv.py "hdr_bev 3x3" > $R/gpurun_out/r2c/pmc1.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/r2c/pmc2 -- python $R/tools/ubench_conv.py "conv_2" > $R/gpurun_out/r2c/pmc2.log 2>&1
find $R/gpurun_out/r2c -name "*counter_collection.csv" | head
