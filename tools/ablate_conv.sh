#!/bin/bash
# Dev: builds diagnostic variants of libsmos_hip.so with one ingredient of the conv_igemm stage removed at a time
# (-DSMOS_CONV_ABLATE bits: 1 barrier, 2 activation requests, 4 weight ring traffic, 8 epilogue stores; results are wrong,
# only the timing means something) into streammos_amd/lib/ablate/.  Run here (hipcc cross-compiles), then on the GPU box:
#   for k in 0 1 2 4 5 7 15; do SMOS_HIP_LIB=streammos_amd/lib/ablate/libsmos_hip_$k.so SMOS_UBENCH_NO_LIB=1 \
#       python tools/ubench_conv.py > gpurun_out/ablate_$k.log; done
set -e
cd "$(dirname "$0")/.."
out=streammos_amd/lib/ablate
mkdir -p $out
objs=$(ls streammos_amd/lib/*.o | grep -v conv_igemm.o)
for k in ${@:-0 1 2 4 5 7 15}; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=on -Iinclude -Istreammos_amd/csrc \
      -DSMOS_CONV_ABLATE=${k%%s*} -DSMOS_CONV_SCHED=$([[ $k == *s* ]] && echo ${k##*s} || echo 0) -c streammos_amd/csrc/conv_igemm.hip -o $out/conv_igemm_$k.o &
done
wait
for k in ${@:-0 1 2 4 5 7 15}; do
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $out/libsmos_hip_$k.so $objs $out/conv_igemm_$k.o
  rm $out/conv_igemm_$k.o
done
ls -la $out
