#!/bin/bash
# Dev: diagnostic variants of libsmos_hip.so with one ingredient of the conv_wino k-step removed at a time
# (-DSMOS_WINO_ABLATE bits: 1 region requests, 2 region stores, 4 weight DMA, 8 barrier, 16 patch reads + transform,
# 32 A-operand reads, 64 output stores, 128 explicit vmcnt wait; results are wrong, only the timing means something) into
# streammos_amd/lib/ablate/.  Run here (hipcc cross-compiles), then on the GPU box:
#   for k in 0 1 ...; do SMOS_HIP_LIB=streammos_amd/lib/ablate/libsmos_wino_$k.so python tools/ubench_wino.py conv_2; done
set -e
cd "$(dirname "$0")/.."
out=streammos_amd/lib/ablate
mkdir -p $out
objs=$(ls streammos_amd/lib/*.o | grep -v conv_wino.o)
for k in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=on -Iinclude -Istreammos_amd/csrc \
      -DSMOS_WINO_ABLATE=$k -c streammos_amd/csrc/conv_wino.hip -o $out/conv_wino_$k.o &
done
wait
for k in "$@"; do
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $out/libsmos_wino_$k.so $objs $out/conv_wino_$k.o
  rm $out/conv_wino_$k.o
done
ls $out
