#!/bin/bash
# Dev: diagnostic variants of libsmos_hip.so with one ingredient of point_head removed (-DSMOS_HEAD_ABLATE bits: 1 row loads, 2 layer-1 MFMAs, 4 layers 2 and 3, 8 logit stores;
# results are wrong, only the timing means something) into streammos_amd/lib/ablate/.  Run here, then on the
# GPU box:  for k in 1 2 4; do SMOS_HIP_LIB=$PWD/streammos_amd/lib/ablate/libsmos_head_$k.so python tools/ubench_head.py; done
set -e
cd "$(dirname "$0")/.."
out=streammos_amd/lib/ablate
mkdir -p $out
objs=$(ls streammos_amd/lib/*.o | grep -v "/point_head.o")
for k in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=on -Iinclude -Istreammos_amd/csrc \
      -DSMOS_HEAD_ABLATE=$k -c streammos_amd/csrc/point_head.hip -o $out/head_$k.o &
done
wait
for k in "$@"; do
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $out/libsmos_head_$k.so $objs $out/head_$k.o
  rm $out/head_$k.o
done
ls $out
