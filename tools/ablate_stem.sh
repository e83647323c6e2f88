#!/bin/bash
# Dev: diagnostic variants of libsmos_hip.so with one ingredient of stem_gemm removed (-DSMOS_STEM_ABLATE bits: 1 row loads,
# 2 MFMAs, 4 Y stores; results are wrong, only the timing means something) into streammos_amd/lib/ablate/.  Run here, then on the
# GPU box:  for k in 1 2 4; do SMOS_HIP_LIB=$PWD/streammos_amd/lib/ablate/libsmos_stem_$k.so python tools/ubench_stem.py; done
set -e
cd "$(dirname "$0")/.."
out=streammos_amd/lib/ablate
mkdir -p $out
objs=$(ls streammos_amd/lib/*.o | grep -v "/stem.o")
for k in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=on -Iinclude -Istreammos_amd/csrc \
      -DSMOS_STEM_ABLATE=$k -c streammos_amd/csrc/stem.hip -o $out/stem_$k.o &
done
wait
for k in "$@"; do
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $out/libsmos_stem_$k.so $objs $out/stem_$k.o
  rm $out/stem_$k.o
done
ls $out
