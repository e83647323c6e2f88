#!/bin/bash
# Dev: samples package power and shader clock while the bench's timed region runs (is the step power-limited?).
# usage (GPU box): bash tools/power_probe.sh
cd $GRAFT_REPO_ROOT
python3 bench.py --steps 6000 --warmup 5 --train-steps 0 --cpu-scans 0 --no-raw > gpurun_out/power_probe_bench.log 2>&1 &
BP=$!
for i in $(seq 1 45); do
  sleep 1
  P=$(rocm-smi --showpower 2>/dev/null | grep -i "Package Power" | head -1 | sed 's/.*: //')
  C=$(rocm-smi --showclocks 2>/dev/null | grep -i "sclk" | head -1 | sed 's/.*(\(.*\))/\1/')
  echo "t=$i s  power $P W  sclk $C"
done
wait $BP
grep "^{" gpurun_out/power_probe_bench.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
