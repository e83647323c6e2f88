#!/bin/bash
# Dev: samples package power and shader clock while the bench's timed region runs (is the step power-limited?).
# usage (GPU box): bash tools/power_probe.sh
cd $GRAFT_REPO_ROOT
python3 bench.py --steps 2000 --warmup 5 --train-steps 0 --cpu-scans 0 --no-raw > gpurun_out/power_probe_bench.log 2>&1 &
BP=$!
sleep 25
for i in 1 2 3 4 5 6; do
  rocm-smi --showpower --showclocks --showmaxpower --showtemp 2>&1 | grep -i -E "power|sclk|mclk|fclk|Temperature \(Sensor (junction|edge)" | head -12
  echo "--"
  sleep 1
done
wait $BP
grep "^{" gpurun_out/power_probe_bench.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
