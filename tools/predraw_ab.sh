#!/bin/bash
# pre-drawn OU normals (F_PREDRAW) on / off by batch size, one box: bench.py with GAQ_PREDRAW=0|1 (GAQ_NT left to the size rule)
# usage: bash tools/predraw_ab.sh <out-name> [reps] ["cfg1|cfg2|..."]
out=gpurun_out/${1:-predraw}; mkdir -p $out
reps=${2:-2}
IFS='|' read -ra CFGS <<< "${3:---envs 65536 --steps 1000|--envs 131072 --steps 1000|--envs 262144 --steps 1000||--model Crazyflie --steps 600 --warmup 600|--model Crazyflie --randomize --steps 600 --warmup 600|--envs 65536 --steps 1000 --model Crazyflie --randomize}"
for rep in $(seq $reps); do
for cfg in "${CFGS[@]}"; do
  for pd in 0 1; do
    GAQ_PREDRAW=$pd timeout -k 10 300 python bench.py --no-cpu-baseline $cfg 2>>$out/err.log | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('predraw=$pd %-62s %8.2f us frac %.3f' % ('$cfg', d['ms_per_step']*1e3, d['roofline']['frac']))" | tee -a $out/ab.txt || exit 1
  done
done
done
