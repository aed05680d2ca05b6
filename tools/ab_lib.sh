#!/bin/bash
# A/B of two builds on ONE box: bash tools/ab_lib.sh <out-name> <variant .so> "<bench args>" ["<bench args>" ...]
out=gpurun_out/${1:?name}; lib=${2:?variant library}; shift 2; mkdir -p $out
for rep in 1 2 3; do
  for cfg in "$@"; do
    for which in in-tree variant; do
      L=""; [ $which = variant ] && L=$PWD/$lib
      GAQ_LIB=$L timeout -k 10 300 python bench.py --no-cpu-baseline $cfg 2>>$out/err.log | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('%-8s %-70s %8.2f us' % ('$which', '$cfg', d['ms_per_step']*1e3))" | tee -a $out/ab.txt || exit 1
    done
  done
done
