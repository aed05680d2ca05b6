#!/bin/bash
# A/B of two builds on ONE box over cases of tools/variant_rates.py (N = 2^20, device tensors): bash tools/ab_cases.sh <out-name> <variant .so> "<case substring>" ...
out=gpurun_out/${1:?name}; lib=${2:?variant library}; shift 2; mkdir -p $out
for rep in 1 2; do
  for c in "$@"; do
    for which in in-tree variant; do
      L=""; [ $which = variant ] && L=$PWD/$lib
      GAQ_LIB=$L timeout -k 10 300 python tools/variant_rates.py "$c" 600 2>>$out/err.log | python -c "
import json,sys
for k,v in json.load(sys.stdin).items(): print('%-8s %7.2f us  v%-5d %s' % ('$which', v['us_per_step'], v['kernel_variant'], k))" | tee -a $out/ab.txt || exit 1
    done
  done
done
