#!/bin/bash
# A/B/C... of several builds on ONE box, interleaved: bash tools/ab_multi.sh <out-name> "<bench args>" lib1.so lib2.so ...   ("-" = the in-tree library)
out=gpurun_out/${1:?name}; cfg=${2?bench args}; shift 2; mkdir -p $out
for rep in 1 2 3; do
  for lib in "$@"; do
    L=""; [ "$lib" != "-" ] && L=$PWD/$lib
    GAQ_LIB=$L timeout -k 10 300 python bench.py --no-cpu-baseline --no-layouts $cfg 2>>$out/err.log | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('%-40s %-50s %8.2f us  kernel %8.2f us' % ('$lib', '$cfg', d['ms_per_step']*1e3, d['roofline']['kernel_ms']*1e3))" | tee -a $out/ab.txt || exit 1
  done
done
