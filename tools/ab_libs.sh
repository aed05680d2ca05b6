#!/bin/bash
# A/B of kernel builds on ONE box: GAQ_LIB=<variant> python bench.py ... for each library under build/aux/ (and the in-tree one)
out=gpurun_out/${1:-ab}; mkdir -p $out
for rep in 1 2; do
for lib in "" build/aux/libgaq_head.so; do
  for cfg in "--envs 65536 --steps 1000 --model Crazyflie --randomize" "--model Crazyflie --randomize --steps 600 --warmup 600"; do
    GAQ_LIB=${lib:+$PWD/$lib} timeout -k 10 300 python bench.py --no-cpu-baseline $cfg 2>>$out/err.log | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('%-34s %-58s %8.2f us frac %.3f' % ('${lib:-in-tree}', '$cfg', d['ms_per_step']*1e3, d['roofline']['frac']))" | tee -a $out/ab.txt || exit 1
  done
done
done
for cfg in "--randomize-every 1" "--randomize-every 4" "--randomize-every 1 --model RandomQuad"; do
  m="--model Crazyflie --randomize"; case "$cfg" in *RandomQuad*) m="";; esac
  timeout -k 10 300 python bench.py --no-cpu-baseline $m --steps 600 --warmup 600 --stagger $cfg 2>>$out/err.log | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('staggered %-40s %8.2f us' % ('$cfg', d['ms_per_step']*1e3))" | tee -a $out/ab.txt || exit 1
done
