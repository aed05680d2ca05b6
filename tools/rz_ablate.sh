#!/bin/bash
# timing-only ablations of the per-episode re-randomisation inside the step kernel (GAQ_ABLATE bits: 1 arithmetic, 2 the promotion's plane
# copy, 4 the whole promotion): results are wrong by construction, only the kernel time is read
# needs the measurement build: make -C gym_art_amd/csrc EXTRA=-DGAQ_DIAG_BUILD OUT=$PWD/build/variants/libgaq_diag.so OBJ=$PWD/build/variants/obj_diag
out=gpurun_out/${1:-rzab}; mkdir -p $out
for rep in 1 2 3; do
for ab in 0 2 4; do
  GAQ_LIB=$PWD/build/variants/libgaq_diag.so GAQ_ABLATE=$ab timeout -k 10 300 python bench.py --no-cpu-baseline --model Crazyflie --randomize --steps 600 --warmup 600 --stagger --randomize-every 1 2>>$out/err.log | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('rz every=1 ablate=$ab %8.2f us kern %8.2f' % (d['ms_per_step']*1e3, d['roofline']['kernel_ms']*1e3))" || exit 1
done
done
