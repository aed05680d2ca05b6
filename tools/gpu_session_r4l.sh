#!/bin/bash
# round 4, session L: the aux row / observation variants / per-env goals under the Mellinger controller on the split state (F_MELL | F_AUXP
# [| F_ENVX]): the whole GPU suite, the rates beside the generic kernels they replace, and the two occupancy floors (variant: without)
set -o pipefail
R=$PWD
O=$R/gpurun_out/${1:-r4l}
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/gputest.log 2>&1; echo "pytest rc=$?" | tee -a $O/gputest.log
tail -15 $O/gputest.log
for c in "with the Mellinger controller"; do
  timeout -k 10 600 python tools/variant_rates.py "$c" 600 2>>$O/err.log | python -c "
import json,sys
for k,v in json.load(sys.stdin).items(): print('%7.2f us  v%-6d %s' % (v['us_per_step'], v['kernel_variant'], k))" | tee -a $O/rates.txt || exit 1
done
echo "--- occupancy floors of <214036> / <82962> (169 -> 168 VGPRs, 4 spilled): in-tree = with, variant = without" | tee -a $O/rates.txt
bash tools/ab_cases.sh $(basename $O)/ab_floors build/variants/libgaq_nofloors.so "excite=True with the Mellinger controller (what" "info=True with the Mellinger controller, Crazyflie" || exit 1
cat $O/ab_floors/ab.txt >> $O/rates.txt
exit 0
