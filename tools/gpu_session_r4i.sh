#!/bin/bash
# round 4, session I: per-env goals (resample_goal, excite) and the gyro-bias walk on the split state (F_ENVX): the whole GPU suite, then
# the rates of the new kernels beside the generic ones on fp64 planes they replace
set -o pipefail
R=$PWD
O=$R/gpurun_out/${1:-r4i}
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/gputest.log 2>&1; echo "pytest rc=$?" | tee -a $O/gputest.log
tail -15 $O/gputest.log
for c in "resample_goal=True" "excite=True" "gyro-bias random walk" "info=True: aux row" "default configuration, class default"; do
  timeout -k 10 300 python tools/variant_rates.py "$c" 600 2>>$O/err.log | python -c "
import json,sys
for k,v in json.load(sys.stdin).items(): print('%7.2f us  v%-6d %s' % (v['us_per_step'], v['kernel_variant'], k))" | tee -a $O/rates.txt || exit 1
done
exit 0
