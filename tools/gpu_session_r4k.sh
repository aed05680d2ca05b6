#!/bin/bash
# round 4, session K: Mellinger with per-env models in the specialised kernels (F_MELL | per-env): the whole GPU suite, then the rates beside
# the full generic kernel they replace (GAQ_FORCE_GENERIC=1)
set -o pipefail
R=$PWD
O=$R/gpurun_out/${1:-r4k}
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/gputest.log 2>&1; echo "pytest rc=$?" | tee -a $O/gputest.log
tail -15 $O/gputest.log
for g in 0 1; do
for c in "Mellinger controller with per-env"; do
  GAQ_FORCE_GENERIC=$g timeout -k 10 600 python tools/variant_rates.py "$c" 300 2>>$O/err.log | python -c "
import json,sys
for k,v in json.load(sys.stdin).items(): print('force_generic=$g %7.2f us  v%-6d %s' % (v['us_per_step'], v['kernel_variant'], k))" | tee -a $O/rates.txt || exit 1
done
done
exit 0
