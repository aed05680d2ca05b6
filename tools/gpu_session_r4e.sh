#!/bin/bash
# round 4, session E: the -m gpu suite and the variant rates after the aux values went straight into their LDS rows (F_AUXP: 180 -> 158 VGPRs,
# three waves per SIMD) and the two pointer-select stack objects (omega_dot, the noise quaternion) went away
set -o pipefail
R=$PWD
O=$R/gpurun_out/${1:-r4e}
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/gputest.log 2>&1; echo "pytest rc=$?" >> $O/gputest.log; tail -6 $O/gputest.log
python tools/variant_rates.py > $O/variant_rates.json 2> $O/variant_rates.err || { tail -5 $O/variant_rates.err; exit 1; }
python - <<PY
import json
for k, v in json.load(open("$O/variant_rates.json")).items(): print("%7.2f us  v%-5d %s" % (v["us_per_step"], v["kernel_variant"], k))
PY
grep -q "rc=0" $O/gputest.log || exit 1
exit 0
