#!/bin/bash
# round 4, session C: does v_cvt_f32_f64 obey MODE.FP_ROUND (tools/rtz_check), the -m gpu suite on the new kernels (aux rows / observation
# rows of the F_AUXP kernels through LDS, non-temporal observation copy, hardware sqrt in the reward norms, fp32 t2w / t2t map), variant
# rates, A/B of the reward's hardware sqrt
set -o pipefail
R=$PWD
O=$R/gpurun_out/${1:-r4c}
mkdir -p $O
tools/rtz_check > $O/rtz_check.json 2> $O/rtz_check.err; echo "rtz_check rc=$?"; cat $O/rtz_check.json
rm -f $O/coverage.json
KERNEL_COVERAGE_OUT=$O/coverage.json timeout -k 10 900 python -m pytest tests -m gpu -q > $O/gputest.log 2>&1; echo "pytest rc=$?" >> $O/gputest.log; tail -6 $O/gputest.log
python tools/kernel_coverage.py $O/coverage.json > $O/kernel_coverage.txt 2>&1; head -3 $O/kernel_coverage.txt
python tools/variant_rates.py > $O/variant_rates.json 2> $O/variant_rates.err || { tail -5 $O/variant_rates.err; exit 1; }
python - <<PY
import json
for k, v in json.load(open("$O/variant_rates.json")).items(): print("%7.2f us  v%-5d %s" % (v["us_per_step"], v["kernel_variant"], k))
PY
echo "--- reward norms: v_sqrt_f32 (in-tree) vs sqrtf (variant)"
bash tools/ab_cases.sh $(basename $O)/ab_sqrt build/variants/libgaq_ieeesqrt.so "default configuration, alias_obs=True" "sense_noise=default (split" "Crazyflie + sense_noise" || exit 1
bash tools/ab_lib.sh $(basename $O)/ab_sqrt_small build/variants/libgaq_ieeesqrt.so "--envs 65536 --steps 1000 --no-layouts" || exit 1
grep -q "rc=0" $O/gputest.log || exit 1
exit 0
