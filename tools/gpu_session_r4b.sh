#!/bin/bash
# round 4, session B: the whole -m gpu suite on the new sources (fast Box-Muller, F_AUXP kernels, gaq_get_params as a read, the sharded
# handle / device_ids env), the variant rates, the F_AUXP A/B (same library, GAQ_NO_AUXP=1 = round 3's kernel choice) and PMC of the
# two VALU-bound cases
set -o pipefail
R=$PWD
O=$R/gpurun_out/${1:-r4b}
mkdir -p $O
rm -f $O/coverage.json
KERNEL_COVERAGE_OUT=$O/coverage.json timeout -k 10 900 python -m pytest tests -m gpu -q > $O/gputest.log 2>&1; echo "pytest rc=$?" >> $O/gputest.log; tail -15 $O/gputest.log
python tools/kernel_coverage.py $O/coverage.json > $O/kernel_coverage.txt 2>&1; head -4 $O/kernel_coverage.txt
grep -q "rc=0" $O/gputest.log || exit 1
python tools/variant_rates.py > $O/variant_rates.json 2> $O/variant_rates.err || { tail -5 $O/variant_rates.err; exit 1; }
python - <<PY
import json
for k, v in json.load(open("$O/variant_rates.json")).items(): print("%7.2f us  v%-5d %s" % (v["us_per_step"], v["kernel_variant"], k))
PY
echo "--- F_AUXP A/B (GAQ_NO_AUXP=1: the light generic kernel of round 3)"
for rep in 1 2; do
  for c in "info=True" "xyz_vxyz_quat_omega" "t2w_t2t"; do
    for no in 0 1; do
      GAQ_NO_AUXP=$no timeout -k 10 300 python tools/variant_rates.py "$c" 600 2>>$O/ab_auxp.err | python -c "
import json,sys
for k,v in json.load(sys.stdin).items(): print('no_auxp=$no %7.2f us  v%-5d %s' % (v['us_per_step'], v['kernel_variant'], k))" | tee -a $O/ab_auxp.txt || exit 1
    done
  done
done
echo "--- observation rows of the F_PACK kernels packed straight into LDS (variant) vs held in registers (in-tree)"
bash tools/ab_cases.sh $(basename $O)/ab_rowslds build/variants/libgaq_rowslds.so "sense_noise=default (split" "Crazyflie + sense_noise" "info=True" \
  "obs xyz_vxyz_R_omega_acc_act" "Mellinger controller, obs xyz_vxyz_R_omega_h" || exit 1
echo "--- cache policy of the caller's observation copy (library-owned heads layout): nt / sc1 against the default"
bash tools/ab_cases.sh $(basename $O)/ab_copynt build/variants/libgaq_copynt.so "default configuration, class default" "re-randomised every episode, class default" "Mellinger controller, class default" || exit 1
bash tools/ab_cases.sh $(basename $O)/ab_copysc1 build/variants/libgaq_copysc1.so "default configuration, class default" "re-randomised every episode, class default" || exit 1
bash tools/pmc_case.sh $(basename $O)/pmc_cf_sense "Crazyflie + sense_noise" 352 || exit 1
bash tools/pmc_case.sh $(basename $O)/pmc_info "info=True" 352 || exit 1
exit 0
