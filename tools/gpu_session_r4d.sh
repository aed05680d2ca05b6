#!/bin/bash
# round 4, session D: the -m gpu suite with the round-toward-zero head conversion; A/Bs: RTZ heads off (variant); the uniform model read
# from memory at the point of use -- from the kernel-argument segment by scalar loads in every split-state kernel
# (mk2_all), from a per-wave LDS copy in the lag + packed-observation kernels (mlds_packlag) -- with PMC of <1046> for both
set -o pipefail
R=$PWD
O=$R/gpurun_out/${1:-r4d}
mkdir -p $O
rm -f $O/coverage.json
KERNEL_COVERAGE_OUT=$O/coverage.json timeout -k 10 900 python -m pytest tests -m gpu -q > $O/gputest.log 2>&1; echo "pytest rc=$?" >> $O/gputest.log; tail -6 $O/gputest.log
echo "--- heads by round-toward-zero conversion (in-tree) vs round-to-nearest + fix-up (variant)"
bash tools/ab_cases.sh $(basename $O)/ab_rtz build/variants/libgaq_nortz.so "default configuration, alias_obs=True" "default configuration, class default" "sense_noise=default (split" "Crazyflie + sense_noise" "Crazyflie uniform" || exit 1
bash tools/ab_lib.sh $(basename $O)/ab_rtz_small build/variants/libgaq_nortz.so "--envs 65536 --steps 1000 --no-layouts" "--envs 131072 --steps 1000 --no-layouts" || exit 1
echo "--- uniform model by scalar loads from the kernel-argument segment at the point of use, every split-state kernel (variant)"
bash tools/ab_cases.sh $(basename $O)/ab_mk2 build/variants/libgaq_mk2_all.so "default configuration, alias_obs=True" "default configuration, class default" "sense_noise=default (split" "Crazyflie + sense_noise" "Crazyflie uniform" \
   "Mellinger controller, class default" "Mellinger controller, Crazyflie" "info=True" "xyz_vxyz_quat_omega" "obs xyz_vxyz_R_omega_acc_act" || exit 1
bash tools/ab_lib.sh $(basename $O)/ab_mk2_small build/variants/libgaq_mk2_all.so "--envs 65536 --steps 1000 --no-layouts" "--envs 131072 --steps 1000 --no-layouts" "--envs 262144 --steps 1000 --no-layouts" || exit 1
echo "--- uniform model from a per-wave LDS copy: lag + packed-observation kernels (variant)"
bash tools/ab_cases.sh $(basename $O)/ab_mlds_packlag build/variants/libgaq_mlds_packlag.so "Crazyflie + sense_noise" || exit 1
bash tools/pmc_case.sh $(basename $O)/pmc_cf_sense_intree "Crazyflie + sense_noise" 352 || exit 1
GAQ_LIB=$R/build/variants/libgaq_mk2_all.so bash tools/pmc_case.sh $(basename $O)/pmc_cf_sense_mk2 "Crazyflie + sense_noise" 352 || exit 1
GAQ_LIB=$R/build/variants/libgaq_mlds_packlag.so bash tools/pmc_case.sh $(basename $O)/pmc_cf_sense_mlds "Crazyflie + sense_noise" 352 || exit 1
grep -q "rc=0" $O/gputest.log || exit 1
exit 0
