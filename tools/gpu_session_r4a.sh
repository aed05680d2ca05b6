#!/bin/bash
# round 4, session A: baseline rates of every variant case on this box, A/B of the fast Box-Muller build, and a trial of rocprofv3's
# PC sampling on the VALU-bound kernels (where do the waves of <1046> / <1044> spend their time?)
set -o pipefail
R=$PWD
O=$R/gpurun_out/${1:-r4a}
mkdir -p $O
python tools/variant_rates.py > $O/variant_rates.json 2> $O/variant_rates.err || { tail -5 $O/variant_rates.err; exit 1; }
python - <<EOF
import json
for k, v in json.load(open("$O/variant_rates.json")).items(): print("%7.2f us  v%-5d %s" % (v["us_per_step"], v["kernel_variant"], k))
EOF
echo "--- fast Box-Muller A/B"
bash tools/ab_cases.sh $(basename $O)/ab_fastbm build/variants/libgaq_fastbm.so "default configuration, alias_obs=True" "default configuration, class default" \
  "sense_noise=default (split" "Crazyflie + sense_noise" "Crazyflie uniform" "info=True" || exit 1
bash tools/ab_lib.sh $(basename $O)/ab_fastbm_small build/variants/libgaq_fastbm.so "--envs 65536 --steps 1000 --no-layouts" "--envs 131072 --steps 1000 --no-layouts" || exit 1
echo "--- PC sampling trial"
cd /tmp && export TMPDIR=/tmp
export ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1
for c in "Crazyflie + sense_noise" "sense_noise=default (split"; do
  key=$(echo "$c" | tr -c 'A-Za-z0-9' '_' | cut -c1-24)
  timeout -k 10 150 rocprofv3 --kernel-trace --pc-sampling-beta-enabled --pc-sampling-unit time --pc-sampling-method host_trap --pc-sampling-interval 1 \
    --output-format csv -d $O/pcs_$key -- python3 $R/tools/variant_rates.py "$c" 300 > $O/pcs_$key.log 2>&1
  rc=$?; echo "pc sampling '$c' rc=$rc"; tail -3 $O/pcs_$key.log
  [ $rc -ne 0 ] && break
done
cd $R
python3 tools/pcs_summary.py $O/pcs_* > $O/pcs_summary.txt 2>&1; head -60 $O/pcs_summary.txt
# keep the merged-back payload small: the raw sample files can be hundreds of MB
find $O -name "*pc_sampling*.csv" -size +20M -delete
exit 0
