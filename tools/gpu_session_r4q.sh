#!/bin/bash
# round 4, session Q: PMC traffic / instruction counts of the round's last kernels (per-env goals, bias walk, Mellinger with the aux row /
# chasing goals / on per-env models) and of the sensor-noise and aux-row kernels beside them
set -o pipefail
O=${1:-r4q}
bash tools/pmc_case.sh $O/envx_goal "resample_goal=True (per-env goals; class" || exit 1
bash tools/pmc_case.sh $O/envx_bias "gyro-bias random walk (class default" || exit 1
bash tools/pmc_case.sh $O/mell_auxp "info=True with the Mellinger controller (class" || exit 1
bash tools/pmc_case.sh $O/mell_envx "excite=True with the Mellinger controller (what" || exit 1
bash tools/pmc_case.sh $O/mell_per_env "Mellinger controller with per-env randomized Crazyflie (the" 480 || exit 1
bash tools/pmc_case.sh $O/sense "sense_noise=default (split" || exit 1
bash tools/pmc_case.sh $O/auxp_info "info=True: aux row" || exit 1
python3 - <<PY
import json
for k in ("envx_goal", "envx_bias", "mell_auxp", "mell_envx", "mell_per_env", "sense", "auxp_info"):
    d = json.load(open("gpurun_out/$O/%s/summary.json" % k))["_derived"]
    print("%-13s %s: %.1f B/env-step (fetch %.1f + write %.1f), %s VALU/wave, wave cycles %s" % (k, d["kernel"][:40], d["traffic_bytes_per_env_step"], d["fetch_bytes_per_env_step"], d["write_bytes_per_env_step"], d.get("valu_insts_per_wave"), d.get("wave_cycles_per_wave(quad-cycles x4)")))
PY
