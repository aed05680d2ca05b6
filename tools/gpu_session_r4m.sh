#!/bin/bash
# round 4, session M: occupancy floors (three waves per SIMD asked of the allocator) for kernels a few VGPRs above the 168 line:
# in-tree = with the floors, variant = -DGAQ_NO_FLOORS
set -o pipefail
O=gpurun_out/${1:-r4m}; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_round3.py -q -p no:cacheprovider -k "handle_reports" 2>&1 | tail -3
bash tools/ab_cases.sh $(basename $O)/ab_floors build/variants/libgaq_nofloors.so "Crazyflie + sense_noise=default, thrust noise off" "resample_goal=True, Crazyflie, thrust noise off" "excite=True with the Mellinger controller, Crazyflie, thrust noise off" || exit 1
bash tools/ab_lib.sh $(basename $O)/ab_floors_swarm build/variants/libgaq_nofloors.so "--no-layouts --swarm 8 --steps 300 --warmup 100" || exit 1
exit 0
