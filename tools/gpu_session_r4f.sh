#!/bin/bash
# round 4, session F (timing only): would a third wave per SIMD help <1046>?  The variant is a what-if build (u / w parked in LDS, lag / OU state stored
# right after the sub-step loop, scalar model reload, amdgpu_waves_per_eu(3): 167 VGPRs, 4 spilled) whose in-kernel resets are NOT correct
set -o pipefail
R=$PWD
O=$R/gpurun_out/${1:-r4f}
mkdir -p $O
bash tools/ab_cases.sh $(basename $O)/ab_probe3w build/variants/libgaq_probe3w.so "Crazyflie + sense_noise" || exit 1
exit 0
