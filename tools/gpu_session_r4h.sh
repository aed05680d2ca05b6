#!/bin/bash
# round 4, session H: the scalar model reload (quad_core.hpp kModelMem) in the uniform-model GENERIC kernels too? (variant: -DGAQ_MODEL_MEM_GENERIC=1;
# static: <520> 4513 -> 198 spill-lane instructions, <584> 1171 -> 122, <8> 1177 -> 184, <72> 787 -> 122)
set -o pipefail
R=$PWD
O=$R/gpurun_out/${1:-r4h}
mkdir -p $O
bash tools/ab_cases.sh $(basename $O)/ab_mmgen build/variants/libgaq_mmgen.so "resample_goal=True" "gyro-bias random walk" "info=True on fp64 planes" "info=True with the Mellinger" "Mellinger controller, alias_obs=False" || exit 1
exit 0
