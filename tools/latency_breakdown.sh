#!/bin/bash
# builds the measurement library (-DGAQ_DIAG_BUILD: the only kind that honours GAQ_ABLATE) next to the product one and runs
# tools/latency_breakdown.py with it: bash tools/latency_breakdown.sh > profiles/rNN_latency_breakdown.json
set -e
R=$(cd $(dirname $0)/.. && pwd)
make -s -j8 -C $R/gym_art_amd/csrc ARCH=gfx950 EXTRA=-DGAQ_DIAG_BUILD OUT=$R/build/variants/libgaq_diag.so OBJ=$R/build/variants/obj_diag 1>&2
GAQ_LIB=$R/build/variants/libgaq_diag.so python3 $R/tools/latency_breakdown.py
