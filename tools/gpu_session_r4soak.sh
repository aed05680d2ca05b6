#!/bin/bash
# round 4: tools/soak.py on the final kernels (long runs, invariants)
set -o pipefail
O=$PWD/gpurun_out/${1:-r4soak}; mkdir -p $O
timeout -k 10 1000 python tools/soak.py > $O/soak.json 2> $O/soak.err; rc=$?; tail -12 $O/soak.err; exit $rc
