#!/bin/bash
# the GPU suite + kernel coverage alone (the first half of tools/gpu_session_r4final.sh a1)
set -o pipefail
O=$PWD/gpurun_out/${1:-r4tests}; mkdir -p $O
rm -f $O/coverage.json
KERNEL_COVERAGE_OUT=$O/coverage.json timeout -k 10 900 python -m pytest tests -m gpu -q > $O/gputest.log 2>&1; echo "pytest rc=$?" >> $O/gputest.log; tail -5 $O/gputest.log
python tools/kernel_coverage.py $O/coverage.json > $O/kernel_coverage.txt 2>&1; head -3 $O/kernel_coverage.txt
grep -q "rc=0" $O/gputest.log
