#!/bin/bash
# long random-configuration hunts on the GPU (the fuzz tests of tests/ with other seeds and counts); one line per run in
# hunt.txt, the FULL pytest log of every run beside it (seed, failing configuration, any abort / fault message), and the loop
# STOPS at the first run that fails, times out or dies: a GPU fault is diagnosed from that log, not ridden over.
# usage: bash tools/hunt.sh <first seed> <last seed>
out=gpurun_out/hunt; mkdir -p $out
for seed in $(seq ${1:-21} ${2:-26}); do
  # (the last two: every kernel instantiation against the generic kernel and every env operation on every kind of handle, flown from
  #  this seed -- their configuration count is fixed, the third field is not used)
  for spec in "tests/test_gpu_parity.py random_configurations_on_the_device 3000" "tests/test_gpu_round2.py class_level_random 3000" "tests/test_gpu_round2.py fused_rollouts_equal 2000" \
              "tests/test_gpu_kernel_coverage.py instantiation 0" "tests/test_gpu_api_matrix.py kind_of_handle 0"; do
    set -- $spec
    log=$out/seed${seed}_$2.log
    GAQ_FUZZ_SEED=$seed GAQ_FUZZ_CONFIGS=$3 timeout -k 10 500 python -m pytest $1 -m gpu -q -x -k "$2" > $log 2>&1
    rc=$?
    r=$(grep -E "passed|failed|error" $log | tail -1)
    echo "seed $seed $2 x$3: rc=$rc $r" | tee -a $out/hunt.txt
    if [ $rc -ne 0 ]; then
      echo "STOP: rc=$rc (124 = timeout) -- full log in $log" | tee -a $out/hunt.txt
      tail -40 $log
      exit $rc
    fi
  done
done
