#!/bin/bash
# long random-configuration hunts on the GPU (the fuzz tests of tests/ with other seeds and counts); prints one line per run
# usage: bash tools/hunt.sh <first seed> <last seed>
out=gpurun_out/hunt; mkdir -p $out
for seed in $(seq ${1:-21} ${2:-26}); do
  for spec in "tests/test_gpu_parity.py random_configurations_on_the_device 3000" "tests/test_gpu_round2.py class_level_random 3000" "tests/test_gpu_round2.py fused_rollouts_equal 2000"; do
    set -- $spec
    r=$(GAQ_FUZZ_SEED=$seed GAQ_FUZZ_CONFIGS=$3 timeout -k 10 500 python -m pytest $1 -m gpu -q -x -k "$2" 2>&1 | grep -E "passed|failed" | tail -1)
    echo "seed $seed $2 x$3: $r" | tee -a $out/hunt.txt
  done
done
