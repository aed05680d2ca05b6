#!/bin/bash
# PMC passes (separate runs: SQ group, FETCH_SIZE, WRITE_SIZE) of ONE case of tools/variant_rates.py: bash tools/pmc_case.sh <out-name> "<case substring>" [alg bytes]
# -> gpurun_out/<out-name>/summary.json (tools/pmc_summary.py: per-launch averages of the step kernel, bytes per env-step)
R=$PWD; O=$R/gpurun_out/${1:?name}; CASE=${2:?case substring}; ALG=${3:-352}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/$c -- python3 $R/tools/variant_rates.py "$CASE" 40 > $O/$c.log 2>&1 || { echo "pass $c failed rc=$?"; tail -5 $O/$c.log; exit 1; }
done
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT --output-format csv -d $O/SQ -- python3 $R/tools/variant_rates.py "$CASE" 40 > $O/SQ.log 2>&1; echo "SQ pass rc=$?" >> $O/optional_passes.txt
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM --output-format csv -d $O/SQ2 -- python3 $R/tools/variant_rates.py "$CASE" 40 > $O/SQ2.log 2>&1; echo "SQ2 pass rc=$?" >> $O/optional_passes.txt
cd $R
python3 tools/pmc_summary.py --kernel step_kernel --envs 1048576 --alg-bytes $ALG --label "$CASE" --out $O/summary.json $O/FETCH_SIZE $O/WRITE_SIZE $O/SQ $O/SQ2 | tail -3
find $O -name "*counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete
