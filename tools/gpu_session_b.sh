#!/bin/bash
# round-2 GPU session B: tests, then C3 / default benches and the PMC passes of the C3 kernel
set -o pipefail
R=$PWD
O=$R/gpurun_out/r2b
mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/gputest.log 2>&1; echo "pytest rc=$?" >> $O/gputest.log; tail -15 $O/gputest.log
grep -q "rc=0" $O/gputest.log || exit 1
python bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err || exit 1
python bench.py --no-cpu-baseline --layout shadow > $O/bench_shadow.json 2>> $O/bench_default.err || exit 1
python bench.py --no-cpu-baseline --model Crazyflie --randomize --steps 600 --warmup 500 > $O/bench_c3.json 2> $O/bench_c3.err || exit 1
python bench.py --no-cpu-baseline --model Crazyflie --randomize --steps 600 --warmup 500 --envs 65536 > $O/bench_c3_65536.json 2>> $O/bench_c3.err || exit 1
python bench.py --no-cpu-baseline --model Crazyflie --steps 600 --warmup 500 > $O/bench_cf_uniform.json 2>> $O/bench_c3.err || exit 1
python bench.py --no-cpu-baseline --envs 65536 --steps 600 --warmup 1000 > $O/bench_c2_65536.json 2>> $O/bench_c3.err || exit 1
cat $O/bench_default.json $O/bench_shadow.json $O/bench_c3.json $O/bench_c3_65536.json $O/bench_cf_uniform.json $O/bench_c2_65536.json | python -c "
import sys, json
for ln in sys.stdin:
    d = json.loads(ln); r = d['roofline']
    print('%-100s %.3e  %.2f us  frac %.3f' % (d['config']['workload'][:100], d['value'], r['kernel_ms']*1e3, r['frac']))
"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_c3_fetch -- python3 $R/bench.py --steps 30 --warmup 5 --repeats 1 --no-cpu-baseline --model Crazyflie --randomize > $O/pmc_c3_fetch.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_c3_write -- python3 $R/bench.py --steps 30 --warmup 5 --repeats 1 --no-cpu-baseline --model Crazyflie --randomize > $O/pmc_c3_write.log 2>&1 || exit 1
cd $R
python3 tools/pmc_summary.py --kernel step_kernel --envs 1048576 --alg-bytes 480 --label "step_kernel<23> (C3: per-env CrazyFlie, alias layout, mixed residual rows)" --out $O/pmc_c3.json $O/pmc_c3_fetch $O/pmc_c3_write | tail -12
