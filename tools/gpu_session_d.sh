#!/bin/bash
# round-2 GPU session D: tests, small-batch rates with the pre-drawn-noise kernels, compacted re-randomisation
set -o pipefail
R=$PWD
O=$R/gpurun_out/r2d
mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/gputest.log 2>&1; echo "pytest rc=$?" >> $O/gputest.log; tail -15 $O/gputest.log
grep -q "rc=0" $O/gputest.log || exit 1
python tools/latency_breakdown.py > $O/latency_breakdown.json 2> $O/latency_breakdown.err || { tail -20 $O/latency_breakdown.err; exit 1; }
python -c "
import json; d=json.load(open('$O/latency_breakdown.json'))
for r in d['rows']: print(r['N'], {k: round(v,2) for k,v in r['breakdown_eager_us'].items()}, {k: round(v,2) for k,v in r['breakdown_graph_us'].items()}, round(r['frac_352B_eager'],3), round(r['frac_352B_graph'],3))
"
for extra in "--stagger" "--stagger --randomize-every 1" "--stagger --randomize-every 4"; do
  python bench.py --no-cpu-baseline --model Crazyflie --randomize --steps 600 --warmup 600 --repeats 3 $extra >> $O/bench_rerandomize.jsonl 2>> $O/bench_rerandomize.err || { tail -20 $O/bench_rerandomize.err; exit 1; }
done
python bench.py --no-cpu-baseline --steps 1100 --warmup 2000 --repeats 3 >> $O/bench_predraw_ab.jsonl 2>> $O/bench_predraw_ab.err
GAQ_FORCE_PREDRAW=1 python bench.py --no-cpu-baseline --steps 1100 --warmup 2000 --repeats 3 >> $O/bench_predraw_ab.jsonl 2>> $O/bench_predraw_ab.err
python bench.py --no-cpu-baseline --model Crazyflie --randomize --steps 600 --warmup 600 --repeats 3 >> $O/bench_predraw_ab.jsonl 2>> $O/bench_predraw_ab.err
GAQ_FORCE_PREDRAW=1 python bench.py --no-cpu-baseline --model Crazyflie --randomize --steps 600 --warmup 600 --repeats 3 >> $O/bench_predraw_ab.jsonl 2>> $O/bench_predraw_ab.err
for n in 65536 131072; do
  python bench.py --no-cpu-baseline --envs $n --steps 1000 --warmup 2000 --repeats 3 >> $O/bench_small.jsonl 2>> $O/bench_small.err
  python bench.py --no-cpu-baseline --envs $n --steps 100 --warmup 100 --repeats 3 --graph 32 >> $O/bench_small.jsonl 2>> $O/bench_small.err
  python bench.py --no-cpu-baseline --envs $n --steps 1000 --warmup 2000 --repeats 3 --model Crazyflie --randomize >> $O/bench_small.jsonl 2>> $O/bench_small.err
  python bench.py --no-cpu-baseline --envs $n --steps 100 --warmup 100 --repeats 3 --graph 32 --model Crazyflie --randomize >> $O/bench_small.jsonl 2>> $O/bench_small.err
done
python -c "
import json
for f in ('bench_rerandomize','bench_predraw_ab','bench_small'):
    for ln in open('$O/'+f+'.jsonl'):
        d=json.loads(ln); r=d['roofline']; print('%-18s %.3e  %.2f us/step kern %.2f frac %.3f %s' % (f, d['value'], d['ms_per_step']*1e3/(1 if not 'graph of' in d['config']['workload'] else 1), r['kernel_ms']*1e3, r['frac'], d['config']['workload'][:40]+' ... '+d['config']['workload'][-90:]))
"
