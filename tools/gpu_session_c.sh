#!/bin/bash
# round-2 GPU session C: tests, latency breakdown, per-episode re-randomisation rate, rocprof stats of C3
set -o pipefail
R=$PWD
O=$R/gpurun_out/r2c
mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/gputest.log 2>&1; echo "pytest rc=$?" >> $O/gputest.log; tail -15 $O/gputest.log
grep -q "rc=0" $O/gputest.log || exit 1
python tools/latency_breakdown.py > $O/latency_breakdown.json 2> $O/latency_breakdown.err || { tail -20 $O/latency_breakdown.err; exit 1; }
python -c "
import json; d=json.load(open('$O/latency_breakdown.json'))
for r in d['rows']: print(r['N'], {k: round(v,2) for k,v in r['breakdown_eager_us'].items()}, {k: round(v,2) for k,v in r['breakdown_graph_us'].items()}, round(r['frac_352B_eager'],3), round(r['frac_352B_graph'],3))
"
for extra in "" "--stagger" "--stagger --randomize-every 1" "--stagger --randomize-every 4"; do
  python bench.py --no-cpu-baseline --model Crazyflie --randomize --steps 600 --warmup 600 --repeats 3 $extra >> $O/bench_rerandomize.jsonl 2>> $O/bench_rerandomize.err || { tail -20 $O/bench_rerandomize.err; exit 1; }
done
python -c "
import json
for ln in open('$O/bench_rerandomize.jsonl'):
    d=json.loads(ln); print('%.3e  %.2f us/step  %s' % (d['value'], d['ms_per_step']*1e3, d['config']['workload'][-120:]))
"
