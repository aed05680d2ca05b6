#!/bin/bash
# small-batch kernels (pre-drawn noise + non-temporal policy) against the ordinary instantiations, by batch size
set -o pipefail
R=$PWD
O=$R/gpurun_out/r2f
mkdir -p $O; rm -f $O/sweep.jsonl
python -m pytest tests -m gpu -x -q -k "predrawn or determinism or hummingbird_500 or per_env_randomized or crazyflie_motor_lag or bound_step" > $O/gputest.log 2>&1; echo "pytest rc=$?" >> $O/gputest.log; tail -3 $O/gputest.log
grep -q "rc=0" $O/gputest.log || exit 1
for round in 1 2; do
for n in 16384 32768 65536 131072 262144 524288; do
  for mode in small ordinary; do
    for cfg in "" "--model Crazyflie --randomize"; do
      if [ $mode = small ]; then export GAQ_FORCE_PREDRAW=1; unset GAQ_NO_PREDRAW; else export GAQ_NO_PREDRAW=1; unset GAQ_FORCE_PREDRAW; fi
      python bench.py --no-cpu-baseline --repeats 3 --envs $n --steps 1000 --warmup 1500 $cfg 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'N': $n, 'mode': '$mode', 'cfg': '$cfg', 'us_per_step': d['ms_per_step'] * 1e3, 'frac': d['roofline']['frac'], 'value': d['value']}))
" >> $O/sweep.jsonl || exit 1
    done
  done
done
done
python - <<'PY'
import json, collections
rows = [json.loads(l) for l in open("gpurun_out/r2f/sweep.jsonl")]
t = collections.defaultdict(list)
for r in rows:
    t[(r["cfg"], r["N"], r["mode"])].append(r["us_per_step"])
for c in sorted({r["cfg"] for r in rows}):
    print("cfg:", c or "(default Hummingbird)")
    for n in sorted({r["N"] for r in rows}):
        a, b = min(t[(c, n, "small")]), min(t[(c, n, "ordinary")])
        print("   N=%7d  small-batch kernel %7.2f us   ordinary %7.2f us   (%+.1f %%)" % (n, a, b, 100 * (a / b - 1)))
PY
