#!/bin/bash
# 2 x 2: pre-drawn noise x non-temporal policy, by batch size
set -o pipefail
R=$PWD
O=$R/gpurun_out/r2g
mkdir -p $O; rm -f $O/sweep.jsonl
for round in 1 2 3; do
for n in 16384 65536 131072 262144 1048576; do
  for combo in 00 10 01 11; do
    for cfg in "" "--model Crazyflie --randomize"; do
      export GAQ_PREDRAW=${combo:0:1} GAQ_NT=${combo:1:1}
      python bench.py --no-cpu-baseline --repeats 3 --envs $n --steps 1000 --warmup 1500 $cfg 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'N': $n, 'predraw': ${combo:0:1}, 'nt': ${combo:1:1}, 'cfg': '$cfg', 'us_per_step': d['ms_per_step'] * 1e3, 'frac': d['roofline']['frac']}))
" >> $O/sweep.jsonl || exit 1
    done
  done
done
done
python - <<'PY'
import json, collections
rows = [json.loads(l) for l in open("gpurun_out/r2g/sweep.jsonl")]
t = collections.defaultdict(list)
for r in rows:
    t[(r["cfg"], r["N"], r["predraw"], r["nt"])].append(r["us_per_step"])
for c in sorted({r["cfg"] for r in rows}):
    print("cfg:", c or "(default Hummingbird)", "   columns: predraw/nt = 0/0, 1/0, 0/1, 1/1 (best of 3, us per step)")
    for n in sorted({r["N"] for r in rows}):
        xs = [min(t[(c, n, p, q)]) for p, q in ((0, 0), (1, 0), (0, 1), (1, 1))]
        print("   N=%8d   %s" % (n, "  ".join("%7.2f (%+5.1f%%)" % (x, 100 * (x / xs[0] - 1)) for x in xs)))
PY
