#!/bin/bash
# round 4: the default kernel over batch sizes 2^14 ... 2^23 on the final sources (the sweep was round 1's kernel)
set -o pipefail
O=gpurun_out/${1:-r4sweep}; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-layouts"
( for e in 14 15 16 17 18 19 20 21 22 23; do
    n=$((1 << e)); k=$(( e < 18 ? 2000 : (e < 21 ? 1000 : 300) ))
    $B --envs $n --steps $k --warmup $k || exit 1
  done ) > $O/n_sweep.jsonl 2> $O/n_sweep.err || { tail -5 $O/n_sweep.err; exit 1; }
python - <<PY
import json
for ln in open("$O/n_sweep.jsonl"):
    d = json.loads(ln); r = d["roofline"]
    print("N=%8d  %8.2f us/step  %.3e env-steps/s  frac %.3f  %s" % (d["config"]["total_envs"], d["ms_per_step"] * 1e3, d["value"], r["frac"], r["kernel"]))
PY
