#!/bin/bash
# round 4, session J: bench.py with the `sustained` region -- default arguments and the driver's short regions
set -o pipefail
O=gpurun_out/${1:-r4j}
mkdir -p $O
( time timeout -k 10 500 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | tail -4 || exit 1
( time timeout -k 10 500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_args.json 2> $O/bench_driver_args.err ) 2>&1 | tail -4 || exit 1
python - <<PY
import json
for f in ("bench_default", "bench_driver_args"):
    d = json.loads(open("$O/%s.json" % f).read().strip().splitlines()[-1])
    print(f, "value %.4e  %.2f us  frac %.3f" % (d["value"], d["ms_per_step"] * 1e3, d["roofline"]["frac"]), "sustained", d.get("sustained"))
PY
