#!/bin/bash
# A/B of cache-policy bits on the step kernels' streaming loads / stores (gaq_kernels.hpp GAQ_LD_AUX / GAQ_ST_AUX / GAQ_ACT_AUX:
# 1 = sc0, 2 = nt, 16 = sc1).  Step 1 (build container, no GPU): `bash tools/aux_variants.sh build` compiles one library
# per variant into gpurun_out/aux/ (it travels to the GPU box).  Step 2 (GPU): `bash tools/aux_variants.sh run` benches each.
set -o pipefail
R=$PWD
L=$R/build/aux          # the variant libraries (git-ignored, but they travel to the GPU box)
O=$R/gpurun_out/aux     # results
VARIANTS="base:0:0:0 st_sc1:0:16:0 st_nt:0:2:0 ld_nt:2:0:0 ld_nt_st_sc1:2:16:0 ld_nt_st_nt:2:2:0 act_nt:0:0:2 all_nt_sc1:2:16:2 st_sc0sc1:0:17:0"
if [ "$1" = "build" ]; then
  mkdir -p $L
  for v in $VARIANTS; do
    IFS=: read name ld st act <<< "$v"
    make -s -j8 -C gym_art_amd/csrc OUT=$L/libgaq_$name.so OBJ=$R/build/obj_$name EXTRA="-DGAQ_LD_AUX=$ld -DGAQ_ST_AUX=$st -DGAQ_ACT_AUX=$act" 2>/dev/null
  done
  ls -la $L/*.so
  exit 0
fi
mkdir -p $O; rm -f $O/results.jsonl
for round in 1 2; do
  for v in $VARIANTS; do
    IFS=: read name ld st act <<< "$v"
    for cfg in "" "--envs 65536 --steps 1000" "--model Crazyflie --randomize --steps 600 --warmup 600" "--layout shadow"; do
      GAQ_LIB=$L/libgaq_$name.so python bench.py --no-cpu-baseline --repeats 3 $cfg 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'variant': '$name', 'ld': $ld, 'st': $st, 'act': $act, 'cfg': '$cfg', 'us_per_step': d['ms_per_step'] * 1e3, 'kernel_us': d['roofline']['kernel_ms'] * 1e3, 'value': d['value']}))
" >> $O/results.jsonl || exit 1
    done
  done
done
python - <<'PY'
import json, collections
rows = [json.loads(l) for l in open("gpurun_out/aux/results.jsonl")]
t = collections.defaultdict(list)
for r in rows:
    t[(r["cfg"], r["variant"])].append(r["us_per_step"])
cfgs = sorted({r["cfg"] for r in rows})
for c in cfgs:
    print("cfg:", c or "(default)")
    base = min(t[(c, "base")])
    for v in dict.fromkeys(r["variant"] for r in rows):
        xs = t[(c, v)]
        print("   %-14s %s   best %.2f us  (%+.1f %% vs base)" % (v, " ".join("%.2f" % x for x in xs), min(xs), 100 * (min(xs) / base - 1)))
PY
