#!/bin/bash
# final GPU sessions of round 4: the judged artefacts for the COMMITTED sources, in parts that each fit one gpurun call.
#   bash tools/gpu_session_r4final.sh <part> [name]      part = a1 | a2 | b | hunt
#   a1: pytest -m gpu (+ kernel coverage), bench_variants.jsonl (layouts, C2 / C3 at their own N, graphs, rollouts, re-randomisation, swarm,
#       config 4's per-GPU share with RCCL on one rank, and the LARGE-N lines 2^22 / 2^23: state far above the 256-MB Infinity Cache)
#   a2: variant_rates, latency breakdown (measurement build), PCIe-inclusive rate, parity report, oracle drift
#   b : rocprofv3 --kernel-trace --stats of the default command and of config 3, PMC traffic (separate FETCH_SIZE / WRITE_SIZE passes) of the
#       default kernel in its three layouts and of config 3, then the default bench.py line with those profiles in place
#   hunt: ten seeds of tools/hunt.sh on the final kernels (at most ten per round)
set -o pipefail
R=$PWD
part=${1:?part}
O=$R/gpurun_out/${2:-r4final}
X=r04
mkdir -p $O
B="python bench.py --no-cpu-baseline --no-layouts"
case $part in
a1)
  rm -f $O/coverage.json
  KERNEL_COVERAGE_OUT=$O/coverage.json timeout -k 10 900 python -m pytest tests -m gpu -q > $O/gputest.log 2>&1; echo "pytest rc=$?" >> $O/gputest.log; tail -5 $O/gputest.log
  python tools/kernel_coverage.py $O/coverage.json > $O/kernel_coverage.txt 2>&1; head -3 $O/kernel_coverage.txt
  grep -q "rc=0" $O/gputest.log || exit 1
  ( $B --layout shadow; $B --layout plain
    $B --envs 65536 --steps 1000; $B --envs 65536 --steps 100 --warmup 100 --graph 32; $B --envs 65536 --rollout 64 --steps 30 --warmup 5
    $B --model Crazyflie --randomize --steps 600 --warmup 600; $B --model Crazyflie --randomize --envs 65536 --steps 1000
    $B --model Crazyflie --randomize --envs 65536 --steps 100 --warmup 100 --graph 32
    $B --model Crazyflie --steps 600 --warmup 600
    $B --rollout 64 --steps 30 --warmup 5; $B --model Crazyflie --randomize --rollout 64 --steps 30 --warmup 5
    $B --model Crazyflie --randomize --steps 600 --warmup 600 --stagger; $B --model Crazyflie --randomize --steps 600 --warmup 600 --stagger --randomize-every 1; $B --model RandomQuad --steps 600 --warmup 600 --stagger --randomize-every 1
    $B --envs 131072 --steps 1000; $B --envs 131072 --steps 100 --warmup 100 --graph 32; $B --swarm 8 --steps 300 --warmup 100; $B --swarm 8 --envs 131072 --steps 600 --warmup 200
    $B --envs 4194304 --steps 300 --warmup 300; $B --envs 8388608 --steps 150 --warmup 150; $B --envs 4194304 --layout shadow --steps 300 --warmup 300
    GAQ_BENCH_FORCE_DIST=1 $B --envs 131072 --steps 600 --warmup 300 --repeats 3 ) > $O/bench_variants.jsonl 2> $O/bench_variants.err || { tail -20 $O/bench_variants.err; exit 1; }
  python -c "
import json
for ln in open('$O/bench_variants.jsonl'):
    d=json.loads(ln); r=d['roofline']; print('%.3e  %7.2f us/step kern %7.2f frac %.3f  %s' % (d['value'], d['ms_per_step']*1e3, r['kernel_ms']*1e3, r['frac'], d['config']['workload'][:34]+' ... '+d['config']['workload'][-110:]))
"
  ;;
a2)
  python tools/variant_rates.py > $O/variant_rates.json 2> $O/variant_rates.err || { tail -5 $O/variant_rates.err; exit 1; }
  python -c "
import json
for k, v in json.load(open('$O/variant_rates.json')).items(): print('%7.2f us  v%-5d %s' % (v['us_per_step'], v['kernel_variant'], k))
"
  bash tools/latency_breakdown.sh > $O/latency_breakdown.json 2> $O/latency_breakdown.err || { tail -5 $O/latency_breakdown.err; exit 1; }
  python tools/pcie_rate.py > $O/pcie_rate.json 2> $O/pcie_rate.err || { tail -5 $O/pcie_rate.err; exit 1; }
  python tools/parity_report.py > $O/parity_report.json 2> $O/parity_report.err || { tail -5 $O/parity_report.err; exit 1; }
  python tools/oracle_drift.py > $O/oracle_drift.json 2> $O/oracle_drift.err || { tail -5 $O/oracle_drift.err; exit 1; }
  python -c "
import json
d=json.load(open('$O/oracle_drift.json'))
for k,v in d.items(): print('oracle_drift', k, 'max %.3g' % v['max'], 'frac>1e-6', v['frac_above_1e-6'])
d=json.load(open('$O/parity_report.json'))
for k,v in d.items(): print('parity_report', k, 'max %.3g' % v['max'])
"
  ;;
b)
  cd /tmp && export TMPDIR=/tmp
  # (--no-layouts: the layout regions of the default line launch the SAME instantiation with other arguments and would mix into this kernel's average)
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_default -- python3 $R/bench.py --no-cpu-baseline --no-layouts > $O/bench_under_rocprof.json 2> $O/stats_default.log || exit 1
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c3 -- python3 $R/bench.py --no-cpu-baseline --no-layouts --model Crazyflie --randomize --steps 600 --warmup 600 --repeats 2 > $O/bench_c3_under_rocprof.json 2> $O/stats_c3.log || exit 1
  P="--steps 30 --warmup 5 --repeats 1 --no-cpu-baseline --no-layouts"
  for key in default_alias default_shadow default_plain c3_alias; do
    case $key in
      default_alias) args="";; default_shadow) args="--layout shadow";; default_plain) args="--layout plain";; c3_alias) args="--model Crazyflie --randomize";;
    esac
    for c in FETCH_SIZE WRITE_SIZE; do
      rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_${key}_$c -- python3 $R/bench.py $P $args > $O/pmc_${key}_$c.log 2>&1 || exit 1
    done
  done
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT --output-format csv -d $O/pmc_default_alias_SQ -- python3 $R/bench.py $P > $O/pmc_default_alias_SQ.log 2>&1; echo "SQ pass (optional) default_alias rc=$?" | tee -a $O/optional_passes.txt
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT --output-format csv -d $O/pmc_c3_alias_SQ -- python3 $R/bench.py $P --model Crazyflie --randomize > $O/pmc_c3_alias_SQ.log 2>&1; echo "SQ pass (optional) c3_alias rc=$?" | tee -a $O/optional_passes.txt
  cd $R
  python3 tools/pmc_summary.py --kernel step_kernel --envs 1048576 --alg-bytes 352 --label "step_kernel<148> (default: Hummingbird, alias layout)" --index-key default_alias --out $O/pmc_default_alias.json $O/pmc_default_alias_FETCH_SIZE $O/pmc_default_alias_WRITE_SIZE $O/pmc_default_alias_SQ | grep traffic_bytes_per_env
  python3 tools/pmc_summary.py --kernel step_kernel --envs 1048576 --alg-bytes 352 --label "step_kernel<148> (Hummingbird, library-owned heads + obs copy)" --index-key default_shadow --out $O/pmc_default_shadow.json $O/pmc_default_shadow_FETCH_SIZE $O/pmc_default_shadow_WRITE_SIZE | grep traffic_bytes_per_env
  python3 tools/pmc_summary.py --kernel step_kernel --envs 1048576 --alg-bytes 352 --label "step_kernel<4> (Hummingbird, fp64 planes)" --index-key default_plain --out $O/pmc_default_plain.json $O/pmc_default_plain_FETCH_SIZE $O/pmc_default_plain_WRITE_SIZE | grep traffic_bytes_per_env
  python3 tools/pmc_summary.py --kernel step_kernel --envs 1048576 --alg-bytes 480 --label "step_kernel<151> (C3: per-env CrazyFlie, alias layout, mixed residual rows)" --index-key c3_alias --out $O/pmc_c3_alias.json $O/pmc_c3_alias_FETCH_SIZE $O/pmc_c3_alias_WRITE_SIZE $O/pmc_c3_alias_SQ | grep traffic_bytes_per_env
  find $O -name "*counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete
  # the default line LAST, with the PMC profiles of these very sources in place on this box's copy of the repo, so that it carries
  # roofline.traffic / measured_frac and the layouts' traffic (tools/collect_final.sh repeats the copy at home)
  for k in default_alias default_shadow default_plain c3_alias; do cp $O/pmc_$k.json $R/profiles/${X}_final_pmc_$k.json; done
  python3 - <<PY
import json
d = json.load(open('$O/pmc_index.json'))
for k, v in d.items():
    v['file'] = '${X}_final_' + v['file']
json.dump(d, open('$R/profiles/pmc_index.json', 'w'), indent=1, sort_keys=True)
PY
  python bench.py > $O/bench_default.json 2> $O/bench_default.err || { tail -5 $O/bench_default.err; exit 1; }
  python -c "
import json
d=json.load(open('$O/bench_default.json')); r=d['roofline']
print('DEFAULT value %.4e  %.2f us/step  kernel %.2f us  frac %.3f  measured_frac %s  traffic/env-step %s  peak_measured %s frac_of_measured %s' % (d['value'], d['ms_per_step']*1e3, r['kernel_ms']*1e3, r['frac'], r.get('measured_frac'), r.get('traffic_bytes_per_env_step'), r.get('peak_measured'), r.get('frac_of_measured')))
for k,v in d.get('layouts',{}).items(): print('  layout %-7s %.2f us  frac %.3f  traffic %s' % (k, v['us_per_step'], v['frac'], v['traffic_bytes_per_env_step']))
s=d.get('staggered_episodes'); print('  staggered %.2f us  frac %.3f (%d regions)' % (s['us_per_step'], s['frac'], s['regions']))
print('  cpu_baseline %.3e on %d cores; vs_cpu_baseline %.0f' % (d['cpu_baseline']['value'], d['cpu_baseline']['cores'], d['vs_cpu_baseline']))
"
  # the driver's own invocation, for the record
  python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_args.json 2> $O/bench_driver_args.err || { tail -5 $O/bench_driver_args.err; exit 1; }
  python -c "
import json
d=json.load(open('$O/bench_driver_args.json')); r=d['roofline']
print('DRIVER ARGS value %.4e  %.2f us/step  kernel %.2f us  frac %.3f' % (d['value'], d['ms_per_step']*1e3, r['kernel_ms']*1e3, r['frac']))
"
  ;;
hunt)
  rm -f gpurun_out/hunt/hunt.txt
  bash tools/hunt.sh ${3:-141} ${4:-150} || exit 1
  cp gpurun_out/hunt/hunt.txt $O/hunt.txt
  ;;
*) echo "unknown part $part"; exit 2;;
esac
exit 0
