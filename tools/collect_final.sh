#!/bin/bash
# copy the judged artefacts of `bash tools/gpu_session_final.sh <name>` from gpurun_out/<name> into profiles/ as <prefix>_final_*
O=gpurun_out/${1:?session name}; P=profiles; X=${2:-r03}
for f in bench_default.json bench_under_rocprof.json bench_c3_under_rocprof.json bench_variants.jsonl variant_rates.json latency_breakdown.json pcie_rate.json parity_report.json oracle_drift.json gputest.log pmc_default_alias.json pmc_default_shadow.json pmc_default_plain.json pmc_c3_alias.json; do cp $O/$f $P/${X}_final_$f; done
cp $(ls -t $O/stats_default/runc/*_kernel_stats.csv | head -1) $P/${X}_final_kernel_stats_default.csv
cp $(ls -t $O/stats_c3/runc/*_kernel_stats.csv | head -1) $P/${X}_final_kernel_stats_c3.csv
python3 - <<PY
import json
d = json.load(open('$O/pmc_index.json'))
for k, v in d.items():
    v['file'] = '${X}_final_' + v['file']
json.dump(d, open('$P/pmc_index.json', 'w'), indent=1, sort_keys=True)
PY
[ -f $O/kernel_coverage.txt ] && cp $O/kernel_coverage.txt $P/${X}_kernel_coverage.txt
python3 tools/kernel_resources.py > $P/${X}_kernel_resources.txt 2>&1
