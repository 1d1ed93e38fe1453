#!/bin/bash
# Turn one measurement pass of tests/probe/refresh_profiles_r05.sh (scratch: gpurun_out/<run>) into the tracked summaries profiles/r05_*.
#   bash profiles/collect_r05.sh gpurun_out/r5a
set -e
R=${1:?run directory}
cd "$(dirname "$0")/.."
tail -3 $R/gputests.txt | head -2
python profiles/parse_rocprof.py stats $R/prof_cifar/runc/*_kernel_stats.csv profiles/r05_train_step_kernel_stats.csv 9
python profiles/parse_rocprof.py stats $R/prof_celeba/runc/*_kernel_stats.csv profiles/r05_celeba_train_step_kernel_stats.csv 6
if [ -d $R/prof_cifar_1s ]; then
python profiles/parse_rocprof.py stats $R/prof_cifar_1s/runc/*_kernel_stats.csv profiles/r05_train_step_kernel_stats_one_stream.csv 9
python profiles/parse_rocprof.py stats $R/prof_celeba_1s/runc/*_kernel_stats.csv profiles/r05_celeba_train_step_kernel_stats_one_stream.csv 6
fi
python profiles/parse_rocprof.py traffic $R/pmc_fetch/runc/*_counter_collection.csv $R/pmc_write/runc/*_counter_collection.csv profiles/r05_traffic.json
if [ -d $R/pmc_fetch_celeba ]; then
python profiles/parse_rocprof.py traffic $R/pmc_fetch_celeba/runc/*_counter_collection.csv $R/pmc_write_celeba/runc/*_counter_collection.csv profiles/r05_celeba_traffic.json
fi
[ -s $R/celeba_ddim250.json ] && cp $R/celeba_ddim250.json profiles/r05_celeba_ddim250.json
python profiles/parse_rocprof.py pmc profiles/r05_wino_pmc.json $R/pmc_wino/runc/*_counter_collection.csv $R/pmc_wino2/runc/*_counter_collection.csv
python - $R <<'EOF'
import json, sys
R = sys.argv[1]
d = json.load(open('profiles/r05_wino_pmc.json'))
d['kernels'] = {k: v for k, v in d['kernels'].items() if 'wino' in k or 'grouped' in k or 'true, true>' in k}
json.dump(d, open('profiles/r05_wino_pmc.json', 'w'), indent=1)
for k, v in d['kernels'].items():
    print(k, v.get('duration_us'), v.get('derived_clock_mhz'), v.get('derived_mfma_busy_frac'), v.get('SQ_INSTS_VALU'), v.get('SQ_INSTS_LDS'))
l = [x for x in open(R + '/bench_n1.json') if x.startswith('{')]
j = json.loads(l[-1])
open('profiles/r05_bench_n1.json', 'w').write(l[-1])
print({k: j[k] for k in ('value', 'ms_per_step', 'ms_fwd_bwd_only', 'ms_per_step_with_loss_item', 'hbm_peak_gib')})
r = j['roofline']
print({k: r[k] for k in ('kernel', 'achieved', 'frac', 'traffic', 'algorithmic_tflops', 'avg_launch_ms', 'launches_per_step', 'whole_step')})
print(r['hbm'] and {k: r['hbm'][k] for k in r['hbm'] if k != 'all_hbm_kernels'})
print(j['sampling']['value'], j['sampling']['roofline']['frac'])
print(j['cpu_baseline'])
s = j['secondary']
print(s['value'], s['ms_per_step'], s['hbm_peak_gib'], s['sampling']['value'], s['cpu_baseline']['value'], s['roofline']['kernel'], s['roofline']['frac'],
      s['roofline']['whole_step'])
EOF
( echo "# in-kernel clock stamps of the Winograd kernels (libvdiff_hip_probe.so; tests/probe/wino_phases.py with VD_WINO_PROBE_LIGHT=1, tests/probe/wgrad_clock.py)"
  echo "# one MI355X box, round 5; cycles = s_memtime (shader clock), MHz = s_memtime ticks per s_memrealtime microsecond"
  grep -v amdgpu.ids $R/wino_phases_light.txt; grep -v amdgpu.ids $R/wgrad_clock.txt ) > profiles/r05_wino_clock.txt
[ -s $R/clock_by_kernel.txt ] && grep -v amdgpu.ids $R/clock_by_kernel.txt > profiles/r05_clock_by_kernel.txt
head -16 profiles/r05_train_step_kernel_stats.csv
head -10 profiles/r05_celeba_train_step_kernel_stats.csv
