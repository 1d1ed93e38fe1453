# cold-cache cost and result of bench.py's stock PyTorch-ROCm baseline leg under MIOpen's find modes (each run gets fresh MIOpen db / cache dirs)
for mode in "bench1" "bench0" "fast" "hybrid"; do
  d=$(mktemp -d)
  export MIOPEN_USER_DB_PATH=$d/db MIOPEN_CUSTOM_CACHE_DIR=$d/cache
  mkdir -p $d/db $d/cache
  unset MIOPEN_FIND_MODE VD_TORCH_BASELINE_BENCHMARK
  case $mode in
    bench1) export VD_TORCH_BASELINE_BENCHMARK=1;;
    bench0) export VD_TORCH_BASELINE_BENCHMARK=0;;
    fast) export VD_TORCH_BASELINE_BENCHMARK=1 MIOPEN_FIND_MODE=FAST;;
    hybrid) export VD_TORCH_BASELINE_BENCHMARK=1 MIOPEN_FIND_MODE=HYBRID;;
  esac
  s=$(date +%s)
  python bench.py --no-secondary --no-sample --no-cpu-baseline --no-extras --steps 3 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$mode', d['torch_rocm_baseline'])"
  echo "$mode wall $(( $(date +%s) - s )) s"
done
