set -x
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4p
mkdir -p $OUT
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_unet_gpu.py -q -m gpu -x -k "gn_ or full_size or dropout" 2>&1 | tail -3
python tests/perf_gn.py 2>&1 | grep "TB/s" | cut -c1-120
B="python bench.py --steps 20 --warmup 5 --no-sample --no-cpu-baseline --no-secondary --no-extras"
for i in 1 2 3; do $B 2>/dev/null | python -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('step', j['ms_per_step'])"; done
