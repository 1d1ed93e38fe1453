# Round-4 pass 5: the F(4x4,3x3) forward kernel -- parity at small and bench geometries, whole-network error against the reference goldens,
# same-box A/B of the train step and the sampler with and without it
set -x
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4e}
mkdir -p $OUT
timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_bench_shapes_gpu.py -q -m gpu -x -s -k "wino43_fwd" 2>&1 | tail -40 > $OUT/fwd43_tests.txt
tail -25 $OUT/fwd43_tests.txt
timeout 1800 python -m pytest tests/test_unet_gpu.py -q -m gpu -x -s 2>&1 | tail -60 > $OUT/unet_tests.txt
VD_WINO43_FWD=0 timeout 1800 python -m pytest tests/test_unet_gpu.py -q -m gpu -x -s -k "full_size" 2>&1 | tail -30 > $OUT/unet_tests_f23.txt
grep -i "err\|passed\|failed" $OUT/unet_tests.txt | tail -30; grep -i "err\|passed\|failed" $OUT/unet_tests_f23.txt | tail -10
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-extras"
for i in 1 2; do
  $B 2>/dev/null | python -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('F(4,3) fwd: step', j['ms_per_step'], 'img/s', j['value'], 'sampling', j['sampling']['value'])"
  VD_WINO43_FWD=0 $B 2>/dev/null | python -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('F(2,3) fwd: step', j['ms_per_step'], 'img/s', j['value'], 'sampling', j['sampling']['value'])"
done > $OUT/fwd43_ab.txt 2>&1
grep -v "^+" $OUT/fwd43_ab.txt
timeout 2700 python -m pytest tests -q -m gpu -x 2>&1 | tail -30 > $OUT/gputests.txt
tail -8 $OUT/gputests.txt
