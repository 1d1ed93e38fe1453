# Round-4 pass 8: LDS-only barriers in the epilogue of the F(4x4,3x3) kernel (+ U prefetch distance 2) against the previous build
set -x
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4h}
mkdir -p $OUT
L=$GRAFT_REPO_ROOT/v-diffusion-torch_amd/lib
timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_bench_shapes_gpu.py -q -m gpu -x -k "wino43" 2>&1 | tail -3 > $OUT/w43_tests.txt
cat $OUT/w43_tests.txt
for i in 1 2; do
  echo "== product"; python tests/perf_wino43.py 2>&1 | grep "FORWARD"
  echo "== previous (v2)"; VDIFF_HIP_LIB=$L/exp/libvd_fwd43_v2.so python tests/perf_wino43.py 2>&1 | grep "FORWARD"
done > $OUT/ab.txt 2>&1
grep -v "^+" $OUT/ab.txt | cut -c1-200
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-extras"
for i in 1 2; do
  $B 2>/dev/null | python -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('product: step', j['ms_per_step'], 'sampling', j['sampling']['value'])"
  VDIFF_HIP_LIB=$L/exp/libvd_fwd43_v2.so $B 2>/dev/null | python -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('previous: step', j['ms_per_step'], 'sampling', j['sampling']['value'])"
done > $OUT/step_ab.txt 2>&1
grep -v "^+" $OUT/step_ab.txt
