# Round-4 pass 6: the F(4x4,3x3) forward kernel with the residual loads hoisted in front of the exchange barrier and the factored
# dyadic transforms, against its first version (scratch library)
set -x
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4f}
mkdir -p $OUT
L=$GRAFT_REPO_ROOT/v-diffusion-torch_amd/lib
timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_bench_shapes_gpu.py -q -m gpu -x -k "wino43" 2>&1 | tail -5 > $OUT/w43_tests.txt
cat $OUT/w43_tests.txt
for i in 1 2; do
  echo "== product"; python tests/perf_wino43.py 2>&1 | grep "FORWARD"
  echo "== first version"; VDIFF_HIP_LIB=$L/exp/libvd_fwd43_v1.so python tests/perf_wino43.py 2>&1 | grep "FORWARD"
done > $OUT/fwd43_ab2.txt 2>&1
grep -v "^+" $OUT/fwd43_ab2.txt
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-extras"
for i in 1 2; do
  $B 2>/dev/null | python -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('product: step', j['ms_per_step'], 'sampling', j['sampling']['value'])"
  VDIFF_HIP_LIB=$L/exp/libvd_fwd43_v1.so $B 2>/dev/null | python -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('first version: step', j['ms_per_step'], 'sampling', j['sampling']['value'])"
done > $OUT/step_ab2.txt 2>&1
grep -v "^+" $OUT/step_ab2.txt
