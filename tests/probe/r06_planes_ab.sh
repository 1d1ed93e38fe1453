# Round 6: same-box A/B of the grouped weight-gradient GEMMs -- 128x128 tiles (VD_PLANES256=0) vs wgrad_planes256_kernel, and gemm.hip built
# with / without the SLP vectorizer (scratch library lib/exp/libgemm_noslp.so) -- on the weight-gradient path alone and on the whole train step
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6p
mkdir -p $OUT
L=$GRAFT_REPO_ROOT/v-diffusion-torch_amd/lib/exp/libgemm_noslp.so
python -m pytest tests/test_kernels_gpu.py tests/test_bench_shapes_gpu.py -q -m gpu -x -k "grouped_wgrad or wino43_wgrad or split_operand" 2>&1 | tail -3 > $OUT/tests_planes.txt
VDIFF_HIP_LIB=$L python -m pytest tests/test_kernels_gpu.py tests/test_bench_shapes_gpu.py -q -m gpu -x -k "grouped_wgrad or wino43_wgrad or gemm_kinds or conv1x1 or attention" 2>&1 | tail -3 > $OUT/tests_noslp.txt
for rep in 1 2; do
  VD_PLANES256=0 python tests/perf_wgrad43.py 2>&1 | grep -v amdgpu.ids | sed "s/^/t128 rep$rep: /" >> $OUT/wgrad_ab.txt
  python tests/perf_wgrad43.py 2>&1 | grep -v amdgpu.ids | sed "s/^/p256 rep$rep: /" >> $OUT/wgrad_ab.txt
  VDIFF_HIP_LIB=$L python tests/perf_wgrad43.py 2>&1 | grep -v amdgpu.ids | sed "s/^/p256+noslp rep$rep: /" >> $OUT/wgrad_ab.txt
  B="--steps 20 --warmup 5 --no-sample --no-cpu-baseline --no-secondary --no-extras --no-torch-baseline --no-fp32-ab --no-calibration"
  VD_PLANES256=0 python bench.py $B 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('t128 rep$rep step', j['ms_per_step'])" >> $OUT/step_ab.txt
  python bench.py $B 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('p256 rep$rep step', j['ms_per_step'])" >> $OUT/step_ab.txt
  VDIFF_HIP_LIB=$L python bench.py $B 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('p256+noslp rep$rep step', j['ms_per_step'])" >> $OUT/step_ab.txt
done
cat $OUT/tests_planes.txt $OUT/tests_noslp.txt; cat $OUT/wgrad_ab.txt | cut -c1-230; cat $OUT/step_ab.txt
