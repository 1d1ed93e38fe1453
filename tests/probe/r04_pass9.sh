# Round-4 pass 9: per-image 64-bit bases in the F(4x4,3x3) convolution kernel (tensors beyond 2 GiB), against the previous build
set -x
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4i}
mkdir -p $OUT
L=$GRAFT_REPO_ROOT/v-diffusion-torch_amd/lib
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_bench_shapes_gpu.py -q -m gpu -x -k "wino43 or ddim250" 2>&1 | tail -4 > $OUT/tests.txt
cat $OUT/tests.txt
for i in 1 2; do
  echo "== product"; python tests/perf_wino43.py 2>&1 | grep "FORWARD" | head -3
  echo "== previous"; VDIFF_HIP_LIB=$L/exp/libvd_prev.so python tests/perf_wino43.py 2>&1 | grep "FORWARD" | head -3
done > $OUT/ab.txt 2>&1
grep -v "^+" $OUT/ab.txt | cut -c1-200
timeout 900 python tests/probe/celeba_ddim250.py > $OUT/celeba_ddim250.json 2> $OUT/celeba_ddim250.err
cat $OUT/celeba_ddim250.json
