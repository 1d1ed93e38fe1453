#!/bin/bash
# same-box A/B: split-operand grouped weight-gradient GEMMs with two LDS stages (lib/exp/libvd_noring.so) vs a ring of three (product)
mkdir -p gpurun_out/r4t
NR=$PWD/v-diffusion-torch_amd/lib/exp/libvd_noring.so
for rnd in 1 2; do
  VD_GEMM_SPLIT=0 timeout 600 python tests/perf_wgrad43.py > gpurun_out/r4t/b_fp32_$rnd.txt 2>&1
  VD_GEMM_SPLIT=1 VDIFF_HIP_LIB=$NR timeout 600 python tests/perf_wgrad43.py > gpurun_out/r4t/b_noring_$rnd.txt 2>&1
  VD_GEMM_SPLIT=1 timeout 600 python tests/perf_wgrad43.py > gpurun_out/r4t/b_ring_$rnd.txt 2>&1
done
VD_GEMM_SPLIT=1 timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_bench_shapes_gpu.py -q -m gpu -x -k "wgrad" > gpurun_out/r4t/tests_ring.txt 2>&1
tail -3 gpurun_out/r4t/tests_ring.txt
for f in gpurun_out/r4t/b_*_2.txt; do echo $f; sed 's/F(2,3).*| F(4,3)/F43/' $f | grep -v amdgpu.ids | cut -c1-120; done
