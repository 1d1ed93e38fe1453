#!/bin/bash
# same-box A/B: grouped weight-gradient GEMMs as fp32 MFMA (VD_GEMM_SPLIT=0) vs split-operand bf16 x 6 (VD_GEMM_SPLIT=1)
mkdir -p gpurun_out/r4t
for rnd in 1 2; do
  for s in 0 1; do
    VD_GEMM_SPLIT=$s timeout 600 python tests/perf_wgrad43.py > gpurun_out/r4t/wg43_split${s}_$rnd.txt 2>&1
  done
done
VD_GEMM_SPLIT=1 timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_bench_shapes_gpu.py -q -m gpu -x -k "wgrad" > gpurun_out/r4t/tests_split1.txt 2>&1
tail -3 gpurun_out/r4t/tests_split1.txt
grep "32x32 256->256\|32x32 512" gpurun_out/r4t/wg43_split*_*.txt
