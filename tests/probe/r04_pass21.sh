#!/bin/bash
# after renaming the split forms: GEMM tests, profile refresh, default bench line
mkdir -p gpurun_out/r4n
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "gemm or split_operand or wgrad" > gpurun_out/r4n/gemm_tests.txt 2>&1; tail -2 gpurun_out/r4n/gemm_tests.txt
bash tests/probe/r04_refresh_all.sh > gpurun_out/refresh_r4n.log 2>&1
timeout 1500 python bench.py > gpurun_out/r4n/bench_n1.json 2> gpurun_out/r4n/bench_n1.err
timeout 900 python tests/probe/celeba_ddim250.py > gpurun_out/r4n/celeba_ddim250.json 2> gpurun_out/r4n/celeba_ddim250.err
tail -c 600 gpurun_out/r4n/bench_n1.json
