#!/bin/bash
# persistent grids that fill their rounds evenly (VD_EVEN_ROUNDS=1, default) vs one workgroup per CU (0): Winograd tests, CelebA + CIFAR step A/B
# (record of a measurement: the VD_EVEN_ROUNDS switch existed only in the build this script measured, see FINDINGS.md round 4; it was not kept)
mkdir -p gpurun_out/r4y
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_bench_shapes_gpu.py -q -m gpu -x -k "wino or reserved" > gpurun_out/r4y/tests.txt 2>&1; tail -2 gpurun_out/r4y/tests.txt
for rnd in 1 2; do
  for e in 0 1; do
    VD_EVEN_ROUNDS=$e timeout 300 python bench.py --config celeba --steps 6 --warmup 2 --no-sample --no-cpu-baseline --no-secondary --no-extras > gpurun_out/r4y/celeba_e${e}_$rnd.json 2>/dev/null
    VD_EVEN_ROUNDS=$e timeout 300 python bench.py --steps 20 --warmup 5 --no-sample --no-cpu-baseline --no-secondary --no-extras > gpurun_out/r4y/cifar_e${e}_$rnd.json 2>/dev/null
    python - <<PY
import json
for w in ("celeba","cifar"):
    b=json.loads([l for l in open("gpurun_out/r4y/%s_e${e}_$rnd.json" % w) if l.startswith("{")][-1])
    k=b["roofline"]["all_matmul_kernels"] if "all_matmul_kernels" in b["roofline"] else b["roofline"].get("top_matmul_kernels",{})
    print("even $e round $rnd", w, b["ms_per_step"], {n: v["ms_per_step"] for n, v in k.items() if "wino43_conv_kernel<4" in n})
PY
  done
done
