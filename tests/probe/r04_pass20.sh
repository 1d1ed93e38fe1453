#!/bin/bash
# fused slab reduction + finish of the F(4x4,3x3) weight gradient; CelebA B = 64 step test; split-operand forms test
mkdir -p gpurun_out/r4v
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_bench_shapes_gpu.py -q -m gpu -x -k "wgrad or split_operand or celeba_train_step_b64" -s > gpurun_out/r4v/tests.txt 2>&1
tail -4 gpurun_out/r4v/tests.txt; grep "CelebA B=64" gpurun_out/r4v/tests.txt
for rnd in 1 2; do timeout 600 python tests/perf_wgrad43.py > gpurun_out/r4v/wg43_$rnd.txt 2>&1; done
sed 's/F(2,3).*| F(4,3)/F43/' gpurun_out/r4v/wg43_2.txt | grep -v amdgpu.ids | cut -c1-125
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r4v/bench.json 2> gpurun_out/r4v/bench.err
python - <<PY
import json
b=json.loads(open("gpurun_out/r4v/bench.json").read().strip().splitlines()[-1])
print(b["ms_per_step"], b["value"], b["sampling"]["value"], b["secondary"]["ms_per_step"])
PY
