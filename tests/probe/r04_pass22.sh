#!/bin/bash
# final bench line of the round (reads the refreshed profiles/r04_*traffic.json), wgrad tests after the threshold change
mkdir -p gpurun_out/r4w
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_bench_shapes_gpu.py -q -m gpu -x -k "wgrad" > gpurun_out/r4w/wgrad_tests.txt 2>&1; tail -2 gpurun_out/r4w/wgrad_tests.txt
timeout 1500 python bench.py > gpurun_out/r4w/bench_n1.json 2> gpurun_out/r4w/bench_n1.err
python __graft_entry__.py smoke > gpurun_out/r4w/smoke.txt 2>&1; tail -1 gpurun_out/r4w/smoke.txt
python - <<PY
import json
b=json.loads([l for l in open("gpurun_out/r4w/bench_n1.json") if l.startswith("{")][-1])
print(b["ms_per_step"], b["value"], b["roofline"]["frac"], b["roofline"]["traffic"], b["sampling"]["value"], b["secondary"]["ms_per_step"], b["secondary"]["roofline"]["traffic"])
PY
