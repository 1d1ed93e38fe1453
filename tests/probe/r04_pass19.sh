#!/bin/bash
# split-operand forms on every GEMM kind: full GPU suite under VD_GEMM_SPLIT=1, per-shape table and train step A/B
mkdir -p gpurun_out/r4u
VD_GEMM_SPLIT=1 timeout 1500 python -m pytest tests -q -m gpu -x > gpurun_out/r4u/suite_split1.txt 2>&1
tail -5 gpurun_out/r4u/suite_split1.txt
for s in 0 1; do
  VD_GEMM_SPLIT=$s timeout 300 python tests/probe/gemm_shapes.py cifar10 > gpurun_out/r4u/shapes_split$s.txt 2>&1
done
for rnd in 1 2; do
  for s in 0 1; do
    VD_GEMM_SPLIT=$s timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r4u/bench_split${s}_$rnd.json 2> gpurun_out/r4u/bench_split${s}_$rnd.err
    python - <<PY
import json
try:
    b=json.loads(open("gpurun_out/r4u/bench_split${s}_$rnd.json").read().strip().splitlines()[-1])
    print("split $s round $rnd:", b["ms_per_step"], b["value"], b["sampling"]["value"], b["secondary"]["ms_per_step"], b["secondary"].get("sampling",{}).get("value"))
except Exception as e: print("split $s round $rnd: failed", e)
PY
  done
done
