"""The N > 1 code path of bench.py on ONE MI355X (no 8-GPU node is available to the build): a 1-rank RCCL process group with
the gradient reducer forced on, so every train step issues the real bucketed all-reduces of the 243 MB CIFAR gradient buffer
(8 x 32 MiB buckets, launched from inside backward) and waits for them -- what the multi-GPU runs add on top of the 1-GPU
step except the wire time.  Reports reducer-on vs reducer-off step time and bounds the overhead.  (reference: DDP wrap
train.py:141-148; the 2-rank gloo test of the reducer itself is tests/test_multirank_cpu.py.)"""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(extra_env):
    env = dict(os.environ, **extra_env)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "6", "--warmup", "3", "--no-sample",
                        "--no-cpu-baseline", "--no-secondary"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    return json.loads(line)


def test_reducer_enabled_step_on_one_rank_rccl():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    plain = _bench({})
    forced = _bench({"VD_BENCH_FORCE_REDUCER": "1", "MASTER_PORT": "29541"})
    a, b = plain["ms_per_step"], forced["ms_per_step"]
    print(f"train step: reducer off {a:.2f} ms, reducer on (1-rank RCCL, 8 buckets of 32 MiB) {b:.2f} ms, overhead {100 * (b / a - 1):.1f} %")
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump({"ms_per_step_reducer_off": a, "ms_per_step_reducer_on": b, "overhead_frac": b / a - 1},
              open(os.path.join(ROOT, "gpurun_out", "reducer_overhead.json"), "w"))
    assert abs(forced["config"]["final_loss"] - plain["config"]["final_loss"]) < 1e-6          # a 1-rank sum is the identity
    assert b <= 1.08 * a + 1.0, f"bucketed all-reduce path costs {b - a:.2f} ms per step on one rank"
