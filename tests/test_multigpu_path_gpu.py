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
                        "--no-cpu-baseline", "--no-secondary", "--no-torch-baseline"], env=env, capture_output=True, text=True, timeout=900)
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
    # the self-check block every N > 1 line carries (round-2 review): backend, RCCL rank count, per-rank times, exposed all-reduce
    mg = forced["multi_gpu"]
    assert plain["multi_gpu"] is None and mg["backend"] == "nccl" and mg["rccl_ranks"] == 1, mg
    assert mg["buckets"] == 8 and mg["grad_bytes_per_step"] >= 4 * 60_806_403 and mg["bucket_bytes"] == 32 << 20, mg
    assert abs(mg["allreduce_exposed_ms"] - (b - mg["ms_per_step_no_allreduce"])) < 1e-2
    assert mg["ms_per_step_rank_min"] <= mg["ms_per_step_rank_max"] <= b + 0.5
    assert b <= 1.08 * a + 1.0, f"bucketed all-reduce path costs {b - a:.2f} ms per step on one rank"


def test_two_rank_launch_contract_on_one_gpu():
    """The driver's N > 1 command line (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N`)
    with N = 2 ranks sharing the one GPU of the test box; the collectives travel over gloo (RCCL refuses two ranks on one
    device), everything else is the code the 8-GPU runs execute: per-rank shards and RNG streams, bucketed gradient all-reduce
    issued from inside backward, loss reduce, barrier + MAX-over-ranks timing, one JSON line from rank 0."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = dict(os.environ, VD_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29547", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "32",
           "--no-sample", "--no-cpu-baseline", "--no-secondary", "--no-extras"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, f"expected ONE JSON line from rank 0, got {len(lines)}"
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 3 and j["warmup"] == 1 and j["scaling"] == "weak"
    assert j["config"]["global_batch"] == 64 and j["config"]["parallelism"] == "dp2"
    assert abs(j["value"] - 64 / (j["ms_per_step"] * 1e-3)) <= 1e-3 * j["value"]        # whole-job images/s = global batch / step time
    assert 0.0 < j["config"]["final_loss"] < 10.0


def test_two_ranks_equal_one_rank_on_the_full_batch(tmp_path):
    """End-to-end data-parallel equivalence (reference DDP semantics, train.py:141-148 + train_utils.py:151-169): two ranks, each with
    half of a fixed batch (rows [r B/2, (r+1) B/2), injected t / noise), run two HotPathTrainer.step updates -- bucketed gradient
    all-reduce launched from inside backward, 1 / world folded into the loss seed, clip + AdamW + EMA on the reduced gradient, loss
    reduce to the leader -- and must land on the parameters / EMA shadow of ONE rank stepping on the full batch, and report the same
    loss (the rank mean of shard means = the batch mean).  The ranks share the test box's one GPU; collectives over gloo.
    Tolerances: the all-reduced gradient differs from the one-rank gradient only by fp32 summation order (relative L2 <= 1e-6 of the
    buffer); parameters after two Adam steps at lr = 2e-4 without warm-up: relative L2 <= 1e-6 (Adam's first steps are sign-like, so a
    handful of near-zero gradient entries may flip: measured and bounded on the whole buffer, not per element)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    sys.path[:0] = [os.path.join(ROOT, "tests")]
    import dp_worker
    out = str(tmp_path / "dp2.pt")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29553", os.path.join(ROOT, "tests", "dp_worker.py"), "--out", out, "--batch", "16", "--steps", "2"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    two = torch.load(out)
    one = dp_worker.run(0, 1, 16, 2)
    assert two["reducer_active"] and not one["reducer_active"] and two["buckets"] == 8
    assert two["replicas_identical"], "the two replicas diverged"
    rel = lambda a, b: ((a.double() - b.double()).norm() / b.double().norm()).item()
    eg, ep, ee = rel(two["g1"] * 1.0, one["g1"]), rel(two["p"], one["p"]), rel(two["ema"], one["ema"])
    print(f"2 ranks x 8 rows vs 1 rank x 16 rows: gradient rel-L2 {eg:.2e}, parameters {ep:.2e}, EMA {ee:.2e}; losses {two['losses']} vs {one['losses']}")
    assert eg <= 1e-6 and ep <= 1e-6 and ee <= 1e-6
    # and the update itself: each element moved by ~lr per step, 1e-2 of a typical weight -- the displacement must agree too
    ed = rel(two["p"].double() - two["p0"].double(), one["p"].double() - one["p0"].double())
    assert torch.equal(two["p0"], one["p0"]) and ed <= 1e-3, f"displacement rel-L2 {ed:.2e}"
    for a, b in zip(two["losses"], one["losses"]):
        assert abs(a - b) <= 2e-6 * max(abs(b), 1.0), (two["losses"], one["losses"])


def test_num_accum_microbatches_equal_one_batch(tmp_path):
    """--num-accum (reference train.py:292; train_utils.py:154 ``loss.div(num_accum).backward()``, :257 ``step(..., update=(i + 1) %
    num_accum == 0)``): HotPathTrainer(num_accum=2) fed two micro-batches of B/2 rows -- the flat gradient buffer kept as a running
    sum, clip + AdamW + EMA with the second -- must land on the parameters / EMA shadow of num_accum=1 on the B rows at once, for one
    rank and for two data-parallel ranks (each rank's 8 rows in two micro-batches of 4; gradient buckets all-reduced per micro-batch as
    DDP does without no_sync).  Same tolerances as test_two_ranks_equal_one_rank_on_the_full_batch."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    sys.path[:0] = [os.path.join(ROOT, "tests")]
    import dp_worker
    one = dp_worker.run(0, 1, 16, 2)
    acc = dp_worker.run(0, 1, 16, 2, accum=2)
    out = str(tmp_path / "dp2acc.pt")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29557", os.path.join(ROOT, "tests", "dp_worker.py"), "--out", out, "--batch", "16", "--steps", "2",
           "--accum", "2"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    two = torch.load(out)
    assert two["replicas_identical"], "the two replicas diverged"
    rel = lambda a, b: ((a.double() - b.double()).norm() / b.double().norm()).item()
    for tag, got in (("1 rank x 2 micro-batches of 8", acc), ("2 ranks x 2 micro-batches of 4", two)):
        eg, ep, ee = rel(got["g1"], one["g1"]), rel(got["p"], one["p"]), rel(got["ema"], one["ema"])
        ed = rel(got["p"].double() - got["p0"].double(), one["p"].double() - one["p0"].double())
        print(f"{tag} vs 1 x 16 rows: gradient rel-L2 {eg:.2e}, parameters {ep:.2e}, EMA {ee:.2e}, displacement {ed:.2e}; "
              f"losses {got['losses']} vs {one['losses']}")
        assert eg <= 1e-6 and ep <= 1e-6 and ee <= 1e-6 and ed <= 1e-3
        for a, b in zip(got["losses"], one["losses"]):
            assert abs(a - b) <= 2e-6 * max(abs(b), 1.0), (got["losses"], one["losses"])


def test_two_ranks_mixing_labelled_and_unlabelled_steps(tmp_path):
    """Round-5 advisor: whether the class embedding is updated is ONE decision for all replicas -- a device flag behind the gradients, summed
    over ranks with the last bucket and read by vd_adamw_ema_flagged.  Two ranks over three updates: (rank 0 labelled, rank 1 y = None),
    (both None), (rank 0 None, rank 1 labelled).  The replicas must stay identical, both ranks must count the same class-embedding
    updates (2 of 3), and parameters / EMA must equal ONE rank that runs the same shards as two micro-batches (num_accum = 2: gradients
    summed, the flag summed with them) -- the removed per-update gloo exchange existed for exactly this case."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    sys.path[:0] = [os.path.join(ROOT, "tests")]
    import dp_worker
    plan = ["LN", "NN", "NL"]
    one = dp_worker.run(0, 1, 16, 3, accum=2, plan=plan)
    out = str(tmp_path / "dp2mixed.pt")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29559", os.path.join(ROOT, "tests", "dp_worker.py"), "--out", out, "--batch", "16", "--steps", "3",
           "--plan", ",".join(plan)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    two = torch.load(out)
    assert two["replicas_identical"], "the two replicas diverged"
    assert two["cls_steps_per_rank"] == [2, 2] and one["cls_steps"] == 2 and two["steps"] == one["steps"] == 3, (two["cls_steps_per_rank"], one["cls_steps"])
    rel = lambda a, b: ((a.double() - b.double()).norm() / b.double().norm()).item()
    ep, ee = rel(two["p"], one["p"]), rel(two["ema"], one["ema"])
    ed = rel(two["p"].double() - two["p0"].double(), one["p"].double() - one["p0"].double())
    print(f"2 ranks (labelled / None mixed over 3 updates) vs 1 rank x 2 micro-batches: parameters rel-L2 {ep:.2e}, EMA {ee:.2e}, displacement {ed:.2e}")
    assert ep <= 1e-6 and ee <= 1e-6 and ed <= 1e-3
    for a, b in zip(two["losses"], one["losses"]):
        assert abs(a - b) <= 2e-6 * max(abs(b), 1.0), (two["losses"], one["losses"])


def test_two_rank_rccl_launch_when_two_gpus_are_visible():
    """The same launch line over RCCL, one rank per GPU -- runs wherever the box has at least two devices (the build's own boxes
    have one: skipped there; the driver's multi-GPU node exercises it without further work).  Checks the contract line AND the
    self-check block: two RCCL ranks, the real bucket plan, per-rank step times, exposed all-reduce time."""
    if not torch.cuda.is_available() or torch.cuda.device_count() < 2:
        pytest.skip("needs two visible GPUs (RCCL refuses two ranks on one device)")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("VD_BENCH_BACKEND", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29549", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2",
           "--no-sample", "--no-cpu-baseline", "--no-secondary", "--no-extras"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["config"]["global_batch"] == 256 and j["config"]["parallelism"] == "dp2"
    mg = j["multi_gpu"]
    assert mg["backend"] == "nccl" and mg["rccl_ranks"] == 2 and mg["devices_visible"] >= 2, mg
    assert mg["buckets"] == 8 and mg["ms_per_step_rank_min"] > 0
    # the all-reduce of 243 MB over xGMI must hide behind backward: well under 10 % of the step exposed
    assert mg["allreduce_exposed_ms"] <= 0.10 * j["ms_per_step"] + 1.0, mg
    print("2-rank RCCL:", json.dumps(mg))


def test_groupnorm_sibling_spin_under_co_resident_kernels():
    """Round-5 review item 5 / advisor: the SPLIT form of the GroupNorm backward (csrc/norm.hip: sibling workgroups of one image slab meet
    at an agent-scope counter and spin) must stay correct when OTHER kernels are co-resident -- the persistent convolutions of the side
    stream and, on N > 1 ranks, RCCL's collective kernels launched from inside backward.  No multi-GPU node is available to the build, so
    the collectives come from a 1-rank RCCL group (the real bucketed all-reduce of the 1.07 GB CelebA gradient buffer, launched mid-backward).
    50 CelebA train steps at B = 64 (the per-rank batch of BASELINE configs[4]: every 64x64 GroupNorm backward takes the SPLIT form) from fixed
    weights / inputs, dropout off, in fresh processes on the PROBE library (it exports the longest sibling wait):
      quiet: no reducer;   busy: reducer on, VD_RESERVE_CUS = 0;   quiet8 / busy8: the same pair with 8 CUs reserved for the collectives.
    Parameters and EMA shadow after 50 updates must be BITWISE those of the quiet run (a 1-rank sum is the identity; any torn or late
    sibling sum would change them), every loss finite, and no sibling may wait longer than 1 % of the spin bound (2^22 iterations)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    probe = os.path.join(ROOT, "v-diffusion-torch_amd", "lib", "libvdiff_hip_probe.so")
    assert os.path.exists(probe), "probe library missing: make -C v-diffusion-torch_amd/csrc"

    def run(extra):
        env = dict(os.environ, VDIFF_HIP_LIB=probe, VD_SOAK_STEPS="50", **extra)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "soak_gn_split.py")], env=env, capture_output=True, text=True, timeout=1500)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
        return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])

    quiet = run({})
    busy = run({"VD_SOAK_REDUCER": "1", "MASTER_PORT": "29573"})
    # (reserving CUs changes the split-K slab plans of the weight gradients -- whole residency rounds of the CUs in use -- hence the rounding:
    #  the reserved-CU run is compared with a quiet run under the same reservation)
    quiet8 = run({"VD_RESERVE_CUS": "8"})
    busy8 = run({"VD_SOAK_REDUCER": "1", "VD_RESERVE_CUS": "8", "MASTER_PORT": "29575"})
    for name, j in (("quiet", quiet), ("busy", busy), ("quiet8", quiet8), ("busy8", busy8)):
        assert j["finite"] and all(l == l and abs(l) < 1e3 for l in j["losses"]), (name, j["losses"][:5])
        assert j["max_spin"] is not None and j["max_spin"] < (1 << 22) // 100, f"{name}: a sibling waited {j['max_spin']} spin iterations"
    assert busy["reducer"] and busy8["reducer"] and busy8["reserved_cus"] == 8 and quiet8["reserved_cus"] == 8 and not quiet["reducer"]
    assert busy["params"] == quiet["params"] and busy["ema"] == quiet["ema"], "reducer on: parameters differ from the quiet run"
    assert busy8["params"] == quiet8["params"] and busy8["ema"] == quiet8["ema"], "reducer on, 8 CUs reserved: parameters differ from the quiet run"
    assert busy["losses"] == quiet["losses"] and busy8["losses"] == quiet8["losses"]
    print(f"50 CelebA steps (B = 64): longest sibling wait quiet {quiet['max_spin']}, reducer on {busy['max_spin']}, reducer on + 8 reserved CUs "
          f"{busy8['max_spin']} spin iterations (bound 4 194 304); parameters and EMA bitwise equal")
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump({"max_spin": {"quiet": quiet["max_spin"], "reducer": busy["max_spin"], "quiet_reserve8": quiet8["max_spin"],
                            "reducer_reserve8": busy8["max_spin"]}, "spin_bound": 1 << 22,
               "steps": 50, "batch": 64, "bitwise_equal": True}, open(os.path.join(ROOT, "gpurun_out", "gn_split_soak.json"), "w"))
