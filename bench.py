#!/usr/bin/env python
"""bench.py -- headline benchmark of the hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1], SURVEY 8d "C2"): CIFAR-10 32x32 class-conditional v-prediction UNet
(configs/cifar10_cond.json: 60.8 M parameters, dropout 0.2, cosine log-SNR schedule, snr_trunc loss), per-GPU batch 128,
synthetic data of that shape, random weights of that architecture (zero-initialised tensors re-randomised so no path hides
behind zeros).  A "step" = one reference ``Trainer.step``: t/noise draw, q_sample, UNet forward, v-loss, backward,
gradient mean over ranks (RCCL, bucketed, overlapped), global-norm clip, AdamW, LR warm-up, EMA.
``value`` = global_batch / step time, inputs resident in HBM, barrier + synchronize on both sides, MAX over ranks.

Extra objects on the same JSON line: ``roofline`` (dominant kernel, live HIP-event timing; ``achieved`` / ``frac`` count the MFMA
FLOPs the kernel EXECUTES, so frac <= 1; the algorithmic rate of a Winograd kernel sits beside it), ``cpu_baseline`` (the CPU
oracle on the host cores, rank 0 at N=1 only), ``sampling`` (DDIM-50 + CFG w=1 images/s, the second half of the metric),
``multi_gpu`` (N > 1: backend, rank count, per-rank step times, exposed all-reduce time), ``secondary`` (BASELINE configs[3]:
CelebA 64x64 train step + DDIM-50 CFG sampling + its own cpu_baseline).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "v-diffusion-torch_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch
import torch.distributed as dist

PEAK_FP32_MFMA_TFLOPS = 157.3          # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD
PEAK_BF16_MFMA_TFLOPS = 2517.0         # MI355X_MICROARCH.md: dense bf16 MFMA
# ceiling of the split-operand GEMM forms (gemm_split_kernel): every fp32 product = six bf16 piece products on the 16-bit matrix cores
PEAK_SPLIT_TFLOPS = round(PEAK_BF16_MFMA_TFLOPS / 6.0, 1)


def kernel_peak(kernel_name):
    """the matrix-core ceiling a kernel is priced against, in fp32-equivalent TFLOP/s: 157.3 for the fp32 MFMA instructions, dense bf16 / 6
    = 419.5 for the split-operand (bf16x3) forms of the tile engine"""
    return PEAK_SPLIT_TFLOPS if ("gemm_split" in kernel_name or "planes256" in kernel_name) else PEAK_FP32_MFMA_TFLOPS
CIFAR = dict(in_channels=3, hid_channels=256, out_channels=3, ch_multipliers=[1, 1, 1], num_res_blocks=3,
             apply_attn=[False, True, True], drop_rate=0.2, num_heads=1, num_classes=10, multitags=False)
FWD_GFLOP_PER_IMG = 37.64              # matmul-class FLOPs of one CIFAR UNet forward
# configs/celeba.json merged with defaults.json (num_heads=1) and --model-out-type v (BASELINE configs[3], parity-test /
# secondary workload: `--config celeba`)
CELEBA = dict(in_channels=3, hid_channels=192, out_channels=3, ch_multipliers=[1, 2, 3, 4], num_res_blocks=3,
              apply_attn=[False, True, True, True], embedding_dim=768, drop_rate=0.1, head_dim=64, num_heads=1,
              num_classes=40, multitags=True)
CELEBA_FWD_GFLOP_PER_IMG = 201.3


def build_model(device, seed=1234, cfg=None):
    import v_diffusion
    torch.manual_seed(seed)
    model = v_diffusion.UNet(**(cfg or CIFAR))
    with torch.no_grad():                 # re-randomise the zero-initialised tensors
        for name, p in model.named_parameters():
            if p.ndim >= 2 and float(p.abs().max()) == 0.0:
                fan_in = p[0].numel()
                p.normal_(0.0, fan_in ** -0.5)
    return model.to(device)


def usable_cpus(cap=64):
    """CPUs this process may really use: affinity mask and cgroup quota, capped (torch CPU kernels stop scaling and
    start thrashing far below the 256 hardware threads of the GPU box)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, cap))


def cpu_baseline(wl="cifar10", batch=16, budget_s=18.0):
    """The CPU oracle (oracle/, proved equal to the reference by the golden fixtures) doing the same train step on the
    host cores: bounded sample (one warm-up step, then timed steps until ~budget_s of CPU work),
    reported beside the GPU number, never the thing measured."""
    from oracle import unet_ref, diffusion_ref as dref
    from oracle.unet_ref import param_shapes
    W = WORKLOADS[wl]
    cfg, res = W["cfg"], W["res"]
    ncpu = usable_cpus()
    torch.set_num_threads(ncpu)
    g = torch.Generator().manual_seed(0)
    sd = {}
    for k, shp in param_shapes(cfg).items():
        fan = max(int(torch.tensor(shp[1:]).prod()) if len(shp) > 1 else 1, 1)
        sd[k] = (torch.randn(shp, generator=g) * fan ** -0.5).requires_grad_(True)
    x0 = torch.rand((batch, 3, res, res), generator=g) * 2 - 1
    if cfg.get("multitags"):
        y = (torch.rand((batch, cfg["num_classes"]), generator=g) < 0.2).float()
    else:
        y = torch.randint(1, cfg["num_classes"] + 1, (batch,), generator=g).float()
    den = lambda a, b, c: unet_ref.unet_forward(sd, cfg, a, b, c, train=True)
    sched = dref.make_schedule("cosine")
    times, t_start = [], time.perf_counter()
    while True:                           # first pass warms the allocator / thread pool and is not counted
        t = torch.rand((batch,), dtype=torch.float64, generator=g)
        noise = torch.randn(x0.shape, generator=g)
        t0 = time.perf_counter()
        loss = dref.train_loss(den, sched, x0, t, y, noise, "v", "snr_trunc").mean()
        loss.backward()
        times.append(time.perf_counter() - t0)
        for v in sd.values():
            v.grad = None
        spent = time.perf_counter() - t_start
        if len(times) >= 2 and (spent + times[-1] > budget_s or len(times) >= 9):
            break
        if len(times) == 1 and times[0] > budget_s:      # very slow host: the cold step is all the budget allows
            break
    timed = sorted(times[1:] or times)
    med = timed[len(timed) // 2]
    try:
        model_name = [l.split(":")[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        model_name = "unknown"
    return {"value": round(batch / med, 3), "unit": "images/s", "cores": ncpu, "kind": "port",
            "sample": f"{W['short']} train step (q_sample+fwd+v-loss+bwd, no optimizer), batch {batch}, "
                      f"median of {len(timed)} timed step(s) after {'1 warm-up' if len(times) > 1 else 'no warm-up (bounded)'}, "
                      f"torch CPU fp32 on {ncpu} threads, {model_name}"}


def torch_rocm_baseline(wl="cifar10", batch=128, steps=3, warmup=2, device="cuda"):
    """BASELINE LEG ONLY (after the timed region, like cpu_baseline; never a target, never in the product path): the same train-step
    math -- q_sample + UNet forward + snr_trunc v-loss + backward, no optimizer -- through STOCK PyTorch-ROCm on the same MI355X: the
    oracle's functional restatement (oracle/unet_ref.py, proved equal to the reference by the goldens) moved to the device, i.e. ATen
    convolutions on MIOpen, rocBLAS / hipBLASLt GEMMs, autograd backward; fp32 with TF32-class paths off (train.py:233-237 resolves
    allow_tf32=False on a non-NVIDIA device name).  It stands in for the number BASELINE.json.published lacks ("reference-GPU
    images/sec"): what the reference's own code achieves on this GPU.  Batch 128 (64 on an out-of-memory error)."""
    from oracle import unet_ref, diffusion_ref as dref
    from oracle.unet_ref import param_shapes
    W = WORKLOADS[wl]
    cfg, res = W["cfg"], W["res"]
    torch.backends.cuda.matmul.allow_tf32 = False
    torch.backends.cudnn.allow_tf32 = False
    # train.py:237 sets cudnn.benchmark (MIOpen then times every solver per shape: 774 img/s here, but 6.5 minutes of search on a fresh
    # box -- profiles/r05_torch_baseline_modes.txt); the default bench run must finish within minutes, so the baseline leg runs MIOpen's
    # immediate mode (758 img/s: 2 % less) unless VD_TORCH_BASELINE_BENCHMARK=1
    bench_mode = os.environ.get("VD_TORCH_BASELINE_BENCHMARK", "0") != "0"
    torch.backends.cudnn.benchmark = bench_mode
    g = torch.Generator().manual_seed(0)
    sd = {}
    for k, shp in param_shapes(cfg).items():
        fan = max(int(torch.tensor(shp[1:]).prod()) if len(shp) > 1 else 1, 1)
        sd[k] = (torch.randn(shp, generator=g) * fan ** -0.5).to(device).requires_grad_(True)
    sched = dref.make_schedule("cosine")
    den = lambda a, b, c: unet_ref.unet_forward(sd, cfg, a, b, c, train=True)
    B = batch
    while True:
        try:
            x0 = (torch.rand((B, 3, res, res), generator=g) * 2 - 1).to(device)
            if cfg.get("multitags"):
                y = (torch.rand((B, cfg["num_classes"]), generator=g) < 0.2).float().to(device)
            else:
                y = torch.randint(1, cfg["num_classes"] + 1, (B,), generator=g).float().to(device)
            times = []
            for i in range(warmup + steps):
                t = torch.rand((B,), dtype=torch.float64, generator=g).to(device)
                noise = torch.randn(x0.shape, generator=g).to(device)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                loss = dref.train_loss(den, sched, x0, t, y, noise, "v", "snr_trunc").mean()
                loss.backward()
                torch.cuda.synchronize()
                times.append(time.perf_counter() - t0)
                for v in sd.values():
                    v.grad = None
            break
        except torch.OutOfMemoryError:
            if B <= 16:
                raise
            B //= 2
            torch.cuda.empty_cache()
    timed = sorted(times[warmup:])
    med = timed[len(timed) // 2]
    return {"value": round(B / med, 2), "unit": "images/s", "kind": "port on ATen/MIOpen", "baseline_only": True,
            "ms_per_step": round(med * 1e3, 2), "batch": B, "finite": bool(torch.isfinite(loss)),
            "sample": f"{W['short']} train step WITHOUT optimizer (q_sample+fwd+v-loss+autograd bwd), batch {B}, median of {steps} steps "
                      f"after {warmup} warm-ups, stock torch {torch.__version__} fp32 (TF32 off, MIOpen "
                      f"{'benchmark mode' if bench_mode else 'immediate mode; benchmark mode measured +2 % at 6.5 min of solver search'}), same GPU"}


# fwd_exec_frac: share of the ALGORITHMIC forward FLOPs the matrix cores execute -- FALLBACK ONLY (non-zero ranks, which record no launches):
# the sampling roofline takes the share from the launches two reverse steps of the run record (sample_once).  These constants are the
# all-F(2x2,3x3) values: CIFAR (32.61 * 4/9 + 0.03 + 5.00) / 37.64; CelebA (183.71 * 4/9 + 0.09 + 17.50) / 201.3
WORKLOADS = {
    "cifar10": dict(cfg=CIFAR, res=32, fwd_gflop=FWD_GFLOP_PER_IMG, fwd_exec_frac=0.5187, short="CIFAR-10 cond UNet",
                    name="CIFAR-10 32x32 class-cond v-pred UNet (cifar10_cond.json, 60.8M params)"),
    "celeba": dict(cfg=CELEBA, res=64, fwd_gflop=CELEBA_FWD_GFLOP_PER_IMG, fwd_exec_frac=0.4930, short="CelebA 64x64 UNet (merged config)",
                   name="CelebA 64x64 multitag v-pred UNet (celeba.json+defaults.json, 266.8M params)"),
}
PEAK_HBM_TBS = 8.0                     # MI355X_MICROARCH.md: HBM3E spec peak (6.3 TB/s measured for a streaming copy)


def traffic_table(wl="cifar10"):
    """HBM bytes per launch from the committed PMC passes of THIS workload (profiles/parse_rocprof.py; separate --pmc passes cannot run
    inside the timed bench, so the figure is a tracked measurement of the same command and says which file it came from): newest round
    first; a workload without a PMC pass of its own gets no table (its `traffic` is null, never another workload's number)"""
    names = {"cifar10": ("r06_traffic.json", "r05_traffic.json", "r04_traffic.json", "r03_traffic.json", "r02_traffic.json", "r01_traffic.json"),
             "celeba": ("r06_celeba_traffic.json", "r05_celeba_traffic.json", "r04_celeba_traffic.json")}[wl]
    for name in names:
        try:
            return name, json.load(open(os.path.join(ROOT, "profiles", name)))["kernels"]
        except Exception:
            continue
    return None, {}


def executed_share(kernel_name):
    """MFMA FLOPs a kernel executes per ALGORITHMIC FLOP recorded for it (v_diffusion/_hip.py records 2*M*N*K of the op each
    launch implements): Winograd F(2x2,3x3) kernels 4/9, the F(4x4,3x3) input gradient 1/4; the fused attention backward recomputes the logits in both of its
    kernels and dP in the second (7 products for the 4 the op defines: 7/4); everything else 1"""
    if kernel_name.startswith("wino43_") or kernel_name.endswith("weight gradient]"):
        return 1.0 / 4.0                      # F(4x4,3x3): 36 multiplies per 4x4 output tile where the convolution defines 144 (the weight
                                              # gradient's 36 planes run on the grouped tile engine, tagged by v_diffusion/_hip.py)
    if kernel_name.startswith("wino_"):
        return 4.0 / 9.0
    if kernel_name.startswith("attn_bwd_"):
        return 7.0 / 4.0
    return 1.0


def sample_once(diffusion, model, labels, SB, RES, T, W, device, rank, world, barrier):
    """DDIM-T + classifier-free guidance over SB images per GPU (2*SB UNet rows per reverse step): images/s + roofline"""
    model.eval()
    lab = labels[:SB].clone()
    diffusion.p_sample(model, (8, 3, RES, RES), label=lab[:8], device=device, seed=131071 + rank, use_ddim=True)   # warm-up
    barrier()
    t0 = time.perf_counter()
    out = diffusion.p_sample(model, (SB, 3, RES, RES), label=lab, device=device, seed=131071 + rank, use_ddim=True)
    barrier()
    ds = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([ds], device=device, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        ds = float(tmax.item())
    # share of the algorithmic FLOPs the matrix cores execute in THIS sampler, from the launches two reverse steps record (each launch
    # carries the 2*M*N*K of the op it implements; executed_share() says what its kernel really multiplies)
    exec_frac, rec_gflop, dominant, table = W["fwd_exec_frac"], None, None, None
    if rank == 0:
        from v_diffusion import _hip
        xt = torch.randn((SB, 3, RES, RES), device=device)
        _hip.PROFILE = []
        with torch.inference_mode():
            for st in (T - 1, T // 2):
                xt = diffusion.p_sample_step(model, xt, torch.full((SB,), st, device=device), lab.clone(), use_ddim=True)
        torch.cuda.synchronize()
        rec, _hip.PROFILE = _hip.PROFILE, None
        fl = sum(r[1] for r in rec if not r[0].startswith("hbm:"))
        fe = sum(r[1] * executed_share(r[0]) for r in rec if not r[0].startswith("hbm:"))
        if fl > 0:
            exec_frac, rec_gflop = fe / fl, fl / 2 / (2 * SB) / 1e9      # recorded algorithmic GFLOP per UNet row (fwd_gflop)
        agg = {}
        for name, work, e0, e1 in rec:
            if name.startswith("hbm:"):
                continue
            a = agg.setdefault(name, [0.0, 0.0, 0])
            a[0] += work; a[1] += e0.elapsed_time(e1) * 1e-3; a[2] += 1
        if agg:
            tot = sum(v[1] for v in agg.values())
            table = {k: {"tflops": round(executed_share(k) * v[0] / v[1] / 1e12, 2), "peak": kernel_peak(k),
                         "frac": round(executed_share(k) * v[0] / v[1] / 1e12 / kernel_peak(k), 4), "ms_per_reverse_step": round(v[1] / 2 * 1e3, 3),
                         "launches_per_reverse_step": v[2] // 2, "avg_launch_ms": round(v[1] / v[2] * 1e3, 4)}
                     for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:10] if v[1] > 0}
            dk = max(agg, key=lambda k: agg[k][1])
            dv = agg[dk]
            dominant = {"kernel": dk, "avg_launch_ms": round(dv[1] / dv[2] * 1e3, 4), "launches_per_reverse_step": dv[2] // 2,
                        "achieved": round(executed_share(dk) * dv[0] / dv[1] / 1e12, 2), "peak": kernel_peak(dk), "unit": "TFLOP/s",
                        "frac": round(executed_share(dk) * dv[0] / dv[1] / 1e12 / kernel_peak(dk), 4),
                        "algorithmic_tflops": round(dv[0] / dv[1] / 1e12, 2), "share_of_matmul_time": round(dv[1] / tot, 3)}
    model.train()
    alg = T * 2 * W["fwd_gflop"] * SB / ds / 1e3                      # algorithmic TFLOP/s (2 x T x forward)
    exe = alg * exec_frac
    tname = "r06_sampler_traffic.json" if RES == 32 else "r06_celeba_sampler_traffic.json"
    try:            # HBM bytes per launch of the dominant kernel from the committed PMC passes of the same sampler call (separate passes: not in the timed region)
        tj = json.load(open(os.path.join(ROOT, "profiles", tname)))["kernels"]
        if dominant is not None:
            key = dominant["kernel"].split(" [")[0].split(" (+")[0]
            dominant["traffic"] = tj[key]["hbm_bytes_per_launch"] if key in tj else None
            dominant["traffic_source"] = f"profiles/{tname} (builder box)"
    except Exception:
        tname = None
    return {"metric": f"ddim{T}_cfg_samples_per_sec", "value": round(world * SB / ds, 2), "unit": "images/s",
            "seconds_per_batch": round(ds, 3), "batch_per_gpu": SB, "unet_rows_per_step": 2 * SB, "w_guide": float(diffusion.w_guide),
            "roofline": {"bound": "mfma", "achieved": round(exe, 2), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(exe / PEAK_FP32_MFMA_TFLOPS, 4), "algorithmic_tflops": round(alg, 2),
                         "speedup_vs_direct_roofline": round(alg / PEAK_FP32_MFMA_TFLOPS, 4),
                         "frac_vs_direct_roofline": round(alg / PEAK_FP32_MFMA_TFLOPS, 4),
                         "executed_share_of_algorithmic_flops": round(exec_frac, 4),
                         "recorded_gflop_per_unet_row": None if rec_gflop is None else round(rec_gflop, 2),
                         "dominant_kernel": dominant, "top_kernels": table,
                         "kernel_table": ("profiles/r06_sampler_kernel_stats.csv" if RES == 32 else "profiles/r06_celeba_sampler_kernel_stats.csv") + " (rocprofv3 --kernel-trace --stats of tests/probe/sample_only.py, the same sampler call)",
                         "traffic_table": None if tname is None else f"profiles/{tname}",
                         "note": "achieved = MFMA FLOPs executed over the whole sampler call (wall time, launch gaps included): the share comes from the "
                                 "launches two reverse steps of this run record (F(4x4,3x3) convolutions execute 1/4 of their algorithmic FLOPs, F(2x2,3x3) "
                                 "ones 4/9); algorithmic_tflops = SURVEY 8d count; dominant_kernel / top_kernels: HIP events around every launch of those "
                                 "two steps, each kernel against its own ceiling (157.3 fp32 MFMA, 419.5 split-operand GEMM forms)"},
            "finite": bool(torch.isfinite(out).all())}


def run_training(wl, B, steps, warmup, device, rank, world, barrier, sample_steps, extras=True, uint8_input=False):
    """build the workload's model + trainer, time `steps` train steps, then the live per-kernel roofline of two more"""
    import v_diffusion
    from v_diffusion import _hip
    from v_diffusion.trainer import HotPathTrainer
    W = WORKLOADS[wl]
    celeba = wl == "celeba"
    model = build_model(device, cfg=W["cfg"])
    model.train()
    diffusion = v_diffusion.GaussianDiffusion(v_diffusion.get_logsnr_schedule("cosine", -20.0, 20.0), sample_steps, "v",
                                              "fixed_medium", "snr_trunc", "mse", intp_frac=0.3, w_guide=1.0, p_uncond=0.1)
    trainer = HotPathTrainer(model, diffusion, lr=2e-4, weight_decay=0.001, warmup=1000, grad_norm=1.0, ema_decay=0.9999,
                             use_ema=True, rank=rank, world_size=world)
    if os.environ.get("VD_BENCH_FORCE_REDUCER") and world == 1:
        # multi-GPU code path on one GPU (tests/test_multigpu_path_gpu.py): a 1-rank RCCL group, bucketed all-reduce of the real
        # gradient buffer switched on, so the bucket / launch overhead the N > 1 runs add is measurable without an 8-GPU node
        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            hp = os.environ.get("VD_RCCL_HIGH_PRIORITY", "0") != "0"
            dist.init_process_group("nccl", init_method="env://", world_size=1, rank=0,
                                    pg_options=dist.ProcessGroupNCCL.Options(is_high_priority_stream=True) if hp else None)
        trainer.reducer.active = True
    RES = W["res"]
    g = torch.Generator(device).manual_seed(4321 + rank)
    x = torch.rand((B, 3, RES, RES), device=device, generator=g) * 2 - 1                  # Normalize(0.5, 0.5) range
    if celeba:
        labels = (torch.rand((B, 40), device=device, generator=g) < 0.2).float()           # CelebA attribute tags
    else:
        labels = torch.randint(1, 11, (B,), device=device, generator=g).float()            # target_transform y+1

    if uint8_input:
        # --uint8-input: the batch as the dataset holds it before the reference's CPU transforms (datasets.py:111-126): uint8 HWC + flip
        # decisions, resident in HBM; every step starts with vd_images_from_uint8_hwc (flip + ToTensor + Normalize on the device)
        u8 = torch.randint(0, 256, (B, RES, RES, 3), device=device, generator=g, dtype=torch.uint8)
        flips = torch.rand((B,), device=device, generator=g) < 0.5

    def one_step():
        if uint8_input:
            return trainer.step_uint8(u8, labels.clone(), flip=flips)
        return trainer.step(x, labels.clone())          # y is mutated by the label drop: hand over a fresh copy

    for _ in range(warmup):
        one_step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = one_step()
    torch.cuda.synchronize()
    dt_local = time.perf_counter() - t0              # this rank's own time, before it waits for the slowest rank
    barrier()
    dt = time.perf_counter() - t0
    rank_ms = [dt_local / steps * 1e3]
    if world > 1:
        tmax = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        mine = torch.tensor([rank_ms[0]], device=device, dtype=torch.float64)
        parts = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine)
        rank_ms = [float(q.item()) for q in parts]
    ms_per_step = dt / steps * 1e3
    res = dict(model=model, diffusion=diffusion, labels=labels, ms_per_step=ms_per_step, value=world * B * steps / dt,
               final_loss=float(loss.item()), name=W["name"], res=RES, fwd_gflop=W["fwd_gflop"], W=W, rank_ms=rank_ms)

    # ---- SURVEY 5 metric variants (rank-local, a few steps each): the reference loop's per-step host sync, and fwd+bwd alone
    if extras:
        n = max(3, min(steps, 5))
        barrier(); t0 = time.perf_counter()
        for _ in range(n):
            float(one_step().item())                    # train_utils.py:169 reads loss.item() every step
        barrier()
        res["ms_per_step_with_loss_item"] = round((time.perf_counter() - t0) / n * 1e3, 3)
        tt, nn_ = trainer.draw(x)
        barrier(); t0 = time.perf_counter()
        for _ in range(n):
            diffusion.train_loss(model, x_0=x, t=tt.clone(), y=labels.clone(), noise=nn_).mean().backward()
        barrier()
        res["ms_fwd_bwd_only"] = round((time.perf_counter() - t0) / n * 1e3, 3)

    # ---- live roofline: HIP events around every matmul-shaped and GroupNorm launch of two extra steps
    if rank == 0:
        _hip.PROFILE = []
        _hip.PROFILE_BYTES.clear()
    for _ in range(2):                  # every rank takes part (the steps contain collectives); only rank 0 records
        one_step()
    barrier()
    roofline = None
    if rank == 0:
        rec, _hip.PROFILE = _hip.PROFILE, None
        agg, hbm = {}, {}
        for name, work, e0, e1 in rec:
            tgt = hbm if name.startswith("hbm:") else agg
            a = tgt.setdefault(name, [0.0, 0.0, 0])
            a[0] += work; a[1] += e0.elapsed_time(e1) * 1e-3; a[2] += 1
        total_t = sum(v[1] for v in agg.values())
        dom = max(agg, key=lambda k: agg[k][1])
        fl, tt_, n = agg[dom]
        tname, tj = traffic_table(wl)
        key = dom.split(" [")[0].split(" (+")[0]                             # the rocprof kernel name (tags of v_diffusion/_hip.py stripped)
        traffic = tj[key]["hbm_bytes_per_launch"] if key in tj else None
        ab = _hip.PROFILE_BYTES.get(dom)
        alg_bytes = ab[0] / ab[1] if ab else None
        # `achieved` / `frac` = what the matrix cores EXECUTE (<= peak by construction).  The Winograd kernels execute 4/9 of the
        # algorithmic FLOPs of the convolution they implement (2*M*N*K of the direct form); that algorithmic rate is
        # reported beside it as `algorithmic_tflops`, and its ratio to the peak as `speedup_vs_direct_roofline` (may exceed 1:
        # it is a speed-up over the best possible direct-convolution kernel, not a utilisation)
        exe = executed_share(dom)
        alg_tf = fl / tt_ / 1e12
        step_alg = sum(v[0] for v in agg.values()) / 2                      # algorithmic FLOPs of one step, as launched
        step_exe = sum(v[0] * executed_share(k) for k, v in agg.items()) / 2
        # the step's executed work by instruction class, each against its own ceiling: fp32 MFMA (Winograd convolutions, fused attention, 64-row
        # tiles) and the split-operand (bf16x3) forms of the tile engine; roofline_ms = the time the launches would take at their ceilings
        cls = {"fp32_mfma": [0.0, 0.0], "bf16x3_split": [0.0, 0.0]}
        for k, v in agg.items():
            c = cls["bf16x3_split" if ("gemm_split" in k or "planes256" in k) else "fp32_mfma"]
            c[0] += v[0] * executed_share(k) / 2; c[1] += v[1] / 2
        ideal_ms = sum(c[0] / (peak * 1e12) for c, peak in ((cls["fp32_mfma"], PEAK_FP32_MFMA_TFLOPS), (cls["bf16x3_split"], PEAK_SPLIT_TFLOPS))) * 1e3
        dom_peak = kernel_peak(dom)
        roofline = {"bound": "mfma", "kernel": dom, "achieved": round(exe * alg_tf, 2), "peak": dom_peak,
                    "unit": "TFLOP/s", "frac": round(exe * alg_tf / dom_peak, 4), "traffic": traffic,
                    "algorithmic_tflops": round(alg_tf, 2),
                    "speedup_vs_direct_roofline": round(alg_tf / PEAK_FP32_MFMA_TFLOPS, 4),
                    "frac_vs_direct_roofline": round(alg_tf / PEAK_FP32_MFMA_TFLOPS, 4),      # SURVEY 8d's definition: algorithmic FLOPs / time / peak
                    "executed_share_of_algorithmic_flops": round(exe, 4),
                    "algorithmic_bytes_per_launch": None if alg_bytes is None else round(alg_bytes),
                    "traffic_ratio": None if (alg_bytes is None or traffic is None) else round(traffic / alg_bytes, 3),
                    "note": (f"Winograd kernel: the matrix cores execute {exe:.4f} of the direct convolution's FLOPs (F(2x2,3x3): 4/9, F(4x4,3x3): 1/4; "
                             "exact fp32 products); achieved / frac count executed FLOPs, algorithmic_tflops / frac_vs_direct_roofline the "
                             "direct convolution's 2*M*N*K (SURVEY 8d)") if exe < 1 else None,
                    "traffic_note": (f"HBM+fabric bytes per launch of this kernel in the same bench command, PMC passes (FETCH_SIZE, WRITE_SIZE: "
                                     f"they cannot run inside the timed region) tracked as profiles/{tname}; traffic_ratio = traffic / "
                                     f"algorithmic bytes (x + y + residual + U read / written once)") if tname else None,
                    "clock_note": ("peak = 2.4 GHz figure; in-kernel clock (s_memtime / s_memrealtime beside the running kernel, "
                                   "profiles/r05_clock_by_kernel.txt): 2.37 GHz under the Winograd convolution kernels, 2.27-2.28 GHz under the "
                                   "tile-engine GEMMs, 2.33 GHz over the whole train step"),
                    "peak_note": ("every kernel is priced against its own ceiling: 157.3 TFLOP/s for the fp32 MFMA instructions, dense bf16 / 6 = "
                                  f"{PEAK_SPLIT_TFLOPS} for the split-operand GEMM forms (three bf16 pieces per fp32 operand, six products on the 16-bit "
                                  "matrix cores; they run power-limited at 1.88 GHz, profiles/r05_clock_split.txt)"),
                    "launches_per_step": n // 2, "avg_launch_ms": round(tt_ / n * 1e3, 4),
                    "flops_per_launch": round(exe * fl / n / 1e9, 3), "flops_unit": "GFLOP executed on the matrix cores per launch",
                    "algorithmic_flops_per_launch": round(fl / n / 1e9, 3),
                    "share_of_matmul_time": round(tt_ / total_t, 3),
                    "all_matmul_kernels": {k: {"tflops": round(executed_share(k) * v[0] / v[1] / 1e12, 2), "peak": kernel_peak(k),
                                               "frac": round(executed_share(k) * v[0] / v[1] / 1e12 / kernel_peak(k), 4),
                                               "algorithmic_tflops": round(v[0] / v[1] / 1e12, 2), "ms_per_step": round(v[1] / 2 * 1e3, 2),
                                               "launches_per_step": v[2] // 2} for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])
                                           if v[0] > 0 or v[1] / 2 * 1e3 >= 0.05},
                    "whole_step": {"frac": round(ideal_ms / ms_per_step, 4),
                                   "frac_note": "time the step's matmul launches would take at their own ceilings / measured step time (<= 1 by construction)",
                                   "roofline_ms": round(ideal_ms, 2),
                                   "fp32_mfma": {"executed_gflop_per_step": round(cls["fp32_mfma"][0] / 1e9, 1), "ms_in_kernels": round(cls["fp32_mfma"][1] * 1e3, 2),
                                                 "tflops_in_kernels": round(cls["fp32_mfma"][0] / max(cls["fp32_mfma"][1], 1e-9) / 1e12, 2),
                                                 "peak": PEAK_FP32_MFMA_TFLOPS,
                                                 "frac_in_kernels": round(cls["fp32_mfma"][0] / max(cls["fp32_mfma"][1], 1e-9) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4)},
                                   "bf16x3_split": {"executed_gflop_per_step": round(cls["bf16x3_split"][0] / 1e9, 1),
                                                    "ms_in_kernels": round(cls["bf16x3_split"][1] * 1e3, 2),
                                                    "tflops_in_kernels": round(cls["bf16x3_split"][0] / max(cls["bf16x3_split"][1], 1e-9) / 1e12, 2),
                                                    "peak": PEAK_SPLIT_TFLOPS,
                                                    "frac_in_kernels": round(cls["bf16x3_split"][0] / max(cls["bf16x3_split"][1], 1e-9) / 1e12 / PEAK_SPLIT_TFLOPS, 4)},
                                   "mfma_executed_tflops": round(step_exe / (ms_per_step * 1e-3) / 1e12, 2),
                                   "algorithmic_tflops": round(3 * W["fwd_gflop"] * B / (ms_per_step * 1e-3) / 1e3, 2),
                                   "speedup_vs_direct_roofline": round(3 * W["fwd_gflop"] * B / (ms_per_step * 1e-3) / 1e3 / PEAK_FP32_MFMA_TFLOPS, 4),
                                   "launched_algorithmic_gflop_per_step": round(step_alg / 1e9, 1)}}
        if hbm:     # the dominant HBM-bound kernel: algorithmic bytes (operands read + written once) / live time / 8 TB/s
            hd = max(hbm, key=lambda k: hbm[k][1])
            by, th, nh = hbm[hd]
            hk = hd[4:]
            roofline["hbm"] = {"bound": "hbm", "kernel": hk, "achieved": round(by / th / 1e9, 1), "peak": PEAK_HBM_TBS * 1e3,
                               "unit": "GB/s", "frac": round(by / th / 1e12 / PEAK_HBM_TBS, 4),
                               "traffic": tj[hk]["hbm_bytes_per_launch"] if hk in tj else None,
                               "traffic_ratio": round(tj[hk]["hbm_bytes_per_launch"] / (by / nh), 3) if hk in tj else None,
                               "traffic_source": f"profiles/{tname}" if (tname and hk in tj) else None,
                               "launches_per_step": nh // 2, "avg_launch_ms": round(th / nh * 1e3, 4),
                               "bytes_per_launch": round(by / nh), "ms_per_step": round(th / 2 * 1e3, 3),
                               "all_hbm_kernels": {k[4:]: {"gbs": round(v[0] / v[1] / 1e9, 1), "ms_per_step": round(v[1] / 2 * 1e3, 3),
                                                           "launches_per_step": v[2] // 2} for k, v in sorted(hbm.items(), key=lambda kv: -kv[1][1])}}
    res["roofline"] = roofline

    # ---- multi-GPU self-check (no 8-GPU node was available to the build: the first N > 1 run explains itself): the backend
    # really in use and its rank count, per-rank step times, and how much of the gradient all-reduce is NOT hidden behind
    # backward = step time with the bucketed all-reduce minus step time with the reducer switched off.  Run last: without the
    # all-reduce the replicas' weights drift apart, which no later measurement depends on.
    mg = None
    if world > 1 or trainer.reducer.active:
        n = max(3, min(steps, 5))
        was = trainer.reducer.active
        trainer.reducer.active = False
        one_step(); barrier(); t0 = time.perf_counter()
        for _ in range(n):
            one_step()
        barrier()
        ms_off = (time.perf_counter() - t0) / n * 1e3
        trainer.reducer.active = was
        if world > 1:
            tmax = torch.tensor([ms_off], device=device, dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            ms_off = float(tmax.item())
        # where inside backward every bucket is handed to the collective (events on the compute stream, one traced step)
        trainer.reducer.trace = []
        one_step()
        pts = trainer.reducer.launch_points_ms()
        trainer.reducer.trace = None
        barrier()
        mg = {"backend": dist.get_backend() if dist.is_initialized() else None,
              "rccl_ranks": dist.get_world_size() if (dist.is_initialized() and dist.get_backend() == "nccl") else 0,
              "devices_visible": torch.cuda.device_count(),
              "ms_per_step_rank_min": round(min(rank_ms), 3), "ms_per_step_rank_max": round(max(rank_ms), 3),
              "ms_per_step_no_allreduce": round(ms_off, 3), "allreduce_exposed_ms": round(ms_per_step - ms_off, 3),
              "grad_bytes_per_step": 4 * trainer.flat.numel, "buckets": len(trainer.reducer.bounds),
              "bucket_bytes": 4 * (trainer.reducer.bounds[0][1] - trainer.reducer.bounds[0][0]),
              "reserved_cus": int(_hip.lib().vd_reserved_cus()), "ready_per_block": bool(_hip.READY_PER_BLOCK),
              "bucket_launch_ms_after_backward_start": pts[:-1] if pts else None,
              "backward_ms": pts[-1] if pts else None}
    res["multi_gpu"] = mg
    res["trainer"] = trainer
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=128, help="per-GPU batch (BASELINE config: 128)")
    ap.add_argument("--no-sample", action="store_true", help="skip the DDIM-50 CFG sampling measurement")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the CelebA (BASELINE configs[3]) secondary block")
    ap.add_argument("--no-extras", action="store_true", help="skip the loss.item() / fwd+bwd-only variants (profiling runs: keeps the "
                    "launch count at warmup + steps + 2 train steps)")
    ap.add_argument("--uint8-input", action="store_true", help="feed every train step a uint8 HWC batch + flip mask (the dataset's format "
                    "before the reference's CPU transforms, datasets.py:111-126) through HotPathTrainer.step_uint8")
    ap.add_argument("--no-torch-baseline", action="store_true", help="skip the stock PyTorch-ROCm baseline leg (torch_rocm_baseline)")
    ap.add_argument("--no-fp32-ab", action="store_true", help="skip the fp32-MFMA A/B leg (a fresh child process with VD_GEMM_SPLIT=0 after the timed region)")
    ap.add_argument("--no-calibration", action="store_true", help="skip the 1.5 s MFMA calibration loop in front of the timed region")
    ap.add_argument("--sample-steps", type=int, default=50)
    ap.add_argument("--config", choices=["cifar10", "celeba"], default="cifar10",
                    help="cifar10 = the headline workload (BASELINE configs[1]); celeba = configs[3] as the primary line")
    args = ap.parse_args()
    wl = args.config

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world == 1:
        sys.exit("bench.py --gpus N>1 must be launched with `python -m torch.distributed.run --nproc-per-node N`")
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    dev_index = local_rank % torch.cuda.device_count()     # (== local_rank on a real node; lets a 1-GPU box host a gloo dry run)
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")       # one node: a gloo backend (VD_BENCH_BACKEND=gloo dry runs) needs no hostname lookup
        backend = os.environ.get("VD_BENCH_BACKEND", "nccl")                                  # "nccl" == RCCL on ROCm
        opts = None
        if backend == "nccl" and os.environ.get("VD_RCCL_HIGH_PRIORITY", "0") != "0":
            # opt-in: RCCL's kernels on a high-priority stream (a bucket's all-reduce is dispatched ahead of queued compute workgroups as
            # soon as a CU frees up).  Off by default: on one MI355X (1-rank RCCL) it changed nothing with all CUs in use and cost
            # 12 ms per step together with reserved CUs -- to be re-measured on a multi-GPU node
            opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)
        dist.init_process_group(backend, init_method="env://", world_size=world, rank=rank, pg_options=opts)

    from v_diffusion import _hip
    _hip.lib()

    def barrier():
        if world > 1:
            if dist.get_backend() == "nccl":
                dist.barrier(device_ids=[dev_index])       # (names the device: no "using the current device" guess inside RCCL)
            else:
                dist.barrier()
        torch.cuda.synchronize()

    B = args.batch
    # ---- in-run clock figure (round-5 review 2e): a fixed register-only fp32 MFMA loop for 1.5 s right in front of the timed region; its in-kernel
    # clock (s_memtime / s_memrealtime) and rate say how fast THIS box runs matrix work, so that lines from different boxes can be normalised
    calib = None
    if not args.no_calibration:
        mhz, tf = _hip.mfma_calibrate(1.5)
        calib = {"kernel": "mfma_calibrate_kernel: 256 workgroups x 8 waves, register-only v_mfma_f32_16x16x4_f32", "seconds": 1.5,
                 "in_kernel_clock_mhz": round(mhz, 1), "tflops": round(tf, 2), "frac_of_fp32_mfma_peak": round(tf / PEAK_FP32_MFMA_TFLOPS, 4),
                 "note": "measured on this box immediately before the timed steps; the train step itself holds 2.33 GHz on a 2.40 GHz box "
                         "(profiles/r05_clock_by_kernel.txt) -- scale by in_kernel_clock_mhz / 2400 when comparing boxes"}
    r = run_training(wl, B, args.steps, args.warmup, device, rank, world, barrier, args.sample_steps, extras=not args.no_extras,
                     uint8_input=args.uint8_input)
    model, diffusion, labels, RES = r["model"], r["diffusion"], r["labels"], r["res"]
    fwd_gflop = r["fwd_gflop"]
    hbm_peak = torch.cuda.max_memory_allocated(device)

    # ---- sampling: DDIM-50 + classifier-free guidance (w=1): 2B UNet rows per step
    sampling = None
    if not args.no_sample:
        sampling = sample_once(diffusion, model, labels, B, RES, args.sample_steps, r["W"], device, rank, world, barrier)

    # ---- secondary workload (BASELINE configs[3]): CelebA 64x64, per-GPU batch 128 -- a few train steps, then DDIM-50 + CFG
    # sampling of 128 images (the second half of the north-star metric on CelebA-shaped tensors), then the CPU oracle beside it
    secondary = None
    if wl == "cifar10" and not args.no_secondary and B == 128:
        del model, diffusion
        r.pop("trainer"); r.pop("model"); r.pop("diffusion")
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        torch.cuda.reset_peak_memory_stats(device)
        r2 = run_training("celeba", 128, 5, 2, device, rank, world, barrier, args.sample_steps, extras=False)
        rf = r2["roofline"]
        secondary = {"metric": "train_images_per_sec", "value": round(r2["value"], 2), "unit": "images/s", "steps": 5, "warmup": 2,
                     "ms_per_step": round(r2["ms_per_step"], 3),
                     "config": {"workload": r2["name"] + " full train step (BASELINE configs[3]); second figure: DDIM-50 CFG w=1 sampling",
                                "global_batch": world * 128, "per_gpu_batch": 128, "resolution": 64, "final_loss": round(r2["final_loss"], 5)},
                     "hbm_peak_gib": round(torch.cuda.max_memory_allocated(device) / 2 ** 30, 2),
                     "roofline": None if rf is None else {k: rf[k] for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic",
                                                                              "algorithmic_tflops", "speedup_vs_direct_roofline",
                                                                              "frac_vs_direct_roofline", "algorithmic_bytes_per_launch",
                                                                              "traffic_ratio", "traffic_note", "launches_per_step",
                                                                              "avg_launch_ms", "flops_per_launch", "share_of_matmul_time",
                                                                              "whole_step")}}
        if rf is not None:
            top = sorted(rf["all_matmul_kernels"].items(), key=lambda kv: -kv[1]["ms_per_step"])[:8]
            secondary["roofline"]["top_matmul_kernels"] = dict(top)
            if "hbm" in rf:
                secondary["roofline"]["hbm"] = {k: rf["hbm"][k] for k in ("kernel", "achieved", "unit", "frac", "traffic", "traffic_ratio",
                                                                             "traffic_source", "bytes_per_launch", "ms_per_step")}
        if not args.no_sample:
            r2.pop("trainer")
            gc.collect()
            torch.cuda.empty_cache()
            secondary["sampling"] = sample_once(r2["diffusion"], r2["model"], r2["labels"], 128, 64, args.sample_steps, r2["W"], device,
                                                rank, world, barrier)
        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            secondary["cpu_baseline"] = cpu_baseline("celeba", batch=4, budget_s=12.0)
        del r2

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline()
    # the same math through stock PyTorch-ROCm on this GPU (baseline leg only, after every timed region; rank 0 at N = 1 like cpu_baseline)
    torch_base = None
    if rank == 0 and world == 1 and not args.no_torch_baseline:
        import gc
        r.pop("trainer", None); r.pop("model", None); r.pop("diffusion", None)
        model = diffusion = None
        gc.collect()
        torch.cuda.empty_cache()
        try:
            torch_base = torch_rocm_baseline(wl, batch=B, device=device)
            torch_base["hot_path_over_baseline"] = round(r["value"] / torch_base["value"], 3)
        except Exception as e:                      # a baseline that cannot run must not take the measured line down with it
            torch_base = {"value": None, "kind": "port on ATen/MIOpen", "baseline_only": True, "error": f"{type(e).__name__}: {e}"[:300]}

    # ---- the pure fp32-MFMA figure beside the headline (round-5 review 2b): VD_GEMM_SPLIT is read once per process, so a FRESH CHILD PROCESS
    # (never a re-exec) runs the same timed region with the tile-engine GEMMs on the fp32 MFMA instructions, after every timed region here
    fp32_ab = None
    if rank == 0 and world == 1 and not args.no_fp32_ab and _hip.lib().vd_gemm_split_forms():
        import subprocess
        cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", str(args.steps), "--warmup", str(args.warmup), "--batch", str(B),
               "--config", wl, "--no-sample", "--no-cpu-baseline", "--no-secondary", "--no-extras", "--no-torch-baseline", "--no-fp32-ab",
               "--no-calibration"]
        try:
            cp = subprocess.run(cmd, env=dict(os.environ, VD_GEMM_SPLIT="0"), capture_output=True, text=True, timeout=600)
            jl = [l for l in cp.stdout.splitlines() if l.startswith("{")]
            cj = json.loads(jl[-1])
            fp32_ab = {"env": "VD_GEMM_SPLIT=0 (tile-engine GEMMs on v_mfma_f32_*: fp32 MFMA everywhere)", "dtype": cj["dtype"],
                       "ms_per_step": cj["ms_per_step"], "value": cj["value"], "unit": "images/s", "steps": cj["steps"], "warmup": cj["warmup"],
                       "headline_over_fp32_mfma": round(r["value"] / cj["value"], 4)}
        except Exception as e:
            fp32_ab = {"value": None, "error": f"{type(e).__name__}: {e}"[:300]}

    # the arithmetic the path computes in: fp32 operands and results everywhere; Winograd convolutions and the fused attention on the fp32 MFMA
    # instructions; the tile-engine GEMMs (1x1 convolutions, linears, attention products, weight-gradient planes) by default through split
    # operands -- every fp32 value the exact sum of three bf16 pieces, six piece products on the 16-bit matrix cores, fp32 accumulation,
    # error against fp64 0.6-0.9 of the fp32 MFMA chain's (round-4 review item 6 ii; VD_GEMM_SPLIT=0 = fp32 MFMA everywhere)
    dtype_label = "fp32 (bf16x3 split products in the tile-engine GEMMs)" if _hip.lib().vd_gemm_split_forms() else "fp32"
    if rank == 0:
        line = {"metric": "train_images_per_sec", "value": round(r["value"], 2), "unit": "images/s", "n_gpus": world, "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": round(r["ms_per_step"], 3), "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": dtype_label, "data": "synthetic",
                "config": {"workload": r["name"] + " full train step: q_sample+fwd+snr_trunc v-loss+bwd+grad all-reduce+clip+AdamW+EMA; "
                                       "second figure: DDIM-50 CFG w=1 sampling",
                           "global_batch": world * B, "per_gpu_batch": B, "resolution": RES, "parallelism": f"dp{world}",
                           "final_loss": round(r["final_loss"], 5),
                           "ddim_cfg_samples_per_sec": None if sampling is None else sampling["value"]},
                "sampling_images_per_sec": None if sampling is None else sampling["value"],
                "fp32_mfma_ab": fp32_ab, "calibration": calib,
                "ms_per_step_with_loss_item": r.get("ms_per_step_with_loss_item"), "ms_fwd_bwd_only": r.get("ms_fwd_bwd_only"),
                "hbm_peak_gib": round(hbm_peak / 2 ** 30, 2),
                "roofline": r["roofline"], "cpu_baseline": cpu, "torch_rocm_baseline": torch_base, "sampling": sampling,
                "multi_gpu": r.get("multi_gpu"), "input": "uint8 HWC + flip mask -> vd_images_from_uint8_hwc" if args.uint8_input else "fp32 NCHW",
                "secondary": secondary}
        print(json.dumps(line), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
