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

Extra objects on the same JSON line: ``roofline`` (dominant kernel, live HIP-event timing), ``cpu_baseline`` (the CPU
oracle on the host cores, rank 0 at N=1 only), ``sampling`` (DDIM-50 + CFG w=1 images/s, the second half of the metric).
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


def cpu_baseline(batch=8):
    """The CPU oracle (oracle/, proved equal to the reference by the golden fixtures) doing the same train step on the
    host cores: bounded sample, reported beside the GPU number, never the thing measured."""
    from oracle import unet_ref, diffusion_ref as dref
    from oracle.unet_ref import param_shapes
    ncpu = usable_cpus()
    torch.set_num_threads(ncpu)
    g = torch.Generator().manual_seed(0)
    sd = {}
    for k, shp in param_shapes(CIFAR).items():
        fan = max(int(torch.tensor(shp[1:]).prod()) if len(shp) > 1 else 1, 1)
        sd[k] = (torch.randn(shp, generator=g) * fan ** -0.5).requires_grad_(True)
    x0 = torch.rand((batch, 3, 32, 32), generator=g) * 2 - 1
    y = torch.randint(1, 11, (batch,), generator=g).float()
    den = lambda a, b, c: unet_ref.unet_forward(sd, CIFAR, a, b, c, train=True)
    sched = dref.make_schedule("cosine")
    times = []
    for it in range(2):                   # first pass warms the allocator / thread pool
        if times and times[0] > 25.0:     # bounded sample: do not spend another > 25 s
            break
        t = torch.rand((batch,), dtype=torch.float64, generator=g)
        noise = torch.randn(x0.shape, generator=g)
        t0 = time.perf_counter()
        loss = dref.train_loss(den, sched, x0, t, y, noise, "v", "snr_trunc").mean()
        loss.backward()
        times.append(time.perf_counter() - t0)
        for v in sd.values():
            v.grad = None
    try:
        model_name = [l.split(":")[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        model_name = "unknown"
    return {"value": round(batch / times[-1], 3), "unit": "images/s", "cores": ncpu, "kind": "port",
            "sample": f"CIFAR-10 cond UNet train step (q_sample+fwd+v-loss+bwd, no optimizer), batch {batch}, {'1 timed step after 1 warm-up' if len(times) > 1 else 'single cold step (bounded)'}, "
                      f"torch CPU fp32 on {ncpu} threads, {model_name}"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=128, help="per-GPU batch (BASELINE config: 128)")
    ap.add_argument("--no-sample", action="store_true", help="skip the DDIM-50 CFG sampling measurement")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sample-steps", type=int, default=50)
    ap.add_argument("--config", choices=["cifar10", "celeba"], default="cifar10",
                    help="cifar10 = the headline workload (BASELINE configs[1]); celeba = configs[3], secondary")
    args = ap.parse_args()
    celeba = args.config == "celeba"
    global FWD_GFLOP_PER_IMG
    if celeba:
        FWD_GFLOP_PER_IMG = CELEBA_FWD_GFLOP_PER_IMG

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
        backend = os.environ.get("VD_BENCH_BACKEND", "nccl")                                  # "nccl" == RCCL on ROCm
        dist.init_process_group(backend, init_method="env://", world_size=world, rank=rank)

    import v_diffusion
    from v_diffusion import _hip
    from v_diffusion.trainer import HotPathTrainer
    _hip.lib()

    model = build_model(device, cfg=CELEBA if celeba else CIFAR)
    model.train()
    diffusion = v_diffusion.GaussianDiffusion(v_diffusion.get_logsnr_schedule("cosine", -20.0, 20.0), args.sample_steps, "v",
                                              "fixed_medium", "snr_trunc", "mse", intp_frac=0.3, w_guide=1.0, p_uncond=0.1)
    trainer = HotPathTrainer(model, diffusion, lr=2e-4, weight_decay=0.001, warmup=1000, grad_norm=1.0, ema_decay=0.9999,
                             use_ema=True, rank=rank, world_size=world)
    B = args.batch
    RES = 64 if celeba else 32
    g = torch.Generator(device).manual_seed(4321 + rank)
    x = torch.rand((B, 3, RES, RES), device=device, generator=g) * 2 - 1                  # Normalize(0.5, 0.5) range
    if celeba:
        labels = (torch.rand((B, 40), device=device, generator=g) < 0.2).float()           # CelebA attribute tags
    else:
        labels = torch.randint(1, 11, (B,), device=device, generator=g).float()            # target_transform y+1

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def one_step():
        return trainer.step(x, labels.clone())          # y is mutated by the label drop: hand over a fresh copy

    for _ in range(args.warmup):
        one_step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = one_step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    ms_per_step = dt / args.steps * 1e3
    value = world * B * args.steps / dt
    final_loss = float(loss.item())

    # ---- live roofline of the dominant kernel: HIP events around every matmul-shaped launch of two extra steps
    roofline = None
    if rank == 0:
        _hip.PROFILE = []
    for _ in range(2):                  # every rank takes part (the steps contain collectives); only rank 0 records
        one_step()
    barrier()
    if rank == 0:
        rec, _hip.PROFILE = _hip.PROFILE, None
        agg = {}
        for name, flops, e0, e1 in rec:
            a = agg.setdefault(name, [0.0, 0.0, 0])
            a[0] += flops; a[1] += e0.elapsed_time(e1) * 1e-3; a[2] += 1
        total_t = sum(v[1] for v in agg.values())
        dom = max(agg, key=lambda k: agg[k][1])
        fl, tt, n = agg[dom]
        traffic = None
        try:    # HBM bytes per launch of the dominant kernel from the committed PMC passes (profiles/parse_rocprof.py)
            tj = json.load(open(os.path.join(ROOT, "profiles", "r01_traffic.json")))["kernels"]
            key = dom.split(" (+")[0]
            if key in tj:
                traffic = tj[key]["hbm_bytes_per_launch"]
        except Exception:
            traffic = None
        roofline = {"bound": "mfma", "kernel": dom, "achieved": round(fl / tt / 1e12, 2), "peak": PEAK_FP32_MFMA_TFLOPS,
                    "unit": "TFLOP/s", "frac": round(fl / tt / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4), "traffic": traffic,
                    "traffic_note": "HBM+fabric bytes per launch, PMC (FETCH_SIZE x2 + WRITE_SIZE), profiles/r01_traffic.json",
                    "launches_per_step": n // 2, "avg_launch_ms": round(tt / n * 1e3, 4),
                    "flops_per_launch": round(fl / n / 1e9, 3), "flops_unit": "GFLOP (2*M*N*K of the implicit GEMM)",
                    "share_of_matmul_time": round(tt / total_t, 3),
                    "all_matmul_kernels": {k: {"tflops": round(v[0] / v[1] / 1e12, 2), "ms_per_step": round(v[1] / 2 * 1e3, 2),
                                               "launches_per_step": v[2] // 2} for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])},
                    "step_matmul_tflops": round(3 * FWD_GFLOP_PER_IMG * B / (ms_per_step * 1e-3) / 1e3, 2)}

    # ---- sampling: DDIM-50 + classifier-free guidance (w=1): 2B UNet rows per step
    sampling = None
    if not args.no_sample:
        model.eval()
        SB = B
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
        T = args.sample_steps
        sampling = {"metric": f"ddim{T}_cfg_samples_per_sec", "value": round(world * SB / ds, 2), "unit": "images/s",
                    "seconds_per_batch": round(ds, 3), "batch_per_gpu": SB, "unet_rows_per_step": 2 * SB, "w_guide": 1.0,
                    "matmul_tflops": round(T * 2 * FWD_GFLOP_PER_IMG * SB / ds / 1e3, 2),
                    "frac_of_fp32_mfma_peak": round(T * 2 * FWD_GFLOP_PER_IMG * SB / ds / 1e3 / PEAK_FP32_MFMA_TFLOPS, 4),
                    "finite": bool(torch.isfinite(out).all())}
        model.train()

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline()

    if rank == 0:
        line = {"metric": "train_images_per_sec", "value": round(value, 2), "unit": "images/s", "n_gpus": world, "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": "fp32", "data": "synthetic",
                "config": {"workload": ("CelebA 64x64 multitag v-pred UNet (celeba.json+defaults.json, 266.8M params)" if celeba else
                                        "CIFAR-10 32x32 class-cond v-pred UNet (cifar10_cond.json, 60.8M params)") + " full train step: "
                                       "q_sample+fwd+snr_trunc v-loss+bwd+grad all-reduce+clip+AdamW+EMA; second figure: DDIM-50 CFG w=1 sampling",
                           "global_batch": world * B, "per_gpu_batch": B, "resolution": RES, "parallelism": f"dp{world}",
                           "final_loss": round(final_loss, 5)},
                "frac_of_fp32_mfma_peak_whole_step": round(3 * FWD_GFLOP_PER_IMG * B / (ms_per_step * 1e-3) / 1e3 / PEAK_FP32_MFMA_TFLOPS, 4),
                "hbm_peak_gib": round(torch.cuda.max_memory_allocated(device) / 2 ** 30, 2),
                "roofline": roofline, "cpu_baseline": cpu, "sampling": sampling}
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
