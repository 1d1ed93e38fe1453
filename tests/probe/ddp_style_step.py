"""What the reference's OWN training loop costs on top of the hot path (not a test): `Trainer.step` of train_utils.py:151-169 written out with
the reference's tools -- DDP(model) on a 1-rank RCCL group, loss.backward() through the chain of autograd nodes, nn.utils.clip_grad_norm_,
torch.optim.AdamW, LambdaLR warm-up, a per-parameter EMA as utils.py:144-149 -- around v_diffusion.UNet / GaussianDiffusion, against
HotPathTrainer.step (flat buffers, fused clip + AdamW + EMA) on the same model and batch.   python tests/probe/ddp_style_step.py"""
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "v-diffusion-torch_amd")]
import bench                                             # noqa: E402
import v_diffusion                                       # noqa: E402
from v_diffusion.trainer import HotPathTrainer           # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29571")
torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method="env://", world_size=1, rank=0)
dev = torch.device("cuda", 0)
B = 128
g = torch.Generator(dev).manual_seed(1)
x = torch.rand((B, 3, 32, 32), device=dev, generator=g) * 2 - 1
y = torch.randint(1, 11, (B,), device=dev, generator=g).float()


def gd():
    return v_diffusion.GaussianDiffusion(v_diffusion.get_logsnr_schedule("cosine", -20.0, 20.0), 50, "v", "fixed_medium", "snr_trunc", "mse",
                                         intp_frac=0.3, w_guide=1.0, p_uncond=0.1)


def timeit(step, n=15, warm=4):
    for _ in range(warm):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


# ---- the reference's loop
from torch.nn.parallel import DistributedDataParallel as DDP   # noqa: E402
model = bench.build_model(dev, cfg=bench.CIFAR).train()
ddp = DDP(model, device_ids=[0])
diffusion = gd()
opt = torch.optim.AdamW(ddp.parameters(), lr=2e-4, betas=(0.9, 0.999), weight_decay=0.001)
sched = torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda=lambda t: min((t + 1) / 1000, 1.0))
shadow = [p.detach().clone() for p in model.parameters()]
gen = torch.Generator(dev).manual_seed(8191)
nupd = [0]


def ref_step():
    t = torch.rand((B,), dtype=torch.float64, device=dev, generator=gen)
    noise = torch.empty_like(x).normal_(generator=gen)
    loss = diffusion.train_loss(ddp, x_0=x, t=t, y=y.clone(), noise=noise).mean()
    loss.backward()
    torch.nn.utils.clip_grad_norm_(ddp.parameters(), max_norm=1.0)
    opt.step()
    opt.zero_grad(set_to_none=True)
    sched.step()
    nupd[0] += 1
    decay = min(0.9999, (1 + nupd[0]) / (10 + nupd[0]))
    with torch.no_grad():
        torch._foreach_lerp_(shadow, [p.detach() for p in model.parameters()], 1 - decay)      # (the reference loops per parameter: utils.py:144-149)
    return loss


for chain in (True, False):
    from v_diffusion import _hip
    _hip.AUTOGRAD_CHAIN = chain
    ms = timeit(ref_step)
    print(f"reference-style loop (DDP 1-rank RCCL + clip_grad_norm_ + torch AdamW + EMA), autograd chain {'on' if chain else 'off (single node)'}: "
          f"{ms:.2f} ms per step = {B / ms * 1e3:.0f} img/s", flush=True)
_hip.AUTOGRAD_CHAIN = True
del ddp, opt, shadow, model
torch.cuda.empty_cache()

# ---- the same loop with the two-line opt-in (v_diffusion.optim): FusedAdamW instead of torch.optim.AdamW, v_diffusion.optim.EMA instead of the
# reference's per-parameter EMA; clip_grad_norm_, LambdaLR and DDP stay torch's
from v_diffusion.optim import FusedAdamW, EMA   # noqa: E402
model = bench.build_model(dev, cfg=bench.CIFAR).train()
opt = FusedAdamW(model.parameters(), lr=2e-4, betas=(0.9, 0.999), weight_decay=0.001)
ddp = DDP(model, device_ids=[0])
sched = torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda=lambda t: min((t + 1) / 1000, 1.0))
ema = EMA(model, decay=0.9999)


def fused_step(clip=True):
    t = torch.rand((B,), dtype=torch.float64, device=dev, generator=gen)
    noise = torch.empty_like(x).normal_(generator=gen)
    loss = diffusion.train_loss(ddp, x_0=x, t=t, y=y.clone(), noise=noise).mean()
    loss.backward()
    if clip:
        torch.nn.utils.clip_grad_norm_(ddp.parameters(), max_norm=1.0)
    opt.step()
    opt.zero_grad(set_to_none=True)
    sched.step()
    ema.update()
    return loss


ms = timeit(fused_step)
print(f"reference-style loop with v_diffusion.optim.FusedAdamW + EMA (DDP 1-rank RCCL, torch clip_grad_norm_ kept): {ms:.2f} ms per step = {B / ms * 1e3:.0f} img/s", flush=True)
opt.param_groups[0]["max_grad_norm"] = 1.0
ms = timeit(lambda: fused_step(clip=False))
print(f"  ... with the clip inside FusedAdamW(max_grad_norm=1.0) instead of clip_grad_norm_: {ms:.2f} ms per step = {B / ms * 1e3:.0f} img/s", flush=True)
del ddp, opt, ema, model
torch.cuda.empty_cache()

# ---- the flat-buffer trainer on the same model / batch
model = bench.build_model(dev, cfg=bench.CIFAR).train()
tr = HotPathTrainer(model, gd(), lr=2e-4, weight_decay=0.001, warmup=1000, grad_norm=1.0, ema_decay=0.9999, use_ema=True)
ms = timeit(lambda: tr.step(x, y.clone()))
print(f"HotPathTrainer.step (flat buffers, fused clip + AdamW + EMA): {ms:.2f} ms per step = {B / ms * 1e3:.0f} img/s", flush=True)
dist.destroy_process_group()
