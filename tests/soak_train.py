"""Soak run (not a test): 300 HotPathTrainer steps of the CIFAR-10 UNet on one fixed batch -- the loss must fall, allocated HBM must
not drift, and sampling through the EMA weight swap must stay finite.    python tests/soak_train.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "v-diffusion-torch_amd")]
import torch, v_diffusion
from v_diffusion.trainer import HotPathTrainer
from bench import build_model, CIFAR
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = build_model(dev, cfg=CIFAR).train()
gd = v_diffusion.GaussianDiffusion(v_diffusion.get_logsnr_schedule("cosine"), 50, "v", "fixed_medium", "snr_trunc", "mse", intp_frac=0.3, w_guide=1.0, p_uncond=0.1)
tr = HotPathTrainer(model, gd, lr=2e-4, weight_decay=0.001, warmup=50, grad_norm=1.0, ema_decay=0.9999, use_ema=True)
B = 32
x = torch.rand((B, 3, 32, 32), device=dev) * 2 - 1      # one fixed batch: the loss must fall
lab = torch.randint(1, 11, (B,), device=dev).float()
losses, mem = [], []
for it in range(300):
    l = tr.step(x, lab.clone())
    if it % 25 == 0 or it == 299:
        losses.append(float(l)); mem.append(torch.cuda.memory_allocated(dev) / 2**20)
print("loss", [round(v, 4) for v in losses])
print("MiB ", [round(v) for v in mem])
assert all(torch.isfinite(torch.tensor(losses))) and losses[-1] < 0.5 * losses[0]
assert max(mem[2:]) - min(mem[2:]) < 64, "allocated memory drifts"
with tr.ema_weights():
    model.eval()
    s = gd.p_sample(model, (4, 3, 32, 32), label=lab[:4], device=dev, seed=1, use_ddim=True)
print("sample finite", bool(torch.isfinite(s).all()), float(s.abs().max()))
