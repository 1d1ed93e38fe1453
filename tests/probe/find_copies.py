"""Which host-side calls issue the device-to-device / host-to-device copies of one CIFAR train step (rocprof shows ~80
__amd_rocclr_copyBuffer launches per step)?  Counts aten::copy_ / aten::to calls by Python call site.   python tests/probe/find_copies.py"""
import collections
import os
import sys
import traceback

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "v-diffusion-torch_amd"))
import bench                                        # noqa: E402
import v_diffusion                                   # noqa: E402
from v_diffusion.trainer import HotPathTrainer       # noqa: E402

dev = torch.device("cuda", 0)
model = bench.build_model(dev).train()
diffusion = v_diffusion.GaussianDiffusion(v_diffusion.get_logsnr_schedule("cosine", -20.0, 20.0), 50, "v", "fixed_medium", "snr_trunc",
                                          "mse", intp_frac=0.3, w_guide=1.0, p_uncond=0.1)
trainer = HotPathTrainer(model, diffusion, lr=2e-4, weight_decay=0.001, warmup=1000, grad_norm=1.0, ema_decay=0.9999, use_ema=True)
x = torch.rand((128, 3, 32, 32), device=dev) * 2 - 1
y = torch.randint(1, 11, (128,), device=dev).float()
for _ in range(3):
    trainer.step(x, y.clone())
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    trainer.step(x, y.clone())
    torch.cuda.synchronize()
cnt = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::_to_copy", "aten::clone", "aten::contiguous", "aten::fill_", "aten::zero_", "aten::cat", "aten::stack") or "Memcpy" in ev.name or "copyBuffer" in ev.name:
        st = [s for s in (ev.stack or []) if "v_diffusion" in s or "bench" in s or "trainer" in s]
        cnt[(ev.name, st[0] if st else "?")] += 1
for (n, s), c in cnt.most_common(40):
    print(f"{c:4d}  {n:24s} {s}")
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=12))
