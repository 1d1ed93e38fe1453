"""Per-shape table of the vd_gemm launches of one CIFAR (or CelebA) train step: which (M, N, K, kinds, batch, split-K) run, how
often, at what rate.  Diagnostic for the short-launch GEMMs (not a test):   python tests/probe/gemm_shapes.py [cifar10|celeba] [B]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "v-diffusion-torch_amd"))
import bench                                        # noqa: E402
import v_diffusion                                   # noqa: E402
from v_diffusion import _hip as H                    # noqa: E402
from v_diffusion.trainer import HotPathTrainer       # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "cifar10"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
dev = torch.device("cuda", 0)
W = bench.WORKLOADS[wl]
model = bench.build_model(dev, cfg=W["cfg"]).train()
diffusion = v_diffusion.GaussianDiffusion(v_diffusion.get_logsnr_schedule("cosine", -20.0, 20.0), 50, "v", "fixed_medium", "snr_trunc",
                                          "mse", intp_frac=0.3, w_guide=1.0, p_uncond=0.1)
trainer = HotPathTrainer(model, diffusion, lr=2e-4, weight_decay=0.001, warmup=1000, grad_norm=1.0, ema_decay=0.9999, use_ema=True)
x = torch.rand((B, 3, W["res"], W["res"]), device=dev) * 2 - 1
y = (torch.rand((B, 40), device=dev) < 0.2).float() if wl == "celeba" else torch.randint(1, 11, (B,), device=dev).float()

orig = H.gemm
KN = {0: "R", 1: "C", 2: "I"}


def gemm(A, Bm, Cm, M, N, K, **kw):
    n0 = len(H.PROFILE) if H.PROFILE is not None else 0
    orig(A, Bm, Cm, M, N, K, **kw)
    if H.PROFILE is not None:
        tag = f" M={M} N={N} K={K} {KN[kw.get('a_kind', 0)]}{KN[kw.get('b_kind', 0)]} b={kw.get('batch', 1)} sk={kw.get('splitk', 1)}" \
              f"{' bias' if kw.get('bias') is not None else ''}{' R' if kw.get('R') is not None else ''}{' acc' if kw.get('accumulate') else ''}" \
              f"{' cs' if kw.get('colsum') is not None else ''}{' st' if kw.get('stats') is not None else ''}"
        for i in range(n0, len(H.PROFILE)):
            r = H.PROFILE[i]
            H.PROFILE[i] = (r[0] + tag,) + r[1:]


H.gemm = gemm
for _ in range(3):
    trainer.step(x, y.clone())
torch.cuda.synchronize()
H.PROFILE = []
NS = 3
for _ in range(NS):
    trainer.step(x, y.clone())
torch.cuda.synchronize()
rec, H.PROFILE = H.PROFILE, None
agg = {}
for name, work, e0, e1 in rec:
    if not name.startswith("gemm"):
        continue
    a = agg.setdefault(name, [0.0, 0.0, 0])
    a[0] += work; a[1] += e0.elapsed_time(e1) * 1e-3; a[2] += 1
tot = sum(v[1] for v in agg.values()) / NS * 1e3
print(f"{wl} B={B}: vd_gemm launches {sum(v[2] for v in agg.values()) // NS}/step, {tot:.2f} ms/step")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{v[1] / NS * 1e3:7.3f} ms/step  {v[2] // NS:3d} x {v[1] / v[2] * 1e6:7.1f} us  {v[0] / v[1] / 1e12:6.1f} TF  {k}")
