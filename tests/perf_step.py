"""Per-shape breakdown of the matmul-shaped launches of one train step (not a test): HIP events around every vd_gemm /
vd_conv3x3 / vd_conv3x3_wgrad call, grouped by (entry, shape).    python tests/perf_step.py [celeba]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "v-diffusion-torch_amd")]
import torch
import v_diffusion
from v_diffusion import _hip as H
from v_diffusion.trainer import HotPathTrainer
from bench import build_model, CIFAR, CELEBA

celeba = len(sys.argv) > 1 and sys.argv[1] == "celeba"
dev = torch.device("cuda", 0)
REC = []
_gemm, _conv, _wgrad = H.gemm, H.conv3x3, H.conv3x3_wgrad


def timed(key, flops, fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record()
    t = H.lib().vd_gemm_last_tile()
    REC.append((key + f" tile={(t // 1000) % 1000}x{t % 1000}/kt{(t // 1000000) % 100}{'T' if t // 100000000 else ''}", flops, e0, e1))


def gemm(A, B, Cm, M, N, K, **kw):
    key = f"gemm({kw.get('a_kind', 0)},{kw.get('b_kind', 0)}) M={M} N={N} K={K} batch={kw.get('batch', 1)} splitk={kw.get('splitk', 1)}"
    timed(key, 2.0 * M * N * K * kw.get("batch", 1), lambda: _gemm(A, B, Cm, M, N, K, **kw))


def conv3x3(x, ldx, wpack, bias, y, ldy, nimg, Hh, W, Cin, Cout, **kw):
    timed(f"conv3x3 {Cin}->{Cout} @{Hh}x{W}", 2.0 * nimg * Hh * W * Cout * 9 * Cin,
          lambda: _conv(x, ldx, wpack, bias, y, ldy, nimg, Hh, W, Cin, Cout, **kw))


def conv3x3_wgrad(x, ldx, dy, lddy, nimg, Hh, W, Cin, Cout, dw, Cin_w, Cout_w, **kw):
    timed(f"wgrad   {Cin}->{Cout} @{Hh}x{W}", 2.0 * nimg * Hh * W * Cout * 9 * Cin,
          lambda: _wgrad(x, ldx, dy, lddy, nimg, Hh, W, Cin, Cout, dw, Cin_w, Cout_w, **kw))


model = build_model(dev, cfg=CELEBA if celeba else CIFAR).train()
diffusion = v_diffusion.GaussianDiffusion(v_diffusion.get_logsnr_schedule("cosine", -20.0, 20.0), 50, "v", "fixed_medium", "snr_trunc",
                                          "mse", intp_frac=0.3, w_guide=1.0, p_uncond=0.1)
trainer = HotPathTrainer(model, diffusion, lr=2e-4, weight_decay=0.001, warmup=1000, grad_norm=1.0, ema_decay=0.9999, use_ema=True)
B, RES = 128, (64 if celeba else 32)
x = torch.rand((B, 3, RES, RES), device=dev) * 2 - 1
lab = (torch.rand((B, 40), device=dev) < 0.2).float() if celeba else torch.randint(1, 11, (B,), device=dev).float()
for _ in range(3):
    trainer.step(x, lab.clone())
H.gemm, H.conv3x3, H.conv3x3_wgrad = gemm, conv3x3, conv3x3_wgrad
NS = 3
for _ in range(NS):
    trainer.step(x, lab.clone())
torch.cuda.synchronize()
agg = {}
for key, fl, e0, e1 in REC:
    a = agg.setdefault(key, [0.0, 0.0, 0])
    a[0] += fl; a[1] += e0.elapsed_time(e1); a[2] += 1
tot = sum(v[1] for v in agg.values()) / NS
print(f"matmul-shaped launches: {tot:.2f} ms/step")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:78s} n={v[2] // NS:3d} {v[1] / NS:7.3f} ms/step {v[1] / v[2] * 1e3:8.1f} us {v[0] / v[1] / 1e9:7.1f} TF/s")
