"""Socket power and shader clock while ONE kernel runs back to back (not a test): which launches pull the part towards its power limit,
and what a launch costs in joules.   python tests/probe/power_by_kernel.py
rocm-smi is sampled from a side thread every 0.25 s during a ~4 s loop of each kernel; energy per launch = mean power x mean launch time."""
import os
import re
import subprocess
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "v-diffusion-torch_amd"))
from v_diffusion import _hip as H   # noqa: E402

DEV = "cuda"
g = torch.Generator(DEV).manual_seed(1)


def smi():
    try:
        out = subprocess.run(["/opt/rocm/bin/rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=5).stdout
    except Exception:
        return None, None
    pw = re.search(r"Power \(W\):\s*([\d.]+)", out)
    ck = re.search(r"sclk clock level:.*?\((\d+)Mhz\)", out)
    return (float(pw.group(1)) if pw else None), (float(ck.group(1)) if ck else None)


def measure(name, fn, flops=0.0, nbytes=0.0, seconds=4.0):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    samples, stop = [], threading.Event()

    def sampler():
        while not stop.is_set():
            samples.append(smi())
            time.sleep(0.25)
    th = threading.Thread(target=sampler)
    th.start()
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        n += 20
    dt = time.perf_counter() - t0
    stop.set(); th.join()
    pw = [p for p, c in samples[2:] if p is not None]
    ck = [c for p, c in samples[2:] if c is not None]
    mp = sum(pw) / len(pw) if pw else float("nan")
    mc = sum(ck) / len(ck) if ck else float("nan")
    ms = dt / n * 1e3
    print(f"{name:46s} {ms:8.3f} ms  {mp:7.0f} W  sclk {mc:6.0f} MHz  {mp * ms / 1e3:7.3f} J/launch"
          + (f"  {flops / ms / 1e9:6.1f} TFLOP/s  {flops / (mp * ms / 1e3) / 1e9:6.1f} GFLOP/J" if flops else "")
          + (f"  {nbytes / ms / 1e9:6.2f} TB/s  {nbytes / (mp * ms / 1e3) / 1e9:6.2f} GB/J" if nbytes else ""), flush=True)


print("idle:", smi())
B, R, C = 128, 32, 256
x = torch.nn.functional.silu(torch.randn((B, R, R, C), device=DEV, generator=g))
w = torch.randn((C, C, 3, 3), device=DEV, generator=g) * (9 * C) ** -0.5
bias = torch.randn((C,), device=DEV, generator=g)
res = torch.randn((B, R, R, C), device=DEV, generator=g)
y = torch.empty((B, R, R, C), device=DEV)
fl = 2.0 * B * R * R * C * 9 * C
u43f = torch.empty(H.lib().vd_wino43_u_floats(C, C), device=DEV); H.wino43_pack_fwd(w, C, C, u43f)
u43 = torch.empty(H.lib().vd_wino43_u_floats(C, C), device=DEV); H.wino43_pack(w, C, C, u43)
part = torch.empty(H.stats_part_numel(B, R * R, C), device=DEV)
measure("wino43 fwd 256->256 @32 (+res, stats)", lambda: H.conv3x3_wino43_fwd(x, C, u43f, bias, y, C, B, R, R, C, C, res=res, ldres=C, stats_part=part), fl / 4)
measure("wino43 dgrad 256->256 @32", lambda: H.conv3x3_dgrad_wino43(x, C, u43, y, C, B, R, R, C, C), fl / 4)
uf = torch.empty(16, C, C, device=DEV); H.wino_pack(w, C, C, uf=uf)
measure("wino F(2,3) fwd 256->256 @32", lambda: H.conv3x3_wino(x, C, uf, bias, y, C, B, R, R, C, C), fl * 4 / 9)
dw, db = torch.empty(C, C, 3, 3, device=DEV), torch.empty(C, device=DEV)
measure("wino43 wgrad 256->256 @32 (all phases)", lambda: H.conv3x3_wgrad_wino43(x, C, res, C, B, R, R, C, C, dw, C, C, dbias=db), fl / 4)
M, N, K = B * R * R, 256, 512
A = torch.randn((M, K), device=DEV, generator=g); Bm = torch.randn((N, K), device=DEV, generator=g); Cm = torch.empty((M, N), device=DEV)
measure(f"gemm RR M={M} N={N} K={K} ({'split' if os.environ.get('VD_GEMM_SPLIT') == '1' else 'fp32 MFMA'})",
        lambda: H.gemm(A, Bm, Cm, M, N, K, a_kind=0, b_kind=0, lda=K, ldb=K, ldc=N), 2.0 * M * N * K)
stats = torch.empty(B, 32, 2, device=DEV); coef = torch.empty(B, 4, C, device=DEV)
gamma, beta = torch.randn(C, device=DEV), torch.randn(C, device=DEV)
H.gn_stats(x, C, B, R * R, C, stats)
H.gn_apply(x, C, stats, gamma, beta, None, 1, 0.0, 1234, H.RS_NONE, y, C, B, R, R, C, coef)
measure("gn_apply 32x32x256", lambda: H.gn_apply(x, C, None, gamma, beta, None, 1, 0.0, 1234, H.RS_NONE, y, C, B, R, R, C, coef), nbytes=2 * 4.0 * B * R * R * C)
dx, dg, dbt = torch.empty_like(x), torch.empty(C, device=DEV), torch.empty(C, device=DEV)
measure("gn_bwd_fused 32x32x256", lambda: H.gn_apply_bwd(res, C, x, C, coef, gamma, beta, None, 1, 0.0, 1234, H.RS_NONE, None, 0, dx, C, False, None, dg, dbt, False, B, R, R, C),
        nbytes=3 * 4.0 * B * R * R * C)
print("idle:", smi())
