"""In-kernel shader clock per kernel class of the train step (not a test): each class runs alone, back to back, for >= 2 s on random data while
tests/probe/clock_witness.hip -- one single-lane workgroup per XCD on a second stream -- stamps (s_memtime, s_memrealtime) once a millisecond
beside it.  clock = delta s_memtime / delta s_memrealtime x 100 MHz (MI355X_MICROARCH.md, DVFS give-back item 6).  The product kernels carry no
stamps; this is the evidence the "the step is power-bound" reading rests on (round-4 review: board power and sysfs sclk are not the test).

    python tests/probe/clock_by_kernel.py > gpurun_out/r05_clock_by_kernel.txt
"""
import ctypes
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "v-diffusion-torch_amd"))
from v_diffusion import _hip as H   # noqa: E402  (the PRODUCT library)

DEV = "cuda"
wit = ctypes.CDLL(os.path.join(HERE, "libclock_witness.so"))
wit.launch_clock_witness.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_ulonglong, ctypes.c_void_p]
wit.launch_clock_witness.restype = ctypes.c_int
g = torch.Generator(DEV).manual_seed(1)
side = torch.cuda.Stream()
NWG, NSAMP, PERIOD = 8, 400, 100_000          # one witness per XCD, 400 stamps, 1 ms apart (100 MHz ticks)


ONLY = [w for w in os.environ.get("VD_CLOCK_ONLY", "").split(",") if w]      # substrings of the entries to run (default: all)


def measure(name, fn, flops=0.0, nbytes=0.0, lead_s=1.5):
    if ONLY and not any(w in name for w in ONLY):
        return
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    buf = torch.zeros(NWG * NSAMP * 2, dtype=torch.int64, device=DEV)
    done = torch.cuda.Event()
    n, t0, launched = 0, time.perf_counter(), False
    while True:
        for _ in range(20):
            fn()
        n += 20
        now = time.perf_counter() - t0
        if not launched and now >= lead_s:                 # >= 1.5 s of back-to-back launches before the first stamp
            with torch.cuda.stream(side):
                rc = wit.launch_clock_witness(buf.data_ptr(), NWG, NSAMP, PERIOD, side.cuda_stream)
                assert rc == 0, rc
                done.record(side)
            launched = True
        if launched and done.query():
            break
        if n % 200 == 0:
            torch.cuda.current_stream().synchronize()      # keep the launch queue bounded
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    t = buf.view(NWG, NSAMP, 2).double().cpu()
    clk = (t[:, -1, 0] - t[:, 0, 0]) / (t[:, -1, 1] - t[:, 0, 1]) * 100.0                      # MHz per witness (XCD), whole window
    win = (t[:, 10:, 0] - t[:, :-10, 0]) / (t[:, 10:, 1] - t[:, :-10, 1]) * 100.0              # 10 ms windows
    ms = dt / n * 1e3
    print(f"{name:44s} {ms:8.3f} ms/launch  in-kernel clock {clk.median():6.0f} MHz (XCDs {clk.min():.0f}-{clk.max():.0f}; 10 ms windows "
          f"{win.min():.0f}-{win.max():.0f})"
          + (f"  {flops / ms / 1e9:6.1f} TFLOP/s executed = {flops / ms / 1e9 / (clk.median() / 2400 * 157.3):.3f} of the MFMA peak AT THAT CLOCK"
             if flops else "")
          + (f"  {nbytes / ms / 1e9:5.2f} TB/s" if nbytes else ""), flush=True)


B, R, C = 128, 32, 256
x = torch.nn.functional.silu(torch.randn((B, R, R, C), device=DEV, generator=g))
w = torch.randn((C, C, 3, 3), device=DEV, generator=g) * (9 * C) ** -0.5
bias = torch.randn((C,), device=DEV, generator=g)
res = torch.randn((B, R, R, C), device=DEV, generator=g)
y = torch.empty((B, R, R, C), device=DEV)
fl = 2.0 * B * R * R * C * 9 * C
print(f"# in-kernel shader clock by kernel class, B = {B}, random data; witness: {NWG} single-lane workgroups (one per XCD), {NSAMP} stamps {PERIOD / 100:.0f} us apart,")
print("# started after 1.5 s of back-to-back launches of the class; 2400 MHz = the clock the 157.3 TFLOP/s fp32-MFMA peak is quoted at")
idle = torch.zeros(1, device=DEV)
measure("(idle: a 1-element add)", lambda: idle.add_(1.0), lead_s=0.5)
u43f = torch.empty(H.lib().vd_wino43_u_floats(C, C), device=DEV); H.wino43_pack_fwd(w, C, C, u43f)
u43 = torch.empty(H.lib().vd_wino43_u_floats(C, C), device=DEV); H.wino43_pack(w, C, C, u43)
part = torch.empty(H.stats_part_numel(B, R * R, C), device=DEV)
measure("wino43_conv<8,true> 256->256 @32 (+res, stats)", lambda: H.conv3x3_wino43_fwd(x, C, u43f, bias, y, C, B, R, R, C, C, res=res, ldres=C, stats_part=part), fl / 4)
measure("wino43_conv<8,false> 256->256 @32", lambda: H.conv3x3_dgrad_wino43(x, C, u43, y, C, B, R, R, C, C), fl / 4)
x16 = x[:, :16, :16].contiguous(); y16 = torch.empty_like(x16); res16 = res[:, :16, :16].contiguous()
measure("wino43_conv<4,true> 256->256 @16 (+res, stats)", lambda: H.conv3x3_wino43_fwd(x16, C, u43f, bias, y16, C, B, 16, 16, C, C, res=res16, ldres=C, stats_part=part), fl / 16)
uf = torch.empty(16, C, C, device=DEV); H.wino_pack(w, C, C, uf=uf)
x8 = x[:, :8, :8].contiguous(); y8 = torch.empty_like(x8)
measure("wino_conv F(2,3) 256->256 @8", lambda: H.conv3x3_wino(x8, C, uf, bias, y8, C, B, 8, 8, C, C), fl / 16 * 4 / 9)
dw, db = torch.empty(C, C, 3, 3, device=DEV), torch.empty(C, device=DEV)
nb = H.lib().vd_conv3x3_wgrad_wino43_ws_bytes(B, R, R, C, C)
ws = H.workspace(nb, x.device, "wgrad43")
wargs = (H.ptr(x), C, H.ptr(res), C, B, R, R, C, C, H.ptr(dw), H.ptr(db), C, C, 0, ws.data_ptr(), ws.numel() * 4)
H.conv3x3_wgrad_wino43(x, C, res, C, B, R, R, C, C, dw, C, C, dbias=db)
measure("wino43 wgrad: transform pass 256,256 @32", lambda: H.lib().vd_conv3x3_wgrad_wino43_phase(*wargs, 1, H.stream()), nbytes=4.0 * B * R * R * 2 * C * 3.25)
measure("wino43 wgrad: 36 grouped GEMMs 256x256 @32", lambda: H.lib().vd_conv3x3_wgrad_wino43_phase(*wargs, 2, H.stream()), fl / 4)
M, N, K = B * R * R, 256, 512
A = torch.randn((M, K), device=DEV, generator=g); Bm = torch.randn((N, K), device=DEV, generator=g); Cm = torch.empty((M, N), device=DEV)
measure(f"gemm_dma ROW,ROW M={M} N={N} K={K}", lambda: H.gemm(A, Bm, Cm, M, N, K, a_kind=0, b_kind=0, lda=K, ldb=K, ldc=N), 2.0 * M * N * K)
M2 = B * 64
A2 = A[:M2, :256].contiguous(); B2 = torch.randn((768, 256), device=DEV, generator=g); C2 = torch.empty((M2, 768), device=DEV)
measure(f"gemm_dma ROW,ROW M={M2} N=768 K=256 (8x8 proj_in)", lambda: H.gemm(A2, B2, C2, M2, 768, 256, a_kind=0, b_kind=0, lda=256, ldb=256, ldc=768), 2.0 * M2 * 768 * 256)
stats = torch.empty(B, 32, 2, device=DEV); coef = torch.empty(B, 4, C, device=DEV)
gamma, beta = torch.randn(C, device=DEV), torch.randn(C, device=DEV)
H.gn_stats(x, C, B, R * R, C, stats)
H.gn_apply(x, C, stats, gamma, beta, None, 1, 0.0, 1234, H.RS_NONE, y, C, B, R, R, C, coef)
measure("gn_apply 32x32x256", lambda: H.gn_apply(x, C, None, gamma, beta, None, 1, 0.0, 1234, H.RS_NONE, y, C, B, R, R, C, coef), nbytes=2 * 4.0 * B * R * R * C)
dx, dg, dbt = torch.empty_like(x), torch.empty(C, device=DEV), torch.empty(C, device=DEV)
measure("gn_bwd_fused 32x32x256", lambda: H.gn_apply_bwd(res, C, x, C, coef, gamma, beta, None, 1, 0.0, 1234, H.RS_NONE, None, 0, dx, C, False, None, dg, dbt, False, B, R, R, C),
        nbytes=3 * 4.0 * B * R * R * C)
n = 60_806_403
p_, g_, m_, v_, e_ = (torch.randn(n, device=DEV, generator=g) * 0.01 for _ in range(5))
v_.abs_()
gn = torch.ones(1, device=DEV)
measure("adamw_ema 60.8 M parameters", lambda: H.adamw_ema(p_, g_, m_, v_, e_, gn, 1.0, 1e-4, 0.9, 0.999, 1e-8, 0.001, 0.5, 0.5, 0.999), nbytes=4.0 * n * 9)

# ---- the whole CIFAR-10 train step (bench.py's workload), looped: the clock the MIX of the classes above holds
sys.path.insert(0, ROOT)
import bench                                             # noqa: E402
import v_diffusion                                       # noqa: E402
from v_diffusion.trainer import HotPathTrainer           # noqa: E402
del p_, g_, m_, v_, e_
torch.cuda.empty_cache()
model = bench.build_model(torch.device(DEV), cfg=bench.CIFAR).train()
gd = v_diffusion.GaussianDiffusion(v_diffusion.get_logsnr_schedule("cosine", -20.0, 20.0), 50, "v", "fixed_medium", "snr_trunc", "mse",
                                   intp_frac=0.3, w_guide=1.0, p_uncond=0.1)
tr = HotPathTrainer(model, gd, lr=2e-4, weight_decay=0.001, warmup=1000, grad_norm=1.0, ema_decay=0.9999, use_ema=True)
xb = torch.rand((B, 3, 32, 32), device=DEV, generator=g) * 2 - 1
yb = torch.randint(1, 11, (B,), device=DEV, generator=g).float()
NSAMP = 1200                                             # 1.2 s of stamps: ~20 whole steps
measure("WHOLE CIFAR-10 bs-128 train step", lambda: tr.step(xb, yb.clone()), flops=0.0, lead_s=2.0)
