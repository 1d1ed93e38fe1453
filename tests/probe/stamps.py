"""In-kernel phase timing of the GEMM engine (not a test): VD_GEMM_PROBE bit 4 makes every workgroup write 100 MHz timestamps
{start, main loop done, epilogue issued, stores drained} through the colsum pointer.  python tests/probe/stamps.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "v-diffusion-torch_amd")]
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _probe_lib  # noqa: F401,E402  (loads libvdiff_hip_probe.so: the product library has no probe code)
os.environ["VD_GEMM_PROBE"] = "16"
import torch
from v_diffusion import _hip as H
DEV = "cuda"
def run(name, fn, nblocks):
    dbg = torch.zeros(nblocks * 4 * 2 + 64, dtype=torch.float32, device=DEV)
    for _ in range(3): fn(dbg)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(dbg); e1.record(); torch.cuda.synchronize()
    t = dbg.view(torch.int64)[: nblocks * 4].view(nblocks, 4).cpu().double()
    ok = (t[:, 0] > 0) & (t[:, 3] > t[:, 0])
    t = t[ok]
    kus = e0.elapsed_time(e1) * 1e3
    f = 100.0                                   # s_memrealtime: 100 MHz constant clock
    span = float(t[:, 3].max() - t[:, 0].min()) / f
    conc = float((t[:, 3] - t[:, 0]).sum()) / f / span
    pro, loop, epi = (t[:, 2] - t[:, 1]) / f, (t[:, 1] - t[:, 0]) / f, (t[:, 3] - t[:, 2]) / f
    print(f"{name}: kernel {kus:.1f} us, stamp span {span:.1f} us, mean concurrent blocks {conc:.0f} ({conc/256:.2f}/CU); per block: "
          f"epilogue ISSUE {pro.mean():.2f} (max {pro.max():.1f})  prologue+loop {loop.mean():.2f} (max {loop.max():.1f})  store DRAIN {epi.mean():.2f} (max {epi.max():.1f}) us", flush=True)

for (M, N, K, nm) in [(131072, 768, 256, "qkv@32"), (32768, 768, 256, "qkv@16"), (131072, 256, 512, "skip1x1@32")]:
    A = torch.randn(M, K, device=DEV); B = torch.randn(N, K, device=DEV); C = torch.empty(M, N, device=DEV)
    run(nm, lambda dbg: H.gemm(A, B, C, M, N, K, lda=K, ldb=K, ldc=N, colsum=dbg), (M // 128) * (N // 128))
x = torch.randn(128, 32, 32, 256, device=DEV); w = torch.randn(256, 9, 256, device=DEV) * 0.02; y = torch.empty(128, 32, 32, 256, device=DEV)
import ctypes as C_
def conv(dbg):
    d = H.GemmDesc()
    d.A, d.B, d.C = H.ptr(x), H.ptr(w), H.ptr(y)
    d.M, d.N, d.K, d.a_kind, d.b_kind = 131072, 256, 2304, 2, 0
    d.lda, d.ldb, d.ldc = 256, 2304, 256
    d.batch, d.nh, d.alpha = 1, 1, 1.0
    d.H, d.W, d.Cin = 32, 32, 256
    d.colsum = H.ptr(dbg)
    H._check(H.lib().vd_gemm(C_.byref(d), H.stream()), "vd_gemm")
run("conv256@32", conv, 2048)
