"""Where a wave's main-loop cycles go (not a test): VD_GEMM_PROBE=32 accumulates core-clock cycles spent issuing the tile DMA,
in the LDS-read + MFMA section, and waiting (vmcnt + barrier), per workgroup.  python tests/probe/loop_phases.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "v-diffusion-torch_amd")]
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _probe_lib  # noqa: F401,E402  (loads libvdiff_hip_probe.so: the product library has no probe code)
os.environ["VD_GEMM_PROBE"] = "32"
import ctypes as C_
import torch
from v_diffusion import _hip as H
DEV = "cuda"
B, R, Cc = 128, 32, 256
x = torch.randn(B, R, R, Cc, device=DEV); w = torch.randn(Cc, 9, Cc, device=DEV) * 0.02; y = torch.randn(B, R, R, Cc, device=DEV)

def run(name, fill, nblocks):
    dbg = torch.zeros(nblocks * 32 + 64, dtype=torch.float32, device=DEV)
    for _ in range(4):
        d = H.GemmDesc(); fill(d); d.stats = H.ptr(dbg); d.stats_hw = 1024
        H._check(H.lib().vd_gemm(C_.byref(d), H.stream()), "vd_gemm")
    torch.cuda.synchronize()
    tw = dbg.view(torch.int64)[: nblocks * 16].view(nblocks, 4, 4).cpu().double()
    for wv in range(4):
        tt = tw[:, wv]
        tt = tt[tt[:, 3] > 0]
        print(f"   wave {wv}: vmcnt {float((tt[:,0]/tt[:,3]).mean()):.0f}  lds+mfma {float((tt[:,1]/tt[:,3]).mean()):.0f}  barrier {float((tt[:,2]/tt[:,3]).mean()):.0f}")
    t = tw[:, 0]
    t = t[t[:, 3] > 0]
    tot = t[:, :3].sum(1)
    print(f"{name}: blocks {len(t)}, tiles/block {t[:,3].mean():.0f}; cycles per tile: vmcnt-barrier {float((t[:,0]/t[:,3]).mean()):.0f}  "
          f"lds+mfma {float((t[:,1]/t[:,3]).mean()):.0f}  barrier {float((t[:,2]/t[:,3]).mean()):.0f}  (shares {100*float((t[:,0]/tot).mean()):.1f} / "
          f"{100*float((t[:,1]/tot).mean()):.1f} / {100*float((t[:,2]/tot).mean()):.1f} %)", flush=True)

def conv(d):
    d.A, d.B, d.C = H.ptr(x), H.ptr(w), H.ptr(y)
    d.M, d.N, d.K, d.a_kind, d.b_kind = B * R * R, Cc, 9 * Cc, 2, 0
    d.lda, d.ldb, d.ldc = Cc, 9 * Cc, Cc
    d.batch, d.nh, d.alpha = 1, 1, 1.0
    d.H, d.W, d.Cin = R, R, Cc
ws = torch.empty(64 * (Cc * 9 * Cc + Cc), device=DEV)
def wgrad(d):
    d.A, d.B, d.C = H.ptr(y), H.ptr(x), H.ptr(ws)
    d.M, d.N, d.K, d.a_kind, d.b_kind = Cc, 9 * Cc, B * R * R, 1, 2
    d.lda, d.ldb, d.ldc = Cc, Cc, 9 * Cc
    d.batch, d.nh, d.alpha = 1, 1, 1.0
    d.H, d.W, d.Cin = R, R, Cc
    d.splitk, d.ws, d.ws_bytes, d.tile = 28, ws.data_ptr(), ws.numel() * 4, 128
run("conv fwd 256->256 @32 (KT16)", conv, 2048)
run("conv wgrad 256->256 @32 (KT16, 28 slabs)", wgrad, 2 * 18 * 28)
