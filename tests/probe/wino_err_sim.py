"""CPU simulation (fp32 arithmetic) of the UNet output error when the residual-block 3x3 convolutions run as Winograd F(2x2,3x3) or
F(4x4,3x3) with different interpolation points, against an fp64 evaluation of the same network.  Not shipped; a design aid."""
import sys, types
import numpy as np
import torch
import torch.nn.functional as F
sys.path.insert(0, "/root/repo")
from oracle import unet_ref
from oracle.cases import CIFAR_COND, CELEBA, make_inputs, make_weights

torch.set_num_threads(8)

def mats(points):
    """Cook-Toom matrices for F(m, 3) with the given finite points + infinity: AT (m x n), G (n x 3), BT (n x n), float64"""
    n = len(points) + 1
    m = n - 2
    P = np.array(points, dtype=np.float64)
    AT = np.zeros((m, n)); G = np.zeros((n, 3)); BT = np.zeros((n, n))
    for i, p in enumerate(P):
        AT[:, i] = p ** np.arange(m)
        Ni = np.prod([p - q for j, q in enumerate(P) if j != i])
        G[i] = p ** np.arange(3) / Ni
        poly = np.poly1d([1.0])
        for j, q in enumerate(P):
            if j != i:
                poly = poly * np.poly1d([1.0, -q])
        c = poly.coeffs[::-1]            # ascending powers
        BT[i, :len(c)] = c
    AT[m - 1, n - 1] = 1.0
    G[n - 1] = [0, 0, 1]
    poly = np.poly1d([1.0])
    for q in P:
        poly = poly * np.poly1d([1.0, -q])
    c = poly.coeffs[::-1]
    BT[n - 1, :len(c)] = c
    return AT, G, BT

def check(points):
    AT, G, BT = mats(points)
    m = AT.shape[0]; n = AT.shape[1]
    rng = np.random.default_rng(0)
    d = rng.standard_normal(n); g = rng.standard_normal(3)
    y = AT @ ((G @ g) * (BT @ d))
    ref = np.array([sum(d[i + k] * g[k] for k in range(3)) for i in range(m)])
    assert np.allclose(y, ref), (y, ref)

def rescale(AT, G, BT, scale_rows):
    """row i of BT scaled by s_i, row i of G by 1/s_i"""
    s = np.array(scale_rows, dtype=np.float64)
    return AT, G / s[:, None], BT * s[:, None]

def wino_conv(x, w, b, AT, G, BT):
    """x [B,C,H,W] fp32, w [O,C,3,3]; tiles m x m; all tile arithmetic in fp32, U rounded once from fp64"""
    m, n = AT.shape
    Bn, C, H, W = x.shape
    O = w.shape[0]
    ATt, BTt = torch.tensor(AT, dtype=torch.float32), torch.tensor(BT, dtype=torch.float32)
    U = torch.einsum("ia,ocab,jb->ijoc", torch.tensor(G), w.double(), torch.tensor(G)).float()          # [n,n,O,C]
    xp = F.pad(x, (1, 1, 1, 1))
    th, tw = H // m, W // m
    # patches [B,C,th,tw,n,n]
    pt = xp.unfold(2, n, m).unfold(3, n, m)
    V = torch.einsum("ia,bcyxae->bcyxie", BTt, pt)
    V = torch.einsum("bcyxie,je->bcyxij", V, BTt)                                                       # [B,C,th,tw,n,n]
    Vr = V.permute(4, 5, 0, 2, 3, 1).reshape(n, n, Bn * th * tw, C)
    M = torch.matmul(Vr, U.transpose(2, 3))                                                             # [n,n,T,O] fp32 accumulate
    Y = torch.einsum("ui,ijto->ujto", ATt, M)
    Y = torch.einsum("ujto,vj->uvto", Y, ATt)                                                           # [m,m,T,O]
    Y = Y.reshape(m, m, Bn, th, tw, O).permute(2, 5, 3, 0, 4, 1).reshape(Bn, O, H, W)
    return Y + b.view(1, -1, 1, 1)

MODE = {"kind": "direct"}
real_conv = F.conv2d
def conv_patch(x, w, b=None, stride=1, padding=0, *a, **k):
    if x.dtype == torch.float32 and w.shape[-1] == 3 and padding == 1 and w.shape[1] >= 32 and w.shape[0] >= 32 and MODE["kind"] != "direct":
        H = x.shape[2]
        if MODE["kind"].startswith("f43") and H >= MODE.get("minH", 16) and H % 4 == 0:
            return wino_conv(x, w, b, *MODE["m43"])
        if H % 2 == 0:
            return wino_conv(x, w, b, *MODE["m23"])
    return real_conv(x, w, b, stride, padding, *a, **k)

m23 = mats([0.0, 1.0, -1.0])
for pts in ([0, 1, -1], [0, 1, -1, 2, -2], [0, .75, -.75, 1.5, -1.5], [0, .5, -.5, 1, -1], [0, .5, -.5, 2, -2], [0, 1, -1, .5, -.5], [0, 0.625, -0.625, 1.25, -1.25]):
    check(pts)
print("matrices ok")

which = sys.argv[1] if len(sys.argv) > 1 else "cifar"
cfg = dict(CIFAR_COND if which == "cifar" else CELEBA, drop_rate=0.0)
B, R = (2, 32) if which == "cifar" else (1, 64)
sd = make_weights(cfg)
x, t, y = make_inputs(cfg, B, R, "single" if which == "cifar" else "multi", seed=5)
with torch.no_grad():
    ref = unet_ref.unet_forward(sd, cfg, x, t, y).double()      # the reference's own fp32 evaluation (1.2e-6 from fp64: SURVEY 8c)
    print("output scale", ref.abs().max().item(), "std", ref.std().item())
    unet_ref.F.conv2d = conv_patch
    def run(name, **mode):
        MODE.clear(); MODE.update(mode)
        out = unet_ref.unet_forward(sd, cfg, x, t, y)
        e = (out.double() - ref)
        print(f"{name:46s} max-abs {e.abs().max().item():.3e}  rel-L2 {(e.norm() / ref.norm()).item():.3e}", flush=True)
    run("direct fp32 (torch CPU)", kind="direct")
    run("F(2x2,3x3) everywhere", kind="f23", m23=m23)
    for nm, pts in (("classic {0,+-1,+-2}", [0, 1, -1, 2, -2]), ("{0,+-3/4,+-3/2}", [0, .75, -.75, 1.5, -1.5]), ("{0,+-1/2,+-1}", [0, .5, -.5, 1, -1]),
                    ("{0,+-1/2,+-2}", [0, .5, -.5, 2, -2]), ("{0,+-5/8,+-5/4}", [0, .625, -.625, 1.25, -1.25])):
        run(f"F(4x4,3x3) {nm} at H>=16, F(2,3) below", kind="f43", m23=m23, m43=mats(pts), minH=16)
