"""PMC target (program directly after `rocprofv3 --pmc ... --`): the dominant Winograd kernels of the CIFAR bs-128 train step at
their bench launches (256->256 @32x32) -- wino43_conv_kernel<8, true> (F(4x4,3x3) forward + bias + residual + GroupNorm partials: what the
step runs since round 4), wino43_conv_kernel<8, false> (its input gradient), wino_conv_wide_kernel<16, 640, true> (the F(2x2,3x3)
forward: VD_WINO43_FWD=0 and the 8x8 level's sibling), wino_wgrad_kernel<16, true> and the F(4x4,3x3) weight-gradient path -- 6 launches
each, through the PRODUCT library."""
import os, sys, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "v-diffusion-torch_amd"))
from v_diffusion import _hip as H
DEV = "cuda"
nimg, Hh, Ww, Cin, Cout = 128, 32, 32, 256, 256
x = torch.nn.functional.silu(torch.randn(nimg, Hh, Ww, Cin, device=DEV))
w = torch.randn(Cout, Cin, 3, 3, device=DEV) * (9 * Cin) ** -0.5
b = torch.randn(Cout, device=DEV)
res = torch.randn(nimg, Hh, Ww, Cout, device=DEV)
dy = torch.randn(nimg, Hh, Ww, Cout, device=DEV) * 0.05
uf = torch.empty(16, Cout, Cin, device=DEV)
H.wino_pack(w, Cout, Cin, uf=uf)
y = torch.empty(nimg, Hh, Ww, Cout, device=DEV)
part = torch.empty(H.stats_part_numel(nimg, Hh * Ww, Cout), device=DEV)
u43 = torch.empty(H.lib().vd_wino43_u_floats(Cout, Cin), device=DEV)
H.wino43_pack(w, Cout, Cin, u43)
u43f = torch.empty(H.lib().vd_wino43_u_floats(Cout, Cin), device=DEV)
H.wino43_pack_fwd(w, Cout, Cin, u43f)
dx = torch.empty(nimg, Hh, Ww, Cin, device=DEV)
dw = torch.empty(Cout, Cin, 3, 3, device=DEV)
db = torch.empty(Cout, device=DEV)
for _ in range(6):
    H.conv3x3_wino(x, Cin, uf, b, y, Cout, nimg, Hh, Ww, Cin, Cout, res=res, ldres=Cout, stats_part=part)
    H.conv3x3_wino43_fwd(x, Cin, u43f, b, y, Cout, nimg, Hh, Ww, Cin, Cout, res=res, ldres=Cout, stats_part=part)
    H.conv3x3_wgrad_wino(x, Cin, dy, Cout, nimg, Hh, Ww, Cin, Cout, dw, Cin, Cout, dbias=db)        # fused F(2x2,3x3) kernel (8x8 layers, small batches)
    H.conv3x3_wgrad_wino43(x, Cin, dy, Cout, nimg, Hh, Ww, Cin, Cout, dw, Cin, Cout, dbias=db)      # what the step runs at this shape
    H.conv3x3_dgrad_wino43(dy, Cout, u43, dx, Cin, nimg, Hh, Ww, Cin, Cout)
torch.cuda.synchronize()
