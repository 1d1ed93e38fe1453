import os, sys, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "v-diffusion-torch_amd"))
from v_diffusion import _hip as H
DEV="cuda"
nimg,Hh,Ww,Cin,Cout=128,32,32,256,256
x=torch.randn(nimg,Hh,Ww,Cin,device=DEV); w=torch.randn(Cout,Cin,3,3,device=DEV)*(9*Cin)**-0.5
uf=torch.empty(16,Cout,Cin,device=DEV); H.wino_pack(w,Cout,Cin,uf=uf)
y=torch.empty(nimg,Hh,Ww,Cout,device=DEV)
for _ in range(4): H.conv3x3_wino(x,Cin,uf,None,y,Cout,nimg,Hh,Ww,Cin,Cout)
torch.cuda.synchronize()
