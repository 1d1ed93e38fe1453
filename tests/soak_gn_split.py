"""Child of tests/test_multigpu_path_gpu.py::test_groupnorm_sibling_spin_under_co_resident_kernels (not a test by itself).

Runs STEPS CelebA train steps at B = 64 (the per-rank batch of BASELINE configs[4]; its 64x64 GroupNorm backward launches take the
spin-synchronised SPLIT form) from fixed weights on fixed inputs, dropout off, and prints one JSON line: a checksum of every parameter
and of the EMA shadow, the loss sequence, and the longest sibling wait any SPLIT launch saw (probe library: vd_gn_set_spin_probe).
Environment decides what runs beside the GroupNorm kernels: VD_SOAK_REDUCER=1 puts a 1-rank RCCL group's bucketed all-reduces into
every backward pass (collective kernels launched mid-backward), VD_RESERVE_CUS the CUs left to them, VD_WGRAD_STREAM the side stream."""
import hashlib
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "v-diffusion-torch_amd")]
import v_diffusion                                      # noqa: E402
from v_diffusion import _hip                            # noqa: E402
from v_diffusion.trainer import HotPathTrainer          # noqa: E402
from oracle import detrand                              # noqa: E402
from oracle.cases import CELEBA, make_inputs, make_weights   # noqa: E402

STEPS = int(os.environ.get("VD_SOAK_STEPS", "50"))
B = int(os.environ.get("VD_SOAK_BATCH", "64"))
dev = torch.device("cuda", 0)
cfg = dict(CELEBA, drop_rate=0.0)
model = v_diffusion.UNet(**cfg)
model.load_state_dict(make_weights(cfg))
model.to(dev).train()
gd = v_diffusion.GaussianDiffusion(v_diffusion.get_logsnr_schedule("cosine", -20.0, 20.0), 50, "v", "fixed_medium", "snr_trunc", "mse",
                                   intp_frac=0.3, w_guide=1.0, p_uncond=0.0)
tr = HotPathTrainer(model, gd, lr=2e-4, weight_decay=0.001, warmup=0, grad_norm=1.0, ema_decay=0.9999, use_ema=True, rank=0, world_size=1)
if os.environ.get("VD_SOAK_REDUCER") == "1":
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29571")
    dist.init_process_group("nccl", init_method="env://", world_size=1, rank=0)
    tr.reducer.active = True
x0, t, y = make_inputs(cfg, B, 64, "multi", seed=41)
x0 = x0.clamp(-1, 1).to(dev)
t, y = t.to(dev), y.to(dev)
noise = detrand.normal("noise", tuple(x0.shape), 41).to(dev)
spin = torch.zeros(1, dtype=torch.int32, device=dev)
has_probe = hasattr(_hip.lib(), "vd_gn_set_spin_probe")
if has_probe:
    _hip.lib().vd_gn_set_spin_probe(spin.data_ptr())
losses = []
for s in range(STEPS):
    loss = tr.step(x0, y.clone(), t=t.clone(), noise=noise)
    losses.append(float(loss))
torch.cuda.synchronize()
if has_probe:
    _hip.lib().vd_gn_set_spin_probe(None)


def digest(tensors):
    h = hashlib.sha256()
    for v in tensors:
        h.update(v.detach().contiguous().cpu().numpy().tobytes())
    return h.hexdigest()


finite = all(bool(torch.isfinite(p).all()) for p in model.parameters())
print(json.dumps({"params": digest([tr.flat.p]), "ema": digest([tr.flat.ema]),
                  "losses": losses, "finite": finite, "max_spin": int(spin.item()) if has_probe else None,
                  "reducer": bool(tr.reducer.active), "reserved_cus": int(_hip.lib().vd_reserved_cus())}))
if dist.is_initialized():
    dist.destroy_process_group()
