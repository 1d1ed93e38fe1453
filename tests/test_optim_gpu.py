"""v_diffusion.optim (round-5 review item 7): the two-line opt-in that puts the fused optimizer kernels under the REFERENCE's own training loop
(train.py:158 ``AdamW(model.parameters(), ...)``, train_utils.py:159-168 ``clip_grad_norm_ -> optimizer.step -> zero_grad -> scheduler.step ->
ema.update``, utils.py:123-190 ``EMA``).  Checked against that loop written with torch's own tools on a second copy of the same model."""
import copy
import math
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = "cuda"


class RefEMA:
    """utils.py:123-190, restated: per-parameter shadow, decay = min(decay, (1 + n) / (10 + n))"""

    def __init__(self, model, decay):
        self.named = [(k, v) for k, v in model.named_parameters() if v.requires_grad]
        self.shadow = {k: v.detach().clone() for k, v in self.named}
        self.decay, self.num_updates = decay, 0

    def update(self):
        self.num_updates += 1
        d = min(self.decay, (1 + self.num_updates) / (10 + self.num_updates))
        with torch.no_grad():
            for k, v in self.named:
                self.shadow[k] += (1 - d) * (v.data - self.shadow[k])


def _models():
    import v_diffusion
    from oracle.cases import TINY, make_inputs, make_weights
    from oracle import detrand
    case = TINY["tinyA"]
    cfg = dict(case["cfg"], drop_rate=0.0)
    sd = make_weights(cfg)
    ms = []
    for _ in range(2):
        m = v_diffusion.UNet(**cfg)
        m.load_state_dict({k: v.clone() for k, v in sd.items()})
        ms.append(m.to(DEV).train())
    gd = v_diffusion.GaussianDiffusion(v_diffusion.get_logsnr_schedule("cosine", -20.0, 20.0), 50, "v", "fixed_medium", "snr_trunc", "mse",
                                       intp_frac=0.3, w_guide=1.0, p_uncond=0.0)
    batches = []
    for s in range(5):
        x0, t, y = make_inputs(cfg, 4, case["R"], case["label"], seed=50 + s)
        batches.append((x0.clamp(-1, 1).to(DEV), t.to(DEV), y.to(DEV), detrand.normal("noise", tuple(x0.shape), 50 + s).to(DEV)))
    return ms, gd, batches


def test_fused_adamw_and_ema_follow_the_torch_loop():
    """five updates of the reference's step sequence, the third with y = None (the class embedding receives no gradient: torch.optim.AdamW skips
    it and its step count lags afterwards): parameters, Adam moments and the EMA shadow of the fused pair must equal those of torch.optim.AdamW +
    a per-parameter EMA (relative L2 <= 1e-6; both sides see gradients of the same kernels, so what differs is the optimizer arithmetic only)"""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from v_diffusion.optim import FusedAdamW, EMA
    (ma, mb), gd, batches = _models()
    kw = dict(lr=2e-3, betas=(0.9, 0.999), weight_decay=0.01)
    oa = torch.optim.AdamW(ma.parameters(), **kw)
    ema_b = EMA(mb, decay=0.9999)                               # built BEFORE the optimizer, as a caller might: it moves over at the first update
    ob = FusedAdamW(mb.parameters(), **kw)
    ema_a = RefEMA(ma, 0.9999)
    sa = torch.optim.lr_scheduler.LambdaLR(oa, lr_lambda=lambda t: min((t + 1) / 3, 1.0))
    sb = torch.optim.lr_scheduler.LambdaLR(ob, lr_lambda=lambda t: min((t + 1) / 3, 1.0))
    in_slot = []
    for s, (x0, t, y, noise) in enumerate(batches):
        yy = None if s == 2 else y
        for m in (ma, mb):
            gd.train_loss(m, x_0=x0, t=t.clone(), y=None if yy is None else yy.clone(), noise=noise).mean().backward()
        base = ob.g.data_ptr()
        in_slot.append(all(p.grad is None or p.grad.data_ptr() == base + 4 * off for p, off in zip(ob._params, ob._offs)))
        # the optimizers are compared on IDENTICAL gradients (copied in place, so they stay in the fused optimizer's slots): the networks'
        # own gradients agree only to the conditioning of each tensor once the parameters differ in the last bit
        with torch.no_grad():
            for pa, pb in zip(ma.parameters(), mb.parameters()):
                assert (pa.grad is None) == (pb.grad is None)
                if pa.grad is not None:
                    pb.grad.copy_(pa.grad)
        for m, o, sch, ema in ((ma, oa, sa, ema_a), (mb, ob, sb, ema_b)):
            torch.nn.utils.clip_grad_norm_(m.parameters(), max_norm=1.0)
            o.step()
            o.zero_grad(set_to_none=True)
            sch.step()
            ema.update()
    torch.cuda.synchronize()
    assert all(in_slot), f"gradients were copied into the flat buffer instead of being produced in it: {in_slot}"
    rel = lambda a, b: ((a.double() - b.double()).norm() / max(b.double().norm().item(), 1e-30)).item()
    worst = 0.0
    sda = oa.state_dict()["state"]
    sdb = ob.state_dict()["state"]
    for i, ((k, pa), (_, pb)) in enumerate(zip(ma.named_parameters(), mb.named_parameters())):
        tol = 1e-6 * max(pa.abs().max().item(), 1e-3) + 2e-4 * 2e-3            # 1e-6 of the tensor's scale + 2e-4 of one step (lr = 2e-3)
        e = (pb - pa).abs().max().item()
        worst = max(worst, e / tol)
        assert e <= tol, f"{k}: parameters differ by {e:.2e} (tolerance {tol:.2e})"
        assert (ema_b.shadow[k] - ema_a.shadow[k]).abs().max().item() <= tol, f"{k}: EMA shadow differs"
        assert int(float(sdb[i]["step"])) == int(float(sda[i]["step"])), (k, sdb[i]["step"], sda[i]["step"])
        if pa.abs().max() > 0:
            assert rel(sdb[i]["exp_avg"], sda[i]["exp_avg"]) <= 1e-5 and rel(sdb[i]["exp_avg_sq"], sda[i]["exp_avg_sq"]) <= 1e-4, k
    steps = {int(float(v["step"])) for v in sda.values()}
    assert steps == {4, 5}, steps                                # the class embedding skipped one update on both sides
    print(f"5 updates (one without labels): worst parameter difference fused vs torch.optim.AdamW = {worst:.2f} of the tolerance")
    # EMA context manager: the module runs on the shadow inside, on the weights again outside (utils.py:151-166)
    w0 = {k: v.detach().clone() for k, v in mb.named_parameters()}
    with ema_b:
        for k, v in mb.named_parameters():
            assert torch.equal(v.detach(), ema_b.shadow[k])
    for k, v in mb.named_parameters():
        assert torch.equal(v.detach(), w0[k])
    # checkpoint round trip in torch.optim.AdamW's format (train_utils.py:317-331)
    oc = FusedAdamW([torch.nn.Parameter(p.detach().clone()) for p in mb.parameters()], **kw)
    oc.load_state_dict(ob.state_dict())
    assert oc.steps == ob.steps and oc.lag_steps == ob.lag_steps and oc.lag_range == ob.lag_range
    assert torch.equal(oc.m, ob.m) and torch.equal(oc.v, ob.v)


def test_gradient_slots_are_not_reused_while_a_gradient_is_alive():
    """models/unet.py::_grad_targets hands autograd views of persistent slots; a slot whose previous view is still referenced (gradient
    accumulation without zero_grad; a caller that kept ``param.grad``) must not be overwritten by the next backward"""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    (ma, mb), gd, batches = _models()
    x0, t, y, noise = batches[0]
    x1, t1, y1, n1 = batches[1]
    # two backward passes accumulated into .grad == the sum of the two separate gradients (the second pass must not write into the first's slots)
    gd.train_loss(ma, x_0=x0, t=t.clone(), y=y.clone(), noise=noise).mean().backward()
    g0 = {k: p.grad.detach().clone() for k, p in ma.named_parameters()}
    gd.train_loss(ma, x_0=x1, t=t1.clone(), y=y1.clone(), noise=n1).mean().backward()
    acc = {k: p.grad.detach().clone() for k, p in ma.named_parameters()}
    kept = {k: p.grad for k, p in ma.named_parameters()}                    # (references a caller holds on to across zero_grad)
    ma.zero_grad(set_to_none=True)
    gd.train_loss(ma, x_0=x1, t=t1.clone(), y=y1.clone(), noise=n1).mean().backward()
    g1 = {k: p.grad.detach().clone() for k, p in ma.named_parameters()}
    gmax = max(v.abs().max().item() for v in acc.values())
    for k in g0:
        assert torch.allclose(acc[k], g0[k] + g1[k], rtol=1e-4, atol=1e-6 * gmax), k     # accumulated == sum of the separate gradients
        assert torch.equal(kept[k], acc[k]), k                                                # the kept tensors were not overwritten by the third pass
    # steady state: with .grad reset every step the same storage is handed out again (no new allocation)
    ptrs = []
    del kept
    for _ in range(3):
        ma.zero_grad(set_to_none=True)
        gd.train_loss(ma, x_0=x0, t=t.clone(), y=y.clone(), noise=noise).mean().backward()
        ptrs.append([p.grad.data_ptr() for p in ma.parameters()])
    assert ptrs[1] == ptrs[2]
