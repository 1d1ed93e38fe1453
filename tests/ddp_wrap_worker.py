"""Worker of tests/test_ddp_wrap_gpu.py: the reference's own data-parallel wrap -- ``DDP(model, device_ids=[local_rank])`` around the
UNet (train.py:141-148), loss.backward() (train_utils.py:154) -- on a 1-rank RCCL group, with a communication hook that notes, for every
gradient bucket DDP hands to the collective, whether the network's backward pass was still running.  Then the same step through the
single-node form (VD_AUTOGRAD_CHAIN=0's code path) for a bitwise comparison of every gradient.  Writes a JSON report to --out."""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "v-diffusion-torch_amd")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--config", default="cifar10")
    a = ap.parse_args()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29561")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method="env://", world_size=1, rank=0)
    import v_diffusion
    from v_diffusion import _hip
    from torch.nn.parallel import DistributedDataParallel as DDP
    from torch.distributed.algorithms.ddp_comm_hooks import default_hooks
    from oracle import detrand
    from oracle.cases import CIFAR_COND, CELEBA, make_inputs, make_weights
    cfg = dict(CIFAR_COND if a.config == "cifar10" else CELEBA, drop_rate=0.0)
    R = 32 if a.config == "cifar10" else 64
    model = v_diffusion.UNet(**cfg)
    model.load_state_dict(make_weights(cfg))
    model.cuda().train()
    gd = v_diffusion.GaussianDiffusion(v_diffusion.get_logsnr_schedule("cosine", -20.0, 20.0), 50, "v", "fixed_medium", "snr_trunc",
                                       "mse", intp_frac=0.3, w_guide=1.0, p_uncond=0.0)
    x0, t, y = make_inputs(cfg, a.batch, R, "multi" if cfg.get("multitags") else "single", seed=5)
    x0, noise = x0.clamp(-1, 1).cuda(), detrand.normal("noise", tuple(x0.shape), 5).cuda()
    t = t.cuda()
    y = y.cuda() if cfg.get("multitags") else y.clamp(min=1.0).cuda()
    eng = model.engine()
    ddp = DDP(model, device_ids=[0])
    events = []

    def hook(state, bucket):
        events.append(dict(index=bucket.index(), bytes=bucket.buffer().numel() * 4, backward_running=eng._active_run is not None,
                           t=time.perf_counter(), last=bucket.is_last()))
        return default_hooks.allreduce_hook(state, bucket)
    ddp.register_comm_hook(None, hook)

    def step(net):
        for p in model.parameters():
            p.grad = None
        loss = gd.train_loss(net, x_0=x0, t=t, y=y.clone(), noise=noise).mean()
        t0 = time.perf_counter()
        loss.backward()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        return float(loss), t0, t1

    report = {"iterations": []}
    for it in range(3):             # (iteration 0: DDP's initial bucket plan; it rebuilds the buckets in arrival order afterwards)
        events.clear()
        loss, t0, t1 = step(ddp)
        report["iterations"].append(dict(loss=loss, buckets=len(events), buckets_while_backward_runs=sum(e["backward_running"] for e in events),
                                         bytes_while_backward_runs=sum(e["bytes"] for e in events if e["backward_running"]),
                                         bucket_ms_after_backward_call=[round((e["t"] - t0) * 1e3, 2) for e in events],
                                         backward_call_ms=round((t1 - t0) * 1e3, 2)))
    g_ddp = {k: p.grad.detach().clone() for k, p in model.named_parameters()}
    # the chain without DDP, then the single-node form: all three must agree bit for bit (a 1-rank mean is the identity)
    step(model)
    g_chain = {k: p.grad.detach().clone() for k, p in model.named_parameters()}
    assert eng._active_run is None
    _hip.AUTOGRAD_CHAIN = False
    loss1, _, _ = step(model)
    _hip.AUTOGRAD_CHAIN = True
    g_one = {k: p.grad.detach().clone() for k, p in model.named_parameters()}
    report["loss_single_node"] = loss1
    report["segments"] = len(eng.grad_segments())
    report["params"] = len(g_one)
    report["grad_bytes"] = 4 * sum(v.numel() for v in g_one.values())
    report["bitwise_ddp_vs_single_node"] = all(torch.equal(g_ddp[k], g_one[k]) for k in g_one)
    report["bitwise_chain_vs_single_node"] = all(torch.equal(g_chain[k], g_one[k]) for k in g_one)
    report["mismatch"] = [k for k in g_one if not torch.equal(g_ddp[k], g_one[k])][:8]
    json.dump(report, open(a.out, "w"))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
