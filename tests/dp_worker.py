"""Worker of tests/test_multigpu_path_gpu.py::test_two_ranks_equal_one_rank_on_the_full_batch: one data-parallel rank of
HotPathTrainer.step (reference Trainer.step under DDP: train_utils.py:137-169, train.py:141-148).  Launched by
`python -m torch.distributed.run --nproc-per-node W`; every rank uses the ONE visible GPU, collectives over gloo (RCCL refuses two
ranks on one device).  Rank r takes rows [r B/W, (r+1) B/W) of a fixed batch with injected t / noise, runs --steps updates and the
leader writes parameters, EMA shadow, first-step gradient buffer and the per-step losses it reports to --out."""
import argparse
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "v-diffusion-torch_amd")]


def fixed_batch(cfg, B, R, steps):
    from oracle import detrand
    from oracle.cases import make_inputs
    out = []
    for s in range(steps):
        x0, t, y = make_inputs(cfg, B, R, "single", seed=31 + s)
        out.append((x0.clamp(-1, 1), t, y.clamp(min=1.0), detrand.normal("noise", tuple(x0.shape), 31 + s)))
    return out


def run(rank, world, B, steps, dev="cuda", accum=1, plan=None):
    """plan (or None): per step one letter per SHARD of the batch (shard = rank * accum + micro-batch; world * accum letters): 'L' = the shard
    is stepped with its labels, 'N' = with y = None (no class embedding at all: reference unet.py:289)"""
    import v_diffusion
    from v_diffusion.trainer import HotPathTrainer
    from oracle.cases import CIFAR_COND, make_weights
    cfg = dict(CIFAR_COND, drop_rate=0.0)
    model = v_diffusion.UNet(**cfg)
    model.load_state_dict(make_weights(cfg))
    model.to(dev).train()
    gd = v_diffusion.GaussianDiffusion(v_diffusion.get_logsnr_schedule("cosine", -20.0, 20.0), 50, "v", "fixed_medium", "snr_trunc",
                                       "mse", intp_frac=0.3, w_guide=1.0, p_uncond=0.0)
    tr = HotPathTrainer(model, gd, lr=2e-4, warmup=0, grad_norm=1.0, use_ema=True, rank=rank, world_size=world, num_accum=accum)
    p0 = tr.flat.p.detach().cpu().clone()
    per = B // world
    mb = per // accum
    losses, g1 = [], None
    for s, (x0, t, y, noise) in enumerate(fixed_batch(cfg, B, 32, steps)):
        # --num-accum (train.py:292, train_utils.py:154,257): the rank's rows in `accum` micro-batches, loss / accum each, gradients summed,
        # the optimizer runs with the last one
        part = []
        for a in range(accum):
            rows = slice(rank * per + a * mb, rank * per + (a + 1) * mb)
            labelled = plan is None or plan[s][rank * accum + a] == "L"
            part.append(tr.step(x0[rows].to(dev), y[rows].to(dev) if labelled else None, update=a == accum - 1, t=t[rows].to(dev),
                                noise=noise[rows].to(dev)))
        losses.append(float(sum(part) / accum))
        if s == 0:
            g1 = tr.flat.g.detach().cpu().clone()
    torch.cuda.synchronize()
    return dict(p0=p0, p=tr.flat.p.detach().cpu(), ema=tr.flat.ema.detach().cpu(), g1=g1, losses=losses,
                buckets=len(tr.reducer.bounds), reducer_active=tr.reducer.active, cls_steps=tr.flat.cls_steps, steps=tr.flat.step_count)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--accum", type=int, default=1)
    ap.add_argument("--plan", default="", help="e.g. LN,NN,NL: per step, which shards pass labels (L) and which y = None (N)")
    a = ap.parse_args()
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    res = run(rank, world, a.batch, a.steps, accum=a.accum, plan=a.plan.split(",") if a.plan else None)
    # every rank must hold the same replica after the update (DDP invariant)
    ref = res["p"].clone().cuda()
    dist.broadcast(ref, src=0)
    same = torch.equal(ref.cpu(), res["p"])
    flags = [None] * world
    dist.all_gather_object(flags, (same, res["cls_steps"]))
    if rank == 0:
        res["replicas_identical"] = all(f[0] for f in flags)
        res["cls_steps_per_rank"] = [f[1] for f in flags]
        torch.save(res, a.out)
    dist.destroy_process_group()
