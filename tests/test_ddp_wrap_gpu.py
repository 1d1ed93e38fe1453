"""Row a14 (SURVEY 8a-14, section 5 option (1)): the reference wraps the model in ``DDP`` (train.py:141-148) and relies on its hooks
firing as gradients become ready during ``loss.backward()`` (train_utils.py:154).  ``v_diffusion.UNet`` returns a chain of autograd
nodes cut at the engine's progress points (models/unet.py::_SegFn), so DDP's buckets leave while the rest of backward still runs.
Checked here on a 1-rank RCCL group (the build's boxes have one GPU): at least four buckets are handed to the collective before the
network's backward pass has ended, and every gradient is bitwise the single-node form's."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("config,batch,segments,params,nparam,port", [("cifar10", 16, 9, 414, 60_806_403, "29561"),
                                                                      ("celeba", 4, 11, 572, 266_825_859, "29563")])
def test_ddp_wrapped_unet_hands_buckets_over_during_backward(tmp_path, config, batch, segments, params, nparam, port):
    """CIFAR-10 model (three levels: 9 nodes, 243 MB of gradients) and the CelebA model (four levels, multitag class embedding: 11 nodes, 1.07 GB)"""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    out = str(tmp_path / "ddp.json")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_PORT=port)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "ddp_wrap_worker.py"), "--out", out, "--batch", str(batch), "--config", config],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    rep = json.load(open(out))
    print(json.dumps(rep))
    if config == "cifar10":
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        json.dump(rep, open(os.path.join(ROOT, "gpurun_out", "ddp_wrap_overlap.json"), "w"))
    assert rep["segments"] == segments and rep["params"] == params and rep["grad_bytes"] >= 4 * nparam
    assert rep["bitwise_chain_vs_single_node"], "the chain of nodes changed a gradient"
    assert rep["bitwise_ddp_vs_single_node"], f"DDP-wrapped gradients differ from the single-node form: {rep['mismatch']}"
    for it in rep["iterations"][1:]:                          # (after DDP rebuilt its buckets in arrival order)
        assert it["buckets"] >= 8, it                          # 243 MB / 1.07 GB in 25 MiB buckets
        assert it["buckets_while_backward_runs"] >= 4, f"DDP saw the gradients only at the end of backward: {it}"
        assert it["bytes_while_backward_runs"] >= 0.5 * rep["grad_bytes"], it
        assert abs(it["loss"] - rep["loss_single_node"]) <= 1e-6 * abs(rep["loss_single_node"])
