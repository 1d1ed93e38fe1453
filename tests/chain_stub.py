"""Stand-in for the UNet behind the chain of autograd nodes (v_diffusion/models/unet.py::_SegFn, _BackwardRun) on CPU: three
parameter segments and a generator in place of UNetEngine.backward_steps that logs what ran when.  Used by tests/test_host_cpu.py (the
mechanics) and tests/test_multirank_cpu.py (the reference's DDP(model) wrap, train.py:141-148, over two gloo ranks)."""
import torch


class ChainStub(torch.nn.Module):
    """Parameters in three segments + the three hooks _BackwardRun needs; its "backward pass" is a generator that fills the gradients
    segment by segment and logs what ran when (the real one is UNetEngine.backward_steps)."""

    def __init__(self, scale=1):
        super().__init__()
        self.a = torch.nn.Parameter(torch.ones(3 * scale))
        self.b = torch.nn.Parameter(torch.ones(2 * scale))
        self.c = torch.nn.Parameter(torch.ones(4 * scale))
        self.num_classes = 0
        self.log = []
        self._active = None
        self.segs = [("c", ["c"]), ("b", ["b"]), (None, ["a"])]          # backward order

    def _chain_begin(self, run, tape, dout, need_dx):
        other = self._active() if self._active is not None else None     # (a weak reference, as models/unet.py::_chain_begin holds it)
        if other is not None and other is not run and not other.finished:
            if other.task_id == run.task_id:
                other.drain()
            else:
                other.stop_early()
        G = {k: torch.empty_like(p) for k, p in self.named_parameters()}
        s = float(dout.sum())

        def gen():
            try:
                self.log.append("kernels c")
                G["c"].fill_(1 * s)
                yield "c"
                self.log.append("kernels b-part")
                yield "not a boundary"                                  # finer progress points are passed over
                G["b"].fill_(2 * s)
                self.log.append("kernels b")
                yield "b"
                self.log.append("kernels a")
                G["a"].fill_(3 * s)
                yield None
                return dout * 5 if need_dx else None
            finally:
                self.log.append("closed")
        import weakref
        self._active = weakref.ref(run)
        return gen(), G, False

    def _chain_end(self, run):
        if self._active is not None and self._active() in (run, None):
            self._active = None

    @property
    def active(self):
        return self._active() if self._active is not None else None

    def _chain_token(self, device):
        return torch.zeros(1)

    def forward(self, x):
        from v_diffusion.models.unet import _BackwardRun, _SegFn
        run = _BackwardRun(self, {"tape": 1}, x.requires_grad, self.segs, x.device)
        run.out = x.detach() * 2
        named = dict(self.named_parameters())
        carrier = x
        for j in range(len(self.segs) - 1, -1, -1):
            carrier = _SegFn.apply(run, j, carrier, *[named[k] for k in self.segs[j][1]])
        return carrier
