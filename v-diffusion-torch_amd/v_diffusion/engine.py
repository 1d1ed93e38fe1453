"""Explicit forward/backward engine of the UNet on the HIP kernels (no autograd graph inside, no tracing compiler).

Replaces what ATen/cuDNN/cuBLAS + autograd do for reference ``UNet.forward`` (v_diffusion/models/unet.py:286-322),
``ResidualBlock.forward`` (:137-148) and ``AttentionBlock.forward`` (:55-81).

Data layout in HBM
  * activations: NHWC fp32 ``[B, H, W, C]`` torch buffers; a tensor handed between ops is a *view* whose
    ``stride(2)`` (= per-pixel stride ``ld``) may exceed C.  The concat of reference unet.py:315 is never
    materialised: each up-block owns one ``[B,H,W,Ch+Cs]`` buffer, the producer of ``h`` writes channels
    ``[0,Ch)`` and the producer of the skip tensor (a down-block, run much earlier) writes ``[Ch,Ch+Cs)``
    directly from their conv epilogues; gradients flow back the same way (the down-block's input gradient is
    accumulated into the skip slice of the up-block's input gradient).
  * weights stay in the reference's OIHW / (out,in) storage; the 3x3 kernels of the residual blocks are transformed per call
    into the Winograd domain by ONE launch for the whole network -- U = G w G^T as [16][Cout][Cin] (forward), [16][Cin][Cout]
    (F(2x2,3x3) input gradient) or the 36-plane lane-ordered image of csrc/wino43.hip (F(4x4,3x3) input gradient): 0.2 ms per
    train step, always coherent with whatever mutated the parameters (optimizer, load_state_dict, EMA swap through ``.data``).
    The direct [Cout][tap][Cin] / [Cin][tap'][Cout] packs exist only for geometries the Winograd kernels decline.
  * the tape (what backward needs) is a python list of per-block dicts of buffers; with 288 GB of HBM nothing is
    recomputed except the dropout mask (counter-based Philox, regenerated from (seed, element index)).
"""
import math

import torch

from . import _hip as H

GROUPS = 32
EPS = 1e-6
THIN = 8          # in/out convolutions with at most this many channels on the thin side take the GEMM + tap kernels


def _ld(t):
    return t.stride(2)


def _chk(t):
    B, Hh, Ww, C = t.shape
    ld = t.stride(2)
    assert t.stride(3) == 1 and t.stride(1) == Ww * ld and t.stride(0) == Hh * Ww * ld, "not an NHWC view"
    return B, Hh, Ww, C, ld


def _splitk(M, N, K):
    """split-K factor for the small-output weight-gradient GEMMs (K = batch*pixels)."""
    tiles = ((M + 63) // 64) * ((N + 63) // 64)
    s = max(1, min(64, 1024 // max(tiles, 1), K // 256))
    return s


class _Blk:
    __slots__ = ("prefix", "kind", "cin", "cout", "rs", "attn", "consumes", "res", "att", "level", "dest", "src_hs", "push_hs",
                 "ch_h")


class UNetEngine:
    """Static execution plan + forward/backward drivers for one ``UNet`` module."""

    def __init__(self, model):
        self.m = model
        hid, mult, nrb = model.hid_channels, list(model.ch_multipliers), model.num_res_blocks
        levels = len(mult)
        chs = [hid * k for k in mult]
        plan = []

        def add(prefix, kind, cin, cout, rs, attn, consumes, level, container):
            b = _Blk()
            b.prefix, b.kind, b.cin, b.cout, b.rs, b.attn, b.consumes, b.level = prefix, kind, cin, cout, rs, attn, consumes, level
            if kind == "midattn":
                b.res, b.att = None, container
            elif attn:
                b.res, b.att = container[0], container[1]
            else:
                b.res, b.att = container, None
            plan.append(b)

        for i in range(levels):                                               # reference unet.py:250-263
            mods = model.downsamples[f"level_{i}"]
            prev = chs[i - 1] if i else hid
            for j in range(nrb):
                add(f"downsamples.level_{i}.{j}", "down", prev if j == 0 else chs[i], chs[i], H.RS_NONE, model.apply_attn[i],
                    False, i, mods[j])
            if i != levels - 1:
                add(f"downsamples.level_{i}.{nrb}", "down", chs[i], chs[i], H.RS_DOWN, model.apply_attn[i], False, i, mods[nrb])
        add("middle.0", "mid", chs[-1], chs[-1], H.RS_NONE, False, False, levels - 1, model.middle[0])
        add("middle.1", "midattn", chs[-1], chs[-1], H.RS_NONE, True, False, levels - 1, model.middle[1])
        add("middle.2", "mid", chs[-1], chs[-1], H.RS_NONE, False, False, levels - 1, model.middle[2])
        for i in range(levels - 1, -1, -1):                                   # reference unet.py:265-284
            mods = model.upsamples[f"level_{i}"]
            nxt = hid if i == 0 else chs[i - 1]
            prv = chs[-1] if i == levels - 1 else chs[i + 1]
            cins = [prv + chs[i]] + [2 * chs[i]] * (nrb - 1) + [nxt + chs[i]]
            for j, cin in enumerate(cins):
                add(f"upsamples.level_{i}.{j}", "up", cin, chs[i], H.RS_NONE, model.apply_attn[i], True, i, mods[j])
            if i != 0:
                add(f"upsamples.level_{i}.{nrb + 1}", "up", chs[i], chs[i], H.RS_UP, model.apply_attn[i], False, i, mods[nrb + 1])
        self.plan = plan
        # ---- static skip-stack analysis: which concat buffer does every pushed tensor land in?
        pushes = [["in_conv", hid, None, None]]          # [producer, Cs, consumer block index, Ch]
        sid = [0]
        for bi, b in enumerate(plan):
            b.push_hs = None
            b.src_hs = None
            b.ch_h = None
            if b.kind == "down":
                pushes.append([bi, b.cout, None, None])
                sid.append(len(pushes) - 1)
                b.push_hs = len(pushes) - 1
            elif b.consumes:
                k = sid.pop()
                b.src_hs = k
                b.ch_h = b.cin - pushes[k][1]
                pushes[k][2], pushes[k][3] = bi, b.ch_h
        assert not sid, "skip stack not emptied"
        self.pushes = pushes
        for bi, b in enumerate(plan):                    # where does each block write its output?
            if b.kind == "down":
                k = b.push_hs
                b.dest = ("cat", pushes[k][2], pushes[k][3], pushes[k][1])
            else:
                nxt = plan[bi + 1] if bi + 1 < len(plan) else None
                b.dest = ("cat", bi + 1, 0, b.cout) if (nxt is not None and nxt.consumes) else ("plain",)
        self.levels = levels
        # FiLM projections (reference unet.py:129,143: one Linear per residual block, all fed by the SAME activated
        # embedding): blocks with equal width are run as ONE batched GEMM per pass instead of a launch-latency-bound
        # M = batch GEMM per block.  film_slot: residual-block prefix -> (2*Cout, index inside its group)
        self.film_groups, self.film_slot = {}, {}
        for b in plan:
            if b.res is not None:
                pre = b.prefix + (".0" if b.att is not None else "")
                grp = self.film_groups.setdefault(2 * b.cout, [])
                self.film_slot[pre] = (2 * b.cout, len(grp))
                grp.append(b.res)
        # forward weight packs kept across calls while a sampler holds the weights fixed (set to {} by the sampling loop,
        # None otherwise: training repacks every step because the optimizer rewrites the weights)
        self.pack_cache = None
        self._pack_state = None           # {weight-set key: persistent buffers + device table of the batched weight pack}
        self._last_pack_state = None
        self._packed = None               # {id(weight): (wf, wd)} valid for the weights as of the last forward
        self._norm_channels = sum(p.numel() for k, p in model.named_parameters()
                                  if k.endswith(".weight") and p.ndim == 1)          # all GroupNorm weights (the only 1-D weights)
        self._pgb_arena = self._pgb_table = None
        self._side = self._side_stream = None     # side stream of the 3x3 weight gradients (VD_WGRAD_STREAM)
        self._wq = None                   # {shape key: [(dY, X, dW, dbias)]} of deferred weight-gradient GEMMs (inside backward only)
        self._active_run = None           # the chain-of-nodes backward pass in flight (models/unet.py::_BackwardRun), one at a time
        self._segments = None

    # ------------------------------------------------------------------------------------------ small helpers
    @staticmethod
    def _new(ref, *shape):
        return torch.empty(shape, dtype=torch.float32, device=ref.device)

    def _res_of(self, level, H0, W0):
        return H0 >> level, W0 >> level

    # ---- GroupNorm statistics ride on the producer: the conv / 1x1 epilogue that writes a tensor also leaves per-
    # (image, row-chunk, channel) sums behind, and the consuming norm only runs the tiny finalize (no read pass)
    def _part(self, ref, nimg, HW, C):
        if HW % 32:
            return None                                   # tiny images: the norm computes its own statistics
        return torch.empty(H.stats_part_numel(nimg, HW, C), dtype=torch.float32, device=ref.device)

    @staticmethod
    def _parts(part, C, HW, rows=None):
        """[(partials, channels, chunks per image)] of the launch that just wrote `part`.  ``rows`` = pixel rows per chunk as the launch
        that wrote it reported (what _conv returns); None: a vd_gemm launch made immediately before this call -- its chunk is half the
        row tile it picked (vd_gemm_last_tile)"""
        if part is None:
            return None
        return [(part, C, HW // (rows or (H.last_row_tile() // 2)))]

    def _norm(self, x, x_parts, gn, film, act, p_drop, seed, rs, y, B, Hh, Ww, C):
        """y = resample(dropout(act(FiLM(GroupNorm(x))))); returns the [B][4][C] coefficient table the backward pass needs.
        With producer partials the statistics never exist as a tensor: one tiny kernel goes from partials to the table."""
        coef = self._new(x, B, 4, C)
        stats = None
        # the folded form lets EVERY workgroup of the apply pass re-reduce the partials of the groups it touches: a few chunks (one per image
        # from the F(4x4,3x3) forward kernel, four on 64-wide images) cost nothing, HW / 64 of them (tile-engine / F(2x2,3x3) producers on
        # 32x32 and larger images) would be read back by every 64-pixel workgroup -- those keep the separate finalize launch
        if x_parts is not None and H.GN_FOLD and max(k for _, _, k in x_parts) <= H.GN_FOLD_MAX_CHUNKS:
            H.gn_apply_from_partials(x, _ld(x), x_parts, gn.weight, gn.bias, film, act, p_drop, seed, rs, y, _ld(y), B, Hh, Ww, C, coef,
                                     GROUPS, EPS)
            return coef
        if x_parts is not None:
            H.gn_coef_from_partials(x_parts, B, Hh * Ww, gn.weight, gn.bias, film, coef, GROUPS, EPS)
        else:
            stats = self._new(x, B, GROUPS, 2)
            H.gn_stats(x, _ld(x), B, Hh * Ww, C, stats, GROUPS, EPS)
        H.gn_apply(x, _ld(x), stats, gn.weight, gn.bias, film, act, p_drop, seed, rs, y, _ld(y), B, Hh, Ww, C, coef, GROUPS)
        return coef

    # ---- 1x1-convolution / linear weight gradients dW = dY^T X (skip, proj_in, proj_out, fc): nothing downstream in backward
    # depends on them, so they are queued per shape and run as ONE grouped launch per shape at the end of a UNet level
    # (vd_gemm_grouped_wgrad) instead of one short split-K GEMM + one slab reduction per block.  The queue keeps the operands alive.
    def _wgrad(self, dy, x, dw, db, M, N, K, lda, ldb, ldc):
        if self._wq is None or not H.GROUPED_WGRAD:
            H.gemm(dy, x, dw, M, N, K, a_kind=H.COL, b_kind=H.COL, lda=lda, ldb=ldb, ldc=ldc, splitk=_splitk(M, N, K), colsum=db)
            return
        key = (M, N, K, lda, ldb, ldc, db is not None)
        q = self._wq.setdefault(key, [])
        q.append((dy, x, dw, db))
        if len(q) == H.GROUP_MAX:
            self._wgrad_flush_key(key)

    # ---- weight gradients on a side stream: nothing downstream in backward reads them, so they run beside the input-gradient /
    # GroupNorm chain of the following blocks (an HBM-bound transform or GroupNorm pass next to an MFMA-bound GEMM or convolution instead
    # of after it).  The side stream waits for the main stream before every launch (its operands are final), the operands are marked as
    # used on it (the caching allocator must not hand their memory out while it still reads them), and the main stream waits for it
    # wherever a gradient is declared complete: at every `progress` call when a reducer listens, else once at the end of backward.
    # Results are bitwise those of the one-stream order (every gradient tensor has one writer); per-kernel profiling (H.PROFILE) runs
    # on one stream so that a kernel's events bracket that kernel alone.
    def _cwgrad(self, x, ldx, dy, lddy, *args, **kw):
        side = self._side
        if side is None:
            return H.conv3x3_wgrad(x, ldx, dy, lddy, *args, **kw)
        side.wait_stream(torch.cuda.current_stream())
        x.record_stream(side)
        dy.record_stream(side)
        with torch.cuda.stream(side):
            H.conv3x3_wgrad(x, ldx, dy, lddy, *args, **kw)

    def _side_join(self):
        if self._side is not None:
            torch.cuda.current_stream().wait_stream(self._side)

    def _wgrad_flush_key(self, key):
        q = self._wq.pop(key, None)
        if not q:
            return
        M, N, K, lda, ldb, ldc, _ = key
        # the grouped launch exists on the LDS-DMA tile engine only: every entry needs what that kernel needs (16-byte aligned operands,
        # pitches that are multiples of 4 floats and inside its 32-bit offset range); anything else takes the per-entry launches
        ok = M % 4 == 0 and N % 4 == 0 and lda % 4 == 0 and ldb % 4 == 0 and max(lda, ldb, ldc) * 128 + 128 < 0x70000000 // 4 and \
            all(dy.data_ptr() % 16 == 0 and x.data_ptr() % 16 == 0 for dy, x, _, _ in q)
        if len(q) == 1 or not ok:
            for dy, x, dw, db in q:
                H.gemm(dy, x, dw, M, N, K, a_kind=H.COL, b_kind=H.COL, lda=lda, ldb=ldb, ldc=ldc, splitk=_splitk(M, N, K), colsum=db)
            return
        if H.GROUPED_AUTO_SPLIT:      # slab count that fills whole residency rounds of the chip (vd_gemm_grouped_wgrad_auto_split)
            S = H.lib().vd_gemm_grouped_wgrad_auto_split(len(q), M, N, K, 1, 64)
        else:
            tiles = ((M + 63) // 64) * ((N + 63) // 64) * len(q)
            S = max(1, min(64, 1024 // max(tiles, 1), K // 256))
        H.gemm_grouped_wgrad(q, M, N, K, lda, ldb, ldc, S)

    def _wgrad_flush(self):
        if not self._wq:
            return
        side = self._side
        if side is None:
            for key in list(self._wq):
                self._wgrad_flush_key(key)
            return
        side.wait_stream(torch.cuda.current_stream())
        for q in self._wq.values():
            for dy, x, _, _ in q:
                dy.record_stream(side)
                x.record_stream(side)
        with torch.cuda.stream(side):
            for key in list(self._wq):
                self._wgrad_flush_key(key)

    # ---- GroupNorm parameter gradients: every norm's backward leaves its per-image dgamma / dbeta terms in a slice of one arena and
    # ONE launch at the end of backward sums them over the images for all norms (73 five-microsecond launches per CIFAR step before)
    def _pgb_begin(self, ref, B):
        need = B * 2 * self._norm_channels
        if self._pgb_arena is None or self._pgb_arena.numel() < need or self._pgb_arena.device != ref.device:
            self._pgb_arena = torch.empty(need, dtype=torch.float32, device=ref.device)
            self._pgb_table = None
        self._pgb_rows, self._pgb_off, self._pgb_blk = [], 0, 0

    def _pgb(self, B, C, dgamma, dbeta):
        sl = self._pgb_arena[self._pgb_off: self._pgb_off + B * 2 * C]
        self._pgb_rows.append([sl.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), B, C, 0, 0, self._pgb_blk])
        self._pgb_off += B * 2 * C
        self._pgb_blk += (C + 15) // 16
        return sl

    def _pgb_finish(self):
        key = tuple(tuple(r) for r in self._pgb_rows)
        if self._pgb_table is None or self._pgb_table[0] != key:            # (flat gradient buffers: built once)
            self._pgb_table = (key, torch.tensor(self._pgb_rows, dtype=torch.int64).to(self._pgb_arena.device))
        H.gn_param_sums_batched(self._pgb_table[1], len(self._pgb_rows), self._pgb_blk)

    PACK_STATES_MAX = 4       # weight sets (raw / EMA) x (training / inference) whose packed images are kept at a time

    def _conv_geoms(self, B, H0, W0):
        """[(weight, rows, H, W, Cin, Cout)] of every residual-block convolution for a (B, C, H0, W0) input: the resolution its two
        convolutions run at (a down-sampling block pools in front of conv1, an up-sampling block up-samples in front of it)"""
        out = []
        for b in self.plan:
            if b.res is None:
                continue
            lh, lw = self._res_of(b.level, H0, W0)
            if b.rs == H.RS_DOWN:
                lh, lw = lh // 2, lw // 2
            elif b.rs == H.RS_UP:
                lh, lw = lh * 2, lw * 2
            out.append((b.res.conv1.weight, B, lh, lw, b.cin, b.cout))
            out.append((b.res.conv2.weight, B, lh, lw, b.cout, b.cout))
        return out

    def _pack_all(self, need_d, geom=None):
        """Re-pack the 3x3 kernels of every residual block in ONE launch (vd_wino_pack_batched / vd_pack_conv3x3_batched) into
        persistent buffers; returns {id(weight): (forward pack, dgrad pack or None)}.
        Pack state (buffers + device table) is kept PER WEIGHT SET -- the key is the parameters' storage pointers, so raw
        weights and an EMA view swap (trainer.ema_weights) each keep their own stable buffers and table instead of freeing and
        re-allocating them on every alternation -- in a small LRU.  ``self._last_pack_state`` is the state the last forward
        used: a captured HIP graph keeps a reference to it, so the table / U images its pack and convolution nodes point to
        stay allocated for as long as the graph is cached (diffusion._sample_loop_graph).
        ``geom`` = (B, H0, W0) of the training input: the input gradients of the layers vd_conv3x3_dgrad_wino43 serves at that
        geometry take the F(4x4,3x3) image U43 instead of the rotated F(2x2,3x3) one."""
        ws = [c.weight for b in self.plan if b.res is not None for c in (b.res.conv1, b.res.conv2)]
        key = (bool(need_d), H.WINO, H.WINO43_FWD, geom) + tuple(w.data_ptr() for w in ws)
        if self._pack_state is None:
            self._pack_state = {}
        st = self._pack_state.pop(key, None)             # (re-inserted below as the most recent entry)
        if st is None:
            dev = ws[0].device
            sizes = [w.numel() for w in ws]
            st = dict(n=len(ws))
            if H.WINO:
                # Winograd-domain kernels U = G w G^T (csrc/wino.hip): [16][Cout][Cin] forward, [16][Cin][Cout] input gradient;
                # csrc/wino43.hip: 36 x Cin x Cout in lane order for the input gradients it serves
                # per layer: which F(4x4,3x3) images the training / inference geometry lets it use (forward: u43f, input gradient: u43);
                # a layer served by them gets no F(2x2,3x3) image for that direction (a call that declines packs one on demand: _conv)
                use43, use43f = {}, {}
                if geom is not None:
                    for (w, nb, lh, lw, ci, co) in self._conv_geoms(*geom):
                        use43[id(w)] = need_d and H.wino43_supported(nb, lh, lw, ci, co, co, ci)
                        use43f[id(w)] = H.wino43_fwd_supported(nb, lh, lw, ci, co, ci, co, co)
                nf = sum(n for w, n in zip(ws, sizes) if not use43f.get(id(w)))
                nd = sum(n for w, n in zip(ws, sizes) if not use43.get(id(w))) if need_d else 0
                n43 = sum(n for w, n in zip(ws, sizes) if use43.get(id(w))) + sum(n for w, n in zip(ws, sizes) if use43f.get(id(w)))
                uf_all = torch.empty(16 * nf // 9, dtype=torch.float32, device=dev) if nf else None
                ud_all = torch.empty(16 * nd // 9, dtype=torch.float32, device=dev) if nd else None
                u43_all = torch.empty(36 * n43 // 9, dtype=torch.float32, device=dev) if n43 else None
                wrows, wviews, woff, doff, wblk = [], {}, 0, 0, 0
                rows43, off43, blk43 = [], 0, 0
                for w, n in zip(ws, sizes):
                    co, ci = w.shape[0], w.shape[1]
                    m = 16 * co * ci
                    uf = ud = u43 = u43f = None
                    if use43f.get(id(w)):
                        u43f = u43_all[off43: off43 + 36 * co * ci]
                        rows43.append([w.data_ptr(), u43f.data_ptr(), 1, co, ci, 0, 0, blk43])
                        off43 += 36 * co * ci
                        blk43 += (co // 32) * (ci // 8)
                    else:
                        uf = uf_all[woff: woff + m].view(16, co, ci)
                        woff += m
                    if need_d and use43.get(id(w)):
                        u43 = u43_all[off43: off43 + 36 * co * ci]
                        rows43.append([w.data_ptr(), u43.data_ptr(), 0, co, ci, 0, 0, blk43])
                        off43 += 36 * co * ci
                        blk43 += (ci // 32) * (co // 8)
                    elif need_d:
                        ud = ud_all[doff: doff + m].view(16, ci, co)
                        doff += m
                    wviews[id(w)] = (uf, ud, u43, u43f)
                    if uf is None and ud is None:
                        continue
                    tiled = int(co % 16 == 0 and ci % 16 == 0)
                    wrows.append([w.data_ptr(), uf.data_ptr() if uf is not None else 0, ud.data_ptr() if ud is not None else 0, co, ci,
                                  tiled, 0, wblk])
                    wblk += (co // 16) * (ci // 16) if tiled else (co * ci + 255) // 256
                st.update(uf=uf_all, ud=ud_all, u43=u43_all, wtable=torch.tensor(wrows, dtype=torch.int64).to(dev) if wrows else None,
                          nw=len(wrows), wblocks=wblk,
                          wviews=wviews, table43=torch.tensor(rows43, dtype=torch.int64).to(dev) if rows43 else None,
                          n43=len(rows43), blocks43=blk43)
            else:
                wf_all = torch.empty(sum(sizes), dtype=torch.float32, device=dev)
                wd_all = torch.empty(sum(sizes), dtype=torch.float32, device=dev) if need_d else None
                rows, views, off, blk = [], {}, 0, 0
                for w, n in zip(ws, sizes):
                    co, ci = w.shape[0], w.shape[1]
                    wf = wf_all[off: off + n].view(co, 9, ci)
                    wd = wd_all[off: off + n].view(ci, 9, co) if need_d else None
                    rows.append([w.data_ptr(), wf.data_ptr(), wd.data_ptr() if need_d else 0, co, ci, ci, co, blk])
                    views[id(w)] = (wf, wd)
                    off += n
                    blk += (n + 255) // 256
                st.update(wf=wf_all, wd=wd_all, table=torch.tensor(rows, dtype=torch.int64).to(dev), blocks=blk, views=views)
            while len(self._pack_state) >= self.PACK_STATES_MAX:
                self._pack_state.pop(next(iter(self._pack_state)))       # least recently used
        self._pack_state[key] = st
        self._last_pack_state = st
        if H.WINO:
            # every convolution the Winograd kernels serve needs only U; the direct packs are made per tensor, on demand, by
            # _pack_f / _pack_d for the geometries that fall back (none in the shipped configs)
            if st["nw"]:
                H.wino_pack_batched(st["wtable"], st["nw"], st["wblocks"])
            if st["n43"]:
                H.wino43_pack_batched(st["table43"], st["n43"], st["blocks43"])
            self._wino = st["wviews"]
            return {}
        H.pack_conv3x3_batched(st["table"], st["n"], st["blocks"])
        self._wino = None
        return st["views"]

    def _conv(self, x, ldx, w, bias, y, ldy, B, Hh, Ww, Cin, Cout, dgrad=False, res=None, ldres=0, stats_part=None):
        """3x3 convolution with kernel ``w`` (forward) or its input gradient (``dgrad``: x = dy, Cin/Cout are the GEMM's):
        Winograd F(4x4,3x3) / F(2x2,3x3) wherever the geometry is served, the direct implicit GEMM otherwise.  Returns the pixel rows per
        chunk of the GroupNorm partial sums it left in ``stats_part`` (the consuming norm needs the chunk count: _parts)."""
        wino = getattr(self, "_wino", None)
        if wino is not None and not dgrad and id(w) in wino and wino[id(w)][3] is not None \
                and H.wino43_fwd_supported(B, Hh, Ww, Cin, Cout, ldx, ldy, ldres if res is not None else 0):
            # forward pass through F(4x4,3x3) (csrc/wino43.hip, dyadic interpolation points); its GroupNorm partials come one chunk per
            # (image, work item)
            H.conv3x3_wino43_fwd(x, ldx, wino[id(w)][3], bias, y, ldy, B, Hh, Ww, Cin, Cout, res=res, ldres=ldres, stats_part=stats_part)
            return H.wino43_fwd_chunk_rows(Hh, Ww)
        if wino is not None and dgrad and id(w) in wino and wino[id(w)][2] is not None \
                and H.wino43_supported(B, Hh, Ww, Cout, Cin, ldx, ldy):
            # F(4x4,3x3) input gradient (csrc/wino43.hip): x = dy [.., Cin = conv Cout], y = dx [.., Cout = conv Cin].  The pack-time choice
            # (_pack_all) saw dense pitches; the check above is the one with the pitches of THIS call (dy may be a channel slice of a concat
            # buffer) and of this geometry -- a layer it declines falls through to the F(2x2,3x3) / direct forms below.
            assert res is None and bias is None and stats_part is None
            H.conv3x3_dgrad_wino43(x, ldx, wino[id(w)][2], y, ldy, B, Hh, Ww, Cout, Cin)
            return None
        if wino is not None and id(w) in wino and H.wino_supported(B, Hh, Ww, Cin, Cout, ldx, ldy, ldres if res is not None else 0):
            U = wino[id(w)][1 if dgrad else 0]
            if U is None:
                # a layer packed for an F(4x4,3x3) form whose call declined it (pitches of THIS call): its F(2x2,3x3) image -- rotated for
                # the input gradient -- is made on demand, into a buffer that lives with the pack state (one allocation per weight and
                # direction, stable address: also valid inside a captured HIP graph), and re-packed per call like every other image
                st = self._last_pack_state
                fb = st.setdefault("fallback_u", {}) if st is not None else {}
                U = fb.get((id(w), dgrad))
                if U is None:
                    shp = (16, w.shape[1], w.shape[0]) if dgrad else (16, w.shape[0], w.shape[1])
                    U = fb[(id(w), dgrad)] = torch.empty(shp, dtype=torch.float32, device=w.device)
                if dgrad:
                    H.wino_pack(w, w.shape[0], w.shape[1], ud=U)
                else:
                    H.wino_pack(w, w.shape[0], w.shape[1], uf=U)
            H.conv3x3_wino(x, ldx, U, bias, y, ldy, B, Hh, Ww, Cin, Cout, res=res, ldres=ldres, stats_part=stats_part)
            return H.last_row_tile() // 2              # the F(2x2,3x3) kernels: one chunk per 64 output rows, reported like a 128-row tile
        H.conv3x3(x, ldx, self._pack_d(w) if dgrad else self._pack_f(w), bias, y, ldy, B, Hh, Ww, Cin, Cout, res=res, ldres=ldres,
                  stats_part=stats_part)
        return H.last_row_tile() // 2                  # the tile engine: half its row tile

    def _pack_f(self, w, cin_p=None):
        if self._packed is not None and id(w) in self._packed:
            return self._packed[id(w)][0]
        cache = self.pack_cache
        if cache is not None and id(w) in cache:
            return cache[id(w)]
        co, ci = w.shape[0], w.shape[1]
        cin_p = cin_p or ci
        wf = self._new(w, co, 9, cin_p)
        H.pack_conv3x3(w, co, ci, wf=wf, Cin_p=cin_p)
        if cache is not None:
            cache[id(w)] = wf
        return wf

    def _pack_d(self, w, cout_p=None):
        if self._packed is not None and id(w) in self._packed and self._packed[id(w)][1] is not None:
            return self._packed[id(w)][1]
        co, ci = w.shape[0], w.shape[1]
        cout_p = cout_p or co
        wd = self._new(w, ci, 9, cout_p)
        H.pack_conv3x3(w, co, ci, wd=wd, Cout_p=cout_p)
        return wd

    def _pack_thin(self, w, rows):
        """[rows][Cin] image of a thin-output 3x3 kernel: row co*9+tap = w[co][:, tap], rows beyond 9*Cout are zero"""
        cache = self.pack_cache
        key = ("thin", id(w))
        if cache is not None and key in cache:
            return cache[key]
        co, ci = w.shape[0], w.shape[1]
        wz = torch.zeros((rows, ci), dtype=torch.float32, device=w.device)
        H.pack_conv3x3(w, co, ci, wf=wz, Cin_p=ci)
        if cache is not None:
            cache[key] = wz
        return wz

    @staticmethod
    def _linear(x, w, b, out, accumulate=False):
        """out[M,N] (+)= x[M,K] @ w[N,K]^T + b"""
        M, K = x.shape
        N = w.shape[0]
        H.gemm(x, w, out, M, N, K, a_kind=H.ROW, b_kind=H.ROW, lda=x.stride(0), ldb=w.stride(0), ldc=out.stride(0), bias=b,
               accumulate=accumulate)

    def _linear_bwd(self, x, w, dy, dw, db, dx, dx_accumulate=False):
        """dw[N,K] = dy^T x ; db[N] = colsum(dy) ; dx[M,K] (+)= dy @ w"""
        M, K = x.shape
        N = w.shape[0]
        self._wgrad(dy, x, dw, db, N, K, M, dy.stride(0), x.stride(0), K)      # db = column sums of dy, from the same staged tiles
        if dx is not None:
            H.gemm(dy, w, dx, M, K, N, a_kind=H.ROW, b_kind=H.COL, lda=dy.stride(0), ldb=w.stride(0), ldc=dx.stride(0),
                   accumulate=dx_accumulate)

    # ------------------------------------------------------------------------------------------ embeddings
    def _embed_fwd(self, t, y, tape):
        m = self.m
        B = t.shape[0]
        dev_ref = m.in_conv.weight
        if t.dtype != torch.float64:
            t = t.to(torch.float64)
        te0 = self._new(dev_ref, B, m.hid_channels)
        H.timestep_embedding(t.contiguous(), te0, B, m.hid_channels)
        l0, l2 = m.time_embed[0], m.time_embed[2]
        h0 = self._new(dev_ref, B, m.embedding_dim)
        self._linear(te0, l0.weight, l0.bias, h0)
        a0 = torch.empty_like(h0)
        H.silu(h0, a0)
        te = torch.empty_like(h0)
        self._linear(a0, l2.weight, l2.bias, te)
        yn = None
        if m.num_classes and y is not None:
            y = y.to(torch.float32).contiguous().clone()      # callers mutate y after the forward (diffusion.py:527-529)
            if m.multitags:
                assert y.ndim == 2 and y.shape[1] == m.num_classes
                yn = torch.empty_like(y)
                H.multitag_norm(y, yn, B, m.num_classes)
                self._linear(yn, m.class_embed.weight, m.class_embed.bias, te, accumulate=True)
            else:
                lin = m.class_embed[1]
                H.class_embed(y.reshape(-1), lin.weight, lin.bias, te, B, m.embedding_dim, m.num_classes)
                yn = y.reshape(-1)
        ta = torch.empty_like(te)
        H.silu(te, ta)
        if tape is not None:
            tape["embed"] = dict(te0=te0, h0=h0, a0=a0, te=te, yn=yn)
        return ta

    def _embed_bwd(self, ctx, dta, G):
        m = self.m
        B = dta.shape[0]
        te0, h0, a0, te, yn = ctx["te0"], ctx["h0"], ctx["a0"], ctx["te"], ctx["yn"]
        dte = torch.empty_like(dta)
        H.silu_bwd(te, dta, dte)
        if yn is not None:
            if m.multitags:
                self._linear_bwd(yn, m.class_embed.weight, dte, G["class_embed.weight"], G["class_embed.bias"], None)
            else:
                H.class_embed_bwd(yn, dte, G["class_embed.1.weight"], G["class_embed.1.bias"], B, m.embedding_dim, m.num_classes)
        elif m.num_classes:
            # class-conditional network called without labels: the reference leaves these gradients None and its optimizer
            # skips them; every entry of G must be written (flat buffers are reused across steps), so they are exact zeros
            # here and HotPathTrainer.step tells the fused optimizer kernel to leave that range alone (vd_adamw_ema r_mode 1:
            # no moment decay, no weight decay, no update, step not advanced -- what torch.optim.AdamW does for grad None)
            for k in G:
                if k.startswith("class_embed."):
                    G[k].zero_()
        l0, l2 = m.time_embed[0], m.time_embed[2]
        da0 = torch.empty_like(dte)
        self._linear_bwd(a0, l2.weight, dte, G["time_embed.2.weight"], G["time_embed.2.bias"], da0)
        dh0 = torch.empty_like(da0)
        H.silu_bwd(h0, da0, dh0)
        self._linear_bwd(te0, l0.weight, dh0, G["time_embed.0.weight"], G["time_embed.0.bias"], None)

    # ------------------------------------------------------------------------------------------ FiLM projections
    def _film_fwd(self, ta, tape):
        """film[g][i] = fc_i(ta) for every residual block i of width group g: [nb][B][2*Cout], one launch per group"""
        B, E = ta.shape
        films, stacks = {}, {}
        cache = self.pack_cache
        for c2, mods in self.film_groups.items():
            key = ("film", c2)
            if cache is not None and key in cache:
                W, bv = cache[key]
            else:
                W = torch.stack([mm.fc.weight.detach() for mm in mods])          # [nb][2C][E]
                bv = torch.stack([mm.fc.bias.detach() for mm in mods])           # [nb][2C]
                if cache is not None:
                    cache[key] = (W, bv)
            nb = len(mods)
            out = self._new(ta, nb, B, c2)
            H.gemm(ta, W, out, B, c2, E, a_kind=H.ROW, b_kind=H.ROW, lda=E, ldb=E, ldc=c2, bias=bv, sBias=c2, batch=nb,
                   sA=(0, 0), sB=(c2 * E, 0), sC=(B * c2, 0))
            films[c2], stacks[c2] = out, W
        if tape is not None:
            tape["film_w"] = stacks
        return films

    def _film_bwd(self, dfilms, stacks, dta):
        """dta += sum_i dfilm_i @ W_i : one batched GEMM per width group into per-block partials + one column sum"""
        B, E = dta.shape
        for c2, df in dfilms.items():
            nb, W = df.shape[0], stacks[c2]
            P = self._new(dta, nb, B * E)
            H.gemm(df, W, P, B, E, c2, a_kind=H.ROW, b_kind=H.COL, lda=c2, ldb=E, ldc=E, batch=nb, sA=(B * c2, 0), sB=(c2 * E, 0),
                   sC=(B * E, 0))
            H.colsum(P, B * E, nb, B * E, dta.view(-1), accumulate=True)

    # ------------------------------------------------------------------------------------------ residual block
    def _res_fwd(self, blk, mod, prefix, x, films, dest, p_drop, seed, tape, x_parts=None):
        B, Hh, Ww, Cin, ldx = _chk(x)
        Cout, rs = blk.cout, blk.rs
        Ho, Wo = (Hh // 2, Ww // 2) if rs == H.RS_DOWN else ((Hh * 2, Ww * 2) if rs == H.RS_UP else (Hh, Ww))
        a1 = self._new(x, B, Ho, Wo, Cin)
        coef1 = self._norm(x, x_parts, mod.norm1, None, 1, 0.0, 0, rs, a1, B, Hh, Ww, Cin)
        h1 = self._new(x, B, Ho, Wo, Cout)
        ph = self._part(x, B, Ho * Wo, Cout)
        rows = self._conv(a1, Cin, mod.conv1.weight, mod.conv1.bias, h1, Cout, B, Ho, Wo, Cin, Cout, stats_part=ph)
        h1_parts = self._parts(ph, Cout, Ho * Wo, rows)
        c2, fi = self.film_slot[prefix]
        film = films[c2][fi]                              # [B][2*Cout], contiguous slice of the group's batched GEMM output
        a2 = self._new(x, B, Ho, Wo, Cout)
        coef2 = self._norm(h1, h1_parts, mod.norm2, film, 1, p_drop, seed, H.RS_NONE, a2, B, Ho, Wo, Cout)
        if rs != H.RS_NONE:
            xs = self._new(x, B, Ho, Wo, Cin)
            H.gn_apply(x, ldx, None, None, None, None, 0, 0.0, 0, rs, xs, Cin, B, Hh, Ww, Cin, None, GROUPS)
        else:
            xs = x
        has_skip = Cin != Cout
        if has_skip:
            sk = self._new(x, B, Ho, Wo, Cout)
            w = mod.skip.weight
            H.gemm(xs, w, sk, B * Ho * Wo, Cout, Cin, a_kind=H.ROW, b_kind=H.ROW, lda=_ld(xs), ldb=Cin, ldc=Cout, bias=mod.skip.bias)
        else:
            sk = xs
        pd = self._part(x, B, Ho * Wo, Cout)
        rows = self._conv(a2, Cout, mod.conv2.weight, mod.conv2.bias, dest, _ld(dest), B, Ho, Wo, Cout, Cout, res=sk,
                          ldres=_ld(sk), stats_part=pd)
        out_parts = self._parts(pd, Cout, Ho * Wo, rows)
        if tape is not None:
            tape[prefix] = dict(x=x, coef1=coef1, a1=a1, h1=h1, coef2=coef2, a2=a2, film=film, xs=xs if has_skip else None,
                                seed=seed, p=p_drop)
        return out_parts

    def _res_bwd(self, blk, mod, prefix, ctx, dy, dx, dx_accumulate, ta, dfilms, G):
        x, a1, h1, a2, film = ctx["x"], ctx["a1"], ctx["h1"], ctx["a2"], ctx["film"]
        B, Hh, Ww, Cin, ldx = _chk(x)
        _, Ho, Wo, Cout, lddy = _chk(dy)
        rs = blk.rs
        # conv2
        self._cwgrad(a2, Cout, dy, lddy, B, Ho, Wo, Cout, Cout, G[prefix + ".conv2.weight"], Cout, Cout,
                        dbias=G[prefix + ".conv2.bias"])
        da2 = self._new(x, B, Ho, Wo, Cout)
        self._conv(dy, lddy, mod.conv2.weight, None, da2, Cout, B, Ho, Wo, Cout, Cout, dgrad=True)
        # norm2 + FiLM + SiLU + dropout
        dh1 = self._new(x, B, Ho, Wo, Cout)
        c2, fi = self.film_slot[prefix]
        dfilm = dfilms[c2][fi]
        H.gn_apply_bwd(da2, Cout, h1, Cout, ctx["coef2"], mod.norm2.weight, mod.norm2.bias, film, 1, ctx["p"], ctx["seed"],
                       H.RS_NONE, None, 0, dh1, Cout, False, dfilm, None, None, False, B, Ho, Wo, Cout, GROUPS,
                       pgb_keep=self._pgb(B, Cout, G[prefix + ".norm2.weight"], G[prefix + ".norm2.bias"]))
        del da2
        # conv1
        self._cwgrad(a1, Cin, dh1, Cout, B, Ho, Wo, Cin, Cout, G[prefix + ".conv1.weight"], Cin, Cout,
                        dbias=G[prefix + ".conv1.bias"])
        da1 = self._new(x, B, Ho, Wo, Cin)
        self._conv(dh1, Cout, mod.conv1.weight, None, da1, Cin, B, Ho, Wo, Cout, Cin, dgrad=True)
        del dh1
        # skip path
        if Cin != Cout:
            xs = ctx["xs"]
            P = B * Ho * Wo
            w = mod.skip.weight
            self._wgrad(dy, xs, G[prefix + ".skip.weight"], G[prefix + ".skip.bias"], Cout, Cin, P, lddy, _ld(xs), Cin)
            dsk = self._new(x, B, Ho, Wo, Cin)
            H.gemm(dy, w, dsk, P, Cin, Cout, a_kind=H.ROW, b_kind=H.COL, lda=lddy, ldb=Cin, ldc=Cin)
        else:
            dsk = dy
        if rs != H.RS_NONE:
            addt = self._new(x, B, Hh, Ww, Cin)
            H.gn_apply_bwd(dsk, _ld(dsk), None, 0, None, None, None, None, 0, 0.0, 0, rs, None, 0, addt, Cin, False, None, None,
                           None, False, B, Hh, Ww, Cin, GROUPS)
        else:
            addt = dsk
        # norm1 + SiLU (+ resample) and the sum with the skip-path gradient
        H.gn_apply_bwd(da1, Cin, x, ldx, ctx["coef1"], mod.norm1.weight, mod.norm1.bias, None, 1, 0.0, 0, rs, addt, _ld(addt),
                       dx, _ld(dx), dx_accumulate, None, None, None, False, B, Hh, Ww, Cin, GROUPS,
                       pgb_keep=self._pgb(B, Cin, G[prefix + ".norm1.weight"], G[prefix + ".norm1.bias"]))
        # FiLM projection film = fc(ta): weight/bias gradient here (keeps the gradient-completion order); the embedding
        # gradient of all blocks is one batched GEMM at the end of backward (_film_bwd)
        self._linear_bwd(ta, mod.fc.weight, dfilm, G[prefix + ".fc.weight"], G[prefix + ".fc.bias"], None)

    # ------------------------------------------------------------------------------------------ attention block
    def _attn_fwd(self, mod, prefix, x, dest, tape, x_parts=None):
        B, Hh, Ww, C, ldx = _chk(x)
        L, nh, hd = Hh * Ww, mod.num_heads, mod.head_dim
        hid = nh * hd
        xn = self._new(x, B, Hh, Ww, C)
        coef = self._norm(x, x_parts, mod.norm, None, 0, 0.0, 0, H.RS_NONE, xn, B, Hh, Ww, C)
        qkv = self._new(x, B, L, 3 * hid)
        H.gemm(xn, mod.proj_in.weight, qkv, B * L, 3 * hid, C, a_kind=H.ROW, b_kind=H.ROW, lda=C, ldb=C, ldc=3 * hid,
               bias=mod.proj_in.bias)
        ld = 3 * hid
        alpha = 1.0 / math.sqrt(hd)
        q, k, v = qkv[0, 0, 0:], qkv[0, 0, hid:], qkv[0, 0, 2 * hid:]
        O = self._new(x, B, L, hid)
        S = lse = None
        if B * nh <= 65535 and H.attn_use_fused(L, hd, B * nh, tape is not None):
            # fused kernels (csrc/attn.hip): the [B, nh, L, L] maps never reach HBM; the backward recomputes them from the
            # per-row log-sum-exp.  Which shapes take them in training is decided by measurement (_hip.attn_use_fused).
            if tape is not None:
                lse = self._new(x, B * nh * L)
            H.attn_fwd(q, k, v, ld, O, hid, lse, B, nh, L, hd, alpha)
        else:
            S = self._new(x, B, nh, L, L)
            H.gemm(q, k, S, L, L, hd, a_kind=H.ROW, b_kind=H.ROW, lda=ld, ldb=ld, ldc=L, batch=B * nh, nh=nh, sA=(L * ld, hd),
                   sB=(L * ld, hd), sC=(nh * L * L, L * L), alpha=alpha)
            H.softmax_rows(S, B * nh * L, L)
            H.gemm(S, v, O, L, hd, L, a_kind=H.ROW, b_kind=H.COL, lda=L, ldb=ld, ldc=hid, batch=B * nh, nh=nh, sA=(nh * L * L, L * L),
                   sB=(L * ld, hd), sC=(L * hid, hd))
        pd = self._part(x, B, L, C)
        if pd is not None and L % 64:
            pd = None                                     # the plain GEMM picks its own row tile: only ask when any tile fits
        H.gemm(O, mod.proj_out.weight, dest, B * L, C, hid, a_kind=H.ROW, b_kind=H.ROW, lda=hid, ldb=hid, ldc=_ld(dest),
               bias=mod.proj_out.bias, R=x, ldr=ldx, stats=pd, stats_hw=L)
        out_parts = self._parts(pd, C, L)
        if tape is not None:
            tape[prefix] = dict(x=x, coef=coef, xn=xn, qkv=qkv, P=S, O=O, lse=lse)
        return out_parts

    def _attn_bwd(self, mod, prefix, ctx, dy, dx, dx_accumulate, G):
        x, xn, qkv, P, O = ctx["x"], ctx["xn"], ctx["qkv"], ctx["P"], ctx["O"]
        B, Hh, Ww, C, ldx = _chk(x)
        lddy = _ld(dy)
        L, nh, hd = Hh * Ww, mod.num_heads, mod.head_dim
        hid, ld = nh * hd, 3 * nh * hd
        M = B * L
        # proj_out
        self._wgrad(dy, O, G[prefix + ".proj_out.weight"], G[prefix + ".proj_out.bias"], C, hid, M, lddy, hid, hid)
        dO = self._new(x, B, L, hid)
        H.gemm(dy, mod.proj_out.weight, dO, M, hid, C, a_kind=H.ROW, b_kind=H.COL, lda=lddy, ldb=hid, ldc=hid)
        dqkv = self._new(x, B, L, ld)
        q, k, v = qkv[0, 0, 0:], qkv[0, 0, hid:], qkv[0, 0, 2 * hid:]
        dq, dk, dv = dqkv[0, 0, 0:], dqkv[0, 0, hid:], dqkv[0, 0, 2 * hid:]
        sP, sQ, sO = (nh * L * L, L * L), (L * ld, hd), (L * hid, hd)
        alpha = 1.0 / math.sqrt(hd)
        if P is None:
            # the forward ran fused: probabilities are recomputed from the saved log-sum-exp
            H.attn_bwd(q, k, v, ld, O, hid, dO, hid, ctx["lse"], self._new(x, B * nh * L), dq, dk, dv, ld, B, nh, L, hd, alpha)
        else:
            # dV[j][d] = sum_l P[l][j] dO[l][d]
            H.gemm(P, dO, dv, L, hd, L, a_kind=H.COL, b_kind=H.COL, lda=L, ldb=hid, ldc=ld, batch=B * nh, nh=nh, sA=sP, sB=sO, sC=sQ)
            # dP[l][j] = sum_d dO[l][d] V[j][d]
            dP = self._new(x, B, nh, L, L)
            H.gemm(dO, v, dP, L, L, hd, a_kind=H.ROW, b_kind=H.ROW, lda=hid, ldb=ld, ldc=L, batch=B * nh, nh=nh, sA=sO, sB=sQ, sC=sP)
            H.softmax_rows_bwd(P, dP, B * nh * L, L, alpha)                 # dP <- dS (already scaled by 1/sqrt(hd))
            # dQ[l][d] = sum_j dS[l][j] K[j][d] ; dK[j][d] = sum_l dS[l][j] Q[l][d]
            H.gemm(dP, k, dq, L, hd, L, a_kind=H.ROW, b_kind=H.COL, lda=L, ldb=ld, ldc=ld, batch=B * nh, nh=nh, sA=sP, sB=sQ, sC=sQ)
            H.gemm(dP, q, dk, L, hd, L, a_kind=H.COL, b_kind=H.COL, lda=L, ldb=ld, ldc=ld, batch=B * nh, nh=nh, sA=sP, sB=sQ, sC=sQ)
            del dP
        del dO
        # proj_in
        self._wgrad(dqkv, xn, G[prefix + ".proj_in.weight"], G[prefix + ".proj_in.bias"], ld, C, M, ld, C, C)
        dxn = self._new(x, B, Hh, Ww, C)
        H.gemm(dqkv, mod.proj_in.weight, dxn, M, C, ld, a_kind=H.ROW, b_kind=H.COL, lda=ld, ldb=C, ldc=C)
        # norm (no activation) + the residual branch
        H.gn_apply_bwd(dxn, C, x, ldx, ctx["coef"], mod.norm.weight, mod.norm.bias, None, 0, 0.0, 0, H.RS_NONE, dy, lddy, dx,
                       _ld(dx), dx_accumulate, None, None, None, False, B, Hh, Ww, C, GROUPS,
                       pgb_keep=self._pgb(B, C, G[prefix + ".norm.weight"], G[prefix + ".norm.bias"]))

    # ------------------------------------------------------------------------------------------ whole network
    def forward(self, x_nchw, t, y, training, save):
        """x (B,Cin,H,W) NCHW -> output NHWC ``[B,H,W,Cp]`` (Cp = out_channels rounded up to 4) and the tape."""
        m = self.m
        B, Ci, H0, W0 = x_nchw.shape
        assert Ci == m.in_channels and H0 % (1 << (self.levels - 1)) == 0 and W0 % (1 << (self.levels - 1)) == 0
        tape = {} if save else None
        x_nchw = x_nchw.to(torch.float32).contiguous()
        cache = self.pack_cache
        if cache is not None and "all" in cache:
            self._packed = cache["all"]                  # a sampler holds the weights fixed: packed once per chain
        else:
            self._packed = self._pack_all(need_d=save, geom=(B, H0, W0))
            if cache is not None:
                cache["all"] = self._packed
        ta = self._embed_fwd(t, y, tape)
        films = self._film_fwd(ta, tape)
        p_drop = float(m.drop_rate) if training else 0.0
        base_seed = int(torch.empty((), dtype=torch.int64).random_().item()) if p_drop > 0 else 0
        cip = (Ci + 3) // 4 * 4
        x4 = self._new(x_nchw, B, H0, W0, cip)
        H.nchw_to_nhwc(x_nchw, x4, B, Ci, H0, W0, cip)
        # concat buffers, one per consuming up-block
        cats = {}
        for bi, b in enumerate(self.plan):
            if b.consumes:
                hh, ww = self._res_of(b.level, H0, W0)
                cats[bi] = self._new(x_nchw, B, hh, ww, b.cin)

        def dest_of(spec, hh, ww):
            if spec[0] == "cat":
                _, u, c0, c = spec
                return cats[u][..., c0:c0 + c]
            return self._new(x_nchw, B, hh, ww, spec[1])

        p0 = self.pushes[0]
        d = dest_of(("cat", p0[2], p0[3], p0[1]), H0, W0)
        pp = self._part(x_nchw, B, H0 * W0, m.hid_channels)
        if pp is not None and (H0 * W0) % 64:
            pp = None
        xc = None
        if cip <= THIN:
            # thin input (3 -> hid): im2col of the 4-channel image once, then a plain K = 36 GEMM (see vd_im2col3x3)
            xc = self._new(x_nchw, B * H0 * W0, 9 * cip)
            H.im2col3x3(x4, cip, xc, B, H0, W0, cip)
            H.gemm(xc, self._pack_f(m.in_conv.weight, cip), d, B * H0 * W0, m.hid_channels, 9 * cip, a_kind=H.ROW, b_kind=H.ROW,
                   lda=9 * cip, ldb=9 * cip, ldc=_ld(d), bias=m.in_conv.bias, stats=pp, stats_hw=H0 * W0)
        else:
            H.conv3x3(x4, cip, self._pack_f(m.in_conv.weight, cip), m.in_conv.bias, d, _ld(d), B, H0, W0, cip, m.hid_channels,
                      stats_part=pp)
        cat_parts = {p0[2]: {p0[3]: self._parts(pp, m.hid_channels, H0 * W0)}}      # consumer block -> {channel offset: partials}
        hs_top, h = d, None
        top_parts, h_parts = cat_parts[p0[2]][p0[3]], None
        for bi, b in enumerate(self.plan):
            if b.kind == "down":
                inp, inp_parts = hs_top, top_parts
            elif b.kind in ("mid", "midattn"):
                inp, inp_parts = (hs_top, top_parts) if h is None else (h, h_parts)
            elif b.consumes:
                inp = cats[bi]
                pa, pb = cat_parts.get(bi, {}).get(0), cat_parts.get(bi, {}).get(b.ch_h)
                inp_parts = pa + pb if (pa is not None and pb is not None) else None
            else:
                inp, inp_parts = h, h_parts
            _, ih, iw, _, _ = _chk(inp)
            oh, ow = (ih // 2, iw // 2) if b.rs == H.RS_DOWN else ((ih * 2, iw * 2) if b.rs == H.RS_UP else (ih, iw))
            spec = b.dest if b.dest[0] == "cat" else ("plain", b.cout)
            out = dest_of(spec, oh, ow)
            if b.kind == "midattn":
                out_parts = self._attn_fwd(b.att, b.prefix, inp, out, tape, inp_parts)
            elif b.att is not None:
                mid = self._new(x_nchw, B, oh, ow, b.cout)
                mid_parts = self._res_fwd(b, b.res, b.prefix + ".0", inp, films, mid, p_drop, base_seed + 2 * bi + 1, tape, inp_parts)
                out_parts = self._attn_fwd(b.att, b.prefix + ".1", mid, out, tape, mid_parts)
            else:
                out_parts = self._res_fwd(b, b.res, b.prefix, inp, films, out, p_drop, base_seed + 2 * bi + 1, tape, inp_parts)
            if spec[0] == "cat":
                cat_parts.setdefault(spec[1], {})[spec[2]] = out_parts
            if b.kind == "down":
                hs_top, top_parts = out, out_parts
            else:
                h, h_parts = out, out_parts
        # out_conv: GN -> SiLU -> 3x3
        C0 = m.hid_channels * m.ch_multipliers[0]
        gn, conv = m.out_conv[0], m.out_conv[2]
        a = self._new(h, B, H0, W0, C0)
        coef = self._norm(h, h_parts, gn, None, 1, 0.0, 0, H.RS_NONE, a, B, H0, W0, C0)
        co = m.out_channels
        cop = (co + 3) // 4 * 4
        out = torch.zeros((B, H0, W0, cop), dtype=torch.float32, device=h.device)
        wz = None
        if co <= THIN:
            # thin output (hid -> 3): z[q][co*9+tap] = a[q] . w[co][tap] as a plain GEMM, then the 9-tap gather (vd_tap_gather)
            nz = (9 * co + 3) // 4 * 4
            wz = self._pack_thin(conv.weight, nz)
            z = self._new(h, B * H0 * W0, nz)
            H.gemm(a, wz, z, B * H0 * W0, 9 * co, C0, a_kind=H.ROW, b_kind=H.ROW, lda=C0, ldb=C0, ldc=nz)
            H.tap_gather(z, nz, conv.bias, out, cop, B, H0, W0, co)
            del z
        else:
            H.conv3x3(a, C0, self._pack_f(conv.weight), conv.bias, out, cop, B, H0, W0, C0, co)
        if tape is not None:
            tape["ta"] = ta
            tape["in"] = dict(x4=x4 if xc is None else None, xc=xc, cip=cip)
            tape["out"] = dict(h=h, coef=coef, a=a, wz=wz)
            tape["wino"] = getattr(self, "_wino", None)
        return out, tape

    def new_grads(self):
        return {k: torch.empty_like(p) for k, p in self.m.named_parameters()}

    def backward(self, tape, dout, G, need_dx=False, progress=None):
        """Fills ``G`` (name -> tensor, every entry overwritten) and returns d/dx (NCHW) when asked.  ``progress(name)`` is called whenever
        the gradient of ``name`` and of everything before it in completion_order() is final (gradient-bucket overlap).  The deferred
        weight-gradient queue and the side stream never outlive the call, whatever it raises (backward_steps' ``finally``)."""
        steps = self.backward_steps(tape, dout, G, need_dx, join=progress is not None)
        try:
            while True:
                name = next(steps)
                if progress is not None:
                    progress(name)
        except StopIteration as fin:
            return fin.value

    def progress_points(self):
        """The names ``backward_steps`` yields, in order (static: a function of the plan): the output convolution, the last block of every
        (UNet level, down / middle / up) group -- of every block under VD_READY_PER_BLOCK when a listener joins -- the input convolution,
        and None for the end of backward (GroupNorm parameters and the embeddings finish there)."""
        pts = ["out_conv.2.bias"]
        grp = lambda q: (q.level, "mid" if q.kind.startswith("mid") else q.kind)
        for bi in range(len(self.plan) - 1, -1, -1):
            b = self.plan[bi]
            nxt = self.plan[bi - 1] if bi > 0 else None
            if nxt is None or grp(nxt) != grp(b):
                pts.append(b.prefix + ".proj_in.bias" if b.kind == "midattn" else (b.prefix + (".0" if b.att is not None else "") + ".fc.bias"))
        return pts + ["in_conv.bias", None]

    def grad_segments(self):
        """[(boundary name, [parameter names])] in BACKWARD order: segment j's gradients are final when ``backward_steps`` yields its
        boundary; the last segment (input convolution, every GroupNorm scale / shift, the embeddings) is final when the pass ends
        (boundary None).  The cut points of the chain of autograd nodes of models/unet.py."""
        if self._segments is None:
            bounds = [p for p in self.progress_points() if p not in ("in_conv.bias", None)] + [None]
            order, segs, i = self.completion_order(), [], 0
            for bnd in bounds:
                names = []
                while i < len(order):
                    names.append(order[i])
                    i += 1
                    if order[i - 1] == bnd:
                        break
                segs.append((bnd, names))
            assert i == len(order) and all(n for _, n in segs)
            self._segments = segs
        return self._segments

    def completion_order(self):
        """Parameter names in the order ``backward_steps`` finishes their gradients: block by block from the output back (the
        matmul-shaped gradients, i.e. all the bytes), then the GroupNorm scales / shifts of the whole network (their per-image terms
        are summed by one launch at the end of backward: _pgb_finish), then the embeddings."""
        names = ["out_conv.2.weight", "out_conv.2.bias"]
        norms = ["out_conv.0.weight", "out_conv.0.bias"]
        have = dict(self.m.named_parameters())

        def res(p):
            out = [p + s for s in (".conv2.weight", ".conv2.bias", ".conv1.weight", ".conv1.bias")]
            norms.extend(p + s for s in (".norm2.weight", ".norm2.bias", ".norm1.weight", ".norm1.bias"))
            if p + ".skip.weight" in have:
                out += [p + ".skip.weight", p + ".skip.bias"]
            return out + [p + ".fc.weight", p + ".fc.bias"]

        def att(p):
            norms.extend(p + s for s in (".norm.weight", ".norm.bias"))
            return [p + s for s in (".proj_out.weight", ".proj_out.bias", ".proj_in.weight", ".proj_in.bias")]

        for b in reversed(self.plan):
            if b.kind == "midattn":
                names += att(b.prefix)
            elif b.att is not None:
                names += att(b.prefix + ".1") + res(b.prefix + ".0")
            else:
                names += res(b.prefix)
        names += ["in_conv.weight", "in_conv.bias"] + norms
        seen = set(names)
        names += [k for k in have if k not in seen]          # embeddings: finished last
        assert sorted(names) == sorted(have), "completion order does not cover the parameter set"
        return names

    def backward_steps(self, tape, dout, G, need_dx=False, join=True):
        """Generator form of the backward pass: runs up to the next point at which a prefix of completion_order() is final, yields the
        last finished name (progress_points() lists them), and returns d/dx through StopIteration.  ``join``: a listener consumes the
        gradients at every yield (a gradient reducer, or autograd handing a segment's gradients to DDP's hooks: models/unet.py), so the
        weight-gradient side stream is joined there; without one it is joined once, at the end."""
        try:
            dx = yield from self._backward(tape, dout, G, need_dx, join)
            return dx
        finally:
            self._wq = None
            self._side = None

    def _backward(self, tape, dout, G, need_dx=False, join=False):
        """dout: NHWC ``[B,H,W,Cp]`` gradient of the padded output (padding channels zero)."""
        m = self.m
        self._join_at_progress = bool(join)                    # a listener consumes gradients at every yield: finished means finished on every stream
        B, H0, W0, cop, _ = _chk(dout)
        self._wino = tape.get("wino", getattr(self, "_wino", None))     # the Winograd images THIS forward packed (another forward may have run since)
        ta = tape["ta"]
        dta = torch.zeros_like(ta)
        dfilms = {c2: self._new(ta, len(mods), B, c2) for c2, mods in self.film_groups.items()}
        self._pgb_begin(dout, B)
        self._wq = {}
        self._side = None
        if H.WGRAD_STREAM and H.PROFILE is None:
            if self._side_stream is None or self._side_stream.device != dout.device:
                self._side_stream = torch.cuda.Stream(device=dout.device)
            self._side = self._side_stream
        # ---- out_conv
        C0 = m.hid_channels * m.ch_multipliers[0]
        gn, conv = m.out_conv[0], m.out_conv[2]
        o = tape["out"]
        co = m.out_channels
        P0 = B * H0 * W0
        da = self._new(dout, B, H0, W0, C0)
        if o["wz"] is not None:
            wz = o["wz"]
            nz = wz.shape[0]
            dz = self._new(dout, P0, nz)
            H.tap_spread(dout, cop, dz, nz, B, H0, W0, co)                   # dz[q][co*9+tap] = dout[q - off(tap)][co]
            gz, cs = self._new(dout, nz, C0), self._new(dout, nz)
            H.gemm(dz, o["a"], gz, nz, C0, P0, a_kind=H.COL, b_kind=H.COL, lda=nz, ldb=C0, ldc=C0, splitk=_splitk(nz, C0, P0),
                   colsum=cs)
            H.thin_wgrad_finish(gz, co, C0, C0, G["out_conv.2.weight"], colsum=cs, cs_stride=9, cs_off=4, dbias=G["out_conv.2.bias"])
            H.gemm(dz, wz, da, P0, C0, nz, a_kind=H.ROW, b_kind=H.COL, lda=nz, ldb=C0, ldc=C0)
            del dz
        else:
            H.conv3x3_wgrad(o["a"], C0, dout, cop, B, H0, W0, C0, cop, G["out_conv.2.weight"], C0, co, dbias=G["out_conv.2.bias"])
            H.conv3x3(dout, cop, self._pack_d(conv.weight, cop), None, da, C0, B, H0, W0, cop, C0)
        dh = self._new(dout, B, H0, W0, C0)
        H.gn_apply_bwd(da, C0, o["h"], _ld(o["h"]), o["coef"], gn.weight, gn.bias, None, 1, 0.0, 0, H.RS_NONE, None, 0, dh, C0,
                       False, None, None, None, False, B, H0, W0, C0, GROUPS,
                       pgb_keep=self._pgb(B, C0, G["out_conv.0.weight"], G["out_conv.0.bias"]))
        del da
        yield "out_conv.2.bias"                # (the GroupNorm parameter gradients are finished by ONE launch at the end: _pgb_finish)
        # ---- blocks in reverse
        dskip = {}                     # hs id -> gradient view (written by the consuming up-block)
        dh_cur = dh                    # gradient of the running `h`
        for bi in range(len(self.plan) - 1, -1, -1):
            b = self.plan[bi]
            first_ctx = tape[b.prefix + ".0"] if (b.att is not None and b.kind != "midattn") else tape[b.prefix]
            xin = first_ctx["x"]
            Bx, ih, iw, cin, _ = _chk(xin)
            if b.kind == "down":
                dy = dskip.pop(b.push_hs)                       # total gradient of the tensor this block pushed
            else:
                dy = dh_cur
            # where does the input gradient go?  Blocks that read the top of the skip stack (every down block and
            # middle.0) accumulate into the slot the consuming up-block already filled; everything else gets a fresh buffer.
            if b.kind == "down" or b.prefix == "middle.0":
                dxbuf, acc = dskip[self._hs_feeding(bi)], True
            else:
                dxbuf, acc = self._new(dout, Bx, ih, iw, cin), False
            if b.kind == "midattn":
                self._attn_bwd(b.att, b.prefix, tape[b.prefix], dy, dxbuf, acc, G)
            elif b.att is not None:
                dmid = self._new(dout, *tape[b.prefix + ".1"]["x"].shape)
                self._attn_bwd(b.att, b.prefix + ".1", tape[b.prefix + ".1"], dy, dmid, False, G)
                self._res_bwd(b, b.res, b.prefix + ".0", tape[b.prefix + ".0"], dmid, dxbuf, acc, ta, dfilms, G)
                del dmid
            else:
                self._res_bwd(b, b.res, b.prefix, tape[b.prefix], dy, dxbuf, acc, ta, dfilms, G)
            # the deferred weight-gradient GEMMs of a level run when backward leaves it; only then is everything up to this
            # block's last tensor final (gradient-bucket overlap: trainer.GradReducer.ready)
            nxt = self.plan[bi - 1] if bi > 0 else None
            grp = lambda q: (q.level, "mid" if q.kind.startswith("mid") else q.kind)
            # with a gradient reducer listening and H.READY_PER_BLOCK, the queue is flushed (and the gradients declared final) after EVERY
            # block instead of every UNet level: buckets leave as early as they can (33 report points instead of 7 for CIFAR) at the price
            # of ungrouped 1x1 / linear weight gradients and a side-stream join per block (measured cost: DESIGN section 4)
            if nxt is None or grp(nxt) != grp(b) or (self._join_at_progress and H.READY_PER_BLOCK):
                self._wgrad_flush()
                if self._join_at_progress:
                    self._side_join()
                yield (b.prefix + ".proj_in.bias" if b.kind == "midattn" else (b.prefix + (".0" if b.att is not None else "") + ".fc.bias"))
            if b.kind == "up" and b.consumes:
                dh_cur = dxbuf[..., :b.ch_h]
                dskip[b.src_hs] = dxbuf[..., b.ch_h:]
            elif not acc:
                dh_cur = dxbuf
        # ---- in_conv
        dy0 = dskip.pop(0)
        assert not dskip
        xc, cip = tape["in"]["xc"], tape["in"]["cip"]
        hid = m.hid_channels
        if xc is not None:
            gw = self._new(dout, hid, 9 * cip)
            H.gemm(dy0, xc, gw, hid, 9 * cip, P0, a_kind=H.COL, b_kind=H.COL, lda=_ld(dy0), ldb=9 * cip, ldc=9 * cip,
                   splitk=_splitk(hid, 9 * cip, P0), colsum=G["in_conv.bias"])
            H.thin_wgrad_finish(gw, hid, cip, m.in_channels, G["in_conv.weight"])
        else:
            H.conv3x3_wgrad(tape["in"]["x4"], cip, dy0, _ld(dy0), B, H0, W0, cip, hid, G["in_conv.weight"], m.in_channels, hid,
                            dbias=G["in_conv.bias"])
        yield "in_conv.bias"
        dx = None
        if need_dx:
            d4 = self._new(dout, B, H0, W0, cip)
            if cip != m.in_channels:
                d4.zero_()
            H.conv3x3(dy0, _ld(dy0), self._pack_d(m.in_conv.weight, m.hid_channels), None, d4, cip, B, H0, W0, m.hid_channels,
                      m.in_channels)
            dx = self._new(dout, B, m.in_channels, H0, W0)
            H.nhwc_to_nchw(d4, cip, dx, B, m.in_channels, H0, W0)
        self._pgb_finish()
        self._film_bwd(dfilms, tape["film_w"], dta)
        self._embed_bwd(tape["embed"], dta, G)
        self._wgrad_flush()
        self._side_join()
        self._side = None
        self._wq = None
        yield None
        return dx

    def _hs_feeding(self, bi):
        """id of the skip-stack entry that block ``bi`` reads as its input (the stack top at that time)."""
        b = self.plan[bi]
        if b.kind == "down":
            return b.push_hs - 1
        return len(self.pushes) - 1          # middle.0 reads the last pushed tensor
