"""ctypes binding of libvdiff_hip.so (C ABI: include/vdiff_hip.h).

PyTorch is used only for device memory and streams: every function here takes torch CUDA(HIP) tensors,
passes raw pointers + the current stream to the HIP library and returns nothing the library allocated.
There is NO fallback: a missing library, a CPU tensor or a non-zero return code raises.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VDIFF_HIP_LIB") or os.path.join(os.path.dirname(_HERE), "lib", "libvdiff_hip.so")   # (override: A/B builds)

ROW, COL, IM2COL = 0, 1, 2
RS_NONE, RS_DOWN, RS_UP = 0, 1, 2
OUT_TYPES = {"v": 0, "x0": 1, "eps": 2, "both": 3}
REWEIGHTS = {"constant": 0, "snr": 1, "snr_trunc": 2, "snr_1plus": 3}

_i32, _i64, _f32, _f64, _u64, _vp, _sz = C.c_int32, C.c_int64, C.c_float, C.c_double, C.c_uint64, C.c_void_p, C.c_size_t


class GemmDesc(C.Structure):
    _fields_ = [("A", _vp), ("B", _vp), ("C", _vp), ("bias", _vp), ("R", _vp),
                ("M", _i32), ("N", _i32), ("K", _i32), ("a_kind", _i32), ("b_kind", _i32),
                ("lda", _i64), ("ldb", _i64), ("ldc", _i64), ("ldr", _i64),
                ("batch", _i32), ("nh", _i32),
                ("sAb", _i64), ("sAh", _i64), ("sBb", _i64), ("sBh", _i64),
                ("sCb", _i64), ("sCh", _i64), ("sRb", _i64), ("sRh", _i64),
                ("alpha", _f32), ("accumulate", _i32), ("H", _i32), ("W", _i32), ("Cin", _i32),
                ("splitk", _i32), ("ws", _vp), ("ws_bytes", _i64), ("tile", _i32),
                ("colsum", _vp), ("colsum_accumulate", _i32), ("stats", _vp), ("stats_hw", _i32), ("sBias", _i64)]


_SIGNATURES = {
    "vd_version": (C.c_int, []),
    "vd_last_error": (C.c_char_p, []),
    "vd_set_reserved_cus": (C.c_int, [_i32]),
    "vd_reserved_cus": (C.c_int, []),
    "vd_mfma_calibrate": (C.c_int, [_vp, _vp, _i32, _i32, C.c_uint32, _vp]),
    "vd_gemm": (C.c_int, [C.POINTER(GemmDesc), _vp]),
    "vd_gemm_last_tile": (C.c_int, []),
    "vd_gemm_split_forms": (C.c_int, []),
    "vd_gemm_grouped_wgrad_ws_bytes": (_sz, [_i32, _i32, _i32, _i32]),
    "vd_gemm_grouped_wgrad_auto_split": (C.c_int, [_i32, _i32, _i32, _i32, _i32, _i32]),
    "vd_gemm_grouped_wgrad": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i64, _i64, _i64, _i32, _vp, _sz, _vp]),
    "vd_conv3x3": (C.c_int, [_vp, _i64, _vp, _vp, _vp, _i64, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "vd_conv3x3_wino_supported": (C.c_int, [_i32, _i32, _i32, _i32, _i32, _i64, _i64, _i64]),
    "vd_conv3x3_wino": (C.c_int, [_vp, _i64, _vp, _vp, _vp, _i64, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "vd_wino_pack": (C.c_int, [_vp, _i32, _i32, _vp, _vp, _vp]),
    "vd_conv3x3_wgrad_wino_supported": (C.c_int, [_i32, _i32, _i32, _i32, _i32, _i64, _i64]),
    "vd_conv3x3_wgrad_wino_ws_bytes": (_sz, [_i32, _i32, _i32, _i32, _i32]),
    "vd_conv3x3_wgrad_wino": (C.c_int, [_vp, _i64, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _i32, _i32, _i32, _vp, _sz, _vp]),
    "vd_conv3x3_wgrad_wino_phase": (C.c_int, [_vp, _i64, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _i32, _i32, _i32, _vp, _sz, _i32, _vp]),
    "vd_wino_last_kernel": (C.c_int, []),
    "vd_wino_wgrad_last_kernel": (C.c_int, []),
    "vd_wino_pack_batched": (C.c_int, [_vp, _i32, _i64, _vp]),
    "vd_conv3x3_dgrad_wino43_supported": (C.c_int, [_i32, _i32, _i32, _i32, _i32, _i64, _i64]),
    "vd_wino43_u_floats": (_sz, [_i32, _i32]),
    "vd_conv3x3_dgrad_wino43": (C.c_int, [_vp, _i64, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _vp]),
    "vd_wino43_last_kernel": (C.c_int, []),
    "vd_conv3x3_wino43_fwd_supported": (C.c_int, [_i32, _i32, _i32, _i32, _i32, _i64, _i64, _i64]),
    "vd_conv3x3_wino43_preferred": (C.c_int, [_i32, _i32, _i32, _i32]),
    "vd_wino43_pack_fwd": (C.c_int, [_vp, _i32, _i32, _vp, _vp]),
    "vd_conv3x3_wino43_fwd_chunk_rows": (C.c_int, [_i32, _i32]),
    "vd_conv3x3_wino43_fwd": (C.c_int, [_vp, _i64, _vp, _vp, _vp, _i64, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "vd_wino43_pack": (C.c_int, [_vp, _i32, _i32, _vp, _vp]),
    "vd_wino43_pack_batched": (C.c_int, [_vp, _i32, _i64, _vp]),
    "vd_conv3x3_wgrad_wino43_supported": (C.c_int, [_i32, _i32, _i32, _i32, _i32, _i64, _i64]),
    "vd_conv3x3_wgrad_wino43_ws_bytes": (_sz, [_i32, _i32, _i32, _i32, _i32]),
    "vd_conv3x3_wgrad_wino43": (C.c_int, [_vp, _i64, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _i32, _i32, _i32, _vp, _sz, _vp]),
    "vd_wino43_wgrad_last_kernel": (C.c_int, []),
    "vd_conv3x3_wgrad_wino43_phase": (C.c_int, [_vp, _i64, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _i32, _i32, _i32, _vp, _sz, _i32, _vp]),
    "vd_gn_stats_from_partials": (C.c_int, [_vp, _i32, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _f32, _vp, _vp]),
    "vd_gn_coef_from_partials": (C.c_int, [_vp, _i32, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _f32, _vp, _vp, _vp, _vp, _vp]),
    "vd_conv3x3_wgrad_ws_bytes": (_sz, [_i32, _i32, _i32, _i32, _i32]),
    "vd_conv3x3_wgrad": (C.c_int, [_vp, _i64, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _i32, _i32, _i32, _vp, _sz, _vp]),
    "vd_conv3x3_wgrad_phase": (C.c_int, [_vp, _i64, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _i32, _i32, _i32, _vp, _sz, _i32, _vp]),
    "vd_pack_conv3x3": (C.c_int, [_vp, _i32, _i32, _vp, _i32, _vp, _i32, _vp]),
    "vd_pack_conv3x3_batched": (C.c_int, [_vp, _i32, _i64, _vp]),
    "vd_gn_ws_bytes": (_sz, [_i32, _i32, _i32]),
    "vd_gn_stats": (C.c_int, [_vp, _i64, _i32, _i32, _i32, _i32, _f32, _vp, _vp, _sz, _vp]),
    "vd_gn_apply": (C.c_int, [_vp, _i64, _vp, _vp, _vp, _vp, _i32, _f32, _u64, _i32, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "vd_gn_apply_from_partials": (C.c_int, [_vp, _i64, _vp, _i32, _i32, _vp, _i32, _i32, _vp, _vp, _vp, _i32, _f32, _u64, _i32, _vp, _i64, _i32,
                                            _i32, _i32, _i32, _i32, _f32, _vp, _vp]),
    "vd_gn_apply_bwd": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _i32, _f32, _u64, _i32, _vp, _i64, _vp, _i64, _i32,
                                  _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _sz, _vp]),
    "vd_gn_bwd_last_kernel": (C.c_int, []),
    "vd_gn_apply_bwd_keep": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _i32, _f32, _u64, _i32, _vp, _i64, _vp, _i64, _i32,
                                       _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _sz, _vp]),
    "vd_gn_param_sums_batched": (C.c_int, [_vp, _i32, _i64, _vp]),
    "vd_colsum_ws_bytes": (_sz, [_i64, _i32]),
    "vd_colsum": (C.c_int, [_vp, _i64, _i64, _i32, _vp, _i32, _vp, _sz, _vp]),
    "vd_axpby": (C.c_int, [_vp, _i64, _f32, _vp, _i64, _f32, _i64, _i32, _vp]),
    "vd_silu": (C.c_int, [_vp, _vp, _i64, _vp]),
    "vd_silu_bwd": (C.c_int, [_vp, _vp, _vp, _i64, _i32, _vp]),
    "vd_softmax_rows": (C.c_int, [_vp, _i64, _i32, _vp]),
    "vd_softmax_rows_bwd": (C.c_int, [_vp, _vp, _i64, _i32, _f32, _vp]),
    "vd_attn_supported": (C.c_int, [_i32, _i32, _i32]),
    "vd_attn_fwd": (C.c_int, [_vp, _vp, _vp, _i64, _vp, _i64, _vp, _i32, _i32, _i32, _i32, _f32, _vp]),
    "vd_attn_bwd": (C.c_int, [_vp, _vp, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _f32, _vp]),
    "vd_attn_bwd_phase": (C.c_int, [_vp, _vp, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _f32,
                                    _i32, _vp]),
    "vd_nchw_to_nhwc": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _i64, _vp]),
    "vd_nhwc_to_nchw": (C.c_int, [_vp, _i64, _vp, _i32, _i32, _i32, _i32, _vp]),
    "vd_images_to_uint8_hwc": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _vp]),
    "vd_images_from_uint8_hwc": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "vd_timestep_embedding": (C.c_int, [_vp, _vp, _i32, _i32, _f64, _vp]),
    "vd_class_embed": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp]),
    "vd_class_embed_bwd": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "vd_multitag_norm": (C.c_int, [_vp, _vp, _i32, _i32, _vp]),
    "vd_im2col3x3": (C.c_int, [_vp, _i64, _vp, _i32, _i32, _i32, _i32, _vp]),
    "vd_tap_gather": (C.c_int, [_vp, _i64, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _vp]),
    "vd_tap_spread": (C.c_int, [_vp, _i64, _vp, _i64, _i32, _i32, _i32, _i32, _vp]),
    "vd_thin_wgrad_finish": (C.c_int, [_vp, _i32, _i32, _i32, _vp, _i32, _vp, _i32, _i32, _vp, _vp]),
    "vd_q_sample": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp]),
    "vd_loss_fwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _i32, _i32, _i32, _vp]),
    "vd_loss_bwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _i32, _i32, _i32, _vp]),
    "vd_bpd_terms": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp]),
    "vd_bpd_bwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _i32, _i32, _i32, _vp]),
    "vd_sample_step": (C.c_int, [_vp, _vp, _vp, C.POINTER(_f32), _vp, _i32, _i32, _i32, _i32, _vp, _vp, _i32, _i32, _i32, _vp]),
    "vd_sumsq_ws_bytes": (_sz, [_i64]),
    "vd_sumsq": (C.c_int, [_vp, _i64, _vp, _vp, _sz, _vp]),
    "vd_adamw_ema": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _vp, _f32, _f32, _f32, _f32, _f32, _f32, _f32, _f32, _f32, _i64, _i64, _i32,
                               _f32, _f32, _vp]),
    "vd_adamw_ema_flagged": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _vp, _f32, _f32, _f32, _f32, _f32, _f32, _f32, _f32, _f32, _i64, _i64,
                                       _vp, _vp, _f64, _f64, _vp]),
}
EXPORTS = tuple(_SIGNATURES)
# extra entry points of libvdiff_hip_probe.so (built with -DVD_PROBES; tests/probe/*.py load it through VDIFF_HIP_LIB): bound when
# the loaded library has them, absent from the product library
PROBE_EXPORTS = {"vd_wino_set_probe": (C.c_int, [_vp]), "vd_wino43_set_probe": (C.c_int, [_vp]), "vd_gn_set_spin_probe": (C.c_int, [_vp])}

_lib = None


def lib():
    """Load the HIP library (raises if it has not been built -- there is no CPU fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: build it with `python __graft_entry__.py` (or make -C v-diffusion-torch_amd/csrc). "
                "The v_diffusion hot path has no CPU/PyTorch fallback.")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(l, name)           # AttributeError if the symbol is missing
            fn.restype, fn.argtypes = res, args
        for name, (res, args) in PROBE_EXPORTS.items():
            if hasattr(l, name):
                fn = getattr(l, name)
                fn.restype, fn.argtypes = res, args
        _lib = l
    return _lib


class HipError(RuntimeError):
    pass


def _check(rc, what):
    if rc != 0:
        raise HipError(f"{what} failed (rc={rc}): {lib().vd_last_error().decode(errors='replace')}")


def ptr(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise HipError("v_diffusion HIP op received a CPU tensor: the hot path runs on an MI355X only (no CPU fallback)")
    if t.dtype not in (torch.float32, torch.float64, torch.uint8, torch.int32):
        raise HipError(f"unsupported dtype {t.dtype}")
    return t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


_ws_cache = {}

# When set to a list, the matmul-shaped wrappers append (kernel_name, flops, start_event, end_event) per launch
# (bench.py uses it for the live roofline figure; events sit on the stream the kernels are launched on).  Names are the
# rocprofv3 kernel names: gemm_dma_kernel<BM, BN, a_kind, b_kind, splitk, KT, TR> with kinds 0 = ROW, 1 = COL, 2 = IM2COL
# (TR = transposed-accumulator epilogue, taken by launches without output statistics).
PROFILE = None
PROFILE_BYTES = {}            # kernel name -> [algorithmic bytes (operands read + written once) summed over its recorded launches, launches]
_KIND = {0: "ROW", 1: "COL", 2: "IM2COL"}


def _note_bytes(name, nbytes):
    if PROFILE is not None:
        e = PROFILE_BYTES.setdefault(name, [0.0, 0])
        e[0] += float(nbytes)
        e[1] += 1


class _Timed:
    def __init__(self, name, flops):
        self.name, self.flops = name, flops

    def __enter__(self):
        if PROFILE is not None:
            self.e0, self.e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *exc):
        if PROFILE is not None and exc[0] is None:
            self.e1.record()
            t = lib().vd_gemm_last_tile()
            # hundreds digit of the K-tile group: 1 = transposed epilogue, 2 = split-operand form (VD_GEMM_SPLIT=1), 3 = both
            flags, kt, bm, bn = t // 100000000, (t // 1000000) % 100, (t // 1000) % 1000, t % 1000
            tr, spl = flags & 1, flags >= 2
            name = self.name.format(tile=f"{bm}, {bn}", kt=f"{kt}, {'true' if tr else 'false'}")
            if spl:
                name = name.replace("gemm_dma_kernel", "gemm_split_kernel")
            if spl and (bm, bn) == (256, 256):          # round 6: the 8-wave 256x256 form of the grouped weight-gradient launches (its rocprof name)
                name = "wgrad_planes256_kernel" + name[name.index(">") + 1:]
            if kt == 0:
                name = name.replace("gemm_dma_kernel", "gemm_kernel").replace(", 0, false>", ">")
            PROFILE.append((name, self.flops, self.e0, self.e1))


class _TimedName(_Timed):
    """PROFILE record under a fixed kernel name (launches whose instantiation does not come from vd_gemm_last_tile)"""

    def __exit__(self, *exc):
        if PROFILE is not None and exc[0] is None:
            self.e1.record()
            PROFILE.append((self.name, self.flops, self.e0, self.e1))


def workspace(nbytes, device, tag="default"):
    """Per-(device, stream, tag) scratch buffer, grown on demand.  Kernels on one stream are ordered, so a scratch
    region can be reused by the next launch on the same stream."""
    key = (device, torch.cuda.current_stream(device).cuda_stream, tag)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() * 4 < nbytes:
        buf = torch.empty((max(int(nbytes), 1 << 20) + 3) // 4, dtype=torch.float32, device=device)
        _ws_cache[key] = buf
    return buf


# ----------------------------------------------------------------------------------------------- wrappers
def gemm(A, B, Cm, M, N, K, *, a_kind=ROW, b_kind=ROW, lda, ldb, ldc, bias=None, R=None, ldr=0, batch=1, nh=1,
         sA=(0, 0), sB=(0, 0), sC=(0, 0), sR=(0, 0), alpha=1.0, accumulate=False, splitk=1, tile=0, colsum=None,
         colsum_accumulate=False, stats=None, stats_hw=0, sBias=0):
    d = GemmDesc()
    d.A, d.B, d.C, d.bias, d.R = ptr(A), ptr(B), ptr(Cm), ptr(bias), ptr(R)
    d.M, d.N, d.K, d.a_kind, d.b_kind = M, N, K, a_kind, b_kind
    d.lda, d.ldb, d.ldc, d.ldr = lda, ldb, ldc, ldr
    d.batch, d.nh = batch, nh
    (d.sAb, d.sAh), (d.sBb, d.sBh), (d.sCb, d.sCh), (d.sRb, d.sRh) = sA, sB, sC, sR
    d.alpha, d.accumulate = alpha, int(accumulate)
    d.splitk, d.tile = splitk, tile
    d.colsum, d.colsum_accumulate = ptr(colsum), int(colsum_accumulate)
    d.stats, d.stats_hw = ptr(stats), stats_hw
    d.sBias = sBias
    if splitk > 1:
        ws = workspace(splitk * (M * N + M) * 4, A.device, "splitk")
        d.ws, d.ws_bytes = ws.data_ptr(), ws.numel() * 4
    with _Timed("gemm_dma_kernel<{tile}, " + f"{a_kind}, {b_kind}, " + ("true, {kt}>" if splitk > 1 else "false, {kt}>"),
                2.0 * M * N * K * batch):
        _check(lib().vd_gemm(C.byref(d), stream()), "vd_gemm")


GROUP_MAX = 36
GROUPED_WGRAD = os.environ.get("VD_GROUPED_WGRAD", "1") != "0"   # A/B switch: 0 = one launch per weight-gradient GEMM


def gemm_grouped_wgrad(entries, M, N, K, lda, ldb, ldc, splitk):
    """entries: [(A = dY rows [K][M], B = X rows [K][N], C = dW [M][N], colsum = dbias [M] or None)], at most GROUP_MAX, all of one shape:
    C = A^T B (+ colsum) for every entry in ONE launch (vd_gemm_grouped_wgrad)"""
    n = len(entries)
    assert 0 < n <= GROUP_MAX
    arr = lambda i: (_vp * n)(*[ptr(e[i]) for e in entries])
    has_cs = entries[0][3] is not None
    assert all((e[3] is not None) == has_cs for e in entries)
    S = max(1, int(splitk))
    ws = workspace(lib().vd_gemm_grouped_wgrad_ws_bytes(n, M, N, S), entries[0][0].device, "splitk")
    with _Timed("gemm_dma_kernel<{tile}, 1, 1, true, 32, true, true>", 2.0 * M * N * K * n):
        _check(lib().vd_gemm_grouped_wgrad(arr(0), arr(1), arr(2), arr(3) if has_cs else None, n, M, N, K, lda, ldb, ldc, S,
                                           ws.data_ptr(), ws.numel() * 4, stream()), "vd_gemm_grouped_wgrad")


def last_row_tile():
    """BM of the calling thread's last vd_gemm / vd_conv3x3 launch (chunk size of its output statistics = BM/2)"""
    return (lib().vd_gemm_last_tile() // 1000) % 1000


def stats_part_numel(nimg, HW, Cc):
    """floats needed for the GroupNorm partials of an (nimg, HW, Cc) output, whatever row tile the launch picks"""
    return nimg * max(HW // 32, 1) * 2 * Cc


def gn_stats_from_partials(parts, nimg, HW, stats, G=32, eps=1e-6):
    """parts: [(part, C, chunks)] for one or two channel-concatenated sources"""
    (p1, c1, k1), (p2, c2, k2) = parts[0], (parts[1] if len(parts) > 1 else (None, 0, 0))
    _check(lib().vd_gn_stats_from_partials(ptr(p1), c1, k1, ptr(p2), c2, k2, nimg, HW, G, eps, ptr(stats), stream()),
           "vd_gn_stats_from_partials")


def gn_coef_from_partials(parts, nimg, HW, gamma, beta, film, coef, G=32, eps=1e-6):
    (p1, c1, k1), (p2, c2, k2) = parts[0], (parts[1] if len(parts) > 1 else (None, 0, 0))
    _check(lib().vd_gn_coef_from_partials(ptr(p1), c1, k1, ptr(p2), c2, k2, nimg, HW, G, eps, ptr(gamma), ptr(beta), ptr(film),
                                          ptr(coef), stream()), "vd_gn_coef_from_partials")


WINO = os.environ.get("VD_WINO", "1") != "0"      # A/B switch: 0 sends every 3x3 convolution to the direct implicit GEMM


def wino_supported(nimg, H, W, Cin, Cout, ldx, ldy, ldres=0):
    return WINO and bool(lib().vd_conv3x3_wino_supported(nimg, H, W, Cin, Cout, ldx, ldy, ldres))


def wino_pack(w, Cout, Cin, uf=None, ud=None):
    _check(lib().vd_wino_pack(ptr(w), Cout, Cin, ptr(uf), ptr(ud), stream()), "vd_wino_pack")


def wino_pack_batched(table, n, total_blocks):
    assert table.is_cuda and table.dtype == torch.int64 and table.is_contiguous()
    _check(lib().vd_wino_pack_batched(table.data_ptr(), n, total_blocks, stream()), "vd_wino_pack_batched")


def conv3x3_wino(x, ldx, U, bias, y, ldy, nimg, H, W, Cin, Cout, res=None, ldres=0, stats_part=None):
    """Winograd F(2x2,3x3) form (see vd_conv3x3_wino); `flops` recorded = the direct convolution's (algorithmic) count, of
    which the matrix cores execute 4/9"""
    with _TimedName("wino_conv_kernel<{tw}, {ns}, {st}, false, 0>", 2.0 * nimg * H * W * Cout * 9 * Cin) as t:
        _check(lib().vd_conv3x3_wino(ptr(x), ldx, ptr(U), ptr(bias), ptr(res), ldres, ptr(y), ldy, nimg, H, W, Cin, Cout,
                                     ptr(stats_part), stream()), "vd_conv3x3_wino")
        if PROFILE is not None:
            k = lib().vd_wino_last_kernel()
            if k < 0:                       # the 128-tile ("wide") form
                k = -k
                t.name = f"wino_conv_wide_kernel<{k // 2000}, {(k // 2) % 1000}, {'true' if k & 1 else 'false'}>"
            else:
                t.name = t.name.format(tw=k // 2000, ns=(k // 2) % 1000, st="true" if k & 1 else "false")
            _note_bytes(t.name, 4.0 * (nimg * H * W * (Cin + Cout + (Cout if res is not None else 0)) + 16 * Cout * Cin))


def mfma_calibrate(seconds=1.5, blocks=256, iters=20000):
    """Run the register-only fp32 MFMA loop of vd_mfma_calibrate back to back for `seconds` (the part's clock settles under load) and
    return (in-kernel clock in MHz, TFLOP/s) of the last launch: a figure bench lines from different boxes can be normalised by."""
    import time
    sink = torch.zeros(1, device="cuda")
    stamps = torch.zeros(2 * blocks, dtype=torch.int64, device="cuda")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t_end = time.perf_counter() + seconds
    ms = 0.0
    while True:
        e0.record()
        _check(lib().vd_mfma_calibrate(sink.data_ptr(), stamps.data_ptr(), blocks, iters, 12345, stream()), "vd_mfma_calibrate")
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        if time.perf_counter() >= t_end:
            break
    st = stamps.view(blocks, 2).double().cpu()
    mhz = float((st[:, 0] / st[:, 1].clamp(min=1)).median()) * 100.0
    tflops = blocks * 8 * iters * 16 * 2048.0 / (ms * 1e-3) / 1e12
    return mhz, tflops


WINO43 = os.environ.get("VD_WINO43", "1") != "0"  # A/B switch: 0 keeps the input gradients on F(2x2,3x3)
WINO43_MIN_W = int(os.environ.get("VD_WINO43_MIN_W", "16"))   # A/B switch: 32 keeps the 16x16 layers on F(2x2,3x3)


# occupancy rule between F(4x4,3x3) and F(2x2,3x3) (vd_conv3x3_wino43_preferred: small batches keep the finer F(2x2,3x3) items);
# VD_WINO43_OCC=0: F(4x4,3x3) wherever served (A/B switch)
WINO43_OCC = os.environ.get("VD_WINO43_OCC", "1") != "0"


def wino43_preferred(nimg, H, W, N):
    return not WINO43_OCC or bool(lib().vd_conv3x3_wino43_preferred(nimg, H, W, N))


def wino43_supported(nimg, H, W, Cin, Cout, lddy, lddx):
    """input gradient of a Cin -> Cout convolution on (nimg, H, W) images through F(4x4,3x3)?"""
    return (WINO and WINO43 and W >= WINO43_MIN_W and bool(lib().vd_conv3x3_dgrad_wino43_supported(nimg, H, W, Cin, Cout, lddy, lddx))
            and wino43_preferred(nimg, H, W, Cin))


def wino43_pack(w, Cout, Cin, U43):
    _check(lib().vd_wino43_pack(ptr(w), Cout, Cin, ptr(U43), stream()), "vd_wino43_pack")


def wino43_pack_batched(table, n, total_blocks):
    assert table.is_cuda and table.dtype == torch.int64 and table.is_contiguous()
    _check(lib().vd_wino43_pack_batched(table.data_ptr(), n, total_blocks, stream()), "vd_wino43_pack_batched")


# forward convolutions of the 16x16 / 32x32 / 64-wide layers through F(4x4,3x3) with the dyadic interpolation points (csrc/wino43.hip:
# 1.78x fewer matrix-core cycles than F(2x2,3x3), ~2x its output error, inside the stated 2e-5 bound); VD_WINO43_FWD=0 keeps F(2x2,3x3)
WINO43_FWD = os.environ.get("VD_WINO43_FWD", "1") != "0"


def wino43_fwd_supported(nimg, H, W, Cin, Cout, ldx, ldy, ldres=0):
    return (WINO and WINO43_FWD and bool(lib().vd_conv3x3_wino43_fwd_supported(nimg, H, W, Cin, Cout, ldx, ldy, ldres))
            and wino43_preferred(nimg, H, W, Cout))


def wino43_pack_fwd(w, Cout, Cin, U43f):
    _check(lib().vd_wino43_pack_fwd(ptr(w), Cout, Cin, ptr(U43f), stream()), "vd_wino43_pack_fwd")


def wino43_fwd_chunk_rows(H, W):
    return int(lib().vd_conv3x3_wino43_fwd_chunk_rows(H, W))


def conv3x3_wino43_fwd(x, ldx, U43f, bias, y, ldy, nimg, H, W, Cin, Cout, res=None, ldres=0, stats_part=None):
    """y = conv3x3(x, w) + bias (+ res) as Winograd F(4x4,3x3) (vd_conv3x3_wino43_fwd); `flops` recorded = the direct convolution's
    (algorithmic) count, of which the matrix cores execute 1/4"""
    name = f"wino43_conv_kernel<{W // 4}, true>"
    with _TimedName(name, 2.0 * nimg * H * W * Cout * 9 * Cin):
        _check(lib().vd_conv3x3_wino43_fwd(ptr(x), ldx, ptr(U43f), ptr(bias), ptr(res), ldres, ptr(y), ldy, nimg, H, W, Cin, Cout,
                                           ptr(stats_part), stream()), "vd_conv3x3_wino43_fwd")
    _note_bytes(name, 4.0 * (nimg * H * W * (Cin + Cout + (Cout if res is not None else 0)) + 36 * Cout * Cin))


def conv3x3_dgrad_wino43(dy, lddy, U43, dx, lddx, nimg, H, W, Cin, Cout):
    """dx = input gradient of the Cin -> Cout 3x3 convolution (Winograd F(4x4,3x3), see vd_conv3x3_dgrad_wino43); `flops` recorded =
    the direct convolution's (algorithmic) count, of which the matrix cores execute 1/4"""
    with _TimedName(f"wino43_conv_kernel<{W // 4}, false>", 2.0 * nimg * H * W * Cout * 9 * Cin):
        _check(lib().vd_conv3x3_dgrad_wino43(ptr(dy), lddy, ptr(U43), ptr(dx), lddx, nimg, H, W, Cin, Cout, stream()),
               "vd_conv3x3_dgrad_wino43")
    _note_bytes(f"wino43_conv_kernel<{W // 4}, false>", 4.0 * (nimg * H * W * (Cin + Cout) + 36 * Cout * Cin))


def conv3x3(x, ldx, wpack, bias, y, ldy, nimg, H, W, Cin, Cout, res=None, ldres=0, accumulate=False, stats_part=None):
    with _Timed("gemm_dma_kernel<{tile}, 2, 0, false, {kt}>", 2.0 * nimg * H * W * Cout * 9 * Cin):
        _check(lib().vd_conv3x3(ptr(x), ldx, ptr(wpack), ptr(bias), ptr(res), ldres, ptr(y), ldy, nimg, H, W, Cin, Cout,
                                int(accumulate), ptr(stats_part), stream()), "vd_conv3x3")


def conv3x3_wgrad_wino(x, ldx, dy, lddy, nimg, H, W, Cin, Cout, dw, Cin_w, Cout_w, accumulate=False, dbias=None):
    nb = lib().vd_conv3x3_wgrad_wino_ws_bytes(nimg, H, W, Cin, Cout)
    ws = workspace(nb, x.device, "wgrad")
    args = (ptr(x), ldx, ptr(dy), lddy, nimg, H, W, Cin, Cout, ptr(dw), ptr(dbias), Cin_w, Cout_w, int(accumulate), ws.data_ptr(),
            ws.numel() * 4)
    if PROFILE is None:
        _check(lib().vd_conv3x3_wgrad_wino(*args, stream()), "vd_conv3x3_wgrad_wino")
        return
    # per-kernel timing: the MFMA kernel and the plane reduction get their own event pairs (same kernels, same order)
    tws = min(W // 2, 16)
    with _TimedName(f"wino_wgrad_kernel<{tws}, {'true' if dbias is not None else 'false'}>", 2.0 * nimg * H * W * Cout * 9 * Cin):
        _check(lib().vd_conv3x3_wgrad_wino_phase(*args, 1, stream()), "vd_conv3x3_wgrad_wino_phase")
    with _TimedName("wino_wgrad_reduce_kernel", 0.0):
        _check(lib().vd_conv3x3_wgrad_wino_phase(*args, 2, stream()), "vd_conv3x3_wgrad_wino_phase")


# weight gradients on a side stream beside the input-gradient / GroupNorm chain of backward (engine._cwgrad): -0.9 ms per CIFAR step,
# -7 ms per CelebA step, same-box A/B; VD_WGRAD_STREAM=0 keeps everything on one stream
WGRAD_STREAM = os.environ.get("VD_WGRAD_STREAM", "1") != "0"
# slab count of the grouped 1x1 / linear weight gradients from vd_gemm_grouped_wgrad_auto_split (whole residency rounds: -0.5 ms per CIFAR
# step against the fixed 1024-workgroup rule, same-box A/B); VD_GROUPED_AUTO_SPLIT=0: that rule
GROUPED_AUTO_SPLIT = os.environ.get("VD_GROUPED_AUTO_SPLIT", "1") != "0"
# data-parallel runs (a gradient reducer listens to backward): report gradients ready after every residual / attention block (1) or at
# UNet-level boundaries only (0: fewer, larger grouped weight-gradient launches; buckets leave in bursts)
GN_FOLD = os.environ.get("VD_GN_FOLD", "1") != "0"       # A/B switch: 0 = statistics finalize and apply as two launches
GN_FOLD_MAX_CHUNKS = int(os.environ.get("VD_GN_FOLD_MAX_CHUNKS", "8"))     # most chunks per image the folded form re-reduces per workgroup
READY_PER_BLOCK = os.environ.get("VD_READY_PER_BLOCK", "0") != "0"
# a training forward outside the flat-buffer trainer returns a CHAIN of autograd nodes cut at the engine's progress points, so that the
# reference's own DDP(model) (train.py:141-148) sees gradients -- and launches its bucket all-reduces -- while backward still runs
# (models/unet.py::_SegFn); 0 = one node for the whole network (A/B switch: same kernels, same results bit for bit, no overlap)
AUTOGRAD_CHAIN = os.environ.get("VD_AUTOGRAD_CHAIN", "1") != "0"
WINO43_WGRAD = os.environ.get("VD_WINO43_WGRAD", "1") != "0"   # A/B switch: 0 keeps every weight gradient on F(2x2,3x3)
# fewest 4x4-output tiles (= K of the 36 GEMMs) it is picked for: 512 = 8x8 images at batch 128 (256 -> 256: x1.17, 768 -> 768: x1.41 over the fused
# F(2x2,3x3) kernel, same-box A/B tests/perf_wgrad43.py); the library serves nothing below 512
WINO43_WGRAD_MIN_TILES = int(os.environ.get("VD_WINO43_WGRAD_MIN_TILES", "512"))


W43_WGRAD_TAG = "[36 planes of the F(4x4,3x3) weight gradient]"


def wgrad43_supported(nimg, H, W, Cin, Cout, ldx, lddy):
    return (WINO and WINO43_WGRAD and nimg * (H // 4) * (W // 4) >= WINO43_WGRAD_MIN_TILES
            and bool(lib().vd_conv3x3_wgrad_wino43_supported(nimg, H, W, Cin, Cout, ldx, lddy)))


def conv3x3_wgrad_wino43(x, ldx, dy, lddy, nimg, H, W, Cin, Cout, dw, Cin_w, Cout_w, accumulate=False, dbias=None):
    """F(4x4,3x3) weight gradient, unfused (vd_conv3x3_wgrad_wino43): transforms -> 36 grouped GEMMs -> finish"""
    nb = lib().vd_conv3x3_wgrad_wino43_ws_bytes(nimg, H, W, Cin, Cout)
    ws = workspace(nb, x.device, "wgrad43")
    args = (ptr(x), ldx, ptr(dy), lddy, nimg, H, W, Cin, Cout, ptr(dw), ptr(dbias), Cin_w, Cout_w, int(accumulate), ws.data_ptr(),
            ws.numel() * 4)
    if PROFILE is None:
        _check(lib().vd_conv3x3_wgrad_wino43(*args, stream()), "vd_conv3x3_wgrad_wino43")
        return
    with _TimedBytes("wino43_wgrad_transform_kernel", 4.0 * nimg * H * W * (Cin + Cout) * (1 + 2.25)):
        _check(lib().vd_conv3x3_wgrad_wino43_phase(*args, 1, stream()), "vd_conv3x3_wgrad_wino43_phase")
    # the 36 planes run as ONE grouped launch of the tile engine: recorded under that instantiation's rocprof name (from vd_gemm_last_tile),
    # tagged so that the bench knows its recorded work is the direct convolution's 2*M*N*K, of which the planes execute 1/4
    with _Timed("gemm_dma_kernel<{tile}, 1, 1, true, {kt}, true> " + W43_WGRAD_TAG, 2.0 * nimg * H * W * Cout * 9 * Cin) as t:
        _check(lib().vd_conv3x3_wgrad_wino43_phase(*args, 2, stream()), "vd_conv3x3_wgrad_wino43_phase")
    # (its operands are the transformed images: 36 planes of [tiles][Cin] and [tiles][Cout], read once, + the 36 x Cout x Cin result)
    _note_bytes(PROFILE[-1][0], 4.0 * (36.0 * nimg * (H // 4) * (W // 4) * (Cin + Cout) + 36.0 * Cout * Cin))
    with _TimedName("wino43_wgrad_finish_kernel", 0.0):
        _check(lib().vd_conv3x3_wgrad_wino43_phase(*args, 4, stream()), "vd_conv3x3_wgrad_wino43_phase")


def conv3x3_wgrad(x, ldx, dy, lddy, nimg, H, W, Cin, Cout, dw, Cin_w, Cout_w, accumulate=False, dbias=None, direct=False):
    """weight (and bias) gradient of the 3x3 convolution: Winograd-domain kernels wherever the geometry is served -- F(4x4,3x3)
    unfused for the large layers, F(2x2,3x3) fused otherwise (VD_WINO=0 or direct=True: the implicit GEMM over pixels)"""
    if not direct and wgrad43_supported(nimg, H, W, Cin, Cout, ldx, lddy):
        return conv3x3_wgrad_wino43(x, ldx, dy, lddy, nimg, H, W, Cin, Cout, dw, Cin_w, Cout_w, accumulate, dbias)
    if WINO and not direct and lib().vd_conv3x3_wgrad_wino_supported(nimg, H, W, Cin, Cout, ldx, lddy):
        return conv3x3_wgrad_wino(x, ldx, dy, lddy, nimg, H, W, Cin, Cout, dw, Cin_w, Cout_w, accumulate, dbias)
    nb = lib().vd_conv3x3_wgrad_ws_bytes(nimg, H, W, Cin, Cout)
    ws = workspace(nb, x.device, "wgrad")
    args = (ptr(x), ldx, ptr(dy), lddy, nimg, H, W, Cin, Cout, ptr(dw), ptr(dbias), Cin_w, Cout_w, int(accumulate), ws.data_ptr(),
            ws.numel() * 4)
    flops = 2.0 * nimg * H * W * Cout * 9 * Cin
    if PROFILE is None:
        _check(lib().vd_conv3x3_wgrad(*args, stream()), "vd_conv3x3_wgrad")
        return
    # per-kernel timing: the MFMA kernel and the slab reduction get their own event pairs (same kernels, same order)
    with _Timed("gemm_dma_kernel<{tile}, 1, 2, true, {kt}>", flops):
        _check(lib().vd_conv3x3_wgrad_phase(*args, 1, stream()), "vd_conv3x3_wgrad_phase")
    with _Timed("reduce_slabs_oihw_kernel", 0.0):
        _check(lib().vd_conv3x3_wgrad_phase(*args, 2, stream()), "vd_conv3x3_wgrad_phase")


def im2col3x3(x, ldx, xc, nimg, H, W, Cc):
    _check(lib().vd_im2col3x3(ptr(x), ldx, ptr(xc), nimg, H, W, Cc, stream()), "vd_im2col3x3")


def tap_gather(z, ldz, bias, out, ldo, nimg, H, W, Cout):
    _check(lib().vd_tap_gather(ptr(z), ldz, ptr(bias), ptr(out), ldo, nimg, H, W, Cout, stream()), "vd_tap_gather")


def tap_spread(dy, lddy, dz, ldz, nimg, H, W, Cout):
    _check(lib().vd_tap_spread(ptr(dy), lddy, ptr(dz), ldz, nimg, H, W, Cout, stream()), "vd_tap_spread")


def thin_wgrad_finish(g, Cout_w, Cin, Cin_w, dw, accumulate=False, colsum=None, cs_stride=1, cs_off=0, dbias=None):
    _check(lib().vd_thin_wgrad_finish(ptr(g), Cout_w, Cin, Cin_w, ptr(dw), int(accumulate), ptr(colsum), cs_stride, cs_off,
                                      ptr(dbias), stream()), "vd_thin_wgrad_finish")


def pack_conv3x3(w, Cout_w, Cin_w, wf=None, Cin_p=0, wd=None, Cout_p=0):
    _check(lib().vd_pack_conv3x3(ptr(w), Cout_w, Cin_w, ptr(wf), Cin_p, ptr(wd), Cout_p, stream()), "vd_pack_conv3x3")


def pack_conv3x3_batched(table, n, total_blocks):
    """table: int64 device tensor [n][8], see vd_pack_conv3x3_batched"""
    assert table.is_cuda and table.dtype == torch.int64 and table.is_contiguous()
    _check(lib().vd_pack_conv3x3_batched(table.data_ptr(), n, total_blocks, stream()), "vd_pack_conv3x3_batched")


def gn_stats(x, ldx, nimg, HW, Cc, stats, G=32, eps=1e-6):
    nb = lib().vd_gn_ws_bytes(nimg, HW, Cc)
    ws = workspace(nb, x.device, "gn")
    _check(lib().vd_gn_stats(ptr(x), ldx, nimg, HW, Cc, G, eps, ptr(stats), ws.data_ptr(), ws.numel() * 4, stream()),
           "vd_gn_stats")


class _TimedBytes:
    """PROFILE record of an HBM-bound launch: ("hbm:<rocprof kernel name>", algorithmic bytes = operands read + written once,
    events).  ``rename`` (optional) is called after the launch and returns the name of the instantiation that ran."""

    def __init__(self, name, nbytes, rename=None):
        self.name, self.nbytes, self.rename = name, nbytes, rename

    def __enter__(self):
        if PROFILE is not None:
            self.e0, self.e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *exc):
        if PROFILE is not None and exc[0] is None:
            self.e1.record()
            PROFILE.append(("hbm:" + (self.rename() if self.rename else self.name), float(self.nbytes), self.e0, self.e1))


def _gn_bwd_name():
    k = lib().vd_gn_bwd_last_kernel()
    if k > 0:
        split, k = k // 100000000, k % 100000000         # (sibling workgroups per image slab: the SPLIT form of round 5)
        nt, k = k >= 1000000, k % 1000000
        return f"gn_bwd_fused_kernel<{k // 10000}, {k % 10000}, {'true' if nt else 'false'}, {'true' if split else 'false'}>"
    return "gn_bwd_apply_kernel (two-pass form)" if k < 0 else "gn_bwd_apply_kernel (resample)"


def gn_apply(x, ldx, stats, gamma, beta, film, act, p_drop, seed, resample, y, ldy, nimg, H, W, Cc, coef, G=32):
    hw_out = H * W // 4 if resample == RS_DOWN else (H * W * 4 if resample == RS_UP else H * W)
    with _TimedBytes("gn_apply_kernel" if gamma is not None else "gn_apply_kernel (resample)", 4.0 * nimg * Cc * (H * W + hw_out)):
        _gn_apply(x, ldx, stats, gamma, beta, film, act, p_drop, seed, resample, y, ldy, nimg, H, W, Cc, coef, G)


def gn_apply_from_partials(x, ldx, parts, gamma, beta, film, act, p_drop, seed, resample, y, ldy, nimg, H, W, Cc, coef, G=32, eps=1e-6):
    """statistics from the producers' partial sums + apply in one launch (vd_gn_apply_from_partials); parts: [(part, C, chunks)] x 1 or 2"""
    (p1, c1, k1), (p2, c2, k2) = parts[0], (parts[1] if len(parts) > 1 else (None, 0, 0))
    hw_out = H * W // 4 if resample == RS_DOWN else (H * W * 4 if resample == RS_UP else H * W)
    with _TimedBytes("gn_apply_kernel", 4.0 * nimg * Cc * (H * W + hw_out)):
        _check(lib().vd_gn_apply_from_partials(ptr(x), ldx, ptr(p1), c1, k1, ptr(p2), c2, k2, ptr(gamma), ptr(beta), ptr(film), int(act),
                                               float(p_drop), int(seed), resample, ptr(y), ldy, nimg, H, W, Cc, G, eps, ptr(coef), stream()),
               "vd_gn_apply_from_partials")


def _gn_apply(x, ldx, stats, gamma, beta, film, act, p_drop, seed, resample, y, ldy, nimg, H, W, Cc, coef, G=32):
    _check(lib().vd_gn_apply(ptr(x), ldx, ptr(stats), ptr(gamma), ptr(beta), ptr(film), int(act), float(p_drop), int(seed),
                             resample, ptr(y), ldy, nimg, H, W, Cc, G, ptr(coef), stream()), "vd_gn_apply")


def gn_apply_bwd(dy, lddy, x, ldx, coef, gamma, beta, film, act, p_drop, seed, resample, add, ldadd, dx, lddx,
                 accumulate_dx, dfilm, dgamma, dbeta, accumulate_params, nimg, H, W, Cc, G=32, pgb_keep=None):
    """pgb_keep ([nimg][2][Cc] floats): leave the per-image dgamma / dbeta terms there for one gn_param_sums_batched launch over
    all norms of the backward pass instead of summing them here (dgamma / dbeta are then not touched)"""
    nb = lib().vd_gn_ws_bytes(nimg, H * W, Cc)
    ws = workspace(nb, dy.device, "gn")
    hw_dy = H * W // 4 if resample == RS_DOWN else (H * W * 4 if resample == RS_UP else H * W)
    nbytes = 4.0 * nimg * Cc * (hw_dy + H * W * (1 + (gamma is not None) + (add is not None) + bool(accumulate_dx)))
    with _TimedBytes("gn_apply_bwd", nbytes, rename=_gn_bwd_name):
        if pgb_keep is not None:
            _check(lib().vd_gn_apply_bwd_keep(ptr(dy), lddy, ptr(x), ldx, ptr(coef), ptr(gamma), ptr(beta), ptr(film), int(act),
                                              float(p_drop), int(seed), resample, ptr(add), ldadd, ptr(dx), lddx, int(accumulate_dx),
                                              ptr(dfilm), ptr(pgb_keep), nimg, H, W, Cc, G, ws.data_ptr(), ws.numel() * 4, stream()),
                   "vd_gn_apply_bwd_keep")
        else:
            _gn_apply_bwd_call(dy, lddy, x, ldx, coef, gamma, beta, film, act, p_drop, seed, resample, add, ldadd, dx, lddx,
                               accumulate_dx, dfilm, dgamma, dbeta, accumulate_params, nimg, H, W, Cc, G, ws)


def gn_param_sums_batched(table, n, total_blocks):
    assert table.is_cuda and table.dtype == torch.int64 and table.is_contiguous()
    _check(lib().vd_gn_param_sums_batched(table.data_ptr(), n, total_blocks, stream()), "vd_gn_param_sums_batched")


def _gn_apply_bwd_call(dy, lddy, x, ldx, coef, gamma, beta, film, act, p_drop, seed, resample, add, ldadd, dx, lddx,
                       accumulate_dx, dfilm, dgamma, dbeta, accumulate_params, nimg, H, W, Cc, G, ws):
    _check(lib().vd_gn_apply_bwd(ptr(dy), lddy, ptr(x), ldx, ptr(coef), ptr(gamma), ptr(beta), ptr(film), int(act),
                                 float(p_drop), int(seed), resample, ptr(add), ldadd, ptr(dx), lddx, int(accumulate_dx),
                                 ptr(dfilm), ptr(dgamma), ptr(dbeta), int(accumulate_params), nimg, H, W, Cc, G,
                                 ws.data_ptr(), ws.numel() * 4, stream()), "vd_gn_apply_bwd")


def colsum(x, ldx, M, N, out, accumulate=False):
    nb = lib().vd_colsum_ws_bytes(M, N)
    ws = workspace(nb, x.device, "colsum")
    _check(lib().vd_colsum(ptr(x), ldx, M, N, ptr(out), int(accumulate), ws.data_ptr(), ws.numel() * 4, stream()), "vd_colsum")


def axpby(x, ldx, alpha, y, ldy, beta, rows, Cc):
    _check(lib().vd_axpby(ptr(x), ldx, alpha, ptr(y), ldy, beta, rows, Cc, stream()), "vd_axpby")


def silu(x, y):
    _check(lib().vd_silu(ptr(x), ptr(y), x.numel(), stream()), "vd_silu")


def silu_bwd(x, dy, dx, accumulate=False):
    _check(lib().vd_silu_bwd(ptr(x), ptr(dy), ptr(dx), x.numel(), int(accumulate), stream()), "vd_silu_bwd")


def softmax_rows(s, rows, L):
    _check(lib().vd_softmax_rows(ptr(s), rows, L, stream()), "vd_softmax_rows")


def softmax_rows_bwd(p, dp, rows, L, alpha):
    _check(lib().vd_softmax_rows_bwd(ptr(p), ptr(dp), rows, L, alpha, stream()), "vd_softmax_rows_bwd")


FUSED_ATTN = os.environ.get("VD_FUSED_ATTN", "1")            # A/B switch: 0 = three-launch attention everywhere, 2 = fused wherever served


def attn_supported(L, hd, backward):
    """is the geometry served by the fused kernels (csrc/attn.hip: L % 64 == 0, head dim 64 / 128 / 256, forward and backward)?"""
    return FUSED_ATTN != "0" and bool(lib().vd_attn_supported(L, hd, int(backward)))


def attn_use_fused(L, hd, rows, training):
    """Which attention path a block takes.  Inference: the fused forward wherever it is served (faster at every measured shape).
    Training (forward + backward with recomputation: 9 products where the three-launch path runs 6) by same-box measurement at
    batch 128 (tests/perf_attn.py, MI355X): head dim 64 fused except at L = 256 (0.144 vs 0.128 ms); head dim 128 / 256 unfused (the
    backward kernels hold Q/dO or K/V fragments of the whole head dim in registers: one wave per SIMD at head dim 256 -- L = 1024:
    5.3 vs 3.6 ms) -- unless the materialised [rows, L, L] maps would not be reasonable to keep (>= 4 GiB each), where the fused
    path is the only sensible one.  VD_FUSED_ATTN=2 forces the fused kernels wherever served (parity tests of the whole step)."""
    if not attn_supported(L, hd, training):
        return False
    if not training or FUSED_ATTN == "2":
        return True
    if 4.0 * rows * L * L >= float(1 << 32):
        return True
    return hd == 64 and L != 256


def attn_fwd(q, k, v, ld, o, ldo, lse, B, nh, L, hd, scale):
    """fused softmax(scale q k^T) v; FLOPs recorded = the two products (4 L^2 hd per image and head)"""
    with _TimedName(f"attn_fwd_kernel<{hd}>", 4.0 * B * nh * L * L * hd):
        _check(lib().vd_attn_fwd(ptr(q), ptr(k), ptr(v), ld, ptr(o), ldo, ptr(lse), B, nh, L, hd, scale, stream()), "vd_attn_fwd")


def attn_bwd(q, k, v, ld, o, ldo, dout, lddo, lse, delta, dq, dk, dv, ldd, B, nh, L, hd, scale):
    """backward of attn_fwd (recomputes the probabilities).  FLOPs recorded are ALGORITHMIC: the four products of the backward
    (dP, dQ to the first kernel, dV, dK to the second; 2 L^2 hd each per image and head) -- the kernels execute 3 + 4 = 7 of
    them (the logits in both, dP again in the second), so their executed MFMA rate is 7/4 of the recorded one"""
    args = (ptr(q), ptr(k), ptr(v), ld, ptr(o), ldo, ptr(dout), lddo, ptr(lse), ptr(delta), ptr(dq), ptr(dk), ptr(dv), ldd, B, nh, L, hd,
            scale)
    if PROFILE is None:
        _check(lib().vd_attn_bwd(*args, stream()), "vd_attn_bwd")
        return
    unit = 2.0 * B * nh * L * L * hd
    with _TimedName(f"attn_bwd_dq_kernel<{hd}>", 2 * unit):
        _check(lib().vd_attn_bwd_phase(*args, 1, stream()), "vd_attn_bwd_phase")
    with _TimedName(f"attn_bwd_dkv_kernel<{hd}>", 2 * unit):
        _check(lib().vd_attn_bwd_phase(*args, 2, stream()), "vd_attn_bwd_phase")


def nchw_to_nhwc(x, y, nimg, Cc, H, W, ldy):
    _check(lib().vd_nchw_to_nhwc(ptr(x), ptr(y), nimg, Cc, H, W, ldy, stream()), "vd_nchw_to_nhwc")


def nhwc_to_nchw(x, ldx, y, nimg, Cc, H, W):
    _check(lib().vd_nhwc_to_nchw(ptr(x), ldx, ptr(y), nimg, Cc, H, W, stream()), "vd_nhwc_to_nchw")


def images_to_uint8_hwc(x, out, n, Cc, HW):
    _check(lib().vd_images_to_uint8_hwc(ptr(x), ptr(out), n, Cc, HW, stream()), "vd_images_to_uint8_hwc")


def images_from_uint8_hwc(u8, flip, out, n, Cc, H, W):
    _check(lib().vd_images_from_uint8_hwc(ptr(u8), ptr(flip), ptr(out), n, Cc, H, W, stream()), "vd_images_from_uint8_hwc")


def timestep_embedding(t, out, n, dim, scale=1000.0):
    if t.dtype != torch.float64:
        raise HipError("timestep_embedding expects fp64 timesteps")
    _check(lib().vd_timestep_embedding(ptr(t), ptr(out), n, dim, scale, stream()), "vd_timestep_embedding")


def class_embed(y, w, bias, temb, n, emb, ncls):
    _check(lib().vd_class_embed(ptr(y), ptr(w), ptr(bias), ptr(temb), n, emb, ncls, stream()), "vd_class_embed")


def class_embed_bwd(y, dtemb, dw, dbias, n, emb, ncls, accumulate=False):
    _check(lib().vd_class_embed_bwd(ptr(y), ptr(dtemb), ptr(dw), ptr(dbias), n, emb, ncls, int(accumulate), stream()),
           "vd_class_embed_bwd")


def multitag_norm(y, out, n, ncls):
    _check(lib().vd_multitag_norm(ptr(y), ptr(out), n, ncls, stream()), "vd_multitag_norm")


def q_sample(x0, eps, logsnr, xt, n, Cc, HW):
    _check(lib().vd_q_sample(ptr(x0), ptr(eps), ptr(logsnr), ptr(xt), n, Cc, HW, stream()), "vd_q_sample")


def loss_fwd(x0, eps, xt, out, logsnr, mot, rw, loss, aux, n, Cc, HW):
    _check(lib().vd_loss_fwd(ptr(x0), ptr(eps), ptr(xt), ptr(out), ptr(logsnr), mot, rw, ptr(loss), ptr(aux), n, Cc, HW,
                             stream()), "vd_loss_fwd")


def loss_bwd(x0, eps, xt, out, logsnr, aux, gloss, mot, rw, dout, n, Cc, HW):
    _check(lib().vd_loss_bwd(ptr(x0), ptr(eps), ptr(xt), ptr(out), ptr(logsnr), ptr(aux), ptr(gloss), mot, rw, ptr(dout),
                             n, Cc, HW, stream()), "vd_loss_bwd")


def bpd_terms(x0, xt, out, coef, mot, clip, kl, nll, pred, mse, n, Cc, HW):
    _check(lib().vd_bpd_terms(ptr(x0), ptr(xt), ptr(out), ptr(coef), mot, int(clip), ptr(kl), ptr(nll), ptr(pred), ptr(mse),
                              n, Cc, HW, stream()), "vd_bpd_terms")


def bpd_bwd(x0, xt, out, coef, use_kl, gloss, mot, clip, dout, n, Cc, HW):
    _check(lib().vd_bpd_bwd(ptr(x0), ptr(xt), ptr(out), ptr(coef), ptr(use_kl), ptr(gloss), mot, int(clip), ptr(dout),
                            n, Cc, HW, stream()), "vd_bpd_bwd")


def sample_step(xt, out, noise, k8, mot, cfg, last, clip, xn, xdup, n, Cc, HW, k_dev=None):
    """k8: 8 host floats, or None with k_dev = device tensor of 8 floats (graph-replayable form)"""
    arr = None if k8 is None else (_f32 * 8)(*[float(v) for v in k8])
    _check(lib().vd_sample_step(ptr(xt), ptr(out), ptr(noise), arr, ptr(k_dev), mot, int(cfg), int(last), int(clip), ptr(xn),
                                ptr(xdup), n, Cc, HW, stream()), "vd_sample_step")


def sumsq(g, out1):
    ws = workspace(lib().vd_sumsq_ws_bytes(g.numel()), g.device, "sumsq")
    _check(lib().vd_sumsq(ptr(g), g.numel(), ptr(out1), ws.data_ptr(), ws.numel() * 4, stream()), "vd_sumsq")


def adamw_ema(p, g, m, v, ema, gnorm_sq, max_norm, lr, b1, b2, eps, wd, bc1, bc2, ema_decay, r_lo=0, r_hi=0, r_mode=0, r_bc1=1.0,
              r_bc2=1.0):
    """r_*: an index range treated apart this step, see vd_adamw_ema (1 = no gradient this step, 2 = own bias corrections)"""
    _check(lib().vd_adamw_ema(ptr(p), ptr(g), ptr(m), ptr(v), ptr(ema), p.numel(), ptr(gnorm_sq), max_norm, lr, b1, b2, eps,
                              wd, bc1, bc2, ema_decay, int(r_lo), int(r_hi), int(r_mode), r_bc1, r_bc2, stream()), "vd_adamw_ema")


def adamw_ema_flagged(p, g, m, v, ema, gnorm_sq, max_norm, lr, b1, b2, eps, wd, bc1, bc2, ema_decay, r_lo, r_hi, r_flag, r_steps):
    """the update with the [r_lo, r_hi) decision made on the device (vd_adamw_ema_flagged): r_flag = device float[1], r_steps = device
    int32[1] (advanced by the call when the flag is set)"""
    assert r_flag.dtype == torch.float32 and r_steps.dtype == torch.int32
    _check(lib().vd_adamw_ema_flagged(ptr(p), ptr(g), ptr(m), ptr(v), ptr(ema), p.numel(), ptr(gnorm_sq), max_norm, lr, b1, b2, eps,
                                      wd, bc1, bc2, ema_decay, int(r_lo), int(r_hi), ptr(r_flag), ptr(r_steps), float(b1), float(b2),
                                      stream()), "vd_adamw_ema_flagged")
