/* vdiff_hip.h -- C ABI of libvdiff_hip.so: the MI355X (gfx950) kernels behind the
 * v-diffusion hot path (UNet forward/backward inside the diffusion train step + the DDIM/CFG
 * sampling loop of tqch/v-diffusion-torch).
 *
 * The reference has no FFI layer: its hot path bottoms out in ATen calls.  Each entry point
 * below replaces one of those call sites (cited as reference file:line) and is what a binding
 * of this path has to import -- see INTEGRATION.md for the ctypes stub.
 *
 * Conventions
 *  - plain pointers and sizes only; no torch types.  All tensors are fp32 unless stated.
 *  - activations are NHWC: [image][y][x][channel] with an explicit per-pixel stride `ld`
 *    (in floats, multiple of 4, base 16-byte aligned) so channel slices of a wider buffer
 *    (the "virtual concat" of reference unet.py:315) can be read/written in place.
 *  - the library never allocates, frees or retains device memory; scratch comes in through
 *    `ws` arguments (size from the matching *_ws_bytes query).  Every launch goes to the
 *    `stream` argument (a hipStream_t passed as void*).  No host synchronisation anywhere,
 *    so every call is HIP-graph capturable.
 *  - return value: 0 on success, non-zero on error; vd_last_error() gives the message of
 *    the calling thread's last failure.
 */
#ifndef VDIFF_HIP_H
#define VDIFF_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VD_VERSION 100

int         vd_version(void);
const char* vd_last_error(void);

/* Compute units the PERSISTENT convolution launches (vd_conv3x3_wino, vd_conv3x3_dgrad_wino43: one workgroup per CU holding all of its
 * LDS and registers for the whole launch, 0.1-1.5 ms) leave free: their grids are sized CUs - n.  A data-parallel rank sets it so that
 * the RCCL kernels of a gradient bucket launched from inside backward (reference DDP overlap, train.py:141-148) find a free CU at once
 * instead of waiting for a persistent launch to tail off.  Process-wide, default 0 or the VD_RESERVE_CUS environment variable;
 * 0 <= n <= CUs - 8; multiples of 8 keep the XCD-aware item order of those kernels.  Returns the previous value, or -1 if refused. */
int         vd_set_reserved_cus(int32_t n);
int         vd_reserved_cus(void);

/* Calibration loop for bench.py (no reference counterpart: a measurement aid): `blocks` workgroups of 8 waves run `iters` x 16
 * register-only v_mfma_f32_16x16x4_f32 per wave (2048 FLOP each per wave) on operands hashed from `seed`; lane 0 of workgroup b writes
 * stamps[2b] = shader cycles (s_memtime) and stamps[2b + 1] = 100 MHz ticks (s_memrealtime) the loop took: the in-kernel clock is their
 * quotient x 100 MHz.  sink: one float, never written in practice. */
int         vd_mfma_calibrate(float* sink, unsigned long long* stamps, int32_t blocks, int32_t iters, uint32_t seed, void* stream);

/* ------------------------------------------------------------------ dense contractions (MFMA fp32)
 * One tile engine (v_mfma_f32_32x32x2_f32, LDS-staged, double buffered) behind every
 * matmul-shaped op of the path.  C[z][m][n] = alpha * sum_k A(z,m,k) * B(z,n,k) (+bias[n]) (+R[z][m][n]) (+C) */
enum { VD_ROW = 0,      /* operand stored [rows][k], k contiguous                                  */
       VD_COL = 1,      /* operand stored [k][rows], rows contiguous                               */
       VD_IM2COL = 2 }; /* A: 3x3 patches of an NHWC image, k = (tap, channel)  (conv forward/dgrad)
                           B: 3x3 patches, k = pixel, n = (tap, channel)         (conv wgrad)       */

typedef struct vd_gemm_desc {
    const float* A; const float* B; float* C;
    const float* bias;          /* [N] or NULL                                                      */
    const float* R;             /* residual, same indexing as C with ldr / sR*, or NULL             */
    int32_t M, N, K;
    int32_t a_kind, b_kind;
    int64_t lda, ldb, ldc, ldr;
    int32_t batch, nh;          /* z in [0,batch): zb = z / nh, zh = z % nh (nh >= 1)               */
    int64_t sAb, sAh, sBb, sBh, sCb, sCh, sRb, sRh;
    float   alpha;
    int32_t accumulate;         /* C += result                                                      */
    int32_t H, W, Cin;          /* image geometry for VD_IM2COL                                     */
    int32_t splitk;             /* >1: K is split over `splitk` slabs in ws, then reduced into C    */
    float*  ws; int64_t ws_bytes;
    int32_t tile;               /* 0 = auto; 128, 64, 12864 (128x64), 64128 (64x128) force the block tile */
    float*  colsum;             /* optional, VD_COL A only: colsum[m] (+)= sum_k A[k][m] (bias gradient of a */
    int32_t colsum_accumulate;  /*   weight-gradient GEMM, computed from the tiles already staged); batch = 1  */
    float*  stats;              /* optional, forward launches: GroupNorm partials of the OUTPUT rows, laid out  */
    int32_t stats_hw;           /*   [image][chunk][2][N] with chunk = BM/2 rows (BM from vd_gemm_last_tile);    */
                                /*   stats_hw = rows (pixels) per image; finalize: vd_gn_stats_from_partials     */
    int64_t sBias;              /* batched launches: bias of entry z is bias + (z / nh) * sBias (0 = one shared bias) */
} vd_gemm_desc;

/* replaces F.linear (modules.py:79-80), 1x1 F.conv2d (modules.py:141-144 <- unet.py:70,71,134), the two
 * einsum contractions of attention (unet.py:57,61-63) and all of their autograd backward GEMMs */
int vd_gemm(const vd_gemm_desc* d, void* stream);
/* ((F*100 + KT)*1000 + BM)*1000 + BN of the calling thread's last vd_gemm / vd_conv3x3* launch (profiling aid: names the
 * kernel instantiation gemm_dma_kernel<BM,BN,a_kind,b_kind,splitk,KT,TR> the launch went to; F bit 0 = TR, transposed-accumulator
 * epilogue (launches without output statistics); F >= 2: split-operand form, below; KT = 0 means the register-staged fallback
 * gemm_kernel<BM,BN,a_kind,b_kind,splitk>).
 * Environment, read once per process: VD_GEMM_SPLIT (default 1 since round 5; 0 = fp32 MFMA everywhere, the A/B form) sends the 128-row tiles of vd_gemm and
 * every vd_gemm_grouped_wgrad launch to the split-operand forms (gemm_split_kernel<...>): each fp32 operand value is split exactly into three bf16 pieces in registers and the six piece
 * products that reach 2^-24 of a.b run on the 16-bit matrix cores with fp32 accumulation.  Same results to fp32 rounding (error against
 * fp64 0.6-0.9 of the fp32 MFMA chain's: tests/test_kernels_gpu.py::test_split_operand_gemm_forms_in_subprocess), 15-26 % shorter
 * launches (in-kernel clock 1.88 GHz instead of 2.28: the 16-bit pipes ARE power-limited on random data), -2.3 % on a train step of either
 * workload (profiles/r05_split_step_ab.txt); vd_gemm_split_forms() tells which form is in effect.  DOMAIN of the split forms: finite operands of
 * magnitude <= the largest bf16 (3.39e38); a larger or infinite operand yields NaN where the fp32 MFMA yields a finite value or Inf
 * (bf16(x) rounds to Inf, the residual x - Inf poisons the products); pieces below the smallest normal bf16 are flushed. */
int vd_gemm_last_tile(void);
int vd_gemm_split_forms(void);   /* 1: the split-operand forms are in effect for this process (VD_GEMM_SPLIT, default 1), 0: fp32 MFMA everywhere */
/* `count` (<= 36) same-shape weight-gradient GEMMs in ONE launch -- the 1x1-convolution / linear weight gradients of the blocks of one
 * UNet level (autograd of modules.py:79-80,141-144 w.r.t. the weight), whose operands live in unrelated buffers:
 *   C[e][M][N] (pitch ldc) = A[e]^T B[e],  A[e] = dY [K][M] (pitch lda), B[e] = X [K][N] (pitch ldb)   (= vd_gemm with COL / COL kinds)
 *   colsum[e][m] = sum_k A[e][k][m]   (the bias gradient; colsum may be NULL)
 * split-K over `splitk` slabs per entry through ws (vd_gemm_grouped_wgrad_ws_bytes), reduced in a fixed order: bitwise reproducible.
 * A, B, C, colsum are HOST arrays of device pointers (they travel in the kernel arguments). */
size_t vd_gemm_grouped_wgrad_ws_bytes(int32_t count, int32_t M, int32_t N, int32_t splitk);
/* slab count in [min_slabs, max_slabs] that fills whole residency rounds of the device best (>= 8 K tiles per slab) */
int vd_gemm_grouped_wgrad_auto_split(int32_t count, int32_t M, int32_t N, int32_t K, int32_t min_slabs, int32_t max_slabs);
int vd_gemm_grouped_wgrad(const float* const* A, const float* const* B, float* const* C, float* const* colsum, int32_t count,
                          int32_t M, int32_t N, int32_t K, int64_t lda, int64_t ldb, int64_t ldc, int32_t splitk, float* ws, size_t ws_bytes,
                          void* stream);

/* 3x3 / stride 1 / pad 1 convolution, NHWC (replaces F.conv2d at modules.py:141-144 <- unet.py:121,125,217,232).
 *   y[b,y,x,co] = bias[co] + res[b,y,x,co] + sum_{tap,ci} xin[b,y+dy,x+dx,ci] * wpack[co][tap][ci]
 * wpack comes from vd_pack_conv3x3 ("forward" pack for the forward pass, "dgrad" pack for the input gradient).
 * Cin must be a multiple of 4 (pad 3 -> 4); only co < Cout is written. */
int vd_conv3x3(const float* xin, int64_t ldx, const float* wpack, const float* bias,
               const float* res, int64_t ldres, float* y, int64_t ldy,
               int32_t nimg, int32_t H, int32_t W, int32_t Cin, int32_t Cout, int32_t accumulate,
               float* stats_part /* optional: GroupNorm partials of y, see vd_gemm_desc.stats */, void* stream);

/* The same convolution as Winograd F(2x2,3x3) on the fp32 matrix cores (csrc/wino.hip): 2.25x fewer MFMA cycles, exact fp32
 * arithmetic, input/output transforms fused into the tile loads / the epilogue (nothing but x, U, y touches HBM).
 *   U = vd_wino_pack(w): uf[16][Cout][Cin] for the forward pass, ud[16][Cin][Cout] (rotated kernel) for the input gradient
 *   (the input gradient of modules.py:141-144 is the same call with xin = dy, U = ud, Cin <-> Cout).
 * vd_conv3x3_wino_supported: 1 when the geometry is served (H, W even powers of two with 4 <= W/2 <= 64, H*W/4 >= 16 tiles per
 * image, Cin % 16 == 0, Cout % 4 == 0, tensors < 2 GiB); otherwise call vd_conv3x3.  stats_part as in vd_conv3x3 with 64-pixel
 * chunks ([img][H*W/64][2][Cout]); vd_gemm_last_tile() reports BM = 128 for it so chunk = BM/2 holds for both paths. */
int vd_conv3x3_wino_supported(int32_t nimg, int32_t H, int32_t W, int32_t Cin, int32_t Cout, int64_t ldx, int64_t ldy, int64_t ldres);
int vd_conv3x3_wino(const float* xin, int64_t ldx, const float* U, const float* bias, const float* res, int64_t ldres,
                    float* y, int64_t ldy, int32_t nimg, int32_t H, int32_t W, int32_t Cin, int32_t Cout,
                    float* stats_part, void* stream);
/* weight gradient of the same convolution in the Winograd domain (dU = sum over tiles of (A dY A^T) (.) (B^T d B), dw = G^T dU G):
 * same arguments and result as vd_conv3x3_wgrad (dw in OIHW, optional dbias), 2.25x fewer MFMA cycles; deterministic (split-K
 * slab planes summed in a fixed order).  Geometry as vd_conv3x3_wino_supported but Cin, Cout only need to be multiples of 4. */
int vd_conv3x3_wgrad_wino_supported(int32_t nimg, int32_t H, int32_t W, int32_t Cin, int32_t Cout, int64_t ldx, int64_t lddy);
size_t vd_conv3x3_wgrad_wino_ws_bytes(int32_t nimg, int32_t H, int32_t W, int32_t Cin, int32_t Cout);
int vd_conv3x3_wgrad_wino(const float* xin, int64_t ldx, const float* dy, int64_t lddy, int32_t nimg, int32_t H, int32_t W,
                          int32_t Cin, int32_t Cout, float* dw_oihw, float* dbias, int32_t Cin_w, int32_t Cout_w,
                          int32_t accumulate, float* ws, size_t ws_bytes, void* stream);
/* the two kernels of vd_conv3x3_wgrad_wino as separate calls (per-kernel timing, bench.py): phase 1 = slab planes (wino_wgrad_kernel),
 * phase 2 = their fixed-order reduction into OIHW (wino_wgrad_reduce_kernel) */
int vd_conv3x3_wgrad_wino_phase(const float* xin, int64_t ldx, const float* dy, int64_t lddy, int32_t nimg, int32_t H, int32_t W,
                                int32_t Cin, int32_t Cout, float* dw_oihw, float* dbias, int32_t Cin_w, int32_t Cout_w,
                                int32_t accumulate, float* ws, size_t ws_bytes, int32_t phase, void* stream);
/* (TW * 1000 + NS) * 2 + stats of the calling thread's last vd_conv3x3_wino launch: names the instantiation
 * wino_conv_kernel<TW, NS, stats, false, 0>; negated when the launch took the 128-tile form wino_conv_wide_kernel<TW, NS, stats>
 * (profiling aid, like vd_gemm_last_tile) */
int vd_wino_last_kernel(void);
/* (TWS * 1000 + split-K slabs) * 2 + dbias of the calling thread's last vd_conv3x3_wgrad_wino launch: names the instantiation
 * wino_wgrad_kernel<TWS, dbias> (TWS = min(W/2, 16)) and its slab count (profiling / test aid) */
int vd_wino_wgrad_last_kernel(void);
#ifdef VD_PROBES
/* libvdiff_hip_probe.so only (built with -DVD_PROBES for tests/probe/; the product library has no timing-probe or timing-experiment
 * code): per-wave phase timestamps of the next vd_conv3x3_wino / vd_conv3x3_wgrad_wino launches into buf (64 x uint64 per
 * workgroup), NULL = off */
int vd_wino_set_probe(unsigned long long* buf);
/* the same for the F(4x4,3x3) forward and input-gradient launches -- tests/probe/w43_phases.py: 16 x uint64 per (workgroup, wave, item round < 4) */
int vd_wino43_set_probe(unsigned long long* buf);
/* the longest sibling wait (spin iterations, atomicMax) of the SPLIT GroupNorm-backward launches that follow -- tests/test_multigpu_path_gpu.py soak */
int vd_gn_set_spin_probe(unsigned* buf);
#endif
int vd_wino_pack(const float* w_oihw, int32_t Cout, int32_t Cin, float* uf /* or NULL */, float* ud /* or NULL */, void* stream);
/* ---- input gradient of the same convolution as Winograd F(4x4, 3x3) (csrc/wino43.hip): 36 multiplies per 4x4 output tile where
 * F(2x2,3x3) needs 64 -- 1.78x fewer matrix-core cycles; with the classic interpolation points {0, +-1, +-2, inf} used here ~7x the
 * rounding error of a direct fp32 sum, an order of magnitude inside the stated bound on gradients (per-tensor relative L2 <= 1e-4;
 * autograd of modules.py:141-144).
 *   U43 = vd_wino43_pack(w): (G rot180(w[co][ci]) G^T)[36] in the order the kernel's lanes read it, vd_wino43_u_floats(Cout, Cin) floats;
 *   dx[nimg][H][W][:Cin] = vd_conv3x3_dgrad_wino43(dy[nimg][H][W][:Cout], U43)   every element written, no accumulation.
 * _supported: 1 for 32x32 images, 64-wide images with H % 16 == 0 and 16x16 images in multiples of four (four per work item),
 * Cout % 8 == 0, Cin % 32 == 0, 16-byte rows, tensors < 2 GiB;
 * otherwise call vd_conv3x3_wino with the rotated F(2x2,3x3) image.  vd_wino43_last_kernel: tiles per row (4 / 8 / 16) of the calling
 * thread's last launch, negative for vd_conv3x3_wino43_fwd = the instantiation wino43_conv_kernel<TWT, FWD> (profiling / test aid). */
int vd_conv3x3_dgrad_wino43_supported(int32_t nimg, int32_t H, int32_t W, int32_t Cin, int32_t Cout, int64_t lddy, int64_t lddx);
size_t vd_wino43_u_floats(int32_t Cout, int32_t Cin);
int vd_conv3x3_dgrad_wino43(const float* dy, int64_t lddy, const float* U43, float* dx, int64_t lddx, int32_t nimg, int32_t H, int32_t W,
                            int32_t Cin, int32_t Cout, void* stream);
int vd_wino43_last_kernel(void);
int vd_wino43_pack(const float* w_oihw, int32_t Cout, int32_t Cin, float* U43, void* stream);
/* all tensors in one launch: items_dev = [n][8] int64 {w, U43, fwd, Cout, Cin, 0, 0, first block}; fwd = 0: the input-gradient image,
 * (Cin/32)*(Cout/8) blocks; fwd = 1: the forward image of vd_wino43_pack_fwd, (Cout/32)*(Cin/8) blocks */
int vd_wino43_pack_batched(const int64_t* items_dev, int32_t n, int64_t total_blocks, void* stream);
/* occupancy rule between the two Winograd orders: 1 when the F(4x4,3x3) convolution kernel (forward: N = Cout; input gradient: N = Cin) is
 * expected to beat the F(2x2,3x3) one on this batch -- both run persistent work items in rounds over the CUs, F(2x2,3x3) has four times as
 * many at 3/8 of the time each (measured at B = 128), so with fewer items than CUs (small-batch sampling) the finer items win.  The
 * engine asks this next to the *_supported queries; the kernels themselves serve every supported geometry regardless. */
int vd_conv3x3_wino43_preferred(int32_t nimg, int32_t H, int32_t W, int32_t N);
/* ---- FORWARD pass of the same convolution through F(4x4,3x3) (replaces F.conv2d of modules.py:141-144 for the residual-block
 * convolutions unet.py:121,125 on the 16x16 / 32x32 / 64-wide layers), with the interpolation points {0, +-3/4, +-3/2, inf}: every
 * coefficient of the data and output transforms is a dyadic rational, exact in fp32, and the whole-network output error is ~2x that of
 * the F(2x2,3x3) forward, inside the stated 2e-5 bound (tests/probe/wino_err_sim.py; measured: tests/test_unet_gpu.py).
 *   U43f = vd_wino43_pack_fwd(w): (G w[co][ci] G^T)[36] in the kernel's lane order, vd_wino43_u_floats(Cout, Cin) floats;
 *   y[nimg][H][W][:Cout] = conv3x3(x[..][:Cin], w) + bias (+ res)     bias / res may be NULL; every element written
 *   stats_part (or NULL): GroupNorm partial sums of y as [nimg][chunks][2][Cout] with ONE CHUNK PER (image, work item):
 *   vd_conv3x3_wino43_fwd_chunk_rows(H, W) pixels per chunk (H*W for 16x16 and 32x32 images, 16*W for 64-wide ones).
 * _supported: the geometries of the input gradient (above) with Cin % 8 == 0, Cout % 32 == 0; otherwise call vd_conv3x3_wino. */
int vd_conv3x3_wino43_fwd_supported(int32_t nimg, int32_t H, int32_t W, int32_t Cin, int32_t Cout, int64_t ldx, int64_t ldy, int64_t ldres);
int vd_wino43_pack_fwd(const float* w_oihw, int32_t Cout, int32_t Cin, float* U43f, void* stream);
int vd_conv3x3_wino43_fwd_chunk_rows(int32_t H, int32_t W);
int vd_conv3x3_wino43_fwd(const float* xin, int64_t ldx, const float* U43f, const float* bias, const float* res, int64_t ldres, float* y,
                          int64_t ldy, int32_t nimg, int32_t H, int32_t W, int32_t Cin, int32_t Cout, float* stats_part, void* stream);
/* weight (and bias) gradient of the same convolution through F(4x4,3x3), UNFUSED: one HBM-bound pass writes dM = A dY A^T and V = B^T d B
 * ([36][tiles][channels], 2.25x the size of dY / x each), the 36 GEMMs over the tile index run as one vd_gemm_grouped_wgrad launch (1.78x fewer
 * MFMA cycles than vd_conv3x3_wgrad_wino), a small kernel folds G^T . G into OIHW; the bias gradient is the column sum of plane (1,1).
 * Same arguments and result as vd_conv3x3_wgrad_wino (accumulate, padded dims, deterministic).  _supported: H, W, Cin, Cout multiples
 * of 4, at least 512 tiles of 4x4 outputs (8x8 images at batch 128).  _phase: 1 = transforms, 2 = GEMMs (+ slab reduction of layers below 147 456 = 384 x 384 weights per plane), 4 = finish (+ slab reduction of wider layers). */
int vd_conv3x3_wgrad_wino43_supported(int32_t nimg, int32_t H, int32_t W, int32_t Cin, int32_t Cout, int64_t ldx, int64_t lddy);
size_t vd_conv3x3_wgrad_wino43_ws_bytes(int32_t nimg, int32_t H, int32_t W, int32_t Cin, int32_t Cout);
int vd_conv3x3_wgrad_wino43(const float* xin, int64_t ldx, const float* dy, int64_t lddy, int32_t nimg, int32_t H, int32_t W,
                            int32_t Cin, int32_t Cout, float* dw_oihw, float* dbias, int32_t Cin_w, int32_t Cout_w,
                            int32_t accumulate, float* ws, size_t ws_bytes, void* stream);
int vd_conv3x3_wgrad_wino43_phase(const float* xin, int64_t ldx, const float* dy, int64_t lddy, int32_t nimg, int32_t H, int32_t W,
                                  int32_t Cin, int32_t Cout, float* dw_oihw, float* dbias, int32_t Cin_w, int32_t Cout_w,
                                  int32_t accumulate, float* ws, size_t ws_bytes, int32_t phase, void* stream);
/* split-K slabs per plane of the calling thread's last vd_conv3x3_wgrad_wino43 launch (test / profiling aid) */
int vd_wino43_wgrad_last_kernel(void);

/* all 3x3 kernels of a network in one launch: items_dev = [n][8] int64 {w, uf, ud, Cout, Cin, tiled, 0, first 256-thread block};
 * tiled = 1 (Cout, Cin multiples of 16): the tensor takes (Cout/16)*(Cin/16) blocks of one 16x16 tile, else ceil(Cout*Cin/256) */
int vd_wino_pack_batched(const int64_t* items_dev, int32_t n, int64_t total_blocks, void* stream);

/* "thin" 3x3 convolutions (3-4 channels on one side: in_conv unet.py:217, out_conv :232).  The taps are moved to the thin
 * side so that the wide side is a plain vd_gemm instead of an implicit GEMM with a 64-wide tile on 3 useful columns:
 *   thin input :  xc = vd_im2col3x3(x) [pixels][9*C]            then  y = vd_gemm(xc, wpack[Cout][9*C])  (+ its transposes)
 *   thin output:  z  = vd_gemm(a, wpack viewed [9*Cout][Cin])    then  y = vd_tap_gather(z, bias)
 *   gradients  :  dz = vd_tap_spread(dy);  da = vd_gemm(dz, wpack);  vd_gemm(dz^T a) -> vd_thin_wgrad_finish -> OIHW */
int vd_im2col3x3(const float* x, int64_t ldx, float* xc /* [nimg*H*W][9*C] */, int32_t nimg, int32_t H, int32_t W, int32_t C, void* stream);
/* out[p][co] = bias[co] + sum_tap z[p + off(tap)][co*9 + tap]   (zero padding; co < Cout) */
int vd_tap_gather(const float* z, int64_t ldz, const float* bias, float* out, int64_t ldo, int32_t nimg, int32_t H, int32_t W, int32_t Cout, void* stream);
/* dz[q][co*9 + tap] = dy[q - off(tap)][co]; columns [9*Cout, ldz) are zero-filled */
int vd_tap_spread(const float* dy, int64_t lddy, float* dz, int64_t ldz, int32_t nimg, int32_t H, int32_t W, int32_t Cout, void* stream);
/* g[co][tap*Cin + ci] -> dw_oihw[co][ci][tap] (+)= (ci < Cin_w);  dbias[co] (+)= colsum[co*cs_stride + cs_off] (optional) */
int vd_thin_wgrad_finish(const float* g, int32_t Cout_w, int32_t Cin, int32_t Cin_w, float* dw_oihw, int32_t accumulate,
                         const float* colsum, int32_t cs_stride, int32_t cs_off, float* dbias, void* stream);

/* weight (and bias) gradient of the same convolution, summed over the whole batch (autograd of F.conv2d):
 *   dw_oihw[co][ci][tap] (+)= sum_{b,y,x} dy[b,y,x,co] * xin[b,y+dy,x+dx,ci]      co < Cout_w, ci < Cin_w
 *   dbias[co]            (+)= sum_{b,y,x} dy[b,y,x,co]                             (dbias may be NULL)
 * xin has Cin (multiple of 4, >= Cin_w) channels, dy has Cout (multiple of 4, >= Cout_w) channels. */
size_t vd_conv3x3_wgrad_ws_bytes(int32_t nimg, int32_t H, int32_t W, int32_t Cin, int32_t Cout);
int vd_conv3x3_wgrad(const float* xin, int64_t ldx, const float* dy, int64_t lddy,
                     int32_t nimg, int32_t H, int32_t W, int32_t Cin, int32_t Cout,
                     float* dw_oihw, float* dbias, int32_t Cin_w, int32_t Cout_w, int32_t accumulate,
                     float* ws, size_t ws_bytes, void* stream);
/* the same in two calls, for per-kernel timing (bench.py's live roofline figure): phase 1 = the split-K MFMA kernel into ws,
 * phase 2 = slab reduction + OIHW transposition into dw_oihw/dbias.  Same stream, same thread, phase 1 first. */
int vd_conv3x3_wgrad_phase(const float* xin, int64_t ldx, const float* dy, int64_t lddy,
                           int32_t nimg, int32_t H, int32_t W, int32_t Cin, int32_t Cout,
                           float* dw_oihw, float* dbias, int32_t Cin_w, int32_t Cout_w, int32_t accumulate,
                           float* ws, size_t ws_bytes, int32_t phase, void* stream);

/* OIHW (Cout_w, Cin_w, 3, 3) -> wf[Cout_w][9][Cin_p]  (forward)  and/or  wd[Cin_w][9][Cout_p] with the taps
 * mirrored (dgrad).  Either output may be NULL.  Padding channels are zero-filled. */
int vd_pack_conv3x3(const float* w_oihw, int32_t Cout_w, int32_t Cin_w,
                    float* wf, int32_t Cin_p, float* wd, int32_t Cout_p, void* stream);
/* the same for many tensors in one launch.  items_dev: DEVICE table of n x 8 int64 {w_oihw, wf (or 0), wd (or 0), Cout_w,
 * Cin_w, Cin_p, Cout_p, first_block}; tensor i owns blocks [first_block_i, first_block_{i+1}) of 256 elements each, enough
 * for max(Cout_w*9*Cin_p, Cin_w*9*Cout_p); total_blocks = end of the last tensor. */
int vd_pack_conv3x3_batched(const int64_t* items_dev, int32_t n, int64_t total_blocks, void* stream);

/* ------------------------------------------------------------------ GroupNorm(32, C, eps) + SiLU + FiLM + dropout + resample
 * (nn.GroupNorm unet.py:28-30, nn.SiLU unet.py:25, FiLM unet.py:145-146, nn.Dropout unet.py:135,147,
 *  nn.AvgPool2d / nn.Upsample unet.py:127-132) */
enum { VD_RS_NONE = 0, VD_RS_DOWN = 1, VD_RS_UP = 2 };

/* per-(image, group) mean and 1/sqrt(var+eps) of x[nimg][HW][C]; stats = [nimg][G][2] */
size_t vd_gn_ws_bytes(int32_t nimg, int32_t HW, int32_t C);
int vd_gn_stats(const float* x, int64_t ldx, int32_t nimg, int32_t HW, int32_t C, int32_t G, float eps,
                float* stats, float* ws, size_t ws_bytes, void* stream);

/* the same statistics from the partial sums a producer epilogue left behind (vd_gemm_desc.stats / vd_conv3x3 stats_part):
 * the normalised tensor is the channel concatenation [C1 | C2] of up to two produced tensors (C2 = 0: one source) */
int vd_gn_stats_from_partials(const float* part1, int32_t C1, int32_t chunks1, const float* part2, int32_t C2, int32_t chunks2,
                              int32_t nimg, int32_t HW, int32_t G, float eps, float* stats, void* stream);

/* ... or straight to the coefficient table vd_gn_apply works from (then call vd_gn_apply with stats = NULL) */
int vd_gn_coef_from_partials(const float* part1, int32_t C1, int32_t chunks1, const float* part2, int32_t C2, int32_t chunks2,
                             int32_t nimg, int32_t HW, int32_t G, float eps, const float* gamma, const float* beta,
                             const float* film, float* coef, void* stream);

/* y = resample( dropout( act( (1+scale) * GN(x) + shift ) ) )
 * film: [nimg][2C] (shift first, scale second, unet.py:145) or NULL; act: 1 = SiLU, 0 = identity;
 * gamma/beta NULL => plain resample of x (the skip path, unet.py:138).
 * coef: [nimg][4][C] floats, kept for the backward pass; written here from `stats`, or -- with stats = NULL -- already
 * filled by vd_gn_coef_from_partials. */
int vd_gn_apply(const float* x, int64_t ldx, const float* stats, const float* gamma, const float* beta,
                const float* film, int32_t act, float p_drop, uint64_t seed,
                int32_t resample, float* y, int64_t ldy,
                int32_t nimg, int32_t H, int32_t W, int32_t C, int32_t G, float* coef, void* stream);

/* vd_gn_coef_from_partials + vd_gn_apply in ONE launch (reference: the same nn.GroupNorm call sites, unet.py:28-30 <- :52,119,123,230):
 * every workgroup of the apply pass derives the statistics of the groups it touches from the producers' partial sums (part1 / part2:
 * layout of vd_gemm `stats`, channel-concatenated sources) and the workgroups of pixel chunk 0 write the [nimg][4][C] table `coef`
 * (NULL: not needed, e.g. inference) that vd_gn_apply_bwd reads. */
int vd_gn_apply_from_partials(const float* x, int64_t ldx, const float* part1, int32_t C1, int32_t chunks1, const float* part2, int32_t C2,
                              int32_t chunks2, const float* gamma, const float* beta, const float* film, int32_t act, float p_drop,
                              uint64_t seed, int32_t resample, float* y, int64_t ldy, int32_t nimg, int32_t H, int32_t W, int32_t C,
                              int32_t G, float eps, float* coef, void* stream);

/* backward of vd_gn_apply.  dy is at the OUTPUT resolution of the forward op.
 *   dx (+)= d/dx ; dfilm [nimg][2C] (written) ; dgamma/dbeta (+)= (accumulate_params)
 * add: optional tensor (input resolution, ld = ldadd) added into dx (the skip-path gradient). */
int vd_gn_apply_bwd(const float* dy, int64_t lddy, const float* x, int64_t ldx, const float* coef,
                    const float* gamma, const float* beta, const float* film, int32_t act, float p_drop, uint64_t seed,
                    int32_t resample, const float* add, int64_t ldadd, float* dx, int64_t lddx, int32_t accumulate_dx,
                    float* dfilm, float* dgamma, float* dbeta, int32_t accumulate_params,
                    int32_t nimg, int32_t H, int32_t W, int32_t C, int32_t G,
                    float* ws, size_t ws_bytes, void* stream);
/* NPT * 10000 + TPB (+ 1000000 for the instantiation with non-temporal loads / stores: slabs of whole 128-byte pixel rows) of the calling
 * thread's last vd_gn_apply_bwd launch when it took the single-pass form gn_bwd_fused_kernel<NPT, TPB, NT>,
 * -1 for the two-pass form (chan_reduce_kernel<1> + gn_bwd_finalize_kernel + gn_bwd_apply_kernel), 0 without a norm (plain resample
 * backward: gn_bwd_apply_kernel).  Profiling aid, like vd_gemm_last_tile. */
int vd_gn_bwd_last_kernel(void);
/* vd_gn_apply_bwd with the sum over images of the per-image dgamma / dbeta terms left to the caller: they are written to
 * pgb_keep[nimg][2][C] ({dgamma, dbeta} terms) and ONE vd_gn_param_sums_batched launch sums them for all norms of a backward pass
 * (the per-norm launch is pure latency: 73 per CIFAR train step).  items_dev = [n][8] int64 {pgb, dgamma, dbeta, nimg, C, accumulate, 0,
 * first 256-thread block}; a norm takes ceil(C / 16) blocks. */
int vd_gn_apply_bwd_keep(const float* dy, int64_t lddy, const float* x, int64_t ldx, const float* coef, const float* gamma,
                         const float* beta, const float* film, int32_t act, float p_drop, uint64_t seed, int32_t resample,
                         const float* add, int64_t ldadd, float* dx, int64_t lddx, int32_t accumulate_dx, float* dfilm, float* pgb_keep,
                         int32_t nimg, int32_t H, int32_t W, int32_t C, int32_t G, float* ws, size_t ws_bytes, void* stream);
int vd_gn_param_sums_batched(const int64_t* items_dev, int32_t n, int64_t total_blocks, void* stream);

/* ------------------------------------------------------------------ small reductions / elementwise
 * out[n] (+)= sum_m x[m][n]   (bias gradients) */
size_t vd_colsum_ws_bytes(int64_t M, int32_t N);
int vd_colsum(const float* x, int64_t ldx, int64_t M, int32_t N, float* out, int32_t accumulate,
              float* ws, size_t ws_bytes, void* stream);
/* y = alpha*x + beta*y over [rows][C] with strides */
int vd_axpby(const float* x, int64_t ldx, float alpha, float* y, int64_t ldy, float beta, int64_t rows, int32_t C, void* stream);
/* SiLU forward / backward on a dense vector (nn.SiLU on t_emb, unet.py:142,203) */
int vd_silu(const float* x, float* y, int64_t n, void* stream);
int vd_silu_bwd(const float* x, const float* dy, float* dx, int64_t n, int32_t accumulate, void* stream);
/* row softmax in place over [rows][L] (torch.softmax unet.py:58-59) and its backward: ds = alpha * p * (dp - sum(dp*p)) */
int vd_softmax_rows(float* s, int64_t rows, int32_t L, void* stream);
int vd_softmax_rows_bwd(const float* p, float* dp, int64_t rows, int32_t L, float alpha, void* stream);

/* fused scaled-dot-product attention (BaseAttentionBlock.scaled_dot_product, unet.py:55-64, and its autograd): o = softmax(scale q k^T) v
 * per (image b, head h) without the [B, nh, L, L] logits in HBM.  Operands are rows of the qkv projection:
 * x[(b L + l) ld + h hd + d].  Served: L % 64 == 0 and hd in {64, 128, 256} forward, {64, 128} backward (vd_attn_supported);
 * everything else takes vd_gemm -> vd_softmax_rows -> vd_gemm.  lse [B nh L] (forward output, may be NULL when no backward
 * follows) and delta [B nh L] (backward scratch) are opaque to the caller.  Fixed summation order: bitwise reproducible. */
int vd_attn_supported(int32_t L, int32_t hd, int32_t backward);
int vd_attn_fwd(const float* q, const float* k, const float* v, int64_t ld, float* o, int64_t ldo, float* lse, int32_t B,
                int32_t nh, int32_t L, int32_t hd, float scale, void* stream);
int vd_attn_bwd(const float* q, const float* k, const float* v, int64_t ld, const float* o, int64_t ldo, const float* dout,
                int64_t lddo, const float* lse, float* delta, float* dq, float* dk, float* dv, int64_t ldd, int32_t B,
                int32_t nh, int32_t L, int32_t hd, float scale, void* stream);
/* the same in two calls, for per-kernel timing: phase 1 = dQ (and delta), phase 2 = dK, dV (after phase 1) */
int vd_attn_bwd_phase(const float* q, const float* k, const float* v, int64_t ld, const float* o, int64_t ldo, const float* dout,
                      int64_t lddo, const float* lse, float* delta, float* dq, float* dk, float* dv, int64_t ldd, int32_t B,
                      int32_t nh, int32_t L, int32_t hd, float scale, int32_t phase, void* stream);

/* layout changes at the boundary of the NCHW call surface */
int vd_nchw_to_nhwc(const float* x, float* y, int32_t nimg, int32_t C, int32_t H, int32_t W, int64_t ldy, void* stream);
int vd_nhwc_to_nchw(const float* x, int64_t ldx, float* y, int32_t nimg, int32_t C, int32_t H, int32_t W, void* stream);

/* the data formats either side of the path:
 *   out_u8[b][y][x][c] = uint8(clamp(x*127.5+127.5, 0, 255))   generate.py:149 (quantise + NCHW->HWC pack, on device)
 *   out[b][c][y][x]    = (u8[b][y][x'][c]/255 - 0.5)/0.5, x' mirrored where flip[b]   datasets.py:115-120 (flip may be NULL) */
int vd_images_to_uint8_hwc(const float* x_nchw, uint8_t* out_hwc, int32_t n, int32_t C, int32_t HW, void* stream);
int vd_images_from_uint8_hwc(const uint8_t* in_hwc, const uint8_t* flip, float* out_nchw, int32_t n, int32_t C,
                             int32_t H, int32_t W, void* stream);

/* ------------------------------------------------------------------ embeddings
 * get_timestep_embedding (functions.py:11-29): fp64 arithmetic, fp32 result [n][dim] */
int vd_timestep_embedding(const double* t, float* out, int32_t n, int32_t dim, double scale, void* stream);
/* OneHot(exclude_zero) + Linear (modules.py:184-201, unet.py:209-215,295): temb[b] += W[:, y-1] (y>0) + bias */
int vd_class_embed(const float* y, const float* w, const float* bias, float* temb, int32_t n, int32_t emb, int32_t ncls, void* stream);
int vd_class_embed_bwd(const float* y, const float* dtemb, float* dw, float* dbias, int32_t n, int32_t emb, int32_t ncls,
                       int32_t accumulate, void* stream);
/* multitag normalisation y / sqrt(max(nnz,1)) (unet.py:290-294) */
int vd_multitag_norm(const float* y, float* out, int32_t n, int32_t ncls, void* stream);

/* ------------------------------------------------------------------ diffusion process (diffusion.py)
 * All images NCHW (the reference call surface); the UNet converts to NHWC at its own boundary.
 * model_out_type: 0 = v, 1 = x0, 2 = eps, 3 = both ; reweight: 0 = constant, 1 = snr, 2 = snr_trunc, 3 = snr_1plus */
/* q_sample (diffusion.py:242-245): xt = sqrt(sigmoid(l_b)) x0 + sqrt(sigmoid(-l_b)) eps */
int vd_q_sample(const float* x0, const float* eps, const float* logsnr, float* xt, int32_t n, int32_t C, int32_t HW, void* stream);
/* train_loss mse branch (diffusion.py:520-541 with the conversions of :466-490): per-sample loss[n];
 * aux[n][2] keeps the two branch means of snr_trunc (x0-mse, eps-mse) for the backward pass.
 * out has C channels (2C for model_out_type "both"). */
int vd_loss_fwd(const float* x0, const float* eps, const float* xt, const float* out, const float* logsnr,
                int32_t model_out_type, int32_t reweight, float* loss, float* aux, int32_t n, int32_t C, int32_t HW, void* stream);
/* dout = gloss[b] * d loss_b / d out */
int vd_loss_bwd(const float* x0, const float* eps, const float* xt, const float* out, const float* logsnr,
                const float* aux, const float* gloss, int32_t model_out_type, int32_t reweight, float* dout,
                int32_t n, int32_t C, int32_t HW, void* stream);
/* variational-bound terms of one step (loss_type "kl": GaussianDiffusion._loss_term_bpd diffusion.py:446-464 with
 * normal_kl / discretized_gaussian_loglik of functions.py:31-67), per sample and in bits per dimension:
 *   kl[b]  = mean KL( q(x_s|x_t,x_0) || p(x_s|x_t) ) / ln 2 ,  nll[b] = mean -log p(x_0|x_t) / ln 2 (8-bit bins)
 * coef[n][8] (device) = {a0, b0x, b0e, c1, c2, true_logvar, model_logvar, 0} per sample: x0_hat = clip?(a0*xt + b0x*out
 * (+ b0e*out_eps)), true_mean = c1*xt + c2*x0, model_mean = c1*xt + c2*x0_hat.  pred (x0_hat, NCHW) and mse[n] are optional. */
int vd_bpd_terms(const float* x0, const float* xt, const float* out, const float* coef, int32_t model_out_type, int32_t clip,
                 float* kl, float* nll, float* pred, float* mse, int32_t n, int32_t C, int32_t HW, void* stream);
/* dout = gloss[b] * d (use_kl[b] != 0 ? kl[b] : nll[b]) / d out   (autograd of the "kl" branch of train_loss, :497-515) */
int vd_bpd_bwd(const float* x0, const float* xt, const float* out, const float* coef, const float* use_kl, const float* gloss,
               int32_t model_out_type, int32_t clip, float* dout, int32_t n, int32_t C, int32_t HW, void* stream);
/* one reverse step (p_mean_var + CFG + noise, diffusion.py:317-392) for a batch that shares the step index.
 * out has n*(1+cfg) images, cond/uncond interleaved when cfg (diffusion.py:369-372).
 * k: 8 HOST floats {a0, b0x, b0e, c1, c2, noise_scale, w_guide, 0}:  x0_hat = clip(a0*xt + b0x*out (+ b0e*out_eps)),
 * mean = last_step ? x0_hat : c1*xt + c2*x0_hat, guided = mean_c + w (mean_c - mean_u), xn = guided + noise_scale*noise.
 * xdup (optional): the new state duplicated to 2n interleaved images = the next CFG UNet input.
 * Exactly one of k (host) / k_dev (device, same 8 floats) is given: the device form keeps the launch replayable from a
 * captured HIP graph with per-step coefficients.  xn may alias xt (element-wise in-place update). */
int vd_sample_step(const float* xt, const float* out, const float* noise, const float* k, const float* k_dev,
                   int32_t model_out_type, int32_t cfg, int32_t last_step, int32_t clip,
                   float* xn, float* xdup, int32_t n, int32_t C, int32_t HW, void* stream);

/* ------------------------------------------------------------------ optimizer tail (train_utils.py:159-168, utils.py:144-149)
 * sum of squares of a flat buffer (global-norm clip), fused clip + AdamW + EMA over flat fp32 buffers */
size_t vd_sumsq_ws_bytes(int64_t n);
int vd_sumsq(const float* g, int64_t n, float* out1, float* ws, size_t ws_bytes, void* stream);
/* [r_lo, r_hi) with r_mode != 0 is an index range treated apart this step (the class-embedding tensors, unet.py:207-215):
 * r_mode 1 = the range received no gradient (class-conditional net called with y = None; the reference leaves .grad None and
 * torch.optim.AdamW skips it: p, m, v untouched, per-parameter step not advanced; the EMA shadow still follows p);
 * r_mode 2 = the range is updated with its own bias corrections r_bc1, r_bc2 (its step count lags). r_mode 0: ignored. */
int vd_adamw_ema(float* p, const float* g, float* m, float* v, float* ema, int64_t n,
                 const float* gnorm_sq, float max_norm, float lr, float beta1, float beta2, float eps, float wd,
                 float bc1, float bc2, float ema_decay, int64_t r_lo, int64_t r_hi, int32_t r_mode, float r_bc1, float r_bc2,
                 void* stream);
/* the same update with the decision about [r_lo, r_hi) made ON THE DEVICE: r_flag[0] > 0 (device float: "some rank's micro-batch of
 * this update carried labels" -- one slot behind the flat gradient buffer, summed over ranks by the last gradient bucket's
 * all-reduce) -> the range is updated with the bias corrections of step count r_steps[0] + 1 (1 - r_beta^k, evaluated in double on
 * the device) and r_steps[0] is advanced; else the range is skipped as r_mode 1 does.  No host ever reads the flag: replaces the
 * per-update host all-reduce the reference's DDP + torch.optim.AdamW pair needs no equivalent of (train_utils.py:159-163:
 * optimizer.step() skips parameters whose .grad is None; under DDP that is the same on every rank by construction). */
int vd_adamw_ema_flagged(float* p, const float* g, float* m, float* v, float* ema, int64_t n,
                         const float* gnorm_sq, float max_norm, float lr, float beta1, float beta2, float eps, float wd,
                         float bc1, float bc2, float ema_decay, int64_t r_lo, int64_t r_hi, const float* r_flag, int32_t* r_steps,
                         double r_beta1, double r_beta2, void* stream);

#ifdef __cplusplus
}
#endif
#endif
