// gemm.hip -- the fp32 MFMA tile engine behind every matmul-shaped op of the hot path:
//   * 3x3 convolution forward / input-gradient (implicit GEMM, A = NHWC patches)   reference modules.py:141-144
//   * 3x3 convolution weight-gradient (implicit GEMM over pixels, split-K slabs)    autograd of the same
//   * 1x1 convolution / linear / attention contractions and their gradients         unet.py:57-63,70-71,134; modules.py:79-80
//
// Design (gfx950 / CDNA4):
//   - v_mfma_f32_32x32x2_f32: exact fp32 (bitwise an fmaf chain), 64 FLOP/clk/SIMD.  A workgroup is 4 waves
//     (one per SIMD), each owning a (BM/2)x(BN/2) sub-tile as 32x32 accumulator blocks.
//   - K is consumed in tiles of 32.  Tiles are staged global -> registers -> LDS (so transforms / zero padding
//     can be applied on the way) and double buffered: the global loads of tile t+1 are in flight while tile t
//     feeds the matrix cores; one barrier per K tile.
//   - operands that are k-contiguous in memory live in LDS as [row][32] with the 16-byte chunk index XOR-swizzled
//     by (row>>1)&7, read back with one conflict-free ds_read_b128 per 4 MFMAs (lane (i,h) takes k = 8s+4h+j,
//     j = 0..3: the MFMA's own k order is a free permutation as long as A and B agree); operands that are
//     row-contiguous live as [32][rows] and are read with ds_read_b32 (lanes = consecutive rows).
//   - all edges are predicated (rows, columns, k, image borders, channel padding), so ragged shapes need no padding
//     beyond 4-float alignment of the contiguous dimension.
#include "common.h"
#include <math.h>
#include <stdlib.h>
#include <type_traits>

namespace {

constexpr int KT = 32;   // K tile
#ifndef VD_GEMM_SPLIT_DEFAULT
#define VD_GEMM_SPLIT_DEFAULT 1      /* round 5: the split-operand forms are the shipped default (VD_GEMM_SPLIT=0: fp32 MFMA everywhere, the A/B form) */
#endif

struct GemmArgs {
    const float* A; const float* B; float* C; const float* bias; const float* R;
    int M, N, K;
    long long lda, ldb, ldc, ldr;
    int nh;
    long long sAb, sAh, sBb, sBh, sCb, sCh, sRb, sRh;
    float alpha; int accumulate;
    int H, W, Cin;
    int kt_total, kt_per_split;
    long long slab_stride;
    float* colsum; int colsum_accumulate;     // COL-kind A only: colsum[m] (+)= sum_k A[k][m]  (bias gradients ride along)
    int probe;                                // -DVD_PROBES builds only (VD_GEMM_PROBE; always 0 in the product library): bit0/bit1 drop the A/B tile loads, bit2 the
                                              // barrier, bit4 = phase timestamps through `colsum`, bit5 = loop-phase cycles through `stats`
    float* stats; int stats_hw;               // GroupNorm partials of the OUTPUT: [img][chunk][2][N], chunk = BM/2 output rows
    long long sBias;                          // bias offset per batch entry zb
    int group_S;                              // GROUPED launches: split-K slabs per entry
    int lgW, lgHW;                            // log2 of W and H*W when both are powers of two, else -1 (shift/mask instead of divisions)
    long long a_kblk, b_kblk;                 // GROUPED COL operands stored in K BLOCKS of 16 rows: row k of an entry sits at (k >> 4) * kblk + (k & 15) * ld
                                              // (0: plain [K][ld] rows).  Lets 36 same-shape operands interleave block by block: [K/16][36][16][ld]
};

// timing-probe bits: compiled out of the product library (VD_PROBE_BUILD is a constant: every `PB(p) & bit` branch folds away);
// -DVD_PROBES builds (libvdiff_hip_probe.so, tests/probe/) read them from VD_GEMM_PROBE
__device__ __forceinline__ int PB(const GemmArgs& p) { return VD_PROBE_BUILD ? p.probe : 0; }

__device__ __forceinline__ int row_swz(int row, int chunk) { return row * KT + ((chunk ^ ((row >> 1) & 7)) << 2); }

template <int BM, int BN, int AK, int BK, bool SPLITK>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const GemmArgs p) {
    __shared__ __attribute__((aligned(16))) float smem[2 * (BM + BN) * KT];
    float* As = smem;
    float* Bs = smem + 2 * BM * KT;

    constexpr int MT = BM / 64, NT = BN / 64;          // 32x32 blocks per wave in m / n
    constexpr int AIT = BM / 32, BIT = BN / 32;        // float4 staged per thread
    constexpr int A_CPR = BM / 4, A_RPP = 256 / A_CPR; // COL-layout tile: chunks per k-row, k-rows per pass
    constexpr int B_CPR = BN / 4, B_RPP = 256 / B_CPR;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int wm = (wave >> 1) * (BM / 2), wn = (wave & 1) * (BN / 2);
    const int m0 = blockIdx.y * BM;
    int n0 = blockIdx.x * BN;
    int tapN = 0, ci0 = 0;
    if (BK == VD_IM2COL) {
        const int cchN = (p.Cin + BN - 1) / BN;
        tapN = blockIdx.x / cchN;
        ci0 = (blockIdx.x % cchN) * BN;
        n0 = tapN * p.Cin + ci0;
    }
    const float* A = p.A;
    const float* B = p.B;
    float* C = p.C;
    const float* R = p.R;
    const float* biasp = p.bias;
    int kt_begin = 0, kt_end = p.kt_total;
    if (SPLITK) {
        kt_begin = blockIdx.z * p.kt_per_split;
        kt_end = min(kt_begin + p.kt_per_split, p.kt_total);
        C += (long long)blockIdx.z * p.slab_stride;
    } else {
        const int zb = blockIdx.z / p.nh, zh = blockIdx.z % p.nh;
        A += zb * p.sAb + zh * p.sAh;
        B += zb * p.sBb + zh * p.sBh;
        C += zb * p.sCb + zh * p.sCh;
        if (R) R += zb * p.sRb + zh * p.sRh;
        if (biasp) biasp += zb * p.sBias;
    }

    // ---- per-thread staging coordinates
    const int r_row = tid >> 3, r_chunk = tid & 7;       // ROW-layout: row = r_row + 32*it, 16-byte chunk r_chunk
    int ay[AIT], ax[AIT];                                // image coordinates of the A rows (IM2COL A)
    if (AK == VD_IM2COL) {
#pragma unroll
        for (int it = 0; it < AIT; ++it) {
            const int m = m0 + r_row + 32 * it;
            const int rem = m % (p.H * p.W);
            ay[it] = rem / p.W;
            ax[it] = rem % p.W;
        }
    }
    // K-tile cursor of the im2col A operand: (tap, channel chunk), advanced once per load_tiles call (tiles are
    // visited in order), so the main loop has no integer division
    // (taps are the FAST index: the 9 shifted reads of one 32-channel slab are adjacent in time and hit L1/L2)
    int tapA = (AK == VD_IM2COL) ? kt_begin % 9 : 0;
    int ccA = (AK == VD_IM2COL) ? kt_begin / 9 : 0;
    // pixel cursor of the im2col B operand (wgrad): image coordinates of this thread's k rows
    int by[BIT], bx[BIT];
    const int kinc_x = (BK == VD_IM2COL) ? KT % p.W : 0, kinc_y = (BK == VD_IM2COL) ? KT / p.W : 0;
    if (BK == VD_IM2COL) {
#pragma unroll
        for (int it = 0; it < BIT; ++it) {
            const int k = kt_begin * KT + tid / B_CPR + it * B_RPP;
            const int rem = k % (p.H * p.W);
            by[it] = rem / p.W;
            bx[it] = rem % p.W;
        }
    }

    f32x4 ra[AIT], rb[BIT];

    auto load_tiles = [&](int kt) {
        // ---------------- A
        const int tap = tapA, c0 = ccA * KT;
        if (AK == VD_ROW || AK == VD_IM2COL) {
            int kbase, kwidth, dy = 0, dx = 0;
            if (AK == VD_IM2COL) {
                dy = tap / 3 - 1; dx = tap % 3 - 1;
                kbase = c0; kwidth = p.Cin - c0;
            } else { kbase = kt * KT; kwidth = p.K - kbase; }
            const int ck = r_chunk * 4;
#pragma unroll
            for (int it = 0; it < AIT; ++it) {
                const int m = m0 + r_row + 32 * it;
                bool ok = (m < p.M) && (ck < kwidth);
                long long off;
                if (AK == VD_IM2COL) {
                    ok = ok && ((unsigned)(ay[it] + dy) < (unsigned)p.H) && ((unsigned)(ax[it] + dx) < (unsigned)p.W);
                    off = ((long long)m + dy * p.W + dx) * p.lda + kbase + ck;
                } else off = (long long)m * p.lda + kbase + ck;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (ok) v = *reinterpret_cast<const f32x4*>(A + off);
                ra[it] = v;
            }
        } else {   // VD_COL: A stored [k][m]
#pragma unroll
            for (int it = 0; it < AIT; ++it) {
                const int kk = tid / A_CPR + it * A_RPP, cm = (tid % A_CPR) * 4;
                const int k = kt * KT + kk;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (k < p.K && m0 + cm < p.M) v = *reinterpret_cast<const f32x4*>(A + (long long)k * p.lda + m0 + cm);
                ra[it] = v;
            }
        }
        // ---------------- B
        if (BK == VD_ROW) {
            int kbase, kwidth;
            if (AK == VD_IM2COL) { kbase = tap * p.Cin + c0; kwidth = p.Cin - c0; }
            else { kbase = kt * KT; kwidth = p.K - kbase; }
            const int ck = r_chunk * 4;
#pragma unroll
            for (int it = 0; it < BIT; ++it) {
                const int n = n0 + r_row + 32 * it;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (n < p.N && ck < kwidth) v = *reinterpret_cast<const f32x4*>(B + (long long)n * p.ldb + kbase + ck);
                rb[it] = v;
            }
        } else if (BK == VD_COL) {
#pragma unroll
            for (int it = 0; it < BIT; ++it) {
                const int kk = tid / B_CPR + it * B_RPP, cn = (tid % B_CPR) * 4;
                const int k = kt * KT + kk;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (k < p.K && n0 + cn < p.N) v = *reinterpret_cast<const f32x4*>(B + (long long)k * p.ldb + n0 + cn);
                rb[it] = v;
            }
        } else {   // VD_IM2COL B (wgrad): k = pixel, n = (tapN, ci0 + cn)
            const int dy = tapN / 3 - 1, dx = tapN % 3 - 1;
#pragma unroll
            for (int it = 0; it < BIT; ++it) {
                const int kk = tid / B_CPR + it * B_RPP, cn = (tid % B_CPR) * 4;
                const int k = kt * KT + kk;
                const int yy = by[it] + dy, xx = bx[it] + dx;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (k < p.K && (unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W && ci0 + cn < p.Cin)
                    v = *reinterpret_cast<const f32x4*>(B + ((long long)k + dy * p.W + dx) * p.ldb + ci0 + cn);
                rb[it] = v;
                // advance this row's pixel by KT for the next tile
                int nx = bx[it] + kinc_x, ny = by[it] + kinc_y;
                if (nx >= p.W) { nx -= p.W; ++ny; }
                while (ny >= p.H) ny -= p.H;
                bx[it] = nx; by[it] = ny;
            }
        }
        if (AK == VD_IM2COL) { if (++tapA == 9) { tapA = 0; ++ccA; } }
    };

    auto store_tiles = [&](int buf) {
        float* as = As + buf * BM * KT;
        float* bs = Bs + buf * BN * KT;
        if (AK == VD_COL) {
#pragma unroll
            for (int it = 0; it < AIT; ++it) {
                const int kk = tid / A_CPR + it * A_RPP, cm = (tid % A_CPR) * 4;
                *reinterpret_cast<f32x4*>(as + kk * BM + cm) = ra[it];
            }
        } else {
#pragma unroll
            for (int it = 0; it < AIT; ++it)
                *reinterpret_cast<f32x4*>(as + row_swz(r_row + 32 * it, r_chunk)) = ra[it];
        }
        if (BK == VD_ROW) {
#pragma unroll
            for (int it = 0; it < BIT; ++it)
                *reinterpret_cast<f32x4*>(bs + row_swz(r_row + 32 * it, r_chunk)) = rb[it];
        } else {
#pragma unroll
            for (int it = 0; it < BIT; ++it) {
                const int kk = tid / B_CPR + it * B_RPP, cn = (tid % B_CPR) * 4;
                *reinterpret_cast<f32x4*>(bs + kk * BN + cn) = rb[it];
            }
        }
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // bias gradient riding on the weight-gradient GEMM: the A operand of a COL-kind launch is dY, whose column sums are
    // d(bias).  The waves of n-tile 0 / wave column 0 already stream every A fragment through registers.
    float csum[MT];
#pragma unroll
    for (int a = 0; a < MT; ++a) csum[a] = 0.f;
    const bool do_cs = (AK == VD_COL) && p.colsum != nullptr && blockIdx.x == 0 && (wave & 1) == 0;

    auto compute = [&](int buf) {
        const float* as = As + buf * BM * KT;
        const float* bs = Bs + buf * BN * KT;
#pragma unroll
        for (int s = 0; s < KT / 8; ++s) {
            f32x4 fa[MT], fb[NT];
#pragma unroll
            for (int a = 0; a < MT; ++a) {
                const int row = wm + 32 * a + li;
                if (AK == VD_COL) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) fa[a][j] = as[(8 * s + 4 * lh + j) * BM + row];
                    if (do_cs) csum[a] += (fa[a][0] + fa[a][1]) + (fa[a][2] + fa[a][3]);
                } else fa[a] = *reinterpret_cast<const f32x4*>(as + row_swz(row, 2 * s + lh));
            }
#pragma unroll
            for (int b = 0; b < NT; ++b) {
                const int row = wn + 32 * b + li;
                if (BK == VD_ROW) fb[b] = *reinterpret_cast<const f32x4*>(bs + row_swz(row, 2 * s + lh));
                else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) fb[b][j] = bs[(8 * s + 4 * lh + j) * BN + row];
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int a = 0; a < MT; ++a)
#pragma unroll
                    for (int b = 0; b < NT; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a][j], fb[b][j], acc[a][b], 0, 0, 0);
        }
    };

    // ---- main loop: one barrier per K tile
    if (kt_begin < kt_end) {
        load_tiles(kt_begin);
        store_tiles(0);
        __syncthreads();
        int buf = 0;
        for (int kt = kt_begin; kt < kt_end; ++kt) {
            const bool more = kt + 1 < kt_end;
            if (more) load_tiles(kt + 1);
            compute(buf);
            if (more) store_tiles(buf ^ 1);
            __syncthreads();
            buf ^= 1;
        }
    }

    if (AK == VD_COL && do_cs) {
#pragma unroll
        for (int a = 0; a < MT; ++a) {
            const float v = csum[a] + __shfl_xor(csum[a], 32, 64);
            const int m = m0 + wm + 32 * a + li;
            if (lh == 0 && m < p.M) {
                float* o = p.colsum + (SPLITK ? (long long)blockIdx.z * p.M : 0) + m;
                *o = (!SPLITK && p.colsum_accumulate) ? *o + v : v;
            }
        }
    }
    // ---- epilogue: D[row][col], col = lane&31 (n), row = (r&3) + 8*(r>>2) + 4*(lane>>5) (m)
#pragma unroll
    for (int b = 0; b < NT; ++b) {
        const int ncol = wn + 32 * b + li;
        const int n = n0 + ncol;
        const bool nok = (BK == VD_IM2COL) ? (ci0 + ncol < p.Cin) : (n < p.N);
        if (!nok) continue;
        const float bv = (!SPLITK && biasp) ? biasp[n] : 0.f;
        float st1 = 0.f, st2 = 0.f;           // per-column sum / sum of squares of this wave's BM/2 output rows
#pragma unroll
        for (int a = 0; a < MT; ++a) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (m >= p.M) continue;
                float v = acc[a][b][r];
                if (!SPLITK) {
                    v = v * p.alpha + bv;
                    if (R) v += R[(long long)m * p.ldr + n];
                    if (p.accumulate) v += C[(long long)m * p.ldc + n];
                    st1 += v; st2 += v * v;
                }
                C[(long long)m * p.ldc + n] = v;
            }
        }
        // GroupNorm statistics of the tensor being written, for the norm that consumes it next (saves that norm's read
        // pass): the two lane halves hold complementary rows of the wave's BM/2-row slab, which lies inside one image
        if (!SPLITK && p.stats) {
            st1 += __shfl_xor(st1, 32, 64);
            st2 += __shfl_xor(st2, 32, 64);
            const int mrow = m0 + wm;
            if (lh == 0 && mrow < p.M) {
                const int chunks = p.stats_hw / (BM / 2);
                float* o = p.stats + ((long long)(mrow / p.stats_hw) * chunks + (mrow % p.stats_hw) / (BM / 2)) * 2 * p.N + n;
                o[0] = st1; o[p.N] = st2;
            }
        }
    }
}


// ------------------------------------------------------------------------------------------------------------------
// LDS-DMA variant (the one that normally runs): tiles go HBM/L2 -> LDS directly with `buffer_load_dwordx4 ... lds`
// (no VGPR staging, no ds_write pass).  Each wave-instruction writes 1 KiB of LDS linearly (lane x 16 B), so
//   * k-contiguous operands: one piece = 8 tile rows x 128 B; the XOR chunk swizzle is applied to the per-lane SOURCE
//     address (lane p of a row fetches chunk p ^ swz(row)), the fragment reads apply the same involution;
//   * row-contiguous operands: one piece = 1 KiB of consecutive [k][rows] -- already the LDS image.
// Every edge case is a per-lane offset select: an out-of-range voffset makes the DMA write ZEROS (probed on gfx950:
// tests/probe/oob.hip), which is exactly the conv zero padding / ragged-tile fill.  Per K tile a thread issues
// (BM+BN)/32 DMA instructions and a handful of selects; addresses are descriptor-base (SGPR) + constant 32-bit
// voffset + scalar soffset, so the loop has no 64-bit VALU address math and no branches.
typedef __attribute__((address_space(3))) void* lds_ptr_t;
constexpr unsigned OOB = 0x80000000u;      // >= num_records of every descriptor below
#ifndef VD_SCHED_INTERLEAVE
#define VD_SCHED_INTERLEAVE 1
#endif

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const float* base, int records = (int)OOB) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, records, 0x00020000);
}

// k-contiguous LDS image: [row][KT] floats, 16-byte chunk index XOR-swizzled so that a ds_read_b128 of 32 consecutive
// rows at one chunk is bank-conflict free (KT = 32: 8 chunks/row, swizzle (row>>1)&7; KT = 16: 4 chunks/row, (row>>2)&3)

// pixel index -> (y, x) inside its image; shifts when the geometry is a power of two (every shipped config), the divisions
// otherwise.  Prologue VALU instructions compete with the MFMA stream of the co-resident workgroups for issue slots.
__device__ __forceinline__ void pix_yx(const GemmArgs& p, int pix, int& y, int& x) {
    if (p.lgW >= 0) { const int rem = pix & ((1 << p.lgHW) - 1); y = rem >> p.lgW; x = rem & ((1 << p.lgW) - 1); }
    else { const int rem = pix % (p.H * p.W); y = rem / p.W; x = rem % p.W; }
}
template <int KT>
__device__ __forceinline__ int swz_of(int row) { return KT == 32 ? ((row >> 1) & 7) : ((row >> 2) & 3); }
template <int KT>
__device__ __forceinline__ int row_swz_t(int row, int chunk) { return row * KT + ((chunk ^ swz_of<KT>(row)) << 2); }

// KT = K tile.  128x128 tiles use KT = 16: 32 KB of LDS per workgroup -> 4 workgroups (4 waves per SIMD) per CU, which
// fills the MFMA issue slots a barrier-parked wave leaves empty (PMC: 2 waves/SIMD left the matrix pipe 14.5 % idle).
// TR: the MFMA operands are swapped (D = B.A^T), so a lane holds FOUR CONSECUTIVE COLUMNS of one output row per register
// quad instead of four consecutive rows of one column: the epilogue writes (and reads residuals) with 16 dwordx4 buffer
// instructions per wave instead of 64 dword ones.  Used by every launch that does not ask for output statistics (their
// per-column sums need a column per lane) when N, ldc, ldr are multiples of 4.
// GROUPED (split-K weight-gradient launches only): blockIdx.z = entry * group_S + slab, and entry e reads its A / B operands from
// gp.A[e] / gp.B[e] -- up to VD_GROUP_MAX same-shape GEMMs (the 1x1 / linear weight gradients of the blocks of one UNet level, whose
// operands live in unrelated buffers) in ONE launch instead of one 20-120 us launch each.  The pointer table travels in the kernel
// arguments; launches that are not grouped carry an empty struct instead.
constexpr int VD_GROUP_MAX = 36;      // (36: the xi planes of the F(4x4,3x3) weight gradient, wino43.hip)
struct GroupPtrs { const float* A[VD_GROUP_MAX]; const float* B[VD_GROUP_MAX]; };
struct NoGroup {};
struct GroupOut { float* C[VD_GROUP_MAX]; float* cs[VD_GROUP_MAX]; };

// SPL: fp32 products on the 16-bit matrix cores through split operands.  Every fp32 x is the exact sum of three bf16 pieces,
// h = bf16(x), m = bf16(x - h), l = bf16(x - h - m) (round to nearest even), and a.b = hh + hm + mh + mm + hl + lh up to terms below
// 2^-24 |a.b| (ml, lm, ll), accumulated in fp32 by v_mfma_f32_32x32x16_bf16: six instructions of 32 cycles contract 16 k where the fp32
// form needs eight of 64.  The split (5.5 vector instructions per element, in registers, after the same LDS-DMA tiles and fragment
// reads) issues in the shadow of the MFMAs of the other waves of the SIMD.  Measured against fp64 (tests/probe/split_mfma.hip, K = 1 024
// and 8 192, normal / positive / wide-range operands): relative L2 error 0.6-0.9 of the fp32 MFMA chain's, worst element 0.6-1.0 of it;
// three products (hh, hm, mh) are 3-10x worse and are not used.  Same-box rates of the bare loop: 216 TFLOP/s fp32-equivalent against
// 150 for the fp32 instruction (the chip's power limit, not the issue rate, bounds the 16-bit pipes on random data).
// DOMAIN (round-4 advice): finite operands of magnitude <= the largest bf16 (3.39e38).  Beyond it bf16(x) rounds to Inf, the residual
// x - Inf = -Inf and the products give NaN where the fp32 MFMA would give a finite value or Inf; an Inf operand gives NaN (Inf - Inf) instead
// of Inf; NaN stays NaN; residual pieces below the smallest normal bf16 are flushed (error <= 2^-126, far below fp32 rounding of any normal
// result).  The switch is opt-in and process-wide: a run whose activations overflow fp32's top binade has diverged either way, but its
// Inf shows up as NaN one GEMM earlier under VD_GEMM_SPLIT=1 (tests/test_kernels_gpu.py::test_split_operand_gemm_domain).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pk_bf16(float a, float b) {      // {bf16(a), bf16(b)}, a in the low half (v_cvt_pk_bf16_f32)
    const bf16x2 v = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ void split8(const float (&x)[8], bf16x8& H, bf16x8& M, bf16x8& L) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 h, m, l;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float x0 = x[2 * q], x1 = x[2 * q + 1];
        const unsigned hp = pk_bf16(x0, x1);
        const float r0 = x0 - __uint_as_float(hp << 16), r1 = x1 - __uint_as_float(hp & 0xFFFF0000u);      // exact
        const unsigned mp = pk_bf16(r0, r1);
        const float s0 = r0 - __uint_as_float(mp << 16), s1 = r1 - __uint_as_float(mp & 0xFFFF0000u);      // exact
        h[q] = hp; m[q] = mp; l[q] = pk_bf16(s0, s1);
    }
    H = __builtin_bit_cast(bf16x8, h); M = __builtin_bit_cast(bf16x8, m); L = __builtin_bit_cast(bf16x8, l);
}

#ifndef VD_KT16_BLOCKS
#define VD_KT16_BLOCKS 4
#endif
#ifndef VD_SPL_BLOCKS
#define VD_SPL_BLOCKS 2       /* workgroups per CU the split-operand KT = 16 forms are compiled for (round 5, same-box whole-step A/B of 2 / 3 / 4,
                                 profiles/r05_spl_blocks.txt: CIFAR 58.12 / 57.51 vs 58.58 / 57.70 vs 58.50 / 57.65 ms, CelebA 292.7 / 290.1 vs 293.9 / 290.8 vs 295.5 / 297.4) */
#endif
// the kernel body; instantiated by gemm_dma_kernel (SPL = false: fp32 MFMA) and gemm_split_kernel (SPL = true) below
template <int BM, int BN, int AK, int BK, bool SPLITK, int KT, bool TR, bool GROUPED, bool SPL>
__device__ __forceinline__ void gemm_dma_body(const GemmArgs& p, const std::conditional_t<GROUPED, GroupPtrs, NoGroup>& gp) {
    constexpr int NBUF = 2;
    __shared__ __attribute__((aligned(1024))) float smem[NBUF * (BM + BN) * KT];
    constexpr int MT = BM / 64, NT = BN / 64;
    constexpr int AIT = BM * KT / 1024, BIT = BN * KT / 1024;   // DMA pieces (1 KiB) per wave for A / B
    constexpr int A_LPR = BM / 4, B_LPR = BN / 4;        // lanes per k-row of a row-contiguous tile
    constexpr int A_RPP = 64 / A_LPR, B_RPP = 64 / B_LPR;// k-rows per piece
    constexpr int RLPR = KT / 4, RRPP = 64 / RLPR;       // k-contiguous tile: lanes per row, rows per piece
    // Row-contiguous ([k][rows]) operands with two 32-row MFMA blocks per wave: lane i reads rows 2i, 2i+1 with ONE
    // ds_read_b64 and block a takes element a, i.e. MFMA block a owns the interleaved rows {2i + a} instead of
    // {32a + i}.  Halves the LDS instructions of the weight-gradient kernels; only the output row/column map changes.
    constexpr bool A2 = (AK == VD_COL) && (MT == 2);
    constexpr bool B2 = (BK != VD_ROW) && (NT == 2);
    typedef float f32x2 __attribute__((ext_vector_type(2)));

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int wm = (wave >> 1) * (BM / 2), wn = (wave & 1) * (BN / 2);
    // XCD-aware tile order.  Workgroups are dealt round-robin over the 8 XCDs (linear id % 8), each with its own 4 MiB L2.
    // Remapping so that every XCD walks a CONTIGUOUS range of (m-tile, n-tile) pairs puts the n-tiles of one m-tile
    // (same A rows) and consecutive m-tiles (shared conv halo rows) on one L2, instead of fetching them through the
    // fabric once per XCD.  Pure speed/traffic choice: any placement is correct (bijective map, guide T1).
    // (the z index -- split-K slab or batch entry -- is the slowest: all tiles of one slab / batch entry share their
    // K range or their operands, so they are kept on one XCD too)
    int tbx = blockIdx.x, tby = blockIdx.y, tbz = blockIdx.z;
    {
        const unsigned T = gridDim.x * gridDim.y * gridDim.z;
        const unsigned lin = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        if (T >= 16) {
            const unsigned q = T / 8, r = T % 8, xcd = lin % 8, slot = lin / 8;
            const unsigned t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
            tbx = t % gridDim.x; tby = (t / gridDim.x) % gridDim.y; tbz = t / (gridDim.x * gridDim.y);
        }
    }
    const int m0 = tby * BM;
    int n0 = tbx * BN;
    int tapN = 0, ci0 = 0;
    if (BK == VD_IM2COL) {
        const int cchN = (p.Cin + BN - 1) / BN;
        tapN = tbx / cchN;
        ci0 = (tbx % cchN) * BN;
        n0 = tapN * p.Cin + ci0;
    }
    const float* A = p.A;
    const float* B = p.B;
    float* C = p.C;
    const float* R = p.R;
    const float* biasp = p.bias;
    int kt_begin = 0, kt_end = p.kt_total;
    if (SPLITK) {
        int slab = tbz;
        if constexpr (GROUPED) {
            const int e = tbz / p.group_S;
            slab = tbz - e * p.group_S;
            A = gp.A[e];
            B = gp.B[e];
        }
        kt_begin = slab * p.kt_per_split;
        kt_end = min(kt_begin + p.kt_per_split, p.kt_total);
        C += (long long)tbz * p.slab_stride;
    } else {
        const int zb = tbz / p.nh, zh = tbz % p.nh;
        A += zb * p.sAb + zh * p.sAh;
        B += zb * p.sBb + zh * p.sBh;
        C += zb * p.sCb + zh * p.sCh;
        if (R) R += zb * p.sRb + zh * p.sRh;
        if (biasp) biasp += zb * p.sBias;
    }

    // ---- constant per-thread source offsets (bytes) of every piece this thread takes part in
    unsigned voA[AIT], voB[BIT];
    unsigned mkA[AIT];                 // IM2COL A: bit t set <=> tap t of this row is inside the image
    int kcA[AIT], kcB[BIT];            // k-contiguous tiles: first k (floats) of the 16-byte chunk this lane fetches
    int by[BIT], bx[BIT];              // IM2COL B: image coordinates of this lane's pixel row
    // K tiles of the im2col A operand are ordered (channel slab, tap) with the TAP as the fast index: the nine shifted
    // reads of one 32-channel slab of the block's pixel window are adjacent in time, so eight of them hit L1/L2 instead of
    // going back to the fabric (rocprof FETCH_SIZE of the 256->256 @32x32 conv: see profiles/)
    int tapA = (AK == VD_IM2COL) ? kt_begin % 9 : 0, ccA = (AK == VD_IM2COL) ? kt_begin / 9 : 0;
    const int kinc_x = (BK == VD_IM2COL) ? KT % p.W : 0, kinc_y = (BK == VD_IM2COL) ? KT / p.W : 0;
#pragma unroll
    for (int j = 0; j < AIT; ++j) {
        const int q = j * 4 + wave;
        if (AK == VD_COL) {
            const int kk = q * A_RPP + lane / A_LPR, cm = (lane % A_LPR) * 4;
            const int rowoff = (GROUPED && p.a_kblk) ? (kk >> 4) * (int)p.a_kblk + (kk & 15) * (int)p.lda : kk * (int)p.lda;
            voA[j] = (m0 + cm < p.M) ? (unsigned)(rowoff + cm) * 4u : OOB;
            kcA[j] = kk; mkA[j] = 0;
        } else {
            const int row = q * RRPP + lane / RLPR;
            const int c = (lane % RLPR) ^ swz_of<KT>(row);
            const int m = m0 + row;
            kcA[j] = c * 4;
            voA[j] = (m < p.M) ? (unsigned)(row * (int)p.lda + c * 4) * 4u : OOB;      // < 2^31 by use_dma()
            unsigned mk = 0;
            if (AK == VD_IM2COL) {
                int y, x;
                pix_yx(p, m, y, x);
                const unsigned xm = (x > 0 ? 1u : 0u) | 2u | (x < p.W - 1 ? 4u : 0u);          // taps dx = -1, 0, +1 inside the row
                mk = (y > 0 ? xm : 0u) | (xm << 3) | (y < p.H - 1 ? xm << 6 : 0u);             // bit t = tap t inside the image
            }
            mkA[j] = mk;
        }
    }
#pragma unroll
    for (int j = 0; j < BIT; ++j) {
        const int q = j * 4 + wave;
        if (BK == VD_ROW) {
            const int row = q * RRPP + lane / RLPR;
            const int c = (lane % RLPR) ^ swz_of<KT>(row);
            kcB[j] = c * 4;
            voB[j] = (n0 + row < p.N) ? (unsigned)(row * (int)p.ldb + c * 4) * 4u : OOB;
            by[j] = bx[j] = 0;
        } else {
            const int kk = q * B_RPP + lane / B_LPR, cn = (lane % B_LPR) * 4;
            kcB[j] = kk;
            const bool ok = (BK == VD_IM2COL) ? (ci0 + cn < p.Cin) : (n0 + cn < p.N);
            const int rowoff = (GROUPED && BK == VD_COL && p.b_kblk) ? (kk >> 4) * (int)p.b_kblk + (kk & 15) * (int)p.ldb : kk * (int)p.ldb;
            voB[j] = ok ? (unsigned)(rowoff + cn) * 4u : OOB;
            if (BK == VD_IM2COL) pix_yx(p, kt_begin * KT + kk, by[j], bx[j]);
            else by[j] = bx[j] = 0;
        }
    }
    // block-constant parts of the source addresses
    const float* Ablk = (AK == VD_COL) ? A + m0 : A + (long long)m0 * p.lda;
    const float* Bblk = (BK == VD_ROW) ? B + (long long)n0 * p.ldb : ((BK == VD_COL) ? B + n0 : B + ci0);
    const long long tapoffB = (BK == VD_IM2COL) ? (long long)((tapN / 3 - 1) * p.W + (tapN % 3 - 1)) * p.ldb : 0;

    // The per-tile work is split in two: prep_tiles() does all the address / padding arithmetic of a K tile (scalar and
    // vector ALU only, no memory effect) into a handful of registers, issue_tiles() is just the (BM+BN)/32 DMA instructions.
    // prep(t+2) is called from INSIDE compute(t), so its ~55 instructions issue in the shadow of this wave's own MFMAs: a
    // wave that spends 40 % of every tile in a serial SALU chain in front of its MFMAs (loop-phase probe, VD_GEMM_PROBE=32)
    // leaves the matrix pipe idle whenever its three neighbours do the same.
    // where the DMA of tile t+1 is issued: right after the barrier (forward kinds: +1.5 % there, the loads get the whole tile
    // to land) or after the fragment reads inside the MFMA section (weight-gradient kinds: +1 %)
#ifndef VD_ISSUE_IN
#define VD_ISSUE_IN 2        /* 0 = after the barrier, 1 = inside the MFMA section, 2 = per kind (A/B builds: tests/perf_ab.py) */
#endif
    constexpr bool ISSUE_IN = VD_ISSUE_IN == 2 ? (AK == VD_COL) : (VD_ISSUE_IN == 1);
    struct Prep { const float* pA; const float* pB; unsigned vA[AIT], vB[BIT]; };   // (bases, not descriptors: the resource type cannot be a field)
    auto prep_tiles = [&](int kt, Prep& P) {
        // ------------------------------------------------ A
        {
            const int tap = tapA, c0 = ccA * KT;
            long long aoff;            // floats, block- and tile-constant
            int kwidth;
            if (AK == VD_IM2COL) { aoff = (long long)((tap / 3 - 1) * p.W + (tap % 3 - 1)) * p.lda + c0; kwidth = p.Cin - c0; }
            else if (AK == VD_ROW) { aoff = (long long)kt * KT; kwidth = p.K - kt * KT; }
            else { aoff = (GROUPED && p.a_kblk) ? (long long)kt * (KT / 16) * p.a_kblk : (long long)kt * KT * p.lda; kwidth = p.K - kt * KT; }
            P.pA = Ablk + aoff;
#pragma unroll
            for (int j = 0; j < AIT; ++j) {
                unsigned vo = voA[j];
                if (kcA[j] >= kwidth) vo = OOB;
                if (AK == VD_IM2COL && !((mkA[j] >> tap) & 1u)) vo = OOB;
                P.vA[j] = vo;
            }
        }
        // ------------------------------------------------ B
        {
            long long boff;
            int kwidth;
            if (BK == VD_ROW) {
                if (AK == VD_IM2COL) { boff = (long long)tapA * p.Cin + ccA * KT; kwidth = p.Cin - ccA * KT; }
                else { boff = (long long)kt * KT; kwidth = p.K - kt * KT; }
            } else { boff = ((GROUPED && BK == VD_COL && p.b_kblk) ? (long long)kt * (KT / 16) * p.b_kblk : (long long)kt * KT * p.ldb) + tapoffB; kwidth = p.K - kt * KT; }
            P.pB = Bblk + boff;
            const int dy = tapN / 3 - 1, dx = tapN % 3 - 1;
            if (BK == VD_IM2COL && p.lgW >= 0) {
                // power-of-two images: a piece holds B_RPP whole pixels (B_LPR lanes each), so the padding / K-range test is
                // done ONCE PER PIXEL ON THE SCALAR UNIT and applied with a single v_cndmask per piece through a lane mask
#pragma unroll
                for (int j = 0; j < BIT; ++j) {
                    const int k0 = kt * KT + (j * 4 + wave) * B_RPP;                 // first pixel of the piece (uniform)
                    unsigned long long lanes = 0;
#pragma unroll
                    for (int u = 0; u < B_RPP; ++u) {
                        const int k = k0 + u;
                        const int rem = k & ((1 << p.lgHW) - 1), y = (rem >> p.lgW) + dy, x = (rem & ((1 << p.lgW) - 1)) + dx;
                        const bool ok = k < p.K && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
                        lanes |= ok ? (((B_LPR == 64 ? 0ull : (1ull << (B_LPR & 63))) - 1ull) << (u * (B_LPR & 63))) : 0ull;
                    }
                    unsigned vo;
                    const unsigned oob = OOB;
                    asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(vo) : "v"(oob), "v"(voB[j]), "s"(lanes));
                    P.vB[j] = vo;
                }
            } else
#pragma unroll
            for (int j = 0; j < BIT; ++j) {
                unsigned vo = voB[j];
                if (kcB[j] >= kwidth) vo = OOB;
                if (BK == VD_IM2COL) {
                    if ((unsigned)(by[j] + dy) >= (unsigned)p.H || (unsigned)(bx[j] + dx) >= (unsigned)p.W) vo = OOB;
                    int nx = bx[j] + kinc_x, ny = by[j] + kinc_y;
                    if (nx >= p.W) { nx -= p.W; ++ny; }
                    while (ny >= p.H) ny -= p.H;
                    bx[j] = nx; by[j] = ny;
                }
                P.vB[j] = vo;
            }
        }
        if (AK == VD_IM2COL) { if (++tapA == 9) { tapA = 0; ++ccA; } }
    };
    auto issue_tiles = [&](int buf, const Prep& P) {
        float* as = smem + buf * (BM * KT);
        float* bs = smem + NBUF * BM * KT + buf * (BN * KT);
        const __amdgpu_buffer_rsrc_t rsA = make_rsrc(P.pA, PB(p) & 1 ? 0 : (int)OOB);   // probe: timing-only build knob
        const __amdgpu_buffer_rsrc_t rsB = make_rsrc(P.pB, PB(p) & 2 ? 0 : (int)OOB);
#pragma unroll
        for (int j = 0; j < AIT; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr_t)(as + (j * 4 + wave) * 256), 16, (int)P.vA[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < BIT; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr_t)(bs + (j * 4 + wave) * 256), 16, (int)P.vB[j], 0, 0, 0);
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // bias gradient riding on the weight-gradient GEMM: the A operand of a COL-kind launch is dY, whose column sums are
    // d(bias).  The waves of n-tile 0 / wave column 0 already stream every A fragment through registers.
    float csum[MT];
#pragma unroll
    for (int a = 0; a < MT; ++a) csum[a] = 0.f;
    const bool do_cs = (AK == VD_COL) && p.colsum != nullptr && tbx == 0 && (wave & 1) == 0;
    // timing probe (VD_GEMM_PROBE bit 4, tests/probe/stamps.py): per-workgroup 100 MHz timestamps {start, main loop done,
    // epilogue issued, stores drained} written through the colsum pointer
    const bool dbg = (PB(p) & 16) != 0;
    unsigned long long ts0 = 0, ts2 = 0;
    if (dbg) ts0 = __builtin_amdgcn_s_memrealtime();

    // fragment loads of sub-step s (8 k values): MT + NT LDS reads per wave
    auto load_frags = [&](const float* as, const float* bs, int s, f32x4 (&fa)[MT], f32x4 (&fb)[NT]) {
        if (A2) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x2 v = *reinterpret_cast<const f32x2*>(as + (8 * s + 4 * lh + j) * BM + wm + 2 * li);
                fa[0][j] = v[0]; fa[MT - 1][j] = v[1];
            }
        } else {
#pragma unroll
            for (int a = 0; a < MT; ++a) {
                const int row = wm + 32 * a + li;
                if (AK == VD_COL) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) fa[a][j] = as[(8 * s + 4 * lh + j) * BM + row];
                } else fa[a] = *reinterpret_cast<const f32x4*>(as + row_swz_t<KT>(row, 2 * s + lh));
            }
        }
        if (B2) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x2 v = *reinterpret_cast<const f32x2*>(bs + (8 * s + 4 * lh + j) * BN + wn + 2 * li);
                fb[0][j] = v[0]; fb[NT - 1][j] = v[1];
            }
        } else {
#pragma unroll
            for (int b = 0; b < NT; ++b) {
                const int row = wn + 32 * b + li;
                if (BK == VD_ROW) fb[b] = *reinterpret_cast<const f32x4*>(bs + row_swz_t<KT>(row, 2 * s + lh));
                else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) fb[b][j] = bs[(8 * s + 4 * lh + j) * BN + row];
                }
            }
        }
    };
    auto mfma_group = [&](const f32x4 (&fa)[MT], const f32x4 (&fb)[NT]) {
        if (AK == VD_COL && do_cs) {
            asm volatile("" ::: "memory");      // keep this a branch: if-converted, every wave of the launch ran the adds
#pragma unroll
            for (int a = 0; a < MT; ++a) csum[a] += (fa[a][0] + fa[a][1]) + (fa[a][2] + fa[a][3]);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int a = 0; a < MT; ++a)
#pragma unroll
                for (int b = 0; b < NT; ++b)
                    acc[a][b] = TR ? __builtin_amdgcn_mfma_f32_32x32x2f32(fb[b][j], fa[a][j], acc[a][b], 0, 0, 0)
                                   : __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a][j], fb[b][j], acc[a][b], 0, 0, 0);
    };

    // ---- split-operand form: a sub-step is 16 k (one 32x32x16 instruction deep); lane (li, lh) takes k = 16s + 8lh + j, j = 0..7
    auto load_frags8 = [&](const float* as, const float* bs, int s, float (&fa)[MT][8], float (&fb)[NT][8]) {
        if (A2) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const f32x2 v = *reinterpret_cast<const f32x2*>(as + (16 * s + 8 * lh + j) * BM + wm + 2 * li);
                fa[0][j] = v[0]; fa[MT - 1][j] = v[1];
            }
        } else {
#pragma unroll
            for (int a = 0; a < MT; ++a) {
                const int row = wm + 32 * a + li;
                if (AK == VD_COL) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) fa[a][j] = as[(16 * s + 8 * lh + j) * BM + row];
                } else {
                    const f32x4 q0 = *reinterpret_cast<const f32x4*>(as + row_swz_t<KT>(row, 4 * s + 2 * lh));
                    const f32x4 q1 = *reinterpret_cast<const f32x4*>(as + row_swz_t<KT>(row, 4 * s + 2 * lh + 1));
#pragma unroll
                    for (int j = 0; j < 4; ++j) { fa[a][j] = q0[j]; fa[a][4 + j] = q1[j]; }
                }
            }
        }
        if (B2) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const f32x2 v = *reinterpret_cast<const f32x2*>(bs + (16 * s + 8 * lh + j) * BN + wn + 2 * li);
                fb[0][j] = v[0]; fb[NT - 1][j] = v[1];
            }
        } else {
#pragma unroll
            for (int b = 0; b < NT; ++b) {
                const int row = wn + 32 * b + li;
                if (BK == VD_ROW) {
                    const f32x4 q0 = *reinterpret_cast<const f32x4*>(bs + row_swz_t<KT>(row, 4 * s + 2 * lh));
                    const f32x4 q1 = *reinterpret_cast<const f32x4*>(bs + row_swz_t<KT>(row, 4 * s + 2 * lh + 1));
#pragma unroll
                    for (int j = 0; j < 4; ++j) { fb[b][j] = q0[j]; fb[b][4 + j] = q1[j]; }
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) fb[b][j] = bs[(16 * s + 8 * lh + j) * BN + row];
                }
            }
        }
    };
    auto compute_split = [&](int buf, int kt_prep, Prep& P) {
        const float* as = smem + buf * (BM * KT);
        const float* bs = smem + NBUF * BM * KT + buf * (BN * KT);
        constexpr int S = KT / 16;
        float fa[MT][8], fb[NT][8];
        load_frags8(as, bs, 0, fa, fb);
#pragma unroll
        for (int s = 0; s < S; ++s) {
            if (s == 0) {
                if (ISSUE_IN) issue_tiles(buf ^ 1, P);
                prep_tiles(kt_prep, P);
            }
            bf16x8 ah[MT], am[MT], al[MT], bh[NT], bm[NT], bl[NT];
#pragma unroll
            for (int a = 0; a < MT; ++a) split8(fa[a], ah[a], am[a], al[a]);
#pragma unroll
            for (int b = 0; b < NT; ++b) split8(fb[b], bh[b], bm[b], bl[b]);
            if (AK == VD_COL && do_cs) {
                asm volatile("" ::: "memory");
#pragma unroll
                for (int a = 0; a < MT; ++a)
                    csum[a] += ((fa[a][0] + fa[a][1]) + (fa[a][2] + fa[a][3])) + ((fa[a][4] + fa[a][5]) + (fa[a][6] + fa[a][7]));
            }
            if (s + 1 < S) load_frags8(as, bs, s + 1, fa, fb);
#pragma unroll
            for (int a = 0; a < MT; ++a)
#pragma unroll
                for (int b = 0; b < NT; ++b) {
#define VD_MF(X, Y) acc[a][b] = TR ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(Y[b], X[a], acc[a][b], 0, 0, 0) \
                                    : __builtin_amdgcn_mfma_f32_32x32x16_bf16(X[a], Y[b], acc[a][b], 0, 0, 0)
                    VD_MF(al, bh); VD_MF(ah, bl); VD_MF(am, bm); VD_MF(am, bh); VD_MF(ah, bm); VD_MF(ah, bh);
#undef VD_MF
                }
        }
    };

    // software pipeline over the KT/8 sub-steps: the fragments of sub-step s+1 are requested BEFORE the MFMAs of
    // sub-step s, and sched_group_barrier spreads those LDS reads between the MFMAs so no wave sits on an lgkmcnt wait
    // at a sub-step boundary
    auto compute = [&](int buf, int kt_prep, Prep& P) {
        const float* as = smem + buf * (BM * KT);
        const float* bs = smem + 2 * BM * KT + buf * (BN * KT);
        constexpr int S = KT / 8;
        f32x4 fa0[MT], fb0[NT], fa1[MT], fb1[NT];
        load_frags(as, bs, 0, fa0, fb0);
#pragma unroll
        for (int s = 0; s < S; s += 2) {
            if (s + 1 < S) load_frags(as, bs, s + 1, fa1, fb1);
            if (s == 0) {
                // next tile's DMA and the next-but-one tile's address arithmetic go out between this wave's MFMAs.  Both are
                // unconditional (a branch would fence them off from the MFMAs): past the K range every offset is out of
                // range (the DMA writes zeros into the idle buffer), a split-K slab reads one tile of its neighbour; the
                // vmcnt(0) + barrier closing this iteration covers the extra DMA before the buffers can be reused or freed
                if (ISSUE_IN) issue_tiles(buf ^ 1, P);
                prep_tiles(kt_prep, P);
            }
            mfma_group(fa0, fb0);
            if (VD_SCHED_INTERLEAVE && AK != VD_COL && BK == VD_ROW) {
#pragma unroll
                for (int q = 0; q < MT + NT; ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                       // one LDS read
                    __builtin_amdgcn_sched_group_barrier(0x008, (4 * MT * NT) / (MT + NT), 0);   // a slice of the MFMAs
                }
            }
            if (s + 1 < S) {
                if (s + 2 < S) load_frags(as, bs, s + 2, fa0, fb0);
                mfma_group(fa1, fb1);
                if (VD_SCHED_INTERLEAVE && AK != VD_COL && BK == VD_ROW) {
#pragma unroll
                    for (int q = 0; q < MT + NT; ++q) {
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, (4 * MT * NT) / (MT + NT), 0);
                    }
                }
            }
        }
    };

    if (kt_begin < kt_end) {
        Prep P;
        prep_tiles(kt_begin, P);
        issue_tiles(0, P);
        prep_tiles(kt_begin + 1, P);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int buf = 0;
        if (PB(p) & 4) {           // timing probe only (wrong results): no barrier in the main loop
            for (int kt = kt_begin; kt < kt_end; ++kt) {
                if (!ISSUE_IN) issue_tiles(buf ^ 1, P);
                compute(buf, kt + 2, P);
                buf ^= 1;
            }
        } else if (PB(p) & 32) {      // timing probe: core-clock cycles this wave spends issuing DMA / computing / waiting
            unsigned long long t_dma = 0, t_cmp = 0, t_wait = 0;
            for (int kt = kt_begin; kt < kt_end; ++kt) {
                if (!ISSUE_IN) issue_tiles(buf ^ 1, P);
                const unsigned long long c1 = __builtin_amdgcn_s_memtime();
                compute(buf, kt + 2, P);
                const unsigned long long c2 = __builtin_amdgcn_s_memtime();
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const unsigned long long c2b = __builtin_amdgcn_s_memtime();
                __syncthreads();
                const unsigned long long c3 = __builtin_amdgcn_s_memtime();
                t_dma += c2b - c2; t_cmp += c2 - c1; t_wait += c3 - c2b;      // {own DMA not landed yet, LDS reads + MFMA, barrier}
                buf ^= 1;
            }
            if (lane == 0) {
                const unsigned lin = ((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 4 + wave;
                unsigned long long* o = reinterpret_cast<unsigned long long*>(p.stats) + 4ULL * lin;        // (probe: through `stats`)
                o[0] = t_dma; o[1] = t_cmp; o[2] = t_wait; o[3] = (unsigned long long)(kt_end - kt_begin);
            }
        } else
        for (int kt = kt_begin; kt < kt_end; ++kt) {
            if (!ISSUE_IN) issue_tiles(buf ^ 1, P);
            if constexpr (SPL) compute_split(buf, kt + 2, P);
            else compute(buf, kt + 2, P);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            buf ^= 1;
        }
    }

    if (dbg) ts2 = __builtin_amdgcn_s_memrealtime();
    auto stamp_end = [&]() {
        if (dbg && tid == 0) {
            const unsigned long long ts3 = __builtin_amdgcn_s_memrealtime();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned long long ts4 = __builtin_amdgcn_s_memrealtime();
            const unsigned lin = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
            unsigned long long* o = reinterpret_cast<unsigned long long*>(p.colsum) + 4ULL * lin;
            o[0] = ts0; o[1] = ts2; o[2] = ts3; o[3] = ts4;
        }
    };
    if (AK == VD_COL && do_cs) {
#pragma unroll
        for (int a = 0; a < MT; ++a) {
            const float v = csum[a] + __shfl_xor(csum[a], 32, 64);
            const int m = m0 + wm + (A2 ? 2 * li + a : 32 * a + li);
            if (lh == 0 && m < p.M) {
                float* o = p.colsum + (SPLITK ? (long long)tbz * p.M : 0) + m;
                *o = (!SPLITK && p.colsum_accumulate) ? *o + v : v;
            }
        }
    }
    // ---- epilogue.  D[row][col] of a 32x32 MFMA block: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
    // Every output / residual access is a buffer instruction: descriptor = block base, per-lane 32-bit voffset computed ONCE
    // per column block, the row walks in the SCALAR offset.  The wave shares its SIMD with three waves issuing MFMAs
    // back to back, so every VALU instruction of the epilogue waits for a bubble (in-kernel stamps: the flat-pointer form,
    // ~12 VALU per element for 64-bit addresses, took 36-50 us per workgroup = 8 % of a 3x3 conv launch and 40 % of a
    // K = 256 GEMM).  Rows >= M fall outside num_records and are dropped by the hardware; invalid columns get voffset = OOB.
    const int rows_valid = min(BM, p.M - m0);
    const int ldc4 = (int)p.ldc * 4, ldr4 = (int)p.ldr * 4;
    const __amdgpu_buffer_rsrc_t crs = make_rsrc(C + (long long)m0 * p.ldc + n0, rows_valid * ldc4);
    const __amdgpu_buffer_rsrc_t rrs = make_rsrc(R ? R + (long long)m0 * p.ldr + n0 : C, R ? rows_valid * ldr4 : 0);
    if (TR) {
        // lane (li, lh): row = block row li; register quad q of block b holds columns 8q + 4lh .. +3 of that block.  With
        // interleaved column ownership (B2: block b owns columns 2i + b) the four consecutive columns 16q + 8lh + 4h .. +3 of
        // the wave's 64 are {block 0, block 1} x registers {4q + 2h, 4q + 2h + 1}.  Either way 8 quads per row block.
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        const f32x4 al4 = {p.alpha, p.alpha, p.alpha, p.alpha};
        constexpr int NQ = 4 * NT;
        // the 8 column quads of a row block are handled in two halves of 4: bias / residual / offset staging for all 8 at once
        // (72 live registers beside the 64 accumulators) spilled to scratch under the 128-VGPR cap of the 4-workgroups-per-CU
        // KT = 16 forms (11-60 spilled VGPRs, tests/test_host_cpu.py::test_no_kernel_spills_to_scratch)
        constexpr int NQH = NQ > 4 ? NQ / 2 : NQ;
#pragma unroll
        for (int a = 0; a < MT; ++a) {
            const int rowl = wm + (A2 ? 2 * li + a : 32 * a + li);
#pragma unroll
            for (int e0 = 0; e0 < NQ; e0 += NQH) {
                f32x4 bias4[NQH], add4[NQH];
                unsigned vocs[NQH];
#pragma unroll
                for (int ee = 0; ee < NQH; ++ee) {
                    const int e = e0 + ee;
                    const int ncol = B2 ? wn + 16 * (e >> 1) + 8 * lh + 4 * (e & 1) : wn + 32 * (e >> 2) + 8 * (e & 3) + 4 * lh;
                    const int n = n0 + ncol;
                    const bool nok = (BK == VD_IM2COL) ? (ci0 + ncol < p.Cin) : (n < p.N);
                    vocs[ee] = nok ? (unsigned)(rowl * ldc4 + ncol * 4) : OOB;
                    const unsigned vor = nok ? (unsigned)(rowl * ldr4 + ncol * 4) : OOB;
                    bias4[ee] = (!SPLITK && biasp && nok) ? *reinterpret_cast<const f32x4*>(biasp + n) : f32x4{0.f, 0.f, 0.f, 0.f};
                    f32x4 t = {0.f, 0.f, 0.f, 0.f};
                    if (!SPLITK) {
                        if (R) t = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rrs, (int)vor, 0, 0));
                        if (p.accumulate) t += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(crs, (int)vocs[ee], 0, 0));
                    }
                    add4[ee] = t;
                }
#pragma unroll
                for (int ee = 0; ee < NQH; ++ee) {
                    const int e = e0 + ee;
                    f32x4 v;
                    if (B2) {
                        const int r0 = 4 * (e >> 1) + 2 * (e & 1);
                        v = f32x4{acc[a][0][r0], acc[a][NT - 1][r0], acc[a][0][r0 + 1], acc[a][NT - 1][r0 + 1]};
                    } else {
                        const int b = e >> 2, q = e & 3;
                        v = f32x4{acc[a][b][4 * q], acc[a][b][4 * q + 1], acc[a][b][4 * q + 2], acc[a][b][4 * q + 3]};
                    }
                    if (!SPLITK) v = (v * al4 + bias4[ee]) + add4[ee];
                    const u32x4 u = {__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
                    __builtin_amdgcn_raw_buffer_store_b128(u, crs, (int)vocs[ee], 0, 0);
                }
            }
        }
        stamp_end();
        return;
    }
    const int row_lane = wm + (A2 ? 8 : 4) * lh;
    const bool want_stats = !SPLITK && p.stats != nullptr && !(PB(p) & 32), edge = rows_valid < BM;       // both uniform
#pragma unroll
    for (int b = 0; b < NT; ++b) {
        const int ncol = wn + (B2 ? 2 * li + b : 32 * b + li);
        const int n = n0 + ncol;
        const bool nok = (BK == VD_IM2COL) ? (ci0 + ncol < p.Cin) : (n < p.N);
        const unsigned voc = nok ? (unsigned)(row_lane * ldc4 + ncol * 4) : OOB;
        const unsigned vor = nok ? (unsigned)(row_lane * ldr4 + ncol * 4) : OOB;
        const float bv = (!SPLITK && biasp && nok) ? biasp[n] : 0.f;
        float st1 = 0.f, st2 = 0.f;           // per-column sum / sum of squares of this wave's BM/2 output rows
        f32x2 st1p = {0.f, 0.f}, st2p = {0.f, 0.f};
#pragma unroll
        for (int a = 0; a < MT; ++a) {
            // residual / accumulate operands of the whole 32x32 block are fetched BEFORE its first store (C may alias R,
            // and is its own input when accumulating)
            float addv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rs = A2 ? 2 * ((r & 3) + 8 * (r >> 2)) + a : 32 * a + (r & 3) + 8 * (r >> 2);   // uniform row part
                float t = 0.f;
                if (!SPLITK) {
                    if (R) t = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rrs, (int)vor, rs * ldr4, 0));
                    if (p.accumulate) t += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(crs, (int)voc, rs * ldc4, 0));
                }
                addv[r] = t;
            }
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                // two rows at a time in packed fp32 (v_pk_fma_f32 / v_pk_add_f32): half the VALU instructions
                const int rs0 = A2 ? 2 * ((r & 3) + 8 * (r >> 2)) + a : 32 * a + (r & 3) + 8 * (r >> 2);
                const int rs1 = A2 ? rs0 + 2 : rs0 + 1;                               // r + 1 stays inside the same group of four
                f32x2 v = {acc[a][b][r], acc[a][b][r + 1]};
                if (!SPLITK) {
                    const f32x2 al = {p.alpha, p.alpha}, bb = {bv, bv}, ad = {addv[r], addv[r + 1]};
                    v = (v * al + bb) + ad;
                    if (want_stats) {
                        if (!edge) { st1p += v; st2p += v * v; }
                        else {
                            if (rs0 + row_lane < rows_valid) { st1 += v[0]; st2 += v[0] * v[0]; }
                            if (rs1 + row_lane < rows_valid) { st1 += v[1]; st2 += v[1] * v[1]; }
                        }
                    }
                }
                const float v0 = v[0], v1 = v[1];       // (bit_cast straight from a vector element stores element 0 twice: hipcc 7.2)
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v0), crs, (int)voc, rs0 * ldc4, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v1), crs, (int)voc, rs1 * ldc4, 0);
            }
        }
        st1 += st1p[0] + st1p[1];
        st2 += st2p[0] + st2p[1];
        // GroupNorm statistics of the tensor being written, for the norm that consumes it next (saves that norm's read
        // pass): the two lane halves hold complementary rows of the wave's BM/2-row slab, which lies inside one image
        if (want_stats && nok) {
            st1 += __shfl_xor(st1, 32, 64);
            st2 += __shfl_xor(st2, 32, 64);
            const int mrow = m0 + wm;
            if (lh == 0 && mrow < p.M) {
                const int chunks = p.stats_hw / (BM / 2);
                float* o = p.stats + ((long long)(mrow / p.stats_hw) * chunks + (mrow % p.stats_hw) / (BM / 2)) * 2 * p.N + n;
                o[0] = st1; o[p.N] = st2;
            }
        }
    }
    stamp_end();
}

template <int BM, int BN, int AK, int BK, bool SPLITK, int KT, bool TR, bool GROUPED = false>
__global__ __launch_bounds__(256, (KT == 16 ? VD_KT16_BLOCKS : 2)) void gemm_dma_kernel(const GemmArgs p,
                                                                                          const std::conditional_t<GROUPED, GroupPtrs, NoGroup> gp) {
    gemm_dma_body<BM, BN, AK, BK, SPLITK, KT, TR, GROUPED, false>(p, gp);
}
// the split-operand forms (VD_GEMM_SPLIT=1): same tiles, same LDS-DMA staging, same epilogues, 16-bit MFMA on three bf16 pieces per operand
template <int BM, int BN, int AK, int BK, bool SPLITK, int KT, bool TR, bool GROUPED = false>
__global__ __launch_bounds__(256, (KT == 16 ? VD_SPL_BLOCKS : 2)) void gemm_split_kernel(const GemmArgs p,
                                                                                           const std::conditional_t<GROUPED, GroupPtrs, NoGroup> gp) {
    gemm_dma_body<BM, BN, AK, BK, SPLITK, KT, TR, GROUPED, true>(p, gp);
}

// ---- 256x256 tiles for the grouped split-K weight-gradient launches (round 6; round-5 review item 3 i): C[e][m][n] = sum_k A_e[k][m] B_e[k][n]
// with both operands row-contiguous ([k][rows]) -- the 36 planes of the F(4x4,3x3) weight gradient and the grouped 1x1 / linear weight
// gradients whose M and N are multiples of 256.  Against the 128x128 forms of gemm_split_kernel: every operand element is read (and
// split into its three bf16 pieces) ONCE per plane instead of twice, a wave's 64x128 tile carries 48 MFMAs per K tile of 16 (a barrier and
// 4 LDS-DMA pieces per wave every 1 536 matrix cycles instead of every 768), and 6 split8 per 48 MFMAs is 5.5 plain vector instructions per
// MFMA gap -- what hides beside v_mfma_f32_32x32x16_bf16 (tests/probe/mfma_bf16_fill.hip: up to ~5 per gap are free, 8 cost 15 cycles).
// Workgroup = 8 waves (4 along m x 2 along n), one per CU; three LDS stages of (16 k x 256 m + 16 k x 256 n) fp32 = 96 KB, one barrier per K
// tile, the DMA two tiles ahead.  Fragments: A rows interleaved by 2 (ds_read_b64: MFMA block a of a wave owns m = 2 i + a), B columns by 4
// (ds_read_b128: block b owns n = 4 i + b) -- so a lane's four accumulator registers r of the four B blocks are four CONSECUTIVE columns of
// one output row: the slab leaves as dwordx4 stores.  Same arithmetic as gemm_split_kernel (six bf16 piece products, fp32 accumulation in
// the same k order inside a slab), same slab layout, same colsum partials: the reduce kernels and callers are unchanged.
__global__ __launch_bounds__(512) void wgrad_planes256_kernel(const GemmArgs p, const GroupPtrs gp) {
    constexpr int BM = 256, BN = 256, KTT = 16, NBUF = 3, STG = (BM + BN) * KTT;      // floats per stage
    __shared__ __attribute__((aligned(1024))) float smem[NBUF * STG];
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 128;
    const int tbx = blockIdx.x, tby = blockIdx.y, tbz = blockIdx.z;
    const int m0 = tby * BM, n0 = tbx * BN;
    const int e = tbz / p.group_S, slab = tbz - e * p.group_S;
    const float* A = gp.A[e] + m0;
    const float* B = gp.B[e] + n0;
    const int kt_begin = slab * p.kt_per_split, kt_end = min(kt_begin + p.kt_per_split, p.kt_total);
    float* C = p.C + (long long)tbz * p.slab_stride;

    // DMA plan: a K tile is 16 rows of A and 16 rows of B, 1 KiB each; wave w copies rows w and w + 8 of both (lane l: floats 4l .. 4l+3)
    auto issue = [&](int kt, int buf) {
        float* as = smem + buf * STG;
        float* bs = as + BM * KTT;
        // the K tile's base goes into the descriptor (64-bit, scalar unit); past the slab's range the descriptor is empty: the DMA writes
        // zeros into a stage nobody reads
        const int rec = kt < kt_end ? (int)OOB : 0;
        const __amdgpu_buffer_rsrc_t rsA = make_rsrc(A + (p.a_kblk ? (long long)kt * p.a_kblk : (long long)kt * KTT * p.lda), rec);
        const __amdgpu_buffer_rsrc_t rsB = make_rsrc(B + (p.b_kblk ? (long long)kt * p.b_kblk : (long long)kt * KTT * p.ldb), rec);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int kk = wave + 8 * h;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr_t)(as + kk * BM), 16, lane * 16, kk * (int)p.lda * 4, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr_t)(bs + kk * BN), 16, lane * 16, kk * (int)p.ldb * 4, 0, 0);
        }
    };

    f32x16 acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    float csum[2] = {0.f, 0.f};
    const bool do_cs = p.colsum != nullptr && tbx == 0 && (wave & 1) == 0;

    if (kt_begin < kt_end) {
        issue(kt_begin, 0);
        issue(kt_begin + 1, 1);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");            // tile kt_begin has landed (this wave's four newest pieces belong to the next one)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        int buf = 0;
        for (int kt = kt_begin; kt < kt_end; ++kt) {
            const float* as = smem + buf * STG;
            const float* bs = as + BM * KTT;
            // stage (buf + 2) % 3 held tile kt - 1: every wave finished reading it before the barrier that closed the previous iteration
            issue(kt + 2, buf >= 1 ? buf - 1 : 2);
            float fa[2][8];
            f32x4 fbq[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const f32x2 v = *reinterpret_cast<const f32x2*>(as + (8 * lh + j) * BM + wm + 2 * li);
                fa[0][j] = v[0]; fa[1][j] = v[1];
                fbq[j] = *reinterpret_cast<const f32x4*>(bs + (8 * lh + j) * BN + wn + 4 * li);
            }
            bf16x8 ah[2], am[2], al[2];
#pragma unroll
            for (int a = 0; a < 2; ++a) split8(fa[a], ah[a], am[a], al[a]);
            if (do_cs) {
                asm volatile("" ::: "memory");      // keep this a branch: if-converted, every wave of the launch ran the adds
#pragma unroll
                for (int a = 0; a < 2; ++a)
                    csum[a] += ((fa[a][0] + fa[a][1]) + (fa[a][2] + fa[a][3])) + ((fa[a][4] + fa[a][5]) + (fa[a][6] + fa[a][7]));
            }
#pragma unroll
            for (int hb = 0; hb < 2; ++hb) {        // B blocks two at a time: 24 registers of pieces live instead of 48
                bf16x8 bh[2], bm[2], bl[2];
#pragma unroll
                for (int b2 = 0; b2 < 2; ++b2) {
                    float fb[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) fb[j] = fbq[j][2 * hb + b2];
                    split8(fb, bh[b2], bm[b2], bl[b2]);
                }
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b2 = 0; b2 < 2; ++b2) {
                        f32x16& d = acc[a][2 * hb + b2];
                        // (operands swapped as in the TR forms of the tile engine: D[i][j], i = B row = column n, j = A row = m)
                        d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl[b2], ah[a], d, 0, 0, 0);
                        d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[b2], al[a], d, 0, 0, 0);
                        d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bm[b2], am[a], d, 0, 0, 0);
                        d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[b2], am[a], d, 0, 0, 0);
                        d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bm[b2], ah[a], d, 0, 0, 0);
                        d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[b2], ah[a], d, 0, 0, 0);
                    }
            }
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");        // tile kt + 1 has landed; tile kt + 2 may still be in flight
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            buf = buf == 2 ? 0 : buf + 1;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // (the zero-fill DMAs past the range: nothing may land after the workgroup is gone)

    if (do_cs) {
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const float v = csum[a] + __shfl_xor(csum[a], 32, 64);
            if (lh == 0) p.colsum[(long long)tbz * p.M + m0 + wm + 2 * li + a] = v;
        }
    }
    // epilogue: D[i][j] of block (a, b): j = lane & 31 -> m = wm + 2 j + a; i = (r & 3) + 8 (r >> 2) + 4 lh -> n = wn + 4 i + b
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t crs = make_rsrc(C + (long long)m0 * p.N + n0);
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const unsigned rowb = (unsigned)(wm + 2 * li + a) * (unsigned)p.N * 4u;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = (r & 3) + 8 * (r >> 2) + 4 * lh;
            const u32x4 u = {__float_as_uint(acc[a][0][r]), __float_as_uint(acc[a][1][r]), __float_as_uint(acc[a][2][r]), __float_as_uint(acc[a][3][r])};
            __builtin_amdgcn_raw_buffer_store_b128(u, crs, (int)(rowb + (unsigned)(wn + 4 * i) * 4u), 0, 0);
        }
    }
}

// out (+)= sum over slabs; optional OIHW transposition for the conv weight gradient
__global__ void reduce_slabs_kernel(const float* slabs, int S, long long slab_stride, int M, int N, float* out,
                                    long long ldo, int accumulate, float alpha, const float* cpart, float* colsum,
                                    int colsum_accumulate) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (colsum && idx < M) {            // bias gradient partials of the same launch
        float c = 0.f;
        for (int z = 0; z < S; ++z) c += cpart[(long long)z * M + idx];
        colsum[idx] = colsum_accumulate ? colsum[idx] + c : c;
    }
    if (idx >= (long long)M * N) return;
    float s = 0.f;
    for (int z = 0; z < S; ++z) s += slabs[z * slab_stride + idx];
    const int m = idx / N, n = idx % N;
    float* o = out + (long long)m * ldo + n;
    s *= alpha;
    *o = accumulate ? *o + s : s;
}

// grouped form: grid.y = entry; entry e sums its S slabs (fixed order) into outs.C[e] and its bias-gradient partials into outs.cs[e]
__global__ void reduce_slabs_grouped_kernel(const float* slabs, int S, long long slab_stride, int M, int N, const GroupOut outs,
                                            long long ldo, const float* cpart) {
    const int e = blockIdx.y;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const float* sl = slabs + (long long)e * S * slab_stride;
    float* cs = outs.cs[e];
    if (cs && idx < M) {
        float c = 0.f;
        for (int z = 0; z < S; ++z) c += cpart[((long long)e * S + z) * M + idx];
        cs[idx] = c;
    }
    if (idx >= (long long)M * N) return;
    float s = 0.f;
    for (int z = 0; z < S; ++z) s += sl[z * slab_stride + idx];
    const int m = idx / N, n = idx % N;
    outs.C[e][(long long)m * ldo + n] = s;
}

__global__ void reduce_slabs_oihw_kernel(const float* slabs, int S, long long slab_stride, int Cout, int Cin,
                                         int Cout_w, int Cin_w, float* dw, int accumulate, const float* cpart, float* dbias) {
    // slab element (co, tap*Cin + ci) -> dw[(co*Cin_w + ci)*9 + tap]
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (dbias && idx < Cout_w) {
        float c = 0.f;
        for (int z = 0; z < S; ++z) c += cpart[(long long)z * Cout + idx];
        dbias[idx] = accumulate ? dbias[idx] + c : c;
    }
    const long long total = (long long)Cout * 9 * Cin;
    if (idx >= total) return;
    const int co = idx / (9 * Cin), rem = idx % (9 * Cin);
    const int tap = rem / Cin, ci = rem % Cin;
    if (co >= Cout_w || ci >= Cin_w) return;
    float s = 0.f;
    for (int z = 0; z < S; ++z) s += slabs[z * slab_stride + idx];
    float* o = dw + ((long long)co * Cin_w + ci) * 9 + tap;
    *o = accumulate ? *o + s : s;
}

__global__ void pack_conv3x3_kernel(const float* w, int Cout_w, int Cin_w, float* wf, int Cin_p, float* wd, int Cout_p) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (wf) {   // wf[co][tap][ci_p]
        const long long tot = (long long)Cout_w * 9 * Cin_p;
        if (idx < tot) {
            const int co = idx / (9 * Cin_p), rem = idx % (9 * Cin_p);
            const int tap = rem / Cin_p, ci = rem % Cin_p;
            wf[idx] = (ci < Cin_w) ? w[((long long)co * Cin_w + ci) * 9 + tap] : 0.f;
        }
    }
    if (wd) {   // wd[ci][tap][co_p] = w[co][ci][8 - tap]
        const long long tot = (long long)Cin_w * 9 * Cout_p;
        if (idx < tot) {
            const int ci = idx / (9 * Cout_p), rem = idx % (9 * Cout_p);
            const int tap = rem / Cout_p, co = rem % Cout_p;
            wd[idx] = (co < Cout_w) ? w[((long long)co * Cin_w + ci) * 9 + (8 - tap)] : 0.f;
        }
    }
}


// all 3x3 kernels of a network in ONE launch (the per-tensor form costs ~110 launches of ~6 us per training step).
// items: device table, 8 x int64 per tensor: {w, wf, wd, Cout, Cin, Cin_p, Cout_p, first block}; blocks of 256 elements.
__global__ void pack_conv3x3_batched_kernel(const long long* items, int n) {
    int lo = 0, hi = n - 1;                                  // last item whose first block <= blockIdx.x
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (items[8 * mid + 7] <= (long long)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const long long* it = items + 8 * lo;
    const float* w = reinterpret_cast<const float*>(it[0]);
    float* wf = reinterpret_cast<float*>(it[1]);
    float* wd = reinterpret_cast<float*>(it[2]);
    const int Cout_w = (int)it[3], Cin_w = (int)it[4], Cin_p = (int)it[5], Cout_p = (int)it[6];
    const long long idx = ((long long)blockIdx.x - it[7]) * blockDim.x + threadIdx.x;
    if (wf) {
        const long long tot = (long long)Cout_w * 9 * Cin_p;
        if (idx < tot) {
            const int co = idx / (9 * Cin_p), rem = idx % (9 * Cin_p);
            const int tap = rem / Cin_p, ci = rem % Cin_p;
            wf[idx] = (ci < Cin_w) ? w[((long long)co * Cin_w + ci) * 9 + tap] : 0.f;
        }
    }
    if (wd) {
        const long long tot = (long long)Cin_w * 9 * Cout_p;
        if (idx < tot) {
            const int ci = idx / (9 * Cout_p), rem = idx % (9 * Cout_p);
            const int tap = rem / Cout_p, co = rem % Cout_p;
            wd[idx] = (co < Cout_w) ? w[((long long)co * Cin_w + ci) * 9 + (8 - tap)] : 0.f;
        }
    }
}


// operands inside the LDS-DMA kernel's range: 32-bit byte offsets inside one block tile must stay below the descriptor range
bool dma_in_range(const GemmArgs& a) {
    const long long lim = 0x70000000LL / 4;
    return 128LL * a.lda + 128 < lim && 128LL * a.ldb + 128 < lim && 128LL * a.ldc + 128 < lim && 128LL * a.ldr + 128 < lim &&
           vd_aligned16(a.A) && vd_aligned16(a.B);
}
bool use_dma(const GemmArgs& a) {
    static const bool legacy = getenv("VD_GEMM_LEGACY") != nullptr;       // forces the register-staged kernel (tests, A/B profiling)
    return !legacy && dma_in_range(a);
}

// ---- block-tile menu.  Rectangular tiles exist for channel counts that are not multiples of 128 (CelebA: 192, 576, 960,
// 1344): a 192-wide N on 128-wide tiles would burn 25 % of the MFMA work on zero padding.
struct TileCfg { int bm, bn; double eff; int per_cu; };
const TileCfg TILES[4] = {{128, 128, 1.00, 2}, {128, 64, 0.92, 3}, {64, 128, 0.92, 3}, {64, 64, 0.80, 5}};

// cheapest tile = least (padded MFMA work / per-tile efficiency), with a penalty when the launch cannot give every CU
// at least two workgroups
int choose_tile(long long M, long long Ncols, bool wgrad, long long zcount, int forced) {
    static const char* env = getenv("VD_GEMM_TILE");        // experiments only
    if (env && forced == 0) forced = atoi(env);
    if (forced == 128) return 0;
    if (forced == 12864) return 1;
    if (forced == 64128) return 2;
    if (forced == 64) return 3;
    int best = 0;
    double best_cost = 1e300;
    for (int t = 0; t < 4; ++t) {
        const long long nm = (M + TILES[t].bm - 1) / TILES[t].bm, nn = (Ncols + TILES[t].bn - 1) / TILES[t].bn;
        const double blocks = (double)nm * nn * (wgrad ? 9 : 1) * zcount;
        double cost = (double)(nm * TILES[t].bm) * (double)(nn * TILES[t].bn) / TILES[t].eff;
        if (blocks < 512.0) cost *= sqrt(512.0 / blocks);
        if (cost < best_cost) { best_cost = cost; best = t; }
    }
    return best;
}

// K tile of the kernel that will run.  The 128x128 LDS-DMA kernel exists with KT = 32 (64 KB LDS, 2 workgroups per CU)
// and KT = 16 (32 KB LDS, ~106 VGPRs: 4 per CU): the deeper occupancy wins (+2-3 %) only when the launch has enough
// workgroups to give every CU four of them (>= 1024); short launches keep the longer K tile.  `wide` = the caller
// (conv weight gradient with >= 64 Ki pixels) sized its split-K slabs for 4 workgroups per CU.
inline int ktile_for(const GemmArgs& a, int t, long long nblocks, bool splitk, bool wide, bool wgrad_unsplit = false) {
    static const char* force = getenv("VD_GEMM_KT");
    if (t != 0 || !use_dma(a)) return KT;
    if (wgrad_unsplit) return KT;                  // (no KT = 16 build of that form, see launch<>)
    if (force) return atoi(force) == 16 ? 16 : 32;
    if (splitk) return wide ? 16 : 32;
    return nblocks >= 1024 ? 16 : 32;           // 1024 workgroups = four per CU, all resident (measured: +0.7 % on the sampler)
}

// conv weight gradients with at least this many pixels (K of the GEMM) run the KT = 16 kernel, 4 workgroups per CU
// (same-box A/B, tests/perf_ab.py: +2-3 % at 128 Ki pixels, +2 % at 32 Ki, -20 % at 8 Ki)
#ifndef VD_WGRAD_WIDE
#define VD_WGRAD_WIDE 32768
#endif
constexpr long long WGRAD_WIDE_PIXELS = VD_WGRAD_WIDE;

// transposed-accumulator epilogue (dwordx4 stores): whenever the launch does not want output statistics and every row is
// 16-byte addressable
bool use_tr(const GemmArgs& a, bool wgrad) {
    static const char* env = getenv("VD_GEMM_TR");               // A/B switch: 0 disables
    if (env && atoi(env) == 0) return false;
    if (a.stats && !(a.probe & 32)) return false;
    const long long ncols = wgrad ? a.Cin : a.N;
    if (ncols % 4 || a.ldc % 4 || !vd_aligned16(a.C)) return false;
    if (a.R && (a.ldr % 4 || !vd_aligned16(a.R) || a.sRb % 4 || a.sRh % 4)) return false;
    if (a.bias && (!vd_aligned16(a.bias) || a.sBias % 4)) return false;
    return a.sCb % 4 == 0 && a.sCh % 4 == 0 && a.slab_stride % 4 == 0;
}

// VD_GEMM_SPLIT (read once): 1 (default since round 5) = the split-operand forms (SPL above) wherever they are built, 0 = fp32 MFMA everywhere
static bool split_forms() {
    static const int v = [] { const char* e = getenv("VD_GEMM_SPLIT"); return e ? atoi(e) : VD_GEMM_SPLIT_DEFAULT; }();
    return v != 0;
}
}  // namespace
extern "C" int vd_gemm_split_forms(void) { return split_forms() ? 1 : 0; }
namespace {
template <int BM, int BN, int AK, int BK, bool SPLITK>
void launch(const GemmArgs& a, dim3 grid, hipStream_t st, int ktile) {
    // the only tile with a KT = 16 instantiation; an unsplit conv weight gradient never takes it (ktile_for): those two
    // instantiations carried 11 spilled VGPRs, so they are not built at all
    constexpr bool has16 = BM == 128 && BN == 128 && !(BK == VD_IM2COL && !SPLITK);
    const bool k16 = has16 && ktile == 16;
    const bool tr = use_dma(a) && use_tr(a, BK == VD_IM2COL);
    // split-operand forms (SPL): the 128-row tiles of every operand kind without im2col addressing
    constexpr bool has_spl = BM == 128 && AK != VD_IM2COL && BK != VD_IM2COL;
    const bool spl = has_spl && use_dma(a) && split_forms();
    vd_g_last_tile = ((((tr ? 1 : 0) * 100 + (spl ? 200 : 0) + (use_dma(a) ? (k16 ? 16 : 32) : 0)) * 1000) + BM) * 1000 + BN;
    if constexpr (has_spl) {
        if (spl) {
            if (k16 && tr) hipLaunchKernelGGL((gemm_split_kernel<BM, BN, AK, BK, SPLITK, (has16 ? 16 : 32), true>), grid, dim3(256), 0, st, a, NoGroup{});
            else if (k16) hipLaunchKernelGGL((gemm_split_kernel<BM, BN, AK, BK, SPLITK, (has16 ? 16 : 32), false>), grid, dim3(256), 0, st, a, NoGroup{});
            else if (tr) hipLaunchKernelGGL((gemm_split_kernel<BM, BN, AK, BK, SPLITK, 32, true>), grid, dim3(256), 0, st, a, NoGroup{});
            else hipLaunchKernelGGL((gemm_split_kernel<BM, BN, AK, BK, SPLITK, 32, false>), grid, dim3(256), 0, st, a, NoGroup{});
            return;
        }
    }
    if (!use_dma(a)) hipLaunchKernelGGL((gemm_kernel<BM, BN, AK, BK, SPLITK>), grid, dim3(256), 0, st, a);
    else if (k16 && tr) hipLaunchKernelGGL((gemm_dma_kernel<BM, BN, AK, BK, SPLITK, (has16 ? 16 : 32), true>), grid, dim3(256), 0, st, a, NoGroup{});
    else if (k16) hipLaunchKernelGGL((gemm_dma_kernel<BM, BN, AK, BK, SPLITK, (has16 ? 16 : 32), false>), grid, dim3(256), 0, st, a, NoGroup{});
    else if (tr) hipLaunchKernelGGL((gemm_dma_kernel<BM, BN, AK, BK, SPLITK, 32, true>), grid, dim3(256), 0, st, a, NoGroup{});
    else hipLaunchKernelGGL((gemm_dma_kernel<BM, BN, AK, BK, SPLITK, 32, false>), grid, dim3(256), 0, st, a, NoGroup{});
}

template <int AK, int BK, bool SPLITK>
void launch_tile2(int t, const GemmArgs& a, dim3 grid, hipStream_t st, int ktile) {
    if (t == 0) launch<128, 128, AK, BK, SPLITK>(a, grid, st, ktile);
    else if (t == 1) launch<128, 64, AK, BK, SPLITK>(a, grid, st, ktile);
    else if (t == 2) launch<64, 128, AK, BK, SPLITK>(a, grid, st, ktile);
    else launch<64, 64, AK, BK, SPLITK>(a, grid, st, ktile);
}

template <int AK, int BK>
void launch_tile(int t, bool splitk, const GemmArgs& a, dim3 grid, hipStream_t st, int ktile) {
    if (splitk) launch_tile2<AK, BK, true>(t, a, grid, st, ktile);
    else launch_tile2<AK, BK, false>(t, a, grid, st, ktile);
}

int run_gemm(const vd_gemm_desc& d, hipStream_t st) {
    VD_REQUIRE(d.M > 0 && d.N > 0 && d.K > 0, "vd_gemm: empty problem M=%d N=%d K=%d", d.M, d.N, d.K);
    VD_REQUIRE(d.A && d.B && d.C, "vd_gemm: null operand");
    VD_REQUIRE(vd_aligned16(d.A) && vd_aligned16(d.B), "vd_gemm: A/B must be 16-byte aligned");
    VD_REQUIRE(d.lda % 4 == 0 && d.ldb % 4 == 0, "vd_gemm: lda/ldb must be multiples of 4 floats (lda=%lld ldb=%lld)",
               (long long)d.lda, (long long)d.ldb);
    const int ak = d.a_kind, bk = d.b_kind;
    const bool conv = (ak == VD_IM2COL), wgrad = (bk == VD_IM2COL);
    VD_REQUIRE(!(conv && bk != VD_ROW), "vd_gemm: IM2COL A needs ROW B");
    VD_REQUIRE(!(wgrad && ak != VD_COL), "vd_gemm: IM2COL B needs COL A");
    if (ak == VD_ROW) VD_REQUIRE(d.K % 4 == 0, "vd_gemm: K %% 4 != 0 for k-contiguous A (K=%d)", d.K);
    if (ak == VD_COL) VD_REQUIRE(d.M % 4 == 0, "vd_gemm: M %% 4 != 0 for m-contiguous A (M=%d)", d.M);
    if (bk == VD_COL) VD_REQUIRE(d.N % 4 == 0, "vd_gemm: N %% 4 != 0 for n-contiguous B (N=%d)", d.N);
    if (conv || wgrad) VD_REQUIRE(d.Cin % 4 == 0 && d.H > 0 && d.W > 0, "vd_gemm: bad image geometry");
    const int batch = d.batch > 0 ? d.batch : 1;
    const int splitk = d.splitk > 1 ? d.splitk : 1;
    VD_REQUIRE(!(splitk > 1 && batch > 1), "vd_gemm: split-K and batch are exclusive");
    VD_REQUIRE(!(splitk > 1 && (d.bias || d.R)), "vd_gemm: split-K does not take bias/residual");

    GemmArgs a;
    a.A = d.A; a.B = d.B; a.C = d.C; a.bias = d.bias; a.R = d.R;
    a.M = d.M; a.N = d.N; a.K = d.K;
    a.lda = d.lda; a.ldb = d.ldb; a.ldc = d.ldc; a.ldr = d.ldr;
    a.nh = d.nh > 0 ? d.nh : 1;
    a.sAb = d.sAb; a.sAh = d.sAh; a.sBb = d.sBb; a.sBh = d.sBh; a.sCb = d.sCb; a.sCh = d.sCh; a.sRb = d.sRb; a.sRh = d.sRh;
    a.alpha = d.alpha; a.accumulate = d.accumulate;
    a.H = d.H; a.W = d.W; a.Cin = d.Cin;
    a.kt_total = 0; a.kt_per_split = 0; a.slab_stride = 0;
    a.colsum = d.colsum; a.colsum_accumulate = d.colsum_accumulate;
    a.stats = d.stats; a.stats_hw = d.stats_hw;
    a.sBias = d.sBias;
    a.lgW = a.lgHW = -1;
    if ((conv || wgrad) && (d.W & (d.W - 1)) == 0 && ((d.H * d.W) & (d.H * d.W - 1)) == 0) {
        a.lgW = __builtin_ctz((unsigned)d.W); a.lgHW = __builtin_ctz((unsigned)(d.H * d.W));
    }
    a.probe = 0;
#ifdef VD_PROBES
    { static const char* e = getenv("VD_GEMM_PROBE"); a.probe = e ? atoi(e) : 0; }
#endif
    VD_REQUIRE((a.probe & 16) || !(d.colsum && (ak != VD_COL || batch > 1)), "vd_gemm: colsum needs a COL-kind A operand and batch 1");

    const int tile = choose_tile(d.M, wgrad ? d.Cin : d.N, wgrad, (long long)batch * splitk, d.tile);
    const int tbm = TILES[tile].bm, tbn = TILES[tile].bn;
    const long long nm = (d.M + tbm - 1) / tbm;
    const long long nn = wgrad ? 9LL * ((d.Cin + tbn - 1) / tbn) : (d.N + tbn - 1) / tbn;
    VD_REQUIRE(nm <= 65535, "vd_gemm: too many row tiles (%lld)", nm);
    if (d.stats && !(a.probe & 32)) {
        VD_REQUIRE(batch == 1 && splitk == 1 && !wgrad && bk == VD_ROW && ak != VD_COL, "vd_gemm: output statistics need a plain forward launch");
        VD_REQUIRE(d.stats_hw > 0 && d.stats_hw % (tbm / 2) == 0 && d.M % d.stats_hw == 0,
                   "vd_gemm: output statistics need H*W (%d) to be a multiple of half the row tile (%d)", d.stats_hw, tbm / 2);
    }
    const int ktile = ktile_for(a, tile, nm * nn * batch, splitk > 1, wgrad && (long long)d.K >= WGRAD_WIDE_PIXELS, wgrad && splitk <= 1);
    a.kt_total = conv ? 9 * ((d.Cin + ktile - 1) / ktile) : (d.K + ktile - 1) / ktile;
    a.kt_per_split = a.kt_total;

    float* final_C = d.C;
    if (splitk > 1) {
        a.kt_per_split = (a.kt_total + splitk - 1) / splitk;
        const int used = (a.kt_total + a.kt_per_split - 1) / a.kt_per_split;
        a.slab_stride = (long long)d.M * d.N;
        VD_REQUIRE(d.ws && d.ws_bytes >= (int64_t)((used * a.slab_stride + (d.colsum ? (long long)used * d.M : 0)) * 4),
                   "vd_gemm: split-K workspace too small");
        a.C = d.ws; a.ldc = d.N;
        float* cpart = d.ws + used * a.slab_stride;          // per-slab bias-gradient partials live behind the slabs
        if (d.colsum && !(a.probe & 16)) a.colsum = cpart;      // (probe builds write timestamps through colsum)
        dim3 grid(nn, nm, used);
        if (ak == VD_COL && bk == VD_COL) launch_tile<VD_COL, VD_COL>(tile, true, a, grid, st, ktile);
        else if (ak == VD_COL && bk == VD_IM2COL) launch_tile<VD_COL, VD_IM2COL>(tile, true, a, grid, st, ktile);
        else if (ak == VD_ROW && bk == VD_ROW) launch_tile<VD_ROW, VD_ROW>(tile, true, a, grid, st, ktile);
        else if (ak == VD_ROW && bk == VD_COL) launch_tile<VD_ROW, VD_COL>(tile, true, a, grid, st, ktile);
        else VD_REQUIRE(false, "vd_gemm: split-K not built for kinds (%d,%d)", ak, bk);
        VD_LAUNCH_CHECK("gemm_kernel(splitk)");
        if (!wgrad && !(a.probe & 16)) {   // plain reduce here; the conv wgrad caller reduces with the OIHW transposition itself
            const long long tot = (long long)d.M * d.N;
            hipLaunchKernelGGL(reduce_slabs_kernel, dim3((tot + 255) / 256), dim3(256), 0, st, d.ws, used, a.slab_stride,
                               d.M, d.N, final_C, d.ldc, d.accumulate, d.alpha, cpart, d.colsum, d.colsum_accumulate);
            VD_LAUNCH_CHECK("reduce_slabs_kernel");
        }
        return 0;
    }
    dim3 grid(nn, nm, batch);
    if (ak == VD_ROW && bk == VD_ROW) launch_tile<VD_ROW, VD_ROW>(tile, false, a, grid, st, ktile);
    else if (ak == VD_ROW && bk == VD_COL) launch_tile<VD_ROW, VD_COL>(tile, false, a, grid, st, ktile);
    else if (ak == VD_COL && bk == VD_COL) launch_tile<VD_COL, VD_COL>(tile, false, a, grid, st, ktile);
    else if (ak == VD_COL && bk == VD_ROW) launch_tile<VD_COL, VD_ROW>(tile, false, a, grid, st, ktile);
    else if (ak == VD_IM2COL && bk == VD_ROW) launch_tile<VD_IM2COL, VD_ROW>(tile, false, a, grid, st, ktile);
    else if (ak == VD_COL && bk == VD_IM2COL) launch_tile<VD_COL, VD_IM2COL>(tile, false, a, grid, st, ktile);
    else VD_REQUIRE(false, "vd_gemm: unsupported operand kinds (%d,%d)", ak, bk);
    VD_LAUNCH_CHECK("gemm_kernel");
    return 0;
}

template <int BM, int BN, int KTV = 32>
void launch_grouped(const GemmArgs& a, const GroupPtrs& gp, dim3 grid, hipStream_t st) {
    const bool spl = split_forms();
    vd_g_last_tile = ((((spl ? 3 : 1) * 100 + KTV) * 1000) + BM) * 1000 + BN;
    if (spl) hipLaunchKernelGGL((gemm_split_kernel<BM, BN, VD_COL, VD_COL, true, KTV, true, true>), grid, dim3(256), 0, st, a, gp);
    else hipLaunchKernelGGL((gemm_dma_kernel<BM, BN, VD_COL, VD_COL, true, KTV, true, true>), grid, dim3(256), 0, st, a, gp);
}
// 128x128 tiles of a grouped launch with at least this many workgroups take the KT = 16 form (32 KB of LDS: four workgroups per CU)
constexpr long long GROUPED_K16_MIN_WGS = 768;
// grouped weight-gradient launches whose M and N are multiples of 256 run wgrad_planes256_kernel (split forms only; VD_PLANES256=0: A/B switch)
// when some slab count fills its one-workgroup-per-CU rounds to >= 85 % (same-box A/B, profiles/r06_planes256_ab.txt: -9 ... -10 % per launch at
// 256 -> 256 and 512 -> 256 @32x32, +-0 @16x16; the 8x8 layers -- K = 512 tiles, at most two slabs of 36 ... 648 workgroups -- lose 10-40 % to round
// quantisation and keep the 128x128 tiles, four workgroups per CU)
static bool planes256(int count, int M, int N, int K) {
    static const bool on = !(getenv("VD_PLANES256") && atoi(getenv("VD_PLANES256")) == 0);
    if (!(on && split_forms() && M % 256 == 0 && N % 256 == 0 && K % 16 == 0)) return false;
    const long long blocks = (long long)(M / 256) * (N / 256) * count, slots = vd_cu_count();
    const int kmax = K / 256 > 1 ? K / 256 : 1;
    double best = 0.0;
    for (int sl = 1; sl <= 24 && sl <= kmax; ++sl) {
        const long long wgs = blocks * sl, rounds = (wgs + slots - 1) / slots;
        const double eff = (double)wgs / (double)(rounds * slots);
        if (eff > best) best = eff;
    }
    return best >= 0.85;
}

}  // namespace

/* count same-shape weight-gradient GEMMs in one launch: C[e][M][N] = A[e]^T B[e] (A[e]: [K][M] rows of pitch lda, B[e]: [K][N] rows of
 * pitch ldb, i.e. vd_gemm with a_kind = b_kind = VD_COL), colsum[e][m] = sum_k A[e][k][m] (or NULL), split-K over `splitk` slabs per
 * entry through ws, fixed-order reduction (bitwise reproducible).  A / B / C / colsum are HOST arrays of device pointers. */
/* split-K slab count for vd_gemm_grouped_wgrad that fills whole residency rounds of the device best (all workgroups do equal work and
 * `per_cu` of them fit a CU, so a launch runs in rounds of CUs x per_cu: 576 workgroups on 512 slots take two rounds for 1.125 rounds of
 * work), with at least min_slabs slabs (shorter fp32 accumulation chains: the caller's accuracy budget) and at most max_slabs. */
extern "C" int vd_gemm_grouped_wgrad_auto_split(int32_t count, int32_t M, int32_t N, int32_t K, int32_t min_slabs, int32_t max_slabs) {
    const int tile = choose_tile(M, N, false, (long long)count * 8, 0);
    const bool p256 = planes256(count, M, N, K);
    const long long blocks = p256 ? (long long)(M / 256) * (N / 256) * count
                                  : (long long)((M + TILES[tile].bm - 1) / TILES[tile].bm) * ((N + TILES[tile].bn - 1) / TILES[tile].bn) * count;
    // (128x128 tiles: long launches run the KT = 16 form, four workgroups per CU; 256x256 tiles: one workgroup per CU)
    const long long slots = (long long)vd_cu_count() * (p256 ? 1 : (tile == 0 && blocks * (min_slabs > 1 ? min_slabs : 1) >= GROUPED_K16_MIN_WGS / 2 ? 4 : TILES[tile].per_cu));
    int lo = min_slabs > 1 ? min_slabs : 1, hi = max_slabs > lo ? max_slabs : lo;
    const int kmax = K / 256 > 1 ? K / 256 : 1;                       // at least 8 K tiles per slab
    if (hi > kmax) hi = kmax;
    if (lo > hi) lo = hi;
    int best = lo;
    double best_eff = -1.0;
    for (int s = lo; s <= hi; ++s) {
        const long long wgs = blocks * s, rounds = (wgs + slots - 1) / slots;
        double eff = (double)wgs / (double)(rounds * slots);
        eff *= 1.0 - 0.004 * s;                                       // mild preference for fewer slabs (less reduction traffic)
        if (eff > best_eff) { best_eff = eff; best = s; }
    }
    return best;
}

extern "C" size_t vd_gemm_grouped_wgrad_ws_bytes(int32_t count, int32_t M, int32_t N, int32_t splitk) {
    return (size_t)count * (splitk > 1 ? splitk : 1) * ((size_t)M * N + M) * sizeof(float);
}

/* a_kblk / b_kblk != 0: the A / B rows of every entry are stored in blocks of 16 K rows, block b of an entry at A[e] + b * a_kblk (see
 * GemmArgs): the 36 planes of the F(4x4,3x3) weight gradient interleave block by block so that its transform pass writes one contiguous run */
int vd_gemm_grouped_wgrad_kblk(const float* const* A, const float* const* B, float* const* C, float* const* colsum, int32_t count,
                               int32_t M, int32_t N, int32_t K, int64_t lda, int64_t ldb, int64_t ldc, int32_t splitk, float* ws,
                               size_t ws_bytes, void* stream, int64_t a_kblk, int64_t b_kblk, int32_t slabs_only);

extern "C" int vd_gemm_grouped_wgrad(const float* const* A, const float* const* B, float* const* C, float* const* colsum, int32_t count,
                                     int32_t M, int32_t N, int32_t K, int64_t lda, int64_t ldb, int64_t ldc, int32_t splitk, float* ws,
                                     size_t ws_bytes, void* stream) {
    return vd_gemm_grouped_wgrad_kblk(A, B, C, colsum, count, M, N, K, lda, ldb, ldc, splitk, ws, ws_bytes, stream, 0, 0, 0);
}

/* slabs a grouped launch of this shape fills (<= splitk): the launcher's own plan, for callers that reduce the slabs themselves */
int vd_gemm_grouped_wgrad_used_slabs(int32_t count, int32_t M, int32_t N, int32_t K, int32_t splitk) {
    const int S = splitk > 1 ? splitk : 1;
    const int tile = choose_tile(M, N, false, (long long)count * S, 0);
    const long long nm = (M + TILES[tile].bm - 1) / TILES[tile].bm, nn = (N + TILES[tile].bn - 1) / TILES[tile].bn;
    const bool k16 = planes256(count, M, N, K) || (tile == 0 && nm * nn * count * S >= GROUPED_K16_MIN_WGS);
    const int kt_total = k16 ? (K + 15) / 16 : (K + 31) / 32, per = (kt_total + S - 1) / S;
    return (kt_total + per - 1) / per;
}

/* slabs_only != 0: stop after the split-K launch -- slab z of entry e is left at ws + (e * used + z) * M * N, its column sums at
 * ws + count * used * M * N + (e * used + z) * M, used = vd_gemm_grouped_wgrad_used_slabs(...); C / colsum entries are not written */
int vd_gemm_grouped_wgrad_kblk(const float* const* A, const float* const* B, float* const* C, float* const* colsum, int32_t count,
                               int32_t M, int32_t N, int32_t K, int64_t lda, int64_t ldb, int64_t ldc, int32_t splitk, float* ws,
                               size_t ws_bytes, void* stream, int64_t a_kblk, int64_t b_kblk, int32_t slabs_only) {
    VD_REQUIRE(A && B && C && count > 0 && count <= VD_GROUP_MAX, "vd_gemm_grouped_wgrad: 1..%d entries (got %d)", VD_GROUP_MAX, count);
    VD_REQUIRE(M > 0 && N > 0 && K > 0 && M % 4 == 0 && N % 4 == 0 && lda % 4 == 0 && ldb % 4 == 0 && ldc >= N,
               "vd_gemm_grouped_wgrad: M, N, lda, ldb must be multiples of 4 (M=%d N=%d)", M, N);
    const int S = splitk > 1 ? splitk : 1;
    VD_REQUIRE(ws && ws_bytes >= vd_gemm_grouped_wgrad_ws_bytes(count, M, N, S), "vd_gemm_grouped_wgrad: workspace too small");
    GemmArgs a = {};
    GroupPtrs gp = {};
    GroupOut go = {};
    for (int e = 0; e < count; ++e) {
        VD_REQUIRE(A[e] && B[e] && C[e] && vd_aligned16(A[e]) && vd_aligned16(B[e]), "vd_gemm_grouped_wgrad: entry %d: null / unaligned operand", e);
        gp.A[e] = A[e]; gp.B[e] = B[e]; go.C[e] = C[e]; go.cs[e] = colsum ? colsum[e] : nullptr;
    }
    a.A = A[0]; a.B = B[0];
    a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.ldc = N; a.ldr = 0; a.nh = 1;
    a.alpha = 1.f; a.lgW = a.lgHW = -1; a.probe = 0;
    a.a_kblk = a_kblk; a.b_kblk = b_kblk;
    VD_REQUIRE(a_kblk >= 0 && b_kblk >= 0 && a_kblk % 4 == 0 && b_kblk % 4 == 0 && 2 * a_kblk + 16 * lda + 128 < 0x70000000LL / 4 &&
               2 * b_kblk + 16 * ldb + 128 < 0x70000000LL / 4, "vd_gemm_grouped_wgrad: K-block strides outside the LDS-DMA kernel's range");
    // (the grouped launch exists only on the LDS-DMA kernel: VD_GEMM_LEGACY does not apply to it)
    VD_REQUIRE(dma_in_range(a), "vd_gemm_grouped_wgrad: operands outside the LDS-DMA kernel's range (alignment / row pitch)");
    const int tile = choose_tile(M, N, false, (long long)count * S, 0);
    const int tbm = TILES[tile].bm, tbn = TILES[tile].bn;
    const bool p256 = planes256(count, M, N, K);
    const long long nm = p256 ? M / 256 : (M + tbm - 1) / tbm, nn = p256 ? N / 256 : (N + tbn - 1) / tbn;
    const bool k16 = p256 || (tile == 0 && nm * nn * count * S >= GROUPED_K16_MIN_WGS);
    a.kt_total = k16 ? (K + 15) / 16 : (K + 31) / 32;
    a.kt_per_split = (a.kt_total + S - 1) / S;
    const int used = (a.kt_total + a.kt_per_split - 1) / a.kt_per_split;      // slabs that hold work (<= S)
    a.group_S = used;
    a.slab_stride = (long long)M * N;
    a.C = ws;
    float* cpart = ws + (long long)count * used * a.slab_stride;
    a.colsum = colsum ? cpart : nullptr;
    const dim3 grid((unsigned)nn, (unsigned)nm, (unsigned)(count * used));
    hipStream_t st = (hipStream_t)stream;
    if (p256) {
        vd_g_last_tile = (((3 * 100 + 16) * 1000) + 256) * 1000 + 256;
        hipLaunchKernelGGL(wgrad_planes256_kernel, grid, dim3(512), 0, st, a, gp);
    } else if (tile == 0 && k16) launch_grouped<128, 128, 16>(a, gp, grid, st);
    else if (tile == 0) launch_grouped<128, 128>(a, gp, grid, st);
    else if (tile == 1) launch_grouped<128, 64>(a, gp, grid, st);
    else if (tile == 2) launch_grouped<64, 128>(a, gp, grid, st);
    else launch_grouped<64, 64>(a, gp, grid, st);
    VD_LAUNCH_CHECK("gemm_dma_kernel(grouped)");
    if (slabs_only) {
        VD_REQUIRE(used == vd_gemm_grouped_wgrad_used_slabs(count, M, N, K, splitk), "vd_gemm_grouped_wgrad: slab plan mismatch");
        return 0;
    }
    const long long tot = (long long)M * N;
    hipLaunchKernelGGL(reduce_slabs_grouped_kernel, dim3((unsigned)((tot + 255) / 256), (unsigned)count), dim3(256), 0, st, ws, used,
                       a.slab_stride, M, N, go, (long long)ldc, cpart);
    VD_LAUNCH_CHECK("reduce_slabs_grouped_kernel");
    return 0;
}

thread_local int vd_g_last_tile = 0;
extern "C" int vd_gemm_last_tile(void) { return vd_g_last_tile; }

extern "C" int vd_gemm(const vd_gemm_desc* d, void* stream) {
    VD_REQUIRE(d != nullptr, "vd_gemm: null descriptor");
    return run_gemm(*d, (hipStream_t)stream);
}

extern "C" int vd_conv3x3(const float* xin, int64_t ldx, const float* wpack, const float* bias, const float* res,
                          int64_t ldres, float* y, int64_t ldy, int32_t nimg, int32_t H, int32_t W, int32_t Cin,
                          int32_t Cout, int32_t accumulate, float* stats_part, void* stream) {
    VD_REQUIRE(Cin % 4 == 0, "vd_conv3x3: Cin must be a multiple of 4 (got %d)", Cin);
    vd_gemm_desc d = {};
    d.A = xin; d.B = wpack; d.C = y; d.bias = bias; d.R = res;
    d.M = nimg * H * W; d.N = Cout; d.K = 9 * Cin;
    d.a_kind = VD_IM2COL; d.b_kind = VD_ROW;
    d.lda = ldx; d.ldb = 9LL * Cin; d.ldc = ldy; d.ldr = ldres;
    d.batch = 1; d.nh = 1; d.alpha = 1.f; d.accumulate = accumulate;
    d.H = H; d.W = W; d.Cin = Cin;
    d.stats = stats_part; d.stats_hw = H * W;
    if (stats_part && (H * W) % 64 != 0) d.tile = 64;      // 64-row tiles (32-row slabs) whenever the image allows only that
    return run_gemm(d, (hipStream_t)stream);
}

// Split-K plan of the conv weight gradient.  All blocks of the launch do equal work and `per_cu` of them fit a CU, so the
// launch runs in "rounds" of 256*per_cu resident blocks: pick the slab count that fills whole rounds as exactly as
// possible (a 756-block launch on 512 slots took two rounds for 1.48 rounds of work), keeping >= 512 pixels per slab.
static void wgrad_plan(int nimg, int H, int W, int Cin, int Cout, int* tile_out, int* split_out) {
    const int t = choose_tile(Cout, Cin, true, 64, 0);   // the slab count below fills the chip whatever the tile
    const long long kt = ((long long)nimg * H * W + 31) / 32;
    const long long tiles = ((Cout + TILES[t].bm - 1) / TILES[t].bm) * 9LL * ((Cin + TILES[t].bn - 1) / TILES[t].bn);
    const bool wide = t == 0 && (long long)nimg * H * W >= WGRAD_WIDE_PIXELS;
    const long long slots = 256LL * (wide ? 4 : TILES[t].per_cu);
    int best = 1;
    double best_eff = 0.0;
    for (int s = 1; s <= 64; ++s) {
        if (s > 1 && kt / s < 16) break;
        const long long blocks = tiles * s;
        const long long rounds = (blocks + slots - 1) / slots;
        double eff = (double)blocks / (double)(rounds * slots);
        eff *= 1.0 - 0.004 * s;                       // mild preference for fewer slabs (less reduce traffic)
        if (eff > best_eff) { best_eff = eff; best = s; }
    }
    *tile_out = t; *split_out = best;
}

extern "C" size_t vd_conv3x3_wgrad_ws_bytes(int32_t nimg, int32_t H, int32_t W, int32_t Cin, int32_t Cout) {
    int t, S;
    wgrad_plan(nimg, H, W, Cin, Cout, &t, &S);
    return (size_t)S * ((size_t)Cout * 9 * Cin + Cout) * sizeof(float);
}

// phases: 1 = split-K slabs (the MFMA kernel), 2 = slab reduction + OIHW transposition, 3 = both
static int conv3x3_wgrad_impl(const float* xin, int64_t ldx, const float* dy, int64_t lddy, int32_t nimg, int32_t H, int32_t W,
                              int32_t Cin, int32_t Cout, float* dw_oihw, float* dbias, int32_t Cin_w, int32_t Cout_w,
                              int32_t accumulate, float* ws, size_t ws_bytes, hipStream_t st, int phases) {
    VD_REQUIRE(Cin % 4 == 0 && Cout % 4 == 0, "vd_conv3x3_wgrad: Cin/Cout must be multiples of 4 (%d,%d)", Cin, Cout);
    VD_REQUIRE(Cin_w <= Cin && Cout_w <= Cout, "vd_conv3x3_wgrad: real dims exceed padded dims");
    int t, S;
    wgrad_plan(nimg, H, W, Cin, Cout, &t, &S);
    static const int codes[4] = {128, 12864, 64128, 64};
    VD_REQUIRE(ws && ws_bytes >= (size_t)S * ((size_t)Cout * 9 * Cin + Cout) * 4, "vd_conv3x3_wgrad: workspace too small");
    const long long slab = (long long)Cout * 9 * Cin;
    // slabs phase 1 writes = what run_gemm derives from (K tile, requested split): recomputed here from the arguments alone, so
    // the two phases of vd_conv3x3_wgrad_phase share no hidden state (any thread, any interleaving with other launches)
    int used = 1;
    if (S > 1) {
        GemmArgs ga = {};
        ga.A = dy; ga.B = xin; ga.lda = lddy; ga.ldb = ldx; ga.ldc = 9LL * Cin; ga.ldr = 0;
        const long long K = (long long)nimg * H * W;
        const int ktile = ktile_for(ga, t, 0, true, K >= WGRAD_WIDE_PIXELS);
        const long long kt_total = (K + ktile - 1) / ktile, per = (kt_total + S - 1) / S;
        used = (int)((kt_total + per - 1) / per);
    }
    if (phases & 1) {
        vd_gemm_desc d = {};
        d.A = dy; d.B = xin; d.C = ws;
        d.M = Cout; d.N = 9 * Cin; d.K = nimg * H * W;
        d.a_kind = VD_COL; d.b_kind = VD_IM2COL;
        d.lda = lddy; d.ldb = ldx; d.ldc = 9LL * Cin;
        d.batch = 1; d.nh = 1; d.alpha = 1.f;
        d.H = H; d.W = W; d.Cin = Cin;
        d.splitk = S; d.ws = ws; d.ws_bytes = (int64_t)ws_bytes; d.tile = codes[t];
        if (S > 1) {
            d.colsum = dbias;                  // run_gemm redirects it to the per-slab partial area behind the slabs
        } else {
            d.splitk = 1;
            d.colsum = dbias ? ws + slab : nullptr; d.colsum_accumulate = 0;
        }
        int rc = run_gemm(d, st);
        if (rc) return rc;
    }
    if (phases & 2) {
        float* cpart = ws + used * slab;
        hipLaunchKernelGGL(reduce_slabs_oihw_kernel, dim3((slab + 255) / 256), dim3(256), 0, st, ws, used, slab, Cout, Cin,
                           Cout_w, Cin_w, dw_oihw, accumulate, cpart, dbias);
        VD_LAUNCH_CHECK("reduce_slabs_oihw_kernel");
    }
    return 0;
}

extern "C" int vd_conv3x3_wgrad(const float* xin, int64_t ldx, const float* dy, int64_t lddy, int32_t nimg, int32_t H,
                                int32_t W, int32_t Cin, int32_t Cout, float* dw_oihw, float* dbias, int32_t Cin_w,
                                int32_t Cout_w, int32_t accumulate, float* ws, size_t ws_bytes, void* stream) {
    return conv3x3_wgrad_impl(xin, ldx, dy, lddy, nimg, H, W, Cin, Cout, dw_oihw, dbias, Cin_w, Cout_w, accumulate, ws, ws_bytes,
                              (hipStream_t)stream, 3);
}

extern "C" int vd_conv3x3_wgrad_phase(const float* xin, int64_t ldx, const float* dy, int64_t lddy, int32_t nimg, int32_t H,
                                      int32_t W, int32_t Cin, int32_t Cout, float* dw_oihw, float* dbias, int32_t Cin_w,
                                      int32_t Cout_w, int32_t accumulate, float* ws, size_t ws_bytes, int32_t phase, void* stream) {
    VD_REQUIRE(phase == 1 || phase == 2, "vd_conv3x3_wgrad_phase: phase must be 1 (slabs) or 2 (reduce)");
    return conv3x3_wgrad_impl(xin, ldx, dy, lddy, nimg, H, W, Cin, Cout, dw_oihw, dbias, Cin_w, Cout_w, accumulate, ws, ws_bytes,
                              (hipStream_t)stream, phase);
}

extern "C" int vd_pack_conv3x3(const float* w_oihw, int32_t Cout_w, int32_t Cin_w, float* wf, int32_t Cin_p, float* wd,
                               int32_t Cout_p, void* stream) {
    VD_REQUIRE(w_oihw && (wf || wd), "vd_pack_conv3x3: null pointer");
    VD_REQUIRE((!wf || Cin_p >= Cin_w) && (!wd || Cout_p >= Cout_w), "vd_pack_conv3x3: padded dims too small");
    long long tot = 0;
    if (wf) tot = (long long)Cout_w * 9 * Cin_p;
    if (wd && (long long)Cin_w * 9 * Cout_p > tot) tot = (long long)Cin_w * 9 * Cout_p;
    hipLaunchKernelGGL(pack_conv3x3_kernel, dim3((tot + 255) / 256), dim3(256), 0, (hipStream_t)stream, w_oihw, Cout_w,
                       Cin_w, wf, Cin_p, wd, Cout_p);
    VD_LAUNCH_CHECK("pack_conv3x3_kernel");
    return 0;
}

extern "C" int vd_pack_conv3x3_batched(const int64_t* items_dev, int32_t n, int64_t total_blocks, void* stream) {
    VD_REQUIRE(items_dev && n > 0 && total_blocks > 0 && total_blocks < (1LL << 31), "vd_pack_conv3x3_batched: bad table");
    hipLaunchKernelGGL(pack_conv3x3_batched_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const long long*>(items_dev), n);
    VD_LAUNCH_CHECK("pack_conv3x3_batched_kernel");
    return 0;
}
