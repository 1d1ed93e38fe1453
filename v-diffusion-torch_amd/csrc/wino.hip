// wino.hip -- 3x3 / stride 1 / pad 1 convolution as Winograd F(2x2, 3x3) on the fp32 matrix cores, NHWC.
// Reference op: modules.Conv2d -> F.conv2d (modules.py:141-144 <- unet.py:121,125) and its input gradient (autograd).
//
// Why: 3x3 convolutions are 87-91 % of the FLOPs of the path and gfx950 has no reduced-precision fast path
// for fp32 inputs (v_mfma_f32_* runs at the fp32 vector rate), so the direct implicit GEMM is pinned under 157 TFLOP/s.
// F(2x2,3x3) needs 16 multiplies per 2x2 output tile and (ci, co) pair instead of 36: 2.25x fewer matrix-core cycles for
// the same result in exact fp32 arithmetic (max error ~2x the direct form's, measured against fp64: tests).
//
//   U[xi][co][ci]   = (G w[co][ci] G^T)[xi]                       xi = 4a+b in 0..15      (vd_wino_pack*, once per weight update)
//   V[xi][t][ci]    = (B^T d[t][ci] B)[xi]                        d = 4x4 input patch of 2x2-output tile t   (on the fly)
//   M[xi][t][co]    = sum_ci V[xi][t][ci] U[xi][co][ci]           16 independent GEMMs                         (MFMA)
//   y[t][u][v][co]  = (A^T M[.][t][co] A)[u][v] + bias + residual                                           (epilogue)
//
// Everything between the NHWC input and the NHWC output stays on chip (V and M never exist in HBM):
//   * workgroup = 4 waves = 64 consecutive tiles x 32 output channels x all 16 xi; wave w owns tiles 16w..16w+15 and ALL 16 xi,
//     so the output transform is lane-local (no cross-wave exchange): v_mfma_f32_16x16x4_f32 with U as the row operand gives
//     lane (i, rq) the four consecutive channels 4rq..4rq+3 of tile i in one accumulator quad per xi;
//   * per K tile (16 input channels) the raw input patch (not V: 4x fewer bytes) and the 16 U tiles go HBM/L2 -> LDS with
//     `buffer_load_dwordx4 ... lds`.  The patch image is chunk-major with odd and even input columns in separate runs, so that
//     the 16 tiles of a wave read 16 consecutive 16-byte slots for every patch position (conflict-free ds_read_b128);
//     the U image is [xi][co][16 k] with the chunk XOR-swizzled by (co >> 2) & 2 (conflict-free for the 16x16x4 lane map);
//   * a lane reads its 4x4 patch (16 x b128 = 4 channels each), applies B^T . B in registers (32 adds per channel) and feeds
//     128 MFMAs per K tile; the patch of K tile t+1 is read from LDS while tile t computes (the patch stream runs one tile ahead
//     of the U stream, one barrier per K tile);
//   * epilogue: A^T . A in registers (24 adds per output quad), bias, residual, dwordx4 stores, optional GroupNorm partial sums
//     of the output (16-lane butterflies) in the layout vd_gn_stats_from_partials expects (chunk = the wave's 64 pixels).
#include "common.h"
#include <type_traits>
#include <stdlib.h>

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
constexpr unsigned OOB = 0x80000000u;
constexpr int KT = 16;            // input channels per K tile (one ds_read_b128 per lane and operand)
constexpr int TN = 32;            // output channels per workgroup
constexpr int TILES_WG = 64;      // 2x2-output tiles per workgroup (16 per wave)
constexpr int B_STAGE = 16 * TN * KT;          // floats of U per stage (32 KB)

struct WinoArgs {
    const float* x; long long ldx;             // input  [nimg][H][W][>=K]
    const float* U;                            // [16][Cout][K]
    const float* bias; const float* res; long long ldr;
    float* y; long long ldy;
    float* stats;                              // optional GroupNorm partials [img][HW/64][2][Cout]
    int nimg, H, W, K, Cout;
    int TW, TH, TPI;                           // tiles per row / column / image
    int P;                                     // patch-image row pitch in slots
    int RIN;                                   // input rows per image held by a workgroup (2*tile_rows + 2)
    int NTR;                                   // tile rows per image per workgroup
    int NIW;                                   // images per workgroup (1 unless an image has fewer than 64 tiles)
    int lgTW, lgTPI;                           // log2 (all supported geometries are powers of two)
    float invP2, invRIN;                       // 1 / (2 P), 1 / RIN (prologue index arithmetic without integer division)
    int ntiles;                                // nimg * TPI
    int ncb, ntg;                              // work items: channel blocks x tile groups
    unsigned long long* probe;                 // -DVD_PROBES builds only (libvdiff_hip_probe.so, vd_wino_set_probe): 8 x u64 per workgroup
    int probe_light;                           // -DVD_PROBES builds: stamps at kernel start / end only (VD_WINO_PROBE_LIGHT: no per-barrier s_memtime)
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const float* base, int records = (int)OOB) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, records, 0x00020000);
}

__device__ __forceinline__ int b_swz(int row) { return (row >> 2) & 2; }

// TW = tiles per image row (compile-time geometry: patch pitch, offsets), NS = patch slots per 16-byte channel chunk (multiple of
// 128), STATS = emit GroupNorm partial sums.
//
// Persistent workgroups of 12 waves, one per CU (113-144 KB of LDS): a workgroup walks the work items (64 tiles x 32 channels)
// n * G + w, n = 0, 1, ...
//   * waves 0-7 compute (two per SIMD: with one, the wave's own LDS / transform instructions sit between its MFMAs -- 59 cycles
//     per MFMA instead of 32 in the first version of this kernel).  Wave w owns tile group w & 3 (16 tiles) and HALF of the xi
//     range, a in {2h, 2h+1} with h = w >> 2, for both channel blocks: splitting xi (not channels) between the two waves of a tile
//     group means neither repeats the other's input transform (three of the four patch rows, half of the B^T . B work each).
//     Their partial output transforms meet once per item, through LDS, in the epilogue.
//   * waves 8-11 are loaders: they issue every tile DMA (each `buffer_load ... lds` costs its wave ~100 cycles of issue: spread
//     over the compute waves that was ~700 cycles per wave and K tile with the SIMD's other wave running alone) and run AHEAD
//     across items: during the last K tile of an item they already fetch the first stages of the next one, so neither the DMA
//     latency nor the offset arithmetic of an item's prologue is exposed (11 000 cycles per item before).
// Stage protocol (pb = parity of the item's first stage): patch of K tile t in sA[(pb+t)&1], U of t in sB[(pb+t)&1].  Loader,
// between barrier t-1 and barrier t: U(t+1), patch(t+2) [last tile: head of the next item = patch'(0), patch'(1), U'(0) with
// pb' = (pb + nkt) & 1], wait, barrier t.  Compute, K tile t: steps 0-6 (U fragments of t, the 12 patch reads of t+1), barrier t,
// prefetch of the first U fragments of t+1, step 7.  One more barrier per item separates the epilogue's LDS exchange (in the
// dead U stage) and the read of patch'(0) from the loader's next overwrite.
constexpr int WINO_THREADS = 768;
// PROBE / EXP are instantiated by -DVD_PROBES builds only (libvdiff_hip_probe.so, tests/probe/): the product library has neither.
// EXP != 0: timing experiments only (WRONG results; VD_WINO_EXP, tests/probe/wino_exp.py): 1 = no input-transform arithmetic,
// 2 = also no patch reads, 3 = also no U-fragment reads, 4 = everything but no tile barrier in the compute waves' K loop,
// 5 = like 3 and the loaders issue no DMA (the MFMA + epilogue skeleton alone), 6 = full arithmetic but only half of the U DMA pieces
template <int TW, int NS, bool STATS, bool PROBE = false, int EXP = 0>
__global__ __launch_bounds__(WINO_THREADS) void wino_conv_kernel(const WinoArgs p) {
    unsigned long long ts0 = 0, ts1 = 0, ts2 = 0, twait = 0, rt0 = 0;
    if (PROBE) { ts0 = __builtin_amdgcn_s_memtime(); rt0 = __builtin_amdgcn_s_memrealtime(); }
    constexpr int LGTW = TW == 4 ? 2 : (TW == 8 ? 3 : (TW == 16 ? 4 : (TW == 32 ? 5 : 6)));
    constexpr int P = TW >= 16 ? TW + 1 : (TW == 8 ? 10 : 5);      // pitch: 16 consecutive tiles read 16 distinct bank quads
    constexpr int P2 = 2 * P;
    constexpr int A_STAGE = 4 * NS * 4;                 // floats: 4 chunks x NS slots x 4 floats
    constexpr int NPA = 4 * NS / 64;                    // DMA pieces (1 KiB) of a patch stage
    constexpr int NPB = B_STAGE / 256;                  // 32 pieces of a U stage
    static_assert(NS % 128 == 0, "patch slots per chunk must give every loader wave whole DMA pieces");
    __shared__ __attribute__((aligned(1024))) float smem[2 * A_STAGE + 2 * B_STAGE];
    float* const sA = smem;
    float* const sB = smem + 2 * A_STAGE;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lq = lane >> 4;
    const int tg = wave & 3, ah = (wave >> 2) & 1;     // tile group, xi half (compute waves)
    const bool loader = wave >= 8;
    const int nkt = p.K / KT;
    const int G = gridDim.x, w = blockIdx.x;
    const int nitems = p.ncb * p.ntg;
    // item of round n: linear id n * G + w, re-ordered inside the round so that the workgroups of one XCD (w % 8) take
    // neighbouring items (channel blocks fastest: they share the input patch in that XCD's L2)
    auto item_of = [&](int n, int& tbx, int& tby) -> bool {
        int t = n * G + w;
        // (only in FULL rounds: a ragged last round keeps the plain ids, or permuted and plain ids would collide)
        if ((G & 7) == 0 && (n + 1) * G <= nitems) t = n * G + (w & 7) * (G >> 3) + (w >> 3);
        if (t >= nitems) return false;
        if ((p.ncb & 3) == 0 && p.ncb > 4 && (p.ntg & 7) == 0) {
            // 32 consecutive items (= the items one XCD's workgroups hold at a time) are 4 channel blocks x 8 tile groups rather than
            // 8 x 4: 4 U slices + 8 patches cross the fabric per XCD and round instead of 8 + 4 (a U slice is 1.6x a patch): +1-2 %
            const int c = t >> 5, i = t & 31, ncg = p.ncb >> 2;
            tbx = (c % ncg) * 4 + (i & 3); tby = (c / ncg) * 8 + (i >> 2);
            return true;
        }
        tbx = t % p.ncb; tby = t / p.ncb;
        return true;
    };

    if (loader) {
        // ================================================================= loader waves (4: each issues a quarter of the pieces)
        // Piece q = LW + 4 j.  Patch pieces: chunk = q / (NS/64) (scalar offset), pixel slot group = q % (NS/64): few distinct pixels
        // per lane.  U pieces: half = LW & 1, xi = (LW >> 1) + 2 j (scalar offset).  LW is a compile-time constant per code path so
        // that every register index is static.
        auto run = [&](auto LWc) {
            constexpr int LW = decltype(LWc)::value;
            constexpr int SG = NS / 64, APL = NPA / 4, BPL = NPB / 4;
            unsigned pxo[SG], pxn[SG], vb0 = 0, vb0n = 0;
            const unsigned xi_stride2 = 2u * (unsigned)p.Cout * (unsigned)p.K * 4u;
            auto offsets = [&](int tbx, int tby, unsigned (&px)[SG], unsigned& vb) {
                const int co0 = tbx * TN;
                const int tile0 = tby * TILES_WG;
                const int img0 = tile0 >> p.lgTPI;
                const int trow0 = (tile0 & (p.TPI - 1)) >> LGTW;
                const int y_first = 2 * trow0 - 1;                  // input row held in local row 0 of every image of the workgroup
#pragma unroll
                for (int j = 0; j < APL && j < SG; ++j) {           // (the slot groups repeat with period <= SG)
                    const int g = (LW + 4 * j) % SG;
                    bool seen = false;
                    for (int jj = 0; jj < j; ++jj) seen = seen || ((LW + 4 * jj) % SG == g);
                    if (seen) continue;
                    const int s = g * 64 + lane;                    // slot inside a chunk
                    const int rr = s / P2, rem = s - rr * P2;       // local row over all images of the workgroup, place in the (odd, even) pair
                    const int par = rem >= P ? 1 : 0, idx = rem - par * P;
                    int il = (int)((float)rr * p.invRIN);           // local image, row inside it
                    if (il * p.RIN > rr) --il; else if ((il + 1) * p.RIN <= rr) ++il;
                    const int r = rr - il * p.RIN;
                    const int yy = y_first + r, xx = 2 * idx - 1 + par;   // par 0: odd columns x = 2 idx - 1 ; par 1: even columns x = 2 idx
                    const int img = img0 + il;
                    unsigned vo = OOB;
                    if (il < p.NIW && img < p.nimg && (unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W)
                        vo = (unsigned)((((long long)img * p.H + yy) * p.W + xx) * p.ldx) * 4u;
                    px[g] = vo;
                }
                const int row = lane >> 2, c = (lane & 3) ^ b_swz(row);
                const int co = co0 + (LW & 1) * 16 + row;
                vb = co < p.Cout ? (unsigned)((((long long)(LW >> 1) * p.Cout + co) * p.K + c * 4) * 4) : OOB;
            };
            auto issue_A = [&](int kt, int buf, const unsigned (&px)[SG]) {
                const __amdgpu_buffer_rsrc_t rs = make_rsrc(p.x + kt * KT);
                float* dst = sA + buf * A_STAGE;
#pragma unroll
                for (int j = 0; j < APL; ++j) {
                    constexpr int dummy = 0; (void)dummy;
                    const int q = LW + 4 * j;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(dst + q * 256), 16, (int)px[q % SG], (q / SG) * 16, 0, 0);
                }
            };
            auto issue_B = [&](int kt, int buf, unsigned vb) {
                const __amdgpu_buffer_rsrc_t rs = make_rsrc(p.U + kt * KT);
                float* dst = sB + buf * B_STAGE;
#pragma unroll
                for (int j = 0; j < BPL; ++j) {
                    if (EXP == 6 && (j & 1)) continue;              // timing experiment: half of the U pieces (-28 % of the staged bytes)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(dst + (LW + 4 * j) * 256), 16, (int)vb, (int)(j * xi_stride2), 0, 0);
                }
            };
            int tbx = 0, tby = 0, nbx = 0, nby = 0;
            bool have = item_of(0, tbx, tby);
            int pb = 0;
            if (have) {
                offsets(tbx, tby, pxo, vb0);
                issue_B(0, pb, vb0);
                issue_A(0, pb, pxo);
                if (nkt > 1) issue_A(1, pb ^ 1, pxo);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                                        // [P0] head of the first item landed
            for (int n = 0; have; ++n) {
                const bool next = item_of(n + 1, nbx, nby);
                if (next) offsets(nbx, nby, pxn, vb0n);             // (the loaders have time to spare: off the critical path)
                __syncthreads();                                    // [P2] the compute waves have read patch(0) out of sA[pb]
                for (int kt = 0; kt < nkt; ++kt) {
                    const int buf = (pb + kt) & 1;
                    if (EXP != 5) {
                        if (kt + 1 < nkt) issue_B(kt + 1, buf ^ 1, vb0);
                        if (kt + 2 < nkt) issue_A(kt + 2, buf, pxo);
                    }
                    if (next && kt == nkt - 1 && EXP != 5) {        // head of the next item into the stages this item has left
                        const int pn = (pb + nkt) & 1;
                        issue_B(0, pn, vb0n);
                        issue_A(0, pn, pxn);
                        if (nkt > 1) issue_A(1, pn ^ 1, pxn);
                    }
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    if (EXP != 4) __syncthreads();                  // [kt]
                }
                __syncthreads();                                    // [X] epilogue exchange
                pb = (pb + nkt) & 1;
                have = next;
                if (next) {
#pragma unroll
                    for (int g = 0; g < SG; ++g) pxo[g] = pxn[g];
                    vb0 = vb0n;
                }
            }
        };
        if (wave == 8) run(std::integral_constant<int, 0>{});
        else if (wave == 9) run(std::integral_constant<int, 1>{});
        else if (wave == 10) run(std::integral_constant<int, 2>{});
        else run(std::integral_constant<int, 3>{});
        return;
    }

    // ===================================================================== compute waves
    // patch rows of this xi half in the roles (r0, r1, r2) of   tr[0] = r0 - r2,  tr[1] = sgn * r1 + r2:
    //   h = 0 (a = 0, 1): p0 - p2, p1 + p2  -> rows (0, 1, 2), sgn = +1 ;  h = 1 (a = 2, 3): p2 - p1, p1 - p3 -> rows (2, 3, 1), sgn = -1
    const float sgn = ah ? -1.f : 1.f;
    const int rrow0 = ah ? 2 : 0, rrow1 = ah ? 3 : 1, rrow2 = ah ? 1 : 2;
    auto poff = [](int q) { return ((q & 1) * P + (q >> 1)) * 4; };            // float offset of column q inside a patch row (immediate)
    // U fragment of (xi, cb): lane (n = li, kq = lq) holds U[xi][co0 + 16 cb + n][k0 + 4 kq .. +3]
    const int boff = ah * 8 * 2 * 256 + li * KT + ((lq ^ b_swz(li)) << 2);
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

    int tbx = 0, tby = 0;
    bool have = item_of(0, tbx, tby);
    int pb = 0;
    __syncthreads();                                                // [P0]
    f32x4 ub[2][2];                                                 // U fragments: step x of a K tile uses ub[x & 1], loaded one step ahead
    for (int n = 0; have; ++n) {
        const int co0 = tbx * TN;
        const int tile0 = tby * TILES_WG;                           // first tile of the item
        // Per-item lane constants are RE-DERIVED here from the lane id (a volatile v_mbcnt pair) and from scalars passed through
        // optimisation barriers instead of being kept live across the K loop: the loop runs at the 168-register cap of three
        // waves per SIMD, and everything loop-invariant the compiler hoists out of the item loop ends up parked in scratch.
        int lane_i;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_i));
        const int li_i = lane_i & 15, lq_i = lane_i >> 4;
        int tpim = p.TPI - 1, lgtpi = p.lgTPI, rin = p.RIN;
        asm volatile("" : "+s"(tpim), "+s"(lgtpi), "+s"(rin));
        const int trow0 = (tile0 & tpim) >> LGTW;                   // first tile row inside its image (0 when an item spans images)
        const int tl = 16 * tg + li_i;                              // this lane's tile inside the item
        const int tile = tile0 + tl;
        const int il = tl >> lgtpi;                                 // local image (0 unless the item spans images)
        const int tin = (tile & tpim);                              // tile inside its image
        const int ty = tin >> LGTW, tx = tin & (TW - 1);
        const int slot0 = ((il * rin + 2 * (ty - trow0)) * 2) * P + tx;      // slot of patch position (p=0, q=0), chunk 0
        const float* pbase = sA + (lq_i * NS + slot0) * 4;          // + chunk lq
        const float* pr0 = pbase + rrow0 * P2 * 4;
        const float* pr1 = pbase + rrow1 * P2 * 4;
        const float* pr2 = pbase + rrow2 * P2 * 4;

        f32x4 acc[8][2];
#pragma unroll
        for (int xi = 0; xi < 8; ++xi)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) acc[xi][cb] = f32x4{0.f, 0.f, 0.f, 0.f};

        // patch(0) and the first U fragments (landed: the loader waited in front of the previous barrier)
        f32x4 tr[2][4];
        {
            const int o = pb * A_STAGE;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 r0 = *reinterpret_cast<const f32x4*>(pr0 + o + poff(q)), r1 = *reinterpret_cast<const f32x4*>(pr1 + o + poff(q)),
                            r2 = *reinterpret_cast<const f32x4*>(pr2 + o + poff(q));
                tr[0][q] = r0 - r2;
                tr[1][q] = r1 * sgn + r2;
            }
            if (n == 0) {
                const float* bs = sB + pb * B_STAGE + boff;
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) ub[0][cb] = *reinterpret_cast<const f32x4*>(bs + cb * 256);
            }
        }
        __syncthreads();                                            // [P2]
        if (PROBE && n == 0 && !p.probe_light) ts1 = __builtin_amdgcn_s_memtime();

        for (int kt = 0; kt < nkt; ++kt) {
            const int buf = (pb + kt) & 1;
            const float* bs = sB + buf * B_STAGE + boff;
            const int on = (buf ^ 1) * A_STAGE;                     // stage of the next patch (a dead stage in the last tile: unused)
            f32x4 trn[2][4];
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                f32x4 V[4];
                if ((EXP >= 1 && EXP <= 3) || EXP == 5) { V[0] = tr[a][0]; V[1] = tr[a][1]; V[2] = tr[a][2]; V[3] = tr[a][3]; }
                else { V[0] = tr[a][0] - tr[a][2]; V[1] = tr[a][1] + tr[a][2]; V[2] = tr[a][2] - tr[a][1]; V[3] = tr[a][1] - tr[a][3]; }
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const int xi = 4 * a + b;
                    if (xi < 7 && !(EXP == 3 || EXP == 5)) {
                        // U fragments of step xi + 1 (one step = 8 MFMAs = 256+ cycles ahead of their use)
#pragma unroll
                        for (int cb = 0; cb < 2; ++cb) ub[(xi + 1) & 1][cb] = *reinterpret_cast<const f32x4*>(bs + ((xi + 1) * 2 + cb) * 256);
                    }
                    if (xi < 6 && (EXP == 2 || EXP == 3 || EXP == 5)) {
                        if (xi < 2) { trn[1][2 * xi] = tr[1][2 * xi]; trn[1][2 * xi + 1] = tr[1][2 * xi + 1]; }
                        else if (xi < 4) { trn[0][2 * (xi - 2)] = tr[0][2 * (xi - 2)]; trn[0][2 * (xi - 2) + 1] = tr[0][2 * (xi - 2) + 1]; }
                    } else if (xi < 6 && EXP == 1) {
                        if (xi < 2) { trn[1][2 * xi] = *reinterpret_cast<const f32x4*>(pr2 + on + poff(2 * xi)); trn[1][2 * xi + 1] = *reinterpret_cast<const f32x4*>(pr2 + on + poff(2 * xi + 1)); }
                        else if (xi < 4) { const int q0 = 2 * (xi - 2); trn[0][q0] = *reinterpret_cast<const f32x4*>(pr0 + on + poff(q0)); trn[0][q0 + 1] = *reinterpret_cast<const f32x4*>(pr0 + on + poff(q0 + 1)); }
                        else { const int q0 = 2 * (xi - 4); const f32x4 r1a = *reinterpret_cast<const f32x4*>(pr1 + on + poff(q0)), r1b = *reinterpret_cast<const f32x4*>(pr1 + on + poff(q0 + 1)); asm volatile("" :: "v"(r1a), "v"(r1b)); }
                    } else if (xi < 6) {
                        // 2 of the 12 patch reads of the next K tile (roles r2, r0 first: tr[0] early)
                        if (xi < 2) {
                            const f32x4 r2a = *reinterpret_cast<const f32x4*>(pr2 + on + poff(2 * xi)), r2b = *reinterpret_cast<const f32x4*>(pr2 + on + poff(2 * xi + 1));
                            trn[1][2 * xi] = r2a; trn[1][2 * xi + 1] = r2b;             // (parked: becomes sgn * r1 + r2 below)
                        } else if (xi < 4) {
                            const int q0 = 2 * (xi - 2);
                            const f32x4 r0a = *reinterpret_cast<const f32x4*>(pr0 + on + poff(q0)), r0b = *reinterpret_cast<const f32x4*>(pr0 + on + poff(q0 + 1));
                            trn[0][q0] = r0a - trn[1][q0]; trn[0][q0 + 1] = r0b - trn[1][q0 + 1];
                        } else {
                            const int q0 = 2 * (xi - 4);
                            const f32x4 r1a = *reinterpret_cast<const f32x4*>(pr1 + on + poff(q0)), r1b = *reinterpret_cast<const f32x4*>(pr1 + on + poff(q0 + 1));
                            trn[1][q0] = r1a * sgn + trn[1][q0]; trn[1][q0 + 1] = r1b * sgn + trn[1][q0 + 1];
                        }
                        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                    } else if (xi == 6) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                    if (xi == 7) {
                        // every read of this K tile's stages is done: barrier, then the first U fragments of the next tile (or item)
                        unsigned long long tw = 0;
                        if (PROBE && !p.probe_light) tw = __builtin_amdgcn_s_memtime();
                        if (EXP != 4) __syncthreads();              // [kt]
                        if (PROBE && !p.probe_light) { const unsigned long long te = __builtin_amdgcn_s_memtime(); twait += te - tw; }
                        const float* bn = sB + (buf ^ 1) * B_STAGE + boff;
                        if (EXP != 3 && EXP != 5) {
#pragma unroll
                            for (int cb = 0; cb < 2; ++cb) ub[0][cb] = *reinterpret_cast<const f32x4*>(bn + cb * 256);
                        }
                        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        acc[xi][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ub[xi & 1][0][j], V[b][j], acc[xi][0], 0, 0, 0);
                        acc[xi][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ub[xi & 1][1][j], V[b][j], acc[xi][1], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
                }
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int q = 0; q < 4; ++q) tr[a][q] = trn[a][q];
        }
        if (PROBE && !p.probe_light) ts2 = __builtin_amdgcn_s_memtime();

        // ---------------- epilogue: y = A^T M A (+ bias + residual)      A^T = [1 1 1 0 ; 0 1 -1 -1]
        // lane (li, lq): tile `tile`, channels co0 + 16 cb + 4 lq .. +3.  Column part in registers: s[a][v] = sum_b M[a][b] A[b][v];
        // row part: Y[0][v] = s[0][v] + s[1][v] + s[2][v], Y[1][v] = s[1][v] - s[2][v] - s[3][v] -- this wave holds a in {2h, 2h+1},
        // so it forms its partial PT[u][v] of both channel blocks, hands the block it does not finish to its partner wave through
        // LDS (wave h finishes channel block h) and adds the partner's partial to its own.
        // (the tile coordinates are re-derived here behind an optimisation barrier: kept live across the K loop they cost registers
        //  the loop does not have -- 168 per wave at three waves per SIMD)
        int tby_e = tby, lane_e = lane;
        asm volatile("" : "+s"(tby_e));
        asm volatile("" : "+v"(lane_e));
        const int li_e = lane_e & 15, lq_e = lane_e >> 4;
        const int tile_e = tby_e * TILES_WG + 16 * tg + li_e;
        const int tin_e = tile_e & (p.TPI - 1);
        const int ty_e = tin_e >> LGTW, tx_e = tin_e & (TW - 1);
        const bool tile_ok = tile_e < p.ntiles;
        const int img = tile_e >> p.lgTPI;
        // (32-bit offsets: vd_conv3x3_wino_supported bounds pixels * ld * 4 below 2^31 for x, y and the residual)
        const unsigned pix00 = (unsigned)((img * p.H + 2 * ty_e) * p.W + 2 * tx_e);
        const unsigned ldr_u = (unsigned)p.ldr, ldy_u = (unsigned)p.ldy;
        const __amdgpu_buffer_rsrc_t yrs = make_rsrc(p.y);
        const __amdgpu_buffer_rsrc_t rrs = make_rsrc(p.res ? p.res : p.y, p.res ? (int)OOB : 0);
        const int co = co0 + 16 * ah + 4 * lq_e;
        const bool ok = tile_ok && co < p.Cout;                     // (Cout is a multiple of 4)
        // residual / bias first: their latency hides behind the transform and the exchange
        f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
        if (p.bias && ok) b4 = *reinterpret_cast<const f32x4*>(p.bias + co);
        f32x4 r4[2][2];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int v = 0; v < 2; ++v) {
                const unsigned vor = ok ? ((pix00 + (unsigned)(u * p.W + v)) * ldr_u + (unsigned)co) * 4u : OOB;
                r4[u][v] = p.res ? __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rrs, (int)vor, 0, 0)) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        f32x4 PT[2][2][2];                   // [cb][u][v]
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            f32x4 s0[2], s1[2];
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const f32x4 t1 = acc[4 * a + 1][cb] + acc[4 * a + 2][cb], t2 = acc[4 * a + 1][cb] - acc[4 * a + 2][cb];
                s0[a] = acc[4 * a][cb] + t1;
                s1[a] = t2 - acc[4 * a + 3][cb];
            }
            if (ah == 0) { PT[cb][0][0] = s0[0] + s0[1]; PT[cb][0][1] = s1[0] + s1[1]; PT[cb][1][0] = s0[1]; PT[cb][1][1] = s1[1]; }
            else { PT[cb][0][0] = s0[0]; PT[cb][0][1] = s1[0]; PT[cb][1][0] = -s0[0] - s0[1]; PT[cb][1][1] = -s1[0] - s1[1]; }
        }
        {
            // exchange area = the U stage of the last K tile (dead since its barrier; the next item's head went into the other three
            // stages): [wave][k = 2u+v][lane] float4
            f32x4* xch = reinterpret_cast<f32x4*>(sB + ((pb + nkt - 1) & 1) * B_STAGE);
            const int give = ah ^ 1;        // channel block handed to the partner
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int v = 0; v < 2; ++v) xch[(wave * 4 + 2 * u + v) * 64 + lane_e] = PT[give][u][v];
            __syncthreads();                                        // [X]
            const int partner = wave ^ 4;
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int v = 0; v < 2; ++v) PT[ah][u][v] += xch[(partner * 4 + 2 * u + v) * 64 + lane_e];
        }
        f32x4 a1 = {0.f, 0.f, 0.f, 0.f}, a2 = a1;
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int v = 0; v < 2; ++v) {
                const f32x4 val = (PT[ah][u][v] + b4) + r4[u][v];
                if (STATS) { a1 += val; a2 += val * val; }
                const unsigned voc = ok ? ((pix00 + (unsigned)(u * p.W + v)) * ldy_u + (unsigned)co) * 4u : OOB;
                const u32x4 wv = {__float_as_uint(val[0]), __float_as_uint(val[1]), __float_as_uint(val[2]), __float_as_uint(val[3])};
                __builtin_amdgcn_raw_buffer_store_b128(wv, yrs, (int)voc, 0, 0);
            }
        if (STATS) {
            // sum over the wave's 16 tiles (= lanes with equal lq): fixed-order butterfly, then lane li = 0 writes its 4 channels of
            // the [2][Cout] record of the tile group's 64-pixel chunk
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v1 = a1[j], v2 = a2[j];
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) { v1 += __shfl_xor(v1, o, 16); v2 += __shfl_xor(v2, o, 16); }
                a1[j] = v1; a2[j] = v2;
            }
            const int wtile = tby_e * TILES_WG + 16 * tg;           // the tile group's first tile: all 16 lie in one image
            if (li_e == 0 && wtile < p.ntiles && co < p.Cout) {
                const int wimg = wtile >> p.lgTPI, chunk = (wtile & (p.TPI - 1)) >> 4;
                float* o = p.stats + ((long long)wimg * (p.TPI >> 4) + chunk) * 2 * p.Cout;
                *reinterpret_cast<f32x4*>(o + co) = a1;
                *reinterpret_cast<f32x4*>(o + p.Cout + co) = a2;
            }
        }
        pb = (pb + nkt) & 1;
        have = item_of(n + 1, tbx, tby);
    }
    if (PROBE && p.probe) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long ts3 = __builtin_amdgcn_s_memtime();
        if (lane == 0) {
            unsigned long long* o = p.probe + ((unsigned long long)blockIdx.x * 8 + wave) * 8;
            o[0] = ts0; o[1] = ts1; o[2] = ts2; o[3] = ts3; o[4] = twait; o[5] = rt0; o[6] = (unsigned long long)nkt;
            o[7] = __builtin_amdgcn_s_memrealtime();
        }
    }
}

// ---- the same convolution with work items of 128 tiles x 32 channels ("wide" form).
// Why a second form: these kernels are clock-limited -- what they pay for is bytes and instructions per MFMA,
// not stalls.  A 128-tile item shares one U stage between twice as many tiles (-36 % staged bytes per MFMA; a timing build of the
// kernel above with half of its U pieces gained 4 % of clock at the same cycle count), every wave owns ALL 16 xi of its 16 tiles (each
// patch position is read and row-transformed once instead of by two waves: -33 % patch reads and transform instructions, and the
// output transform is lane-local: no exchange through LDS), at the price of 128 accumulator registers per wave: 8 waves of 256
// registers, no loader waves -- every wave issues a ninth of the next K tile's DMA between its MFMA steps.
// Stage protocol: patch and U of K tile g (a counter that runs on across the items of a workgroup) live in stage g & 1; the DMA of
// g + 1 is issued during g; one barrier per K tile.  The epilogue touches no LDS, so the first K tile of the next item streams in
// behind it.
constexpr int TILES_WIDE = 128;
constexpr int WIDE_THREADS = 512;

// bias + residual, dwordx4 stores of a tile's 2x2 output pixels, optional GroupNorm partial sums (lanes: li = tile of the 16-tile
// group, lq = channel quad)
template <int TW, bool STATS>
__device__ __forceinline__ void wino_store_outputs(const WinoArgs& p, const f32x4 (&F)[2][2], int group_tile0, int li, int co) {
    constexpr int LGTW = TW == 4 ? 2 : (TW == 8 ? 3 : (TW == 16 ? 4 : (TW == 32 ? 5 : 6)));
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const int tile = group_tile0 + li;
    const int tin = tile & (p.TPI - 1);
    const int ty = tin >> LGTW, tx = tin & (TW - 1);
    const int img = tile >> p.lgTPI;
    const unsigned pix00 = (unsigned)((img * p.H + 2 * ty) * p.W + 2 * tx);      // (32-bit: see vd_conv3x3_wino_supported)
    const unsigned ldr_u = (unsigned)p.ldr, ldy_u = (unsigned)p.ldy;
    const __amdgpu_buffer_rsrc_t yrs = make_rsrc(p.y);
    const __amdgpu_buffer_rsrc_t rrs = make_rsrc(p.res ? p.res : p.y, p.res ? (int)OOB : 0);
    const bool ok = tile < p.ntiles && co < p.Cout;                 // (Cout is a multiple of 4)
    f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
    if (p.bias && ok) b4 = *reinterpret_cast<const f32x4*>(p.bias + co);
    f32x4 r4[2][2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            const unsigned vor = ok ? ((pix00 + (unsigned)(u * p.W + v)) * ldr_u + (unsigned)co) * 4u : OOB;
            r4[u][v] = p.res ? __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rrs, (int)vor, 0, 0)) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    f32x4 a1 = {0.f, 0.f, 0.f, 0.f}, a2 = a1;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            const f32x4 val = (F[u][v] + b4) + r4[u][v];
            if (STATS) { a1 += val; a2 += val * val; }
            const unsigned voc = ok ? ((pix00 + (unsigned)(u * p.W + v)) * ldy_u + (unsigned)co) * 4u : OOB;
            const u32x4 wv = {__float_as_uint(val[0]), __float_as_uint(val[1]), __float_as_uint(val[2]), __float_as_uint(val[3])};
            __builtin_amdgcn_raw_buffer_store_b128(wv, yrs, (int)voc, 0, 0);
        }
    if (STATS) {
        // sum over the group's 16 tiles (= lanes with equal lq): fixed-order butterfly, then lane li = 0 writes its 4 channels of
        // the [2][Cout] record of the tile group's 64-pixel chunk (all 16 tiles lie in one image)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v1 = a1[j], v2 = a2[j];
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) { v1 += __shfl_xor(v1, o, 16); v2 += __shfl_xor(v2, o, 16); }
            a1[j] = v1; a2[j] = v2;
        }
        if (li == 0 && group_tile0 < p.ntiles && co < p.Cout) {
            const int wimg = group_tile0 >> p.lgTPI, chunk = (group_tile0 & (p.TPI - 1)) >> 4;
            float* o = p.stats + ((long long)wimg * (p.TPI >> 4) + chunk) * 2 * p.Cout;
            *reinterpret_cast<f32x4*>(o + co) = a1;
            *reinterpret_cast<f32x4*>(o + p.Cout + co) = a2;
        }
    }
}

template <int TW, int NS, bool STATS>
__global__ __launch_bounds__(WIDE_THREADS) void wino_conv_wide_kernel(const WinoArgs p) {
    constexpr int LGTW = TW == 8 ? 3 : (TW == 16 ? 4 : 5);
    constexpr int P = TW >= 16 ? TW + 1 : 10;
    constexpr int P2 = 2 * P;
    constexpr int A_STAGE = 4 * NS * 4;                 // floats
    constexpr int SG = NS / 64;                         // slot groups (DMA pieces) per 16-byte channel chunk
    constexpr int APL = 4 * SG / 8;                     // patch pieces per wave and stage
    static_assert((4 * SG) % 8 == 0, "every wave issues the same number of patch pieces");
    static_assert((2 * A_STAGE + 2 * B_STAGE) * 4 <= 163840, "stages exceed the LDS");
    __shared__ __attribute__((aligned(1024))) float smem[2 * A_STAGE + 2 * B_STAGE];
    float* const sA = smem;
    float* const sB = smem + 2 * A_STAGE;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lq = lane >> 4;
    const int nkt = p.K / KT;
    const int G = gridDim.x, w = blockIdx.x;
    const int nitems = p.ncb * p.ntg;
    unsigned long long pt0 = 0, pr0 = 0;                            // vd_wino_set_probe: clock stamps at kernel start / end
    if (VD_PROBE_BUILD && p.probe) { pt0 = __builtin_amdgcn_s_memtime(); pr0 = __builtin_amdgcn_s_memrealtime(); }
    auto item_of = [&](int n, int& tbx, int& tby) -> bool {
        int t = n * G + w;
        if ((G & 7) == 0 && (n + 1) * G <= nitems) t = n * G + (w & 7) * (G >> 3) + (w >> 3);
        if (t >= nitems) return false;
        if ((p.ncb & 3) == 0 && p.ncb > 4 && (p.ntg & 7) == 0) {
            const int c = t >> 5, i = t & 31, ncg = p.ncb >> 2;
            tbx = (c % ncg) * 4 + (i & 3); tby = (c / ncg) * 8 + (i >> 2);
            return true;
        }
        tbx = t % p.ncb; tby = t / p.ncb;
        return true;
    };

    // ---- DMA: wave w issues patch pieces q = w + 8 j (chunk q / SG by scalar offset, slot group q % SG per lane) and U pieces w + 8 j
    // (16-channel half w & 1, xi = (w >> 1) + 4 j by scalar offset)
    unsigned pxo[APL], vb0 = 0;
    const unsigned xi_stride4 = 4u * (unsigned)p.Cout * (unsigned)p.K * 4u;
    auto offsets = [&](int tbx, int tby) {
        const int co0 = tbx * TN;
        const int tile0 = tby * TILES_WIDE;
        const int img0 = tile0 >> p.lgTPI;
        const int trow0 = (tile0 & (p.TPI - 1)) >> LGTW;
        const int y_first = 2 * trow0 - 1;
#pragma unroll
        for (int j = 0; j < APL; ++j) {
            const int g = (wave + 8 * j) % SG;
            const int s = g * 64 + lane;
            const int rr = s / P2, rem = s - rr * P2;
            const int par = rem >= P ? 1 : 0, idx = rem - par * P;
            int il = (int)((float)rr * p.invRIN);
            if (il * p.RIN > rr) --il; else if ((il + 1) * p.RIN <= rr) ++il;
            const int r = rr - il * p.RIN;
            const int yy = y_first + r, xx = 2 * idx - 1 + par;
            const int img = img0 + il;
            unsigned vo = OOB;
            if (il < p.NIW && img < p.nimg && (unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W)
                vo = (unsigned)((((long long)img * p.H + yy) * p.W + xx) * p.ldx) * 4u;
            pxo[j] = vo;
        }
        const int row = lane >> 2, c = (lane & 3) ^ b_swz(row);
        const int co = co0 + (wave & 1) * 16 + row;
        vb0 = co < p.Cout ? (unsigned)((((long long)(wave >> 1) * p.Cout + co) * p.K + c * 4) * 4) : OOB;
    };
    // piece i of the 4 U + APL patch pieces this wave owes to K tile kt (stage st)
    auto issue_piece = [&](int i, int kt, int st) {
        if (i < 4) {
            const __amdgpu_buffer_rsrc_t rs = make_rsrc(p.U + kt * KT);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(sB + st * B_STAGE + (wave + 8 * i) * 256), 16, (int)vb0, (int)(i * xi_stride4), 0, 0);
        } else {
            const int j = i - 4, q = wave + 8 * j;
            const __amdgpu_buffer_rsrc_t rs = make_rsrc(p.x + kt * KT);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(sA + st * A_STAGE + q * 256), 16, (int)pxo[j], (q / SG) * 16, 0, 0);
        }
    };
    constexpr int NPIECE = 4 + APL;

    auto poff = [](int q) { return ((q & 1) * P + (q >> 1)) * 4; };
    const int boff = li * KT + ((lq ^ b_swz(li)) << 2);

    int tbx = 0, tby = 0, nbx = 0, nby = 0;
    bool have = item_of(0, tbx, tby);
    int gk = 0;                                                     // K tiles done by this workgroup (stage parity)
    if (have) {
        offsets(tbx, tby);
#pragma unroll
        for (int i = 0; i < NPIECE; ++i) issue_piece(i, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int n = 0; have; ++n) {
        const int co0 = tbx * TN;
        const int tile0 = tby * TILES_WIDE;
        const int trow0 = (tile0 & (p.TPI - 1)) >> LGTW;
        const int tl = 16 * wave + li;
        const int tile = tile0 + tl;
        const int il = tl >> p.lgTPI;
        const int tin = (tile & (p.TPI - 1));
        const int ty = tin >> LGTW, tx = tin & (TW - 1);
        const int slot0 = ((il * p.RIN + 2 * (ty - trow0)) * 2) * P + tx;
        const float* pbase = sA + (lq * NS + slot0) * 4;
        const bool next = item_of(n + 1, nbx, nby);

        f32x4 acc[16][2];
#pragma unroll
        for (int xi = 0; xi < 16; ++xi)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) acc[xi][cb] = f32x4{0.f, 0.f, 0.f, 0.f};

        for (int kt = 0; kt < nkt; ++kt, ++gk) {
            const int st = gk & 1;
            const bool last = kt == nkt - 1;
            if (last && next) offsets(nbx, nby);                    // (this item's DMA is all issued: the registers are free)
            const bool dma = !last || next;
            const int nkt_i = last ? 0 : kt + 1;
            const float* pa = pbase + st * A_STAGE;
            const float* bs = sB + st * B_STAGE + boff;
            // Patch rows 0 and 2 first: tr[0] = d0 - d2 is all the first four steps need; rows 1 and 3 arrive behind their MFMAs and
            // become tr[1] = d1 + d2 (into d[0]), tr[2] = d2 - d1 (into d[2]), tr[3] = d1 - d3 (into d[1]) between steps 3 and 4.
            // V of step xi + 1 is formed BEFORE the MFMAs of step xi (gfx90a+ wants two wait states between a VALU write and the MFMA
            // that reads it).  (Tried: v_pk_add_f32 with the negate modifier through inline asm -- hipcc scalarises float4 subtractions
            // into four v_sub_f32 -- gains 4 % only as long as it skips those wait states, i.e. only while it computes garbage.)
            f32x4 d[4][4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                d[0][q] = *reinterpret_cast<const f32x4*>(pa + 0 * P2 * 4 + poff(q));
                d[2][q] = *reinterpret_cast<const f32x4*>(pa + 2 * P2 * 4 + poff(q));
            }
            f32x4 ub[2][2];
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) ub[0][cb] = *reinterpret_cast<const f32x4*>(bs + cb * 256);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                d[1][q] = *reinterpret_cast<const f32x4*>(pa + 1 * P2 * 4 + poff(q));
                d[3][q] = *reinterpret_cast<const f32x4*>(pa + 3 * P2 * 4 + poff(q));
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) d[0][q] -= d[2][q];
            auto vof = [&](int xi) -> f32x4 {                       // V[a][b] from the physical row that holds tr[a]
                const int a = xi >> 2, b = xi & 3, r = a == 0 ? 0 : (a == 1 ? 0 : (a == 2 ? 2 : 1));
                return b == 0 ? d[r][0] - d[r][2] : (b == 1 ? d[r][1] + d[r][2] : (b == 2 ? d[r][2] - d[r][1] : d[r][1] - d[r][3]));
            };
            f32x4 Vn = vof(0);
#pragma unroll
            for (int xi = 0; xi < 16; ++xi) {
                const f32x4 V = Vn;
                if (xi < 15) {
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb) ub[(xi + 1) & 1][cb] = *reinterpret_cast<const f32x4*>(bs + ((xi + 1) * 2 + cb) * 256);
                }
                if (dma && xi < NPIECE) issue_piece(xi, nkt_i, st ^ 1);
                if (xi == 3) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4 d1 = d[1][q], d2 = d[2][q];
                        d[0][q] = d1 + d2; d[2][q] = d2 - d1; d[1][q] = d1 - d[3][q];
                    }
                }
                if (xi < 15) Vn = vof(xi + 1);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[xi][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ub[xi & 1][0][j], V[j], acc[xi][0], 0, 0, 0);
                    acc[xi][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ub[xi & 1][1][j], V[j], acc[xi][1], 0, 0, 0);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }

        // ---------------- epilogue: y = A^T M A (+ bias + residual), all 16 xi in the lane      A^T = [1 1 1 0 ; 0 1 -1 -1]
        int tby_e = tby, lane_e = lane;
        asm volatile("" : "+s"(tby_e));
        asm volatile("" : "+v"(lane_e));
        const int li_e = lane_e & 15, lq_e = lane_e >> 4;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            f32x4 s0[4], s1[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const f32x4 t1 = acc[4 * a + 1][cb] + acc[4 * a + 2][cb], t2 = acc[4 * a + 1][cb] - acc[4 * a + 2][cb];
                s0[a] = acc[4 * a][cb] + t1;
                s1[a] = t2 - acc[4 * a + 3][cb];
            }
            f32x4 F[2][2];
            F[0][0] = s0[0] + s0[1] + s0[2]; F[0][1] = s1[0] + s1[1] + s1[2];
            F[1][0] = s0[1] - s0[2] - s0[3]; F[1][1] = s1[1] - s1[2] - s1[3];
            wino_store_outputs<TW, STATS>(p, F, tby_e * TILES_WIDE + 16 * wave, li_e, co0 + 16 * cb + 4 * lq_e);
        }
        have = next; tbx = nbx; tby = nby;
    }
    if (VD_PROBE_BUILD && p.probe && lane == 0) {
        // same record as the narrow kernel's probe (8 x u64 per wave): start, -, -, end, -, realtime at start, K tiles, realtime at end
        unsigned long long* o = p.probe + ((unsigned long long)blockIdx.x * 8 + wave) * 8;
        o[0] = pt0; o[1] = pt0; o[2] = pt0; o[3] = __builtin_amdgcn_s_memtime(); o[4] = 0; o[5] = pr0; o[6] = (unsigned long long)nkt;
        o[7] = __builtin_amdgcn_s_memrealtime();
    }
}

// ---- weight transform: uf[xi][co][ci] = (G w[co][ci] G^T)[xi] ; ud[xi][ci][co] = (G rot180(w[co][ci]) G^T)[xi]
// G = [1 0 0 ; .5 .5 .5 ; .5 -.5 .5 ; 0 0 1]
__device__ __forceinline__ void g_transform(const float (&w)[3][3], float (&u)[4][4]) {
    float t[4][3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        t[0][j] = w[0][j];
        t[1][j] = 0.5f * (w[0][j] + w[1][j] + w[2][j]);
        t[2][j] = 0.5f * (w[0][j] - w[1][j] + w[2][j]);
        t[3][j] = w[2][j];
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        u[a][0] = t[a][0];
        u[a][1] = 0.5f * (t[a][0] + t[a][1] + t[a][2]);
        u[a][2] = 0.5f * (t[a][0] - t[a][1] + t[a][2]);
        u[a][3] = t[a][2];
    }
}

__device__ __forceinline__ void wino_pack_one(const float* w, float* uf, float* ud, int Cout, int Cin, long long idx) {
    if (idx >= (long long)Cout * Cin) return;
    const int co = idx / Cin, ci = idx % Cin;
    float k[3][3], u[4][4];
    const float* src = w + idx * 9;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) k[i][j] = src[i * 3 + j];
    if (uf) {
        g_transform(k, u);
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) uf[((long long)(4 * a + b) * Cout + co) * Cin + ci] = u[a][b];
    }
    if (ud) {
        float kr[3][3];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) kr[i][j] = k[2 - i][2 - j];
        g_transform(kr, u);
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) ud[((long long)(4 * a + b) * Cin + ci) * Cout + co] = u[a][b];
    }
}

__global__ void wino_pack_kernel(const float* w, float* uf, float* ud, int Cout, int Cin) {
    wino_pack_one(w, uf, ud, Cout, Cin, (long long)blockIdx.x * blockDim.x + threadIdx.x);
}

// 16 (co) x 16 (ci) tile per 256-thread block: uf rows are written 16 consecutive ci at a time; the ud values go through LDS so
// that its rows (consecutive co) are written 16 at a time too (the thread-per-element form writes ud with a 4-byte-per-KB stride
// and moved 1 GB at 1.2 TB/s: 0.8 ms per train step)
__device__ __forceinline__ void wino_pack_tile(const float* w, float* uf, float* ud, int Cout, int Cin, int tile) {
    __shared__ float sh[16][16][17];                // [xi][co_l][ci_l (+1 pad)]
    const int tci = Cin >> 4;
    const int co0 = (tile / tci) * 16, ci0 = (tile % tci) * 16;
    const int a_ = threadIdx.x >> 4, b_ = threadIdx.x & 15;
    {
        const int co = co0 + a_, ci = ci0 + b_;
        float k[3][3], u[4][4];
        const float* src = w + ((long long)co * Cin + ci) * 9;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) k[i][j] = src[i * 3 + j];
        if (uf) {
            g_transform(k, u);
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) uf[((long long)(4 * a + b) * Cout + co) * Cin + ci] = u[a][b];
        }
        if (ud) {
            float kr[3][3];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) kr[i][j] = k[2 - i][2 - j];
            g_transform(kr, u);
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) sh[4 * a + b][a_][b_] = u[a][b];
        }
    }
    if (ud) {
        __syncthreads();
        const int ci = ci0 + a_, co = co0 + b_;      // thread (ci_l = a_, co_l = b_)
#pragma unroll
        for (int xi = 0; xi < 16; ++xi) ud[((long long)xi * Cin + ci) * Cout + co] = sh[xi][b_][a_];
    }
}

// all 3x3 kernels of a network in one launch; items: 8 x int64 per tensor {w, uf, ud, Cout, Cin, tiled, -, first block}
// tiled = 1: blocks are 16x16 (co, ci) tiles (Cout, Cin multiples of 16); 0: 256 (co, ci) pairs per block
__global__ __launch_bounds__(256) void wino_pack_batched_kernel(const long long* items, int n) {
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (items[8 * mid + 7] <= (long long)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const long long* it = items + 8 * lo;
    const float* w = reinterpret_cast<const float*>(it[0]);
    float* uf = reinterpret_cast<float*>(it[1]);
    float* ud = reinterpret_cast<float*>(it[2]);
    const int local = (int)((long long)blockIdx.x - it[7]);
    if (it[5]) wino_pack_tile(w, uf, ud, (int)it[3], (int)it[4], local);
    else wino_pack_one(w, uf, ud, (int)it[3], (int)it[4], (long long)local * blockDim.x + threadIdx.x);
}

#ifdef VD_PROBES
unsigned long long* g_probe = nullptr;  // probe library only (vd_wino_set_probe)
#else
constexpr unsigned long long* g_probe = nullptr;
#endif
thread_local int g_last_wino = 0;       // (TW * 1000 + NS) * 2 + stats of the calling thread's last vd_conv3x3_wino launch (negative: wide form)
thread_local int g_last_wgrad = 0;      // (TWS * 1000 + slabs) * 2 + dbias of the calling thread's last vd_conv3x3_wgrad_wino launch

inline int ilog2(int v) { return 31 - __builtin_clz((unsigned)v); }
inline bool pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

struct Plan { bool ok; int TW, TH, TPI, P, RIN, NTR, NS, nwgimg; };

Plan plan_of(int H, int W) {
    Plan g = {};
    if (H % 2 || W % 2) return g;
    g.TW = W / 2; g.TH = H / 2; g.TPI = g.TW * g.TH;
    if (!pow2(g.TW) || !pow2(g.TH) || g.TW < 4 || g.TW > 64 || g.TPI < 16) return g;
    if (g.TPI >= TILES_WG) { g.nwgimg = 1; g.NTR = TILES_WG / g.TW; if (g.NTR < 1 || g.NTR > g.TH) return g; }
    else { g.nwgimg = TILES_WG / g.TPI; g.NTR = g.TH; }
    g.RIN = 2 * g.NTR + 2;
    g.P = g.TW >= 16 ? g.TW + 1 : (g.TW == 8 ? 10 : 5);       // pitch chosen so 16 consecutive tiles read 16 distinct bank quads
    const int slots = g.nwgimg * g.RIN * 2 * g.P;
    g.NS = (slots + 127) / 128 * 128;
    g.ok = g.NS <= 640;
    return g;
}

// geometry of the wide form (128 tiles per item): TW in {8, 16, 32} with at most 768 patch slots per chunk
Plan plan_wide(int H, int W) {
    Plan g = {};
    if (H % 2 || W % 2) return g;
    g.TW = W / 2; g.TH = H / 2; g.TPI = g.TW * g.TH;
    if (!pow2(g.TW) || !pow2(g.TH) || (g.TW != 8 && g.TW != 16 && g.TW != 32) || g.TPI < 16) return g;
    if (g.TPI >= TILES_WIDE) { g.nwgimg = 1; g.NTR = TILES_WIDE / g.TW; if (g.NTR < 1 || g.NTR > g.TH) return g; }
    else { g.nwgimg = TILES_WIDE / g.TPI; g.NTR = g.TH; }
    g.RIN = 2 * g.NTR + 2;
    g.P = g.TW >= 16 ? g.TW + 1 : 10;
    const int slots = g.nwgimg * g.RIN * 2 * g.P;
    g.NS = (slots + 127) / 128 * 128;
    g.ok = g.NS <= 768;
    return g;
}

}  // namespace

extern "C" int vd_conv3x3_wino_supported(int32_t nimg, int32_t H, int32_t W, int32_t Cin, int32_t Cout, int64_t ldx, int64_t ldy,
                                         int64_t ldres) {
    const Plan g = plan_of(H, W);
    if (!g.ok || Cin % KT || Cout % 4 || nimg <= 0) return 0;
    const long long px = (long long)nimg * H * W;
    const long long lim = 0x7FFFFFF0LL / 4;
    if (px * ldx >= lim || px * ldy >= lim || (ldres > 0 && px * ldres >= lim) || 16LL * Cout * Cin >= lim) return 0;
    if (ldx % 4 || ldy % 4 || ldres % 4) return 0;
    return 1;
}

extern "C" int vd_conv3x3_wino(const float* xin, int64_t ldx, const float* U, const float* bias, const float* res, int64_t ldres,
                               float* y, int64_t ldy, int32_t nimg, int32_t H, int32_t W, int32_t Cin, int32_t Cout,
                               float* stats_part, void* stream) {
    VD_REQUIRE(xin && U && y, "vd_conv3x3_wino: null operand");
    VD_REQUIRE(vd_conv3x3_wino_supported(nimg, H, W, Cin, Cout, ldx, ldy, res ? ldres : 0),
               "vd_conv3x3_wino: unsupported geometry nimg=%d H=%d W=%d Cin=%d Cout=%d (use vd_conv3x3)", nimg, H, W, Cin, Cout);
    VD_REQUIRE(vd_aligned16(xin) && vd_aligned16(U) && vd_aligned16(y) && (!res || vd_aligned16(res)) && (!bias || vd_aligned16(bias)),
               "vd_conv3x3_wino: operands must be 16-byte aligned");
    const int ncu = vd_persistent_cus();      // persistent workgroups: one per CU (the LDS footprint admits no second one), minus the reserved CUs
    {
        // wide form (128-tile items, ~5 % faster per unit of work: same-box A/B in tests/perf_wino.py) wherever its items fill the
        // residency rounds at least as well as the 64-tile items do: rounds x 2 x 0.95 against the narrow form's rounds
        // (576 -> 576 @16x16, 4.5 rounds of wide items, measured 6 % slower than 9 full narrow rounds).  VD_WINO_WIDE=0 / 1 forces.
        static const int wide_env = getenv("VD_WINO_WIDE") ? atoi(getenv("VD_WINO_WIDE")) : -1;
        const Plan gw = plan_wide(H, W);
        const long long ntiles = (long long)nimg * gw.TPI;
        const long long ncbw = (Cout + TN - 1) / TN;
        const long long witems = gw.ok ? ncbw * ((ntiles + TILES_WIDE - 1) / TILES_WIDE) : 0;
        const long long nitems = ncbw * ((ntiles + TILES_WG - 1) / TILES_WG);
        const long long wrounds = (witems + ncu - 1) / ncu, nrounds = (nitems + ncu - 1) / ncu;
        const bool wide = wide_env >= 0 ? wide_env != 0 : (witems >= ncu && 1.9 * (double)wrounds <= (double)nrounds);
        if (wide && gw.ok) {
            WinoArgs a = {};
            a.x = xin; a.ldx = ldx; a.U = U; a.bias = bias; a.res = res; a.ldr = ldres; a.y = y; a.ldy = ldy; a.stats = stats_part;
            a.nimg = nimg; a.H = H; a.W = W; a.K = Cin; a.Cout = Cout;
            a.TW = gw.TW; a.TH = gw.TH; a.TPI = gw.TPI; a.P = gw.P; a.RIN = gw.RIN; a.NTR = gw.NTR; a.NIW = gw.nwgimg;
            a.lgTW = ilog2(gw.TW); a.lgTPI = ilog2(gw.TPI); a.ntiles = (int)ntiles;
            a.invP2 = 1.0f / (float)(2 * gw.P); a.invRIN = 1.0f / (float)gw.RIN;
            a.ncb = (Cout + TN - 1) / TN; a.ntg = (int)((ntiles + TILES_WIDE - 1) / TILES_WIDE);
            a.probe = g_probe;
            const dim3 grid((unsigned)(witems < ncu ? witems : ncu)), blk(WIDE_THREADS);
            hipStream_t st = (hipStream_t)stream;
#define VD_WIDE_LAUNCH(TWV, NSV)                                                                                                   \
            do {                                                                                                                    \
                g_last_wino = -(((TWV) * 1000 + (NSV)) * 2 + (stats_part ? 1 : 0));      /* negative: the wide form */              \
                if (stats_part) hipLaunchKernelGGL((wino_conv_wide_kernel<TWV, NSV, true>), grid, blk, 0, st, a);                    \
                else hipLaunchKernelGGL((wino_conv_wide_kernel<TWV, NSV, false>), grid, blk, 0, st, a);                              \
            } while (0)
            if (gw.TW == 16 && gw.NS <= 640) VD_WIDE_LAUNCH(16, 640);
            else if (gw.TW == 16) VD_WIDE_LAUNCH(16, 768);
            else if (gw.TW == 8) VD_WIDE_LAUNCH(8, 768);
            else VD_WIDE_LAUNCH(32, 768);
#undef VD_WIDE_LAUNCH
            VD_LAUNCH_CHECK("wino_conv_wide_kernel");
            vd_g_last_tile = ((16 * 1000) + 128) * 1000 + TN;
            return 0;
        }
    }
    const Plan g = plan_of(H, W);
    WinoArgs a = {};
    a.x = xin; a.ldx = ldx; a.U = U; a.bias = bias; a.res = res; a.ldr = ldres; a.y = y; a.ldy = ldy; a.stats = stats_part;
    a.nimg = nimg; a.H = H; a.W = W; a.K = Cin; a.Cout = Cout;
    a.TW = g.TW; a.TH = g.TH; a.TPI = g.TPI; a.P = g.P; a.RIN = g.RIN; a.NTR = g.NTR; a.NIW = g.nwgimg;
    a.lgTW = ilog2(g.TW); a.lgTPI = ilog2(g.TPI); a.ntiles = nimg * g.TPI;
    a.invP2 = 1.0f / (float)(2 * g.P); a.invRIN = 1.0f / (float)g.RIN;
    a.ncb = (Cout + TN - 1) / TN; a.ntg = (a.ntiles + TILES_WG - 1) / TILES_WG;
    const long long items = (long long)a.ncb * a.ntg;
    VD_REQUIRE(items < (1LL << 30), "vd_conv3x3_wino: too many work items");
    const dim3 grid((unsigned)(items < ncu ? items : ncu));
    hipStream_t st = (hipStream_t)stream;
    const dim3 blk(WINO_THREADS);
#ifdef VD_PROBES
    a.probe = g_probe;
    { static const bool light = getenv("VD_WINO_PROBE_LIGHT") != nullptr; a.probe_light = light ? 1 : 0; }
#define VD_WINO_LAUNCH(TWV, NSV)                                                                                                   \
    do {                                                                                                                            \
        g_last_wino = ((TWV) * 1000 + (NSV)) * 2 + (stats_part ? 1 : 0);                                                            \
        if (g_probe) {      /* timing probe (tests/probe/wino_phases.py): per-wave phase stamps */                                  \
            if (stats_part) hipLaunchKernelGGL((wino_conv_kernel<TWV, NSV, true, true>), grid, blk, 0, st, a);                       \
            else hipLaunchKernelGGL((wino_conv_kernel<TWV, NSV, false, true>), grid, blk, 0, st, a);                                 \
        } else if (stats_part) hipLaunchKernelGGL((wino_conv_kernel<TWV, NSV, true>), grid, blk, 0, st, a);                          \
        else hipLaunchKernelGGL((wino_conv_kernel<TWV, NSV, false>), grid, blk, 0, st, a);                                           \
    } while (0)
    // timing experiments (WRONG results): probe library only -- the product library has no such instantiation and reads no such knob
    static const int exp_mode = getenv("VD_WINO_EXP") ? atoi(getenv("VD_WINO_EXP")) : 0;
    if (exp_mode && g.TW == 16 && g.NS <= 384 && !stats_part) {
#define VD_WINO_EXP_LAUNCH(E)                                                                                                      \
        do {                                                                                                                        \
            if (g_probe) hipLaunchKernelGGL((wino_conv_kernel<16, 384, false, true, E>), grid, blk, 0, st, a);                      \
            else hipLaunchKernelGGL((wino_conv_kernel<16, 384, false, false, E>), grid, blk, 0, st, a);                             \
        } while (0)
        if (exp_mode == 1) VD_WINO_EXP_LAUNCH(1);
        else if (exp_mode == 2) VD_WINO_EXP_LAUNCH(2);
        else if (exp_mode == 3) VD_WINO_EXP_LAUNCH(3);
        else if (exp_mode == 5) VD_WINO_EXP_LAUNCH(5);
        else if (exp_mode == 6) VD_WINO_EXP_LAUNCH(6);
        else VD_WINO_EXP_LAUNCH(4);
#undef VD_WINO_EXP_LAUNCH
        VD_LAUNCH_CHECK("wino_conv_kernel(exp)");
        g_last_wino = 0;
        return 0;
    }
#else
#define VD_WINO_LAUNCH(TWV, NSV)                                                                                                   \
    do {                                                                                                                            \
        g_last_wino = ((TWV) * 1000 + (NSV)) * 2 + (stats_part ? 1 : 0);                                                            \
        if (stats_part) hipLaunchKernelGGL((wino_conv_kernel<TWV, NSV, true>), grid, blk, 0, st, a);                                 \
        else hipLaunchKernelGGL((wino_conv_kernel<TWV, NSV, false>), grid, blk, 0, st, a);                                           \
    } while (0)
#endif
    // (tiles per row, patch slots): the square images of the shipped configs take the first form of each row
    if (g.TW == 16 && g.NS <= 384) VD_WINO_LAUNCH(16, 384);
    else if (g.TW == 8 && g.NS <= 384) VD_WINO_LAUNCH(8, 384);
    else if (g.TW == 4 && g.NS <= 512) VD_WINO_LAUNCH(4, 512);
    else if (g.TW == 32 && g.NS <= 512) VD_WINO_LAUNCH(32, 512);
    else if (g.TW == 64) VD_WINO_LAUNCH(64, 640);
    else if (g.TW == 16) VD_WINO_LAUNCH(16, 640);
    else if (g.TW == 8) VD_WINO_LAUNCH(8, 640);
    else if (g.TW == 4) VD_WINO_LAUNCH(4, 640);
    else VD_WINO_LAUNCH(32, 640);
#undef VD_WINO_LAUNCH
    VD_LAUNCH_CHECK("wino_conv_kernel");
    vd_g_last_tile = ((16 * 1000) + 128) * 1000 + TN;          // (chunk of the statistics = 64 pixels = BM / 2 with BM = 128)
    return 0;
}

/* timing probe only: device buffer of 64 u64 per workgroup (8 per wave: start, loop start, loop end, end, cycles at the tile
 * barrier, 100 MHz realtime at start, K tiles, realtime at end), or NULL to switch the probe off */
#ifdef VD_PROBES
extern "C" int vd_wino_set_probe(unsigned long long* buf) { g_probe = buf; return 0; }
#endif
extern "C" int vd_wino_last_kernel(void) { return g_last_wino; }
extern "C" int vd_wino_wgrad_last_kernel(void) { return g_last_wgrad; }

extern "C" int vd_wino_pack(const float* w_oihw, int32_t Cout, int32_t Cin, float* uf, float* ud, void* stream) {
    VD_REQUIRE(w_oihw && (uf || ud), "vd_wino_pack: null pointer");
    const long long tot = (long long)Cout * Cin;
    hipLaunchKernelGGL(wino_pack_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w_oihw, uf, ud, Cout, Cin);
    VD_LAUNCH_CHECK("wino_pack_kernel");
    return 0;
}

extern "C" int vd_wino_pack_batched(const int64_t* items_dev, int32_t n, int64_t total_blocks, void* stream) {
    VD_REQUIRE(items_dev && n > 0 && total_blocks > 0 && total_blocks < (1LL << 31), "vd_wino_pack_batched: bad table");
    hipLaunchKernelGGL(wino_pack_batched_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const long long*>(items_dev), n);
    VD_LAUNCH_CHECK("wino_pack_batched_kernel");
    return 0;
}

// ==================================================================================================================
// Weight gradient in the Winograd domain (autograd of modules.py:141-144 with respect to the kernel):
//   dU[xi][co][ci] = sum_t dM[xi][t][co] V[xi][t][ci]        dM = A dY_t A^T (2x2 -> 4x4),  V = B^T d_t B as in the forward pass
//   dw[co][ci]     = G^T dU[.][co][ci] G                     (4x4 -> 3x3)
// i.e. 16 GEMMs whose K index is the tile (pixels / 4): 2.25x fewer MFMA cycles than the implicit GEMM over pixels.
// Workgroup = 8 waves = a 64 (co) x 64 (ci) block of all 16 xi over a range of tile groups (split-K slab); wave w owns xi half
// ah = w >> 2 (a in {2ah, 2ah+1}), co half (w >> 1) & 1 and ci half w & 1: 8 xi x (2x2 blocks of 16x16) = 128 accumulator registers.
// One stage = 16 consecutive tiles: their 64 dY pixels and their input patch (both with all 64 channels of the block, 256 B per
// pixel slot) go HBM -> LDS by DMA; a K step is 4 tiles (the k index of v_mfma_f32_16x16x4_f32 = lane >> 4 picks the tile), each
// lane reads the 2 channels it feeds as MFMA rows / columns with ds_read_b64 (the two 128-byte halves of odd slots are swapped
// so that the tiles of a lane pair fall on different banks), transforms them in registers and issues 32 MFMAs.
// Signs: A has a row (0 -1) and G-side sums use -dU[.][3]; both are folded into the epilogue (the operands carry |coefficients|).
// Epilogue: G^T . G in registers; each xi-half wave writes its partial 3x3 block to its own slab plane; wino_wgrad_reduce sums the
// planes of all slabs in a fixed order into OIHW (bitwise reproducible, no atomics).
constexpr int WG_T = 16;                    // tiles per stage
constexpr int WG_NSX = 136;                 // patch slots per stage (4 rows x (17 odd + 17 even) is the largest geometry)
constexpr int WG_DY_BYTES = 64 * 256, WG_X_BYTES = WG_NSX * 256, WG_STAGE_BYTES = WG_DY_BYTES + WG_X_BYTES;
constexpr int WG_PIECES = WG_STAGE_BYTES / 1024;    // 50
constexpr int WG_PPW = (WG_PIECES + 7) / 8;         // 7 DMA pieces per wave (the last round is partly idle)

struct WgradArgs {
    const float* x; long long ldx; const float* dy; long long lddy;
    float* slabs;                      // [S][9][Cout][Cin]
    float* cpart;                      // [S][Cout] bias-gradient partials, or NULL
    int nimg, H, W, Cin, Cout;
    int TW, TH, TPI, lgTW, lgTPI;
    int nstages, per;                  // stages in all, stages per slab
    unsigned long long* probe;         // vd_wino_set_probe: 4 x u64 per workgroup (s_memtime / s_memrealtime at start and end), or NULL
};

// TWS = tiles of a stage per tile row = min(W/2, 16); the stage then spans NR = 16 / TWS tile rows of one image
template <int TWS, bool DBIAS>
__global__ __launch_bounds__(512) void wino_wgrad_kernel(const WgradArgs p) {
    constexpr int NR = WG_T / TWS, P = TWS + 1, P2 = 2 * P, LG = TWS == 16 ? 4 : (TWS == 8 ? 3 : 2);
    static_assert((2 * NR + 2) * P2 <= WG_NSX, "patch image too large");
    unsigned long long pt0 = 0, pr0 = 0;
    if (VD_PROBE_BUILD && p.probe) { pt0 = __builtin_amdgcn_s_memtime(); pr0 = __builtin_amdgcn_s_memrealtime(); }
    __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * WG_STAGE_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lm = lane & 15, kq = lane >> 4;
    const int ah = wave >> 2, mh = (wave >> 1) & 1, nh = wave & 1;
    // XCD-aware order: workgroups are dealt round-robin over the 8 XCDs (linear id % 8).  All (co, ci) blocks of one slab read
    // the same tiles of x and dY: put them on ONE XCD (slab z lives on XCD z % 8) so those tiles cross the fabric once per slab
    // instead of once per channel block (PMC: 1.02 GB per launch against 0.27 GB of x + dY).  Any placement is correct.
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    {
        const unsigned nb = gridDim.x * gridDim.y, S = gridDim.z, T = nb * S;
        const unsigned lin = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        if (S % 8 == 0) {
            const unsigned xcd = lin % 8, slot = lin / 8;           // slot-th workgroup of that XCD
            const unsigned zz = xcd + 8 * (slot / nb), blk = slot % nb;
            bz = (int)zz; bx = (int)(blk % gridDim.x); by = (int)(blk / gridDim.x);
        }
        (void)T;
    }
    const int ci0 = bx * 64, co0 = by * 64, z = bz;
    const int st_begin = z * p.per, st_end = min(st_begin + p.per, p.nstages);

    // ---------------- DMA: per-lane source offsets relative to the stage (constant) and border classes
    // piece q < 16: dY, slot = 4 q + lane / 16 = (2u+v) * 16 + t ; piece q >= 16: patch, slot = 4 (q - 16) + lane / 16.
    // A slot is one pixel x 64 channels = 16 granules of 16 bytes; odd slots hold their two 128-byte halves swapped.
    unsigned vo[WG_PPW];               // byte offset from the stage's descriptor base, or OOB (never valid)
    unsigned bits[WG_PPW];             // border class of a patch pixel: 1 left pad column, 2 right pad column, 4 top pad row, 8 bottom pad row
#pragma unroll
    for (int j = 0; j < WG_PPW; ++j) {
        const int q = wave + 8 * j;
        unsigned v = OOB, bb = 0;
        if (q < WG_PIECES) {
            const int slot = (q < 16 ? 4 * q : 4 * (q - 16)) + (lane >> 4);
            const int gl = (lane & 15) ^ ((slot & 1) << 3);               // logical granule (4 channels) this lane fetches
            if (q < 16) {
                const int uv = slot >> 4, t = slot & 15;
                const int dty = t >> LG, dtx = t & (TWS - 1);
                if (co0 + gl * 4 < p.Cout)
                    v = (unsigned)((((long long)(2 * dty + (uv >> 1)) * p.W + 2 * dtx + (uv & 1)) * p.lddy + gl * 4) * 4);
            } else {
                const int rr = slot / P2, rem = slot - rr * P2;           // local input row (0 = one above the stage), place in the row pair
                const int par = rem >= P ? 1 : 0, idx = rem - par * P;
                const int xr = 2 * idx + par;                             // column, 0 = one left of the stage  (par 0: odd image columns)
                if (rr < 2 * NR + 2 && ci0 + gl * 4 < p.Cin) {
                    v = (unsigned)((((long long)rr * p.W + xr) * p.ldx + gl * 4) * 4);   // descriptor base = pixel (-1, -1) of the stage
                    bb = (xr == 0 ? 1u : 0u) | (xr > 2 * TWS ? 2u : 0u) | (rr == 0 ? 4u : 0u) | (rr > 2 * NR ? 8u : 0u);
                }
            }
        }
        vo[j] = v; bits[j] = bb;
    }
    auto issue = [&](int st, int buf, bool live) {
        // stage st = tiles [16 st, 16 st + 16): image, first tile row / column (uniform)
        const int tile = st * WG_T;
        const int img = tile >> p.lgTPI, tin = tile & (p.TPI - 1);
        const int ty0 = tin >> p.lgTW, tx0 = tin & (p.TW - 1);
        const unsigned sbits = (tx0 == 0 ? 1u : 0u) | (tx0 + TWS == p.TW ? 2u : 0u) | (ty0 == 0 ? 4u : 0u) | (ty0 + NR == p.TH ? 8u : 0u);
        const long long pix = ((long long)img * p.H + 2 * ty0) * p.W + 2 * tx0;
        const __amdgpu_buffer_rsrc_t rd = make_rsrc(p.dy + pix * p.lddy + co0, live ? (int)OOB : 0);
        // (the patch descriptor starts one row and one column early: it may point in front of the tensor, the lanes that would
        //  read there are exactly the pad lanes, which are switched off)
        const __amdgpu_buffer_rsrc_t rx = make_rsrc(p.x + (pix - p.W - 1) * p.ldx + ci0, live ? (int)OOB : 0);
        unsigned char* dst = smem + buf * WG_STAGE_BYTES;
#pragma unroll
        for (int j = 0; j < WG_PPW; ++j) {
            const int q = wave + 8 * j;
            if (j < 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rd, (lds_ptr_t)(dst + q * 1024), 16, (int)vo[j], 0, 0, 0);
            else if (q < WG_PIECES) {
                const unsigned v = (bits[j] & sbits) ? OOB : vo[j];
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr_t)(dst + q * 1024), 16, (int)v, 0, 0, 0);
            }
        }
    };

    // ---------------- LDS reads.  K step ks feeds tile t = 4 ks + kq of the stage to lane group kq.
    // dY slot = (2u+v) * 16 + t;  patch slot of position (pr, q) of this xi half = ((2 dty + ah + pr) * 2 + (q & 1)) * P + dtx + (q >> 1).
    // Lane part (kq, halves, channel pair) in two base registers per operand, the rest is an immediate offset.
    const int dbase = kq * 256 + ((mh ^ (kq & 1)) << 7) + lm * 8;
    const int xb_even = WG_DY_BYTES + (kq + ah * P2) * 256 + ((nh ^ (kq & 1)) << 7) + lm * 8;
    const int xb_odd = WG_DY_BYTES + (kq + ah * P2) * 256 + ((nh ^ (kq & 1) ^ 1) << 7) + lm * 8;

    f32x4 acc[8][2][2];
#pragma unroll
    for (int xi = 0; xi < 8; ++xi)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[xi][a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x2 bsum = {0.f, 0.f};
    const bool do_bias = DBIAS && ah == 0 && nh == 0 && bx == 0;

    auto kstep = [&](const unsigned char* sb, const int ks) {
        constexpr int dummy = 0; (void)dummy;
        const int dty = (TWS == 16) ? 0 : (TWS == 8 ? (ks >> 1) : ks);
        const int dtx0 = (TWS == 16) ? 4 * ks : (TWS == 8 ? 4 * (ks & 1) : 0);
        f32x2 dyv[2][2], L[3][4];
#pragma unroll
        for (int uv = 0; uv < 4; ++uv) dyv[uv >> 1][uv & 1] = *reinterpret_cast<const f32x2*>(sb + dbase + (uv * 16 + 4 * ks) * 256);
#pragma unroll
        for (int pr = 0; pr < 3; ++pr)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int C = ((2 * dty + pr) * 2 + (q & 1)) * P + dtx0 + (q >> 1);
                L[pr][q] = *reinterpret_cast<const f32x2*>(sb + ((C & 1) ? xb_odd : xb_even) + C * 256);
            }
        if (do_bias) bsum += (dyv[0][0] + dyv[0][1]) + (dyv[1][0] + dyv[1][1]);
        // A dY A^T restricted to this wave's rows a; |coefficients| only (signs: epilogue)
        f32x2 m[2][2], tr[2][4];
        if (ah == 0) {
#pragma unroll
            for (int v = 0; v < 2; ++v) { m[0][v] = dyv[0][v]; m[1][v] = dyv[0][v] + dyv[1][v]; }
#pragma unroll
            for (int q = 0; q < 4; ++q) { tr[0][q] = L[0][q] - L[2][q]; tr[1][q] = L[1][q] + L[2][q]; }
        } else {
#pragma unroll
            for (int v = 0; v < 2; ++v) { m[0][v] = dyv[0][v] - dyv[1][v]; m[1][v] = dyv[1][v]; }
#pragma unroll
            for (int q = 0; q < 4; ++q) { tr[0][q] = L[1][q] - L[0][q]; tr[1][q] = L[0][q] - L[2][q]; }
        }
        // all 16 operands first, then the 32 MFMAs: a VALU result consumed by the very next MFMA costs wait states
        f32x2 A4[2][4], V[2][4];
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            A4[a][0] = m[a][0]; A4[a][1] = m[a][0] + m[a][1]; A4[a][2] = m[a][0] - m[a][1]; A4[a][3] = m[a][1];
            V[a][0] = tr[a][0] - tr[a][2]; V[a][1] = tr[a][1] + tr[a][2]; V[a][2] = tr[a][2] - tr[a][1]; V[a][3] = tr[a][1] - tr[a][3];
        }
        __builtin_amdgcn_sched_barrier(0x0180 | 0x0004 | 0x0010 | 0x0020 | 0x0040 | 0x0200);     // (VALU and MFMA stay on their sides)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb)
                        acc[4 * a + b][mb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(A4[a][b][mb], V[a][b][nb], acc[4 * a + b][mb][nb], 0, 0, 0);
    };

    if (st_begin < st_end) {
        issue(st_begin, 0, true);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int st = st_begin; st < st_end; ++st) {
            const int buf = (st - st_begin) & 1;
            const unsigned char* sb = smem + buf * WG_STAGE_BYTES;
            // the two waves of a SIMD issue their DMA bursts half a stage apart (see wino_conv_kernel)
            if (ah == 0) issue(st + 1, buf ^ 1, st + 1 < st_end);
            kstep(sb, 0);
            kstep(sb, 1);
            if (ah != 0) issue(st + 1, buf ^ 1, st + 1 < st_end);
            kstep(sb, 2);
            kstep(sb, 3);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    }

    // ---------------- epilogue: partial G^T dU G of this wave's two rows a.   G^T = [1 .5 .5 0 ; 0 .5 -.5 0 ; 0 .5 .5 1]
    // true dU[a][b] = sa(a) sb(b) acc with sa(3) = sb(3) = -1 (folded signs)
    // lane (n = lm, rq = kq), register r: co = co0 + 32 mh + 2 (4 rq + r) + mb, ci = ci0 + 32 nh + 2 n + nb
    // The xi-half-1 wave of every (co half, ci half) pair hands its partial to its partner through LDS (the stages are dead: the K
    // loop ended on a barrier), which adds and writes ONE plane per slab -- half the plane traffic of writing both.
    float* plane = p.slabs + (long long)z * 9 * p.Cout * p.Cin;
    f32x2* xch = reinterpret_cast<f32x2*>(smem) + (size_t)(mh * 2 + nh) * (4 * 9 * 64);        // [pair][r][k = 3i+j][lane]
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
        f32x2 w9[4][3][3];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                float c[2][3];
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    const float d0 = acc[4 * a][mb][nb][r], d1 = acc[4 * a + 1][mb][nb][r], d2 = acc[4 * a + 2][mb][nb][r],
                                d3 = -acc[4 * a + 3][mb][nb][r];
                    const float h = 0.5f * (d1 + d2);
                    c[a][0] = d0 + h; c[a][1] = 0.5f * (d1 - d2); c[a][2] = h + d3;
                }
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    if (ah == 0) { const float h = 0.5f * c[1][j]; w9[r][0][j][nb] = c[0][j] + h; w9[r][1][j][nb] = h; w9[r][2][j][nb] = h; }
                    else { const float h = 0.5f * c[0][j]; w9[r][0][j][nb] = h; w9[r][1][j][nb] = -h; w9[r][2][j][nb] = h - c[1][j]; }
                }
            }
        }
        if (ah != 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int k = 0; k < 9; ++k) xch[(r * 9 + k) * 64 + lane] = w9[r][k / 3][k % 3];
        }
        __syncthreads();
        if (ah == 0) {
            const int ci = ci0 + 32 * nh + 2 * lm;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = co0 + 32 * mh + 2 * (4 * kq + r) + mb;
                if (co < p.Cout && ci < p.Cin) {
#pragma unroll
                    for (int k = 0; k < 9; ++k)
                        *reinterpret_cast<f32x2*>(plane + ((long long)k * p.Cout + co) * p.Cin + ci) = w9[r][k / 3][k % 3] + xch[(r * 9 + k) * 64 + lane];
                }
            }
        }
        __syncthreads();
    }
    if (DBIAS && do_bias && p.cpart) {
        // sum over the 4 tiles of a K step (lanes with equal lm): lanes kq = 0 write channels co0 + 32 mh + 2 lm + {0, 1}
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            float v = bsum[e];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            bsum[e] = v;
        }
        const int co = co0 + 32 * mh + 2 * lm;
        if (kq == 0 && co < p.Cout) *reinterpret_cast<f32x2*>(p.cpart + (long long)z * p.Cout + co) = bsum;
    }
    if (VD_PROBE_BUILD && p.probe && threadIdx.x == 0) {
        unsigned long long* o = p.probe + 4ull * ((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x);
        o[0] = pt0; o[1] = pr0; o[2] = __builtin_amdgcn_s_memtime(); o[3] = __builtin_amdgcn_s_memrealtime();
    }
}

// dw[co][ci][tap] (+)= sum over the S planes (fixed order); dbias[co] (+)= sum over the S partial rows
__global__ void wino_wgrad_reduce_kernel(const float* slabs, int planes, int Cout, int Cin, int Cout_w, int Cin_w, float* dw, int accumulate,
                                         const float* cpart, int S, float* dbias) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (dbias && idx < Cout_w) {
        float c = 0.f;
        for (int q = 0; q < S; ++q) c += cpart[(long long)q * Cout + idx];
        dbias[idx] = accumulate ? dbias[idx] + c : c;
    }
    if (idx >= (long long)Cout * Cin) return;
    const int co = idx / Cin, ci = idx % Cin;
    if (co >= Cout_w || ci >= Cin_w) return;
    const long long plane = 9LL * Cout * Cin;
    float s[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) s[t] = 0.f;
    for (int q = 0; q < planes; ++q) {
        const float* src = slabs + q * plane + idx;
#pragma unroll
        for (int t = 0; t < 9; ++t) s[t] += src[(long long)t * Cout * Cin];
    }
    float* o = dw + ((long long)co * Cin_w + ci) * 9;
#pragma unroll
    for (int t = 0; t < 9; ++t) o[t] = accumulate ? o[t] + s[t] : s[t];
}

struct WgPlan { bool ok; int TW, TH, TPI, TWs, NR, nstages, S, per; };

static WgPlan wgrad_plan_wino(int nimg, int H, int W, int Cin, int Cout) {
    WgPlan g = {};
    const int ncu = vd_cu_count();
    if (H % 2 || W % 2 || Cin % 4 || Cout % 4 || nimg <= 0) return g;
    g.TW = W / 2; g.TH = H / 2;
    g.TPI = g.TW * g.TH;
    if (!pow2(g.TW) || !pow2(g.TH) || g.TW < 4 || g.TPI < WG_T) return g;
    g.TWs = g.TW < WG_T ? g.TW : WG_T;
    g.NR = WG_T / g.TWs;
    if (g.NR > g.TH) return g;
    g.nstages = nimg * g.TPI / WG_T;
    // slabs: one workgroup per CU and round (100 KB of LDS each); pick the count that fills whole rounds of the device's CUs best, >= 4 stages each
    const long long blocks = ((Cout + 63) / 64) * (long long)((Cin + 63) / 64);
    int best = 1;
    double best_eff = 0.0;
    for (int S = 1; S <= 64 && S <= g.nstages; ++S) {
        if (S > 1 && g.nstages / S < 4) break;
        const long long wgs = blocks * S, rounds = (wgs + ncu - 1) / ncu;
        double eff = (double)wgs / (double)(rounds * ncu);
        eff *= 1.0 - 0.002 * S;                        // mild preference for fewer slabs (less reduce traffic)
        if (eff > best_eff) { best_eff = eff; best = S; }
    }
    g.per = (g.nstages + best - 1) / best;
    g.S = (g.nstages + g.per - 1) / g.per;
    g.ok = true;
    return g;
}

extern "C" int vd_conv3x3_wgrad_wino_supported(int32_t nimg, int32_t H, int32_t W, int32_t Cin, int32_t Cout, int64_t ldx, int64_t lddy) {
    const WgPlan g = wgrad_plan_wino(nimg, H, W, Cin, Cout);
    if (!g.ok || ldx % 4 || lddy % 4) return 0;
    // 32-bit offsets inside one stage (a few rows of one image) must stay far below 2^31
    const long long lim = 0x40000000LL / 4;
    if ((long long)(2 * g.NR + 4) * (W + 2) * (ldx > lddy ? ldx : lddy) >= lim) return 0;
    return 1;
}

extern "C" size_t vd_conv3x3_wgrad_wino_ws_bytes(int32_t nimg, int32_t H, int32_t W, int32_t Cin, int32_t Cout) {
    const WgPlan g = wgrad_plan_wino(nimg, H, W, Cin, Cout);
    if (!g.ok) return 0;
    return ((size_t)g.S * 9 * Cout * Cin + (size_t)g.S * Cout) * sizeof(float);
}

static int wgrad_wino_impl(const float* xin, int64_t ldx, const float* dy, int64_t lddy, int32_t nimg, int32_t H, int32_t W,
                           int32_t Cin, int32_t Cout, float* dw_oihw, float* dbias, int32_t Cin_w, int32_t Cout_w,
                           int32_t accumulate, float* ws, size_t ws_bytes, void* stream, int phases) {
    VD_REQUIRE(xin && dy && dw_oihw && ws, "vd_conv3x3_wgrad_wino: null operand");
    VD_REQUIRE(vd_conv3x3_wgrad_wino_supported(nimg, H, W, Cin, Cout, ldx, lddy), "vd_conv3x3_wgrad_wino: unsupported geometry "
               "nimg=%d H=%d W=%d Cin=%d Cout=%d (use vd_conv3x3_wgrad)", nimg, H, W, Cin, Cout);
    VD_REQUIRE(Cin_w <= Cin && Cout_w <= Cout, "vd_conv3x3_wgrad_wino: real dims exceed padded dims");
    VD_REQUIRE(vd_aligned16(xin) && vd_aligned16(dy), "vd_conv3x3_wgrad_wino: operands must be 16-byte aligned");
    VD_REQUIRE(ws_bytes >= vd_conv3x3_wgrad_wino_ws_bytes(nimg, H, W, Cin, Cout), "vd_conv3x3_wgrad_wino: workspace too small");
    const WgPlan g = wgrad_plan_wino(nimg, H, W, Cin, Cout);
    WgradArgs a = {};
    a.x = xin; a.ldx = ldx; a.dy = dy; a.lddy = lddy; a.slabs = ws;
    a.cpart = dbias ? ws + (size_t)g.S * 9 * Cout * Cin : nullptr;
    a.nimg = nimg; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
    a.TW = g.TW; a.TH = g.TH; a.TPI = g.TPI; a.lgTW = ilog2(g.TW); a.lgTPI = ilog2(g.TPI);
    a.nstages = g.nstages; a.per = g.per; a.probe = g_probe;
    const dim3 grid((Cin + 63) / 64, (Cout + 63) / 64, g.S);
    hipStream_t st = (hipStream_t)stream;
#define VD_WG_LAUNCH(T)                                                                          \
    do {                                                                                         \
        if (dbias) hipLaunchKernelGGL((wino_wgrad_kernel<T, true>), grid, dim3(512), 0, st, a);  \
        else hipLaunchKernelGGL((wino_wgrad_kernel<T, false>), grid, dim3(512), 0, st, a);       \
    } while (0)
    if (phases & 1) {
        g_last_wgrad = (g.TWs * 1000 + g.S) * 2 + (dbias ? 1 : 0);
        if (g.TWs == 16) VD_WG_LAUNCH(16);
        else if (g.TWs == 8) VD_WG_LAUNCH(8);
        else VD_WG_LAUNCH(4);
        VD_LAUNCH_CHECK("wino_wgrad_kernel");
    }
#undef VD_WG_LAUNCH
    if (phases & 2) {
        const long long tot = (long long)Cout * Cin;
        hipLaunchKernelGGL(wino_wgrad_reduce_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, ws, g.S, Cout, Cin, Cout_w,
                           Cin_w, dw_oihw, accumulate, a.cpart, g.S, dbias);
        VD_LAUNCH_CHECK("wino_wgrad_reduce_kernel");
    }
    return 0;
}

extern "C" int vd_conv3x3_wgrad_wino(const float* xin, int64_t ldx, const float* dy, int64_t lddy, int32_t nimg, int32_t H, int32_t W,
                                     int32_t Cin, int32_t Cout, float* dw_oihw, float* dbias, int32_t Cin_w, int32_t Cout_w,
                                     int32_t accumulate, float* ws, size_t ws_bytes, void* stream) {
    return wgrad_wino_impl(xin, ldx, dy, lddy, nimg, H, W, Cin, Cout, dw_oihw, dbias, Cin_w, Cout_w, accumulate, ws, ws_bytes, stream, 3);
}

/* the two kernels of vd_conv3x3_wgrad_wino as separate calls (per-kernel timing): phase 1 = slab planes, 2 = reduction.  The plan
 * is a pure function of the arguments: the phases share no hidden state. */
extern "C" int vd_conv3x3_wgrad_wino_phase(const float* xin, int64_t ldx, const float* dy, int64_t lddy, int32_t nimg, int32_t H,
                                           int32_t W, int32_t Cin, int32_t Cout, float* dw_oihw, float* dbias, int32_t Cin_w,
                                           int32_t Cout_w, int32_t accumulate, float* ws, size_t ws_bytes, int32_t phase, void* stream) {
    VD_REQUIRE(phase == 1 || phase == 2, "vd_conv3x3_wgrad_wino_phase: phase must be 1 (slab planes) or 2 (reduce)");
    return wgrad_wino_impl(xin, ldx, dy, lddy, nimg, H, W, Cin, Cout, dw_oihw, dbias, Cin_w, Cout_w, accumulate, ws, ws_bytes, stream, phase);
}
