// wino43.hip -- the 3x3 / stride 1 / pad 1 convolution as Winograd F(4x4, 3x3) on the fp32 matrix cores, NHWC: its INPUT GRADIENT
// (round 3), its FORWARD pass (round 4) and, unfused, its WEIGHT GRADIENT (further down).
// Reference op: modules.Conv2d -> F.conv2d (modules.py:141-144 <- unet.py:121,125) and its autograd.
//
// Why a second Winograd order: F(4x4,3x3) needs 36 multiplies per 4x4 output tile and (ci, co) pair where F(2x2,3x3) (wino.hip) needs
// 16 per 2x2 tile -- 2.25 instead of 4 per output, 1.78x fewer matrix-core cycles again -- at a larger rounding error that depends on the
// interpolation points.  With the classic points {0, +-1, +-2, inf} a layer carries ~7x the error of a direct fp32 sum: an order of
// magnitude inside the 1e-4 relative bound on gradients, so the INPUT GRADIENT uses them (FWD = false).  The FORWARD pass (FWD = true) uses
// {0, +-3/4, +-3/2, inf} -- every coefficient of B^T and A^T a dyadic rational, exact in fp32 -- with which the whole-network output error
// is ~2x that of the F(2x2,3x3) forward (fp32 simulation of both networks against the reference's own fp32 evaluation,
// tests/probe/wino_err_sim.py: CIFAR 4.2e-6 vs 2.0e-6, CelebA 5.7e-6 vs 2.3e-6 max-abs; classic points 8.8e-6 / 9.2e-6) and stays inside
// the stated 2e-5 bound on the UNet output (measured on MI355X: tests/test_unet_gpu.py).  VD_WINO43_FWD=0 keeps
// the forward convolutions on F(2x2,3x3).
//
//   U[xi][n][k]   = (G rot180(w[k][n]) G^T)[xi]        xi = 6a+b in 0..35, n = conv input channel (GEMM column), k = conv output
//                                                     channel (GEMM K)                            (vd_wino43_pack*, per weight update)
//   V[xi][t][k]   = (B^T d[t][k] B)[xi]                d = 6x6 patch of dy around 4x4-output tile t               (on the fly)
//   M[xi][t][n]   = sum_k V[xi][t][k] U[xi][n][k]      36 independent GEMMs                                        (MFMA)
//   dx[t][u][v][n] = (A^T M[.][t][n] A)[u][v]                                                                      (epilogue)
//
// Work item = 64 tiles (a 32x32 image, 16 pixel rows of a 64-wide one, or four 16x16 images) x 32 channels x all 36 xi; persistent workgroups of 8
// waves (two per SIMD, <= 256 registers), one per CU (156-160 KB of LDS):
//   * wave w owns tile group w & 3 (16 tiles) and HALF of xi: rows a in {3h .. 3h+2}, h = w >> 2, all six b -- 18 xi x 2 channel
//     blocks = 144 accumulator registers.  Splitting xi by rows splits the input transform cleanly: the row pass of half h needs
//     five of the six patch rows and half of the arithmetic, the column pass is per row.  The two halves of a tile group meet
//     once per item, in the epilogue, through LDS (the stages are dead by then).
//   * K tile = 8 channels (two 16-byte granules per pixel): a lane feeds channels {2 kq, 2 kq + 1} (kq = lane >> 4 = the k slot of
//     v_mfma_f32_16x16x4_f32) as two MFMA k-steps.  Per K tile the raw patch image (not V) and the 36 U tiles go HBM/L2 -> LDS with
//     `buffer_load_dwordx4 ... lds`; two stages, the DMA of tile t+1 is issued between the MFMA steps of tile t, one barrier per tile.
//   * patch image: [granule][pixel row][4 runs of columns x == r (mod 4)][slot] 16-byte slots, so that the 16 tiles of a wave (4
//     pixels apart) read consecutive slots; the row pitch is chosen so that tiles of different tile rows fall on different banks
//     (conflict-free ds_read_b64 for 8 or 16 tiles per row).
//   * U image: packed by vd_wino43_pack in exactly the order the lanes read it -- [32-channel block][K tile][xi][n][kq][cb][j] -- so
//     (lane 16 kq + n of the MFMA row operand owns floats 4 lane .. 4 lane + 3) a stage is ONE contiguous 36 KB run of global
//     memory (36 linear DMA pieces) and a lane's fragment for both 16-channel blocks
//     and both k-steps is one conflict-free ds_read_b128.
#include "common.h"
#include <type_traits>

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
constexpr unsigned OOB = 0x80000000u;
constexpr int KT = 8;                         // channels per K tile
constexpr int TN = 32;                        // output channels per work item
constexpr int TILES = 64;                     // 4x4-output tiles per work item
constexpr int U_STAGE = 36 * 4 * 16 * 2 * 2;  // floats of U per stage (36 KB): [xi][kq][n][cb][j] (lane = 16 kq + n reads 16 bytes)
constexpr int THREADS = 512;

struct Args43 {
    const float* x; long long ldx;            // GEMM A side: dy  [nimg][H][W][>= K]
    const float* U;                           // packed [N/32][K/8][36][4 kq][16 n][2 cb][2 j]
    float* y; long long ldy;                  // dx      [nimg][H][W][>= N]
    int nimg, H, W, K, N;
    int items_per_img;                        // 1 (32x32) or H/16 (64 wide); 16x16 images: FOUR IMAGES per item (tile groups = nimg / 4)
    int ngrp;                                 // tile groups (64 tiles each)
    int ncb, nitems;
    // forward pass only (FWD): y = conv + bias (+ res), GroupNorm partial sums of y per (image, chunk of an item's pixels, channel)
    const float* bias; const float* res; long long ldr;
    float* stats;                             // [nimg][chunks_per_img][2][N] or NULL: chunk = the pixels one work item holds of one image
    int chunks_per_img;                       // 1 (32x32, 16x16) or H/16 (64 wide)
    unsigned long long* probe;                // -DVD_PROBES builds only (vd_wino43_set_probe): 16 x u64 per (workgroup, wave, item round < 4)
};

// workgroup barrier that orders LDS traffic only: __syncthreads() also waits for every outstanding global load / store of the wave
// (vmcnt(0)), which in the epilogue would expose the latency of the residual loads and the drain of the output stores at each barrier
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const float* base, int records = (int)OOB) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, records, 0x00020000);
}

// 1-D input transform B^T of F(4,3), rows {0,1,2} (HALF = 0) or {3,4,5} (HALF = 1).
//   B^T = [4 0 -5 0 1 0 ; 0 -4 -4 1 1 0 ; 0 4 -4 -1 1 0 ; 0 -2 -1 2 1 0 ; 0 2 -1 -2 1 0 ; 0 4 0 -5 0 1]
// The transforms run on the two channels of a lane as packed fp32 instructions (v_pk_fma_f32 / v_pk_add_f32); measured against scalar
// instructions twice (round 3: 0.460 vs 0.464 ms at 256 -> 256 @32x32; round 6, tests/probe/mfma_valu_fill.hip: a lump of packed
// instructions costs the SIMD 5.1 cycles each beside v_mfma_f32_16x16x4_f32, a lump of plain ones 2.6 -- the same per element).
//
// What the K loop's time is made of (round 6: profiles/r06_wino43_ktile_timeline.txt, FINDINGS.md): on gfx950 the fp32 matrix
// instruction and vector instructions of EITHER wave of a SIMD exclude each other, so a K tile costs 144 x 32 matrix cycles plus
// every transform instruction; A/B builds of this loop that spread the transforms over the MFMA gaps (+3 %), aligned the two waves'
// transform lumps with extra barriers (+-1 %), skewed the workgroups, changed the DMA order or lead, read U from global memory or
// changed the scheduler strategy (rounds 4-5) all measured flat or slower.  The constants below are what those A/B builds kept.
constexpr int SB_STEP = 12;                   // MFMA step (of 18 per K tile) the tile barrier sits in front of
constexpr int U_AHEAD = 2;                    // MFMA steps a U fragment is read ahead of its use (1 / 2 / 3 measured: 2 and 3 ~1 % ahead of 1)
// DY: the dyadic point set {0, +-3/4, +-3/2, inf} of the forward pass (the rows of bt6 below) instead of the classic {0, +-1, +-2, inf}
template <int HALF, bool DY, typename T>
__device__ __forceinline__ void bt_half(const T (&d)[6], T& o0, T& o1, T& o2) {
    if (DY) {
        // rows 1, 2 = (d4 - 9/4 d2) +- 3/4 (d3 - 9/4 d1), rows 3, 4 = (d4 - 9/16 d2) +- 3/2 (d3 - 9/16 d1): as many instructions as the classic set
        if (HALF == 0) {
            const T e1 = d[4] - 2.25f * d[2], f1 = d[3] - 2.25f * d[1];
            o0 = 1.265625f * d[0] + (d[4] - 2.8125f * d[2]);
            o1 = e1 + 0.75f * f1;
            o2 = e1 - 0.75f * f1;
        } else {
            const T e2 = d[4] - 0.5625f * d[2], f2 = d[3] - 0.5625f * d[1];
            o0 = e2 + 1.5f * f2;
            o1 = e2 - 1.5f * f2;
            o2 = 1.265625f * d[1] + (d[5] - 2.8125f * d[3]);
        }
    } else if (HALF == 0) {
        const T t = d[4] - 4.f * d[2], u = d[3] - 4.f * d[1];
        o0 = 4.f * d[0] + (d[4] - 5.f * d[2]);
        o1 = t + u;
        o2 = t - u;
    } else {
        const T c = d[4] - d[2], e = d[3] - d[1];
        o0 = c + 2.f * e;
        o1 = c - 2.f * e;
        o2 = 4.f * d[1] + (d[5] - 5.f * d[3]);
    }
}
// all six outputs (the column pass)
template <bool DY, typename T>
__device__ __forceinline__ void bt_full(const T (&r)[6], T (&v)[6]) {
    if (DY) {
        const T e1 = r[4] - 2.25f * r[2], f1 = r[3] - 2.25f * r[1];
        const T e2 = r[4] - 0.5625f * r[2], f2 = r[3] - 0.5625f * r[1];
        v[0] = 1.265625f * r[0] + (r[4] - 2.8125f * r[2]);
        v[1] = e1 + 0.75f * f1;
        v[2] = e1 - 0.75f * f1;
        v[3] = e2 + 1.5f * f2;
        v[4] = e2 - 1.5f * f2;
        v[5] = 1.265625f * r[1] + (r[5] - 2.8125f * r[3]);
        return;
    }
    const T t = r[4] - 4.f * r[2], u = r[3] - 4.f * r[1], c = r[4] - r[2], e = r[3] - r[1];
    v[0] = 4.f * r[0] + (r[4] - 5.f * r[2]);
    v[1] = t + u;
    v[2] = t - u;
    v[3] = c + 2.f * e;
    v[4] = c - 2.f * e;
    v[5] = 4.f * r[1] + (r[5] - 5.f * r[3]);
}
// the two channels of a lane: f32x2 in, transform per channel (scalar build) or on the pair (packed build)
template <int HALF, bool DY>
__device__ __forceinline__ void bt_half2(const f32x2 (&d)[6], f32x2& o0, f32x2& o1, f32x2& o2) { bt_half<HALF, DY, f32x2>(d, o0, o1, o2); }
template <bool DY>
__device__ __forceinline__ void bt_full2(const f32x2 (&r)[6], f32x2 (&v)[6]) { bt_full<DY, f32x2>(r, v); }

// TWT = tiles per image row (8: 32-wide images, 16: 64-wide images, 4: 16x16 images, four of them per item)
// 16x16 images: the item's patch image stacks its four images vertically with ONE shared zero row between neighbours (rows 17 i are the
// halo of image i-1 below and of image i above: 69 rows instead of 72), and the column classes keep only the slots that exist (x + 1 in
// 0..17: classes 0, 1 have five slots, classes 2, 3 four) -- 19 slots per row, 42 KB per stage: the same footprint as a 32x32 image.
// FWD: the forward pass -- dyadic interpolation points, bias / residual / GroupNorm partial sums in the epilogue; else the input gradient
template <int TWT, bool FWD>
__global__ __launch_bounds__(THREADS) void wino43_conv_kernel(const Args43 p) {
    constexpr bool QUAD = TWT == 4;
    constexpr int RUN = TWT + 1;                           // slots of one column class in a pixel row (QUAD: of classes 0 and 1)
    constexpr int RUNS = QUAD ? 4 * RUN - 2 : 4 * RUN;     // slots of a pixel row that hold pixels
    // pixel-row pitch in slots: the tiles a wave reads together (16 / TWT tile rows of TWT tiles) must fall on different 16-byte bank
    // groups: 4 * RP * 16 B == TWT * 16 B (mod 256 B)
    constexpr int RP = QUAD ? RUNS + 1 : RUNS + ((TWT / 4) & 3);
    constexpr int NTR = TILES / TWT;                       // tile rows per item (8 or 4; QUAD: 16 = 4 images x 4)
    constexpr int RIN = QUAD ? 4 * 17 + 1 : 4 * NTR + 2;   // pixel rows held per item
    constexpr int NS = RIN * RP;                           // patch slots per granule
    constexpr int NPG = (NS + 63) / 64;                    // DMA pieces (1 KiB) per granule
    constexpr int A_STAGE = 2 * NPG * 256;                 // floats
    constexpr int STAGE = A_STAGE + U_STAGE;
    constexpr int LGT = TWT == 4 ? 2 : TWT == 8 ? 3 : 4;
    static_assert((4 * RP * 16) % 256 == (TWT * 16) % 256 || (QUAD && (4 * RP * 16) % 256 == 192), "row pitch leaves bank conflicts");
    static_assert(2 * STAGE * 4 <= 163840, "stages exceed the LDS");
    static_assert(8 * 16 * 64 * 16 + 8 * 4 * 8 * 4 <= 2 * STAGE * 4, "epilogue exchange + statistics area exceeds the stages");
    __shared__ __attribute__((aligned(1024))) float smem[2 * STAGE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lq = lane >> 4;
    const int tg = wave & 3;
    const int nkt = p.K / KT;
    const int G = gridDim.x;
    // phase stamps of the probe library (tests/probe/w43_phases.py): slot s of item round n of this wave; no code in the product build
    int probe_n = 0;
    auto stamp = [&](int slot) {
        if (VD_PROBE_BUILD && p.probe && probe_n < 4) {
            const unsigned long long t = slot == 12 ? __builtin_amdgcn_s_memrealtime() : __builtin_amdgcn_s_memtime();
            if (lane == 0) p.probe[(((long long)blockIdx.x * 8 + wave) * 4 + probe_n) * 16 + slot] = t;
        }
    };
    auto stamp_val = [&](int slot, unsigned long long v) {
        if (VD_PROBE_BUILD && p.probe && probe_n < 4 && lane == 0) p.probe[(((long long)blockIdx.x * 8 + wave) * 4 + probe_n) * 16 + slot] = v;
    };

    // ---- DMA plan of this wave.  Patch pieces: granule wave & 1, slot groups (wave >> 1) + 4 j; U pieces wave + 8 j (linear copy).
    constexpr int APL = (NPG + 3) / 4;
    unsigned pxo[APL];
    const float* xitem = p.x;                                       // QUAD: first image of the item (the lane offsets pxo are item-invariant)
    auto offsets = [&](int grp) {                                   // tile group = (image, part of the image) / QUAD: four images
        const int img = grp / p.items_per_img;
        const int part = grp - img * p.items_per_img;
        const int y_first = 4 * NTR * part - 1;
#pragma unroll
        for (int j = 0; j < APL; ++j) {
            const int s = (wave >> 1) + 4 * j;
            const int slot = s * 64 + lane;
            const int r = slot / RP, cs = slot - r * RP;
            int c, idx, yy, im;
            if (QUAD) {
                c = cs >= 3 * RUN - 1 ? 3 : cs >= 2 * RUN ? 2 : cs >= RUN ? 1 : 0;
                idx = cs - (c * RUN - (c == 3 ? 1 : 0));
                const int i4 = r / 17;
                im = i4; yy = r - 17 * i4 - 1;                      // row 17 i: the shared zero row (yy = -1); r = 68: i4 = 4 (no image)
                if (i4 >= 4) yy = -1;                               // (offsets relative to the item's first image: xitem below)
            } else {
                c = cs / RUN; idx = cs - c * RUN;
                im = 0; yy = y_first + r;                           // (relative to the item's image: xitem carries the 64-bit part, so tensors beyond 2 GiB are fine)
            }
            const int xx = 4 * idx + c - 1;
            unsigned vo = OOB;
            if (s < NPG && r < RIN && cs < RUNS && (unsigned)xx < (unsigned)p.W && (unsigned)yy < (unsigned)p.H)
                vo = (unsigned)(((im * p.H + yy) * p.W + xx) * (int)p.ldx) * 4u;
            pxo[j] = vo;
        }
    };
    // piece i of this wave for K tile kt of channel block cb into stage st: i < 5: U piece wave + 8 i; else patch piece
    auto issue_piece = [&](int i, int kt, int st, int cb) {
        float* sb = smem + st * STAGE;
        if (i < 5) {
            const int q = wave + 8 * i;
            if (q < 36) {
                const float* src = p.U + ((long long)cb * nkt + kt) * U_STAGE;
                const __amdgpu_buffer_rsrc_t rs = make_rsrc(src);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(sb + A_STAGE + q * 256), 16, lane * 16, q * 1024, 0, 0);
            }
        } else {
            const int j = i - 5;
            const int s = (wave >> 1) + 4 * j, g = wave & 1;
            if (s < NPG) {
                const __amdgpu_buffer_rsrc_t rs = make_rsrc(xitem + kt * KT);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(sb + (g * NPG + s) * 256), 16, (int)pxo[j], g * 16, 0, 0);
            }
        }
    };
    constexpr int NPIECE = 5 + APL;
    // the tile barrier sits in front of MFMA step SB of 18; the NAFTER steps behind it carry the row pass of the next tile and one DMA
    // piece each of the tile after that
    constexpr int SB = SB_STEP, NAFTER = 18 - SB, PAFTER = NAFTER;
    constexpr int UPF = U_AHEAD;
    static_assert(PAFTER <= NPIECE && NPIECE - PAFTER <= SB, "DMA schedule does not fit the tile");

    // ---- LDS read addresses (floats): patch position (pr, q) of this lane's tile, channels {2 lq, 2 lq + 1}
    const int tl = 16 * tg + li;
    const int tyl = tl >> LGT, tx = tl & (TWT - 1);
    const int prow0 = QUAD ? 17 * (tyl >> 2) + 4 * (tyl & 3) : 4 * tyl;          // first patch row of this lane's tile
    const int pbase = ((lq >> 1) * NPG * 64 + prow0 * RP + tx) * 4 + (lq & 1) * 2;
    auto poff = [](int pr, int q) { return (pr * RP + ((q & 3) * RUN - (QUAD && (q & 3) == 3 ? 1 : 0)) + (q >> 2)) * 4; };
    const int uoff = A_STAGE + lane * 4;

    // item of round n: linear id n * G + w, re-ordered inside FULL rounds so that the workgroups of one XCD (w % 8) take 32
    // neighbouring ids, and 32 neighbouring ids are 4 channel blocks x 8 tile groups: the 4 blocks of a tile group share its patch
    // stream in that XCD's L2, the 8 tile groups of a block share its U stream (as in wino.hip)
    const int ngrp = p.ngrp;                                        // tile groups (64 tiles each)
    auto item_of = [&](int n, int& cb, int& grp) -> bool {
        int t = n * G + (int)blockIdx.x;
        if ((G & 7) == 0 && (n + 1) * G <= p.nitems) t = n * G + ((int)blockIdx.x & 7) * (G >> 3) + ((int)blockIdx.x >> 3);
        if (t >= p.nitems) return false;
        if ((p.ncb & 3) == 0 && (ngrp & 7) == 0) {
            const int c = t >> 5, i = t & 31, ncg = p.ncb >> 2;
            cb = (c % ncg) * 4 + (i & 3); grp = (c / ncg) * 8 + (i >> 2);
            return true;
        }
        cb = t % p.ncb; grp = t / p.ncb;
        return true;
    };

    auto run = [&](auto Hc) {
        constexpr int HALF = decltype(Hc)::value;
        // row pass of one K tile: R[a'][q] = sum_p B^T[3 HALF + a'][p] d[p][q]  (five of the six patch rows), columns q0 .. q1-1
        auto rowpass = [&](const float* sa, f32x2 (&Rn)[3][6], int q0, int q1) {
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                if (q < q0 || q >= q1) continue;
                f32x2 d[6];
#pragma unroll
                for (int pr = 0; pr < 6; ++pr) {
                    if ((HALF == 0 && pr == 5) || (HALF == 1 && pr == 0)) d[pr] = f32x2{0.f, 0.f};
                    // (volatile: keeps each read a ds_read_b64 -- banks (a/4) mod 64, 32-lane groups, conflict-free on this image.  Left to
                    //  itself hipcc pairs them into ds_read2_b64, which is served in 16-lane groups on 32 banks at half the rate: the two
                    //  tile rows of a wave then collide 2-way.  PMC round 3: SQ_LDS_BANK_CONFLICT 2.7e7 per launch, 0 in every other kernel)
                    else d[pr] = *(const volatile __attribute__((address_space(3))) f32x2*)((const __attribute__((address_space(3))) float*)sa + poff(pr, q));
                }
                bt_half2<HALF, FWD>(d, Rn[0][q], Rn[1][q], Rn[2][q]);
            }
        };
        // an item's FIRST stage goes out before the item starts: for the first item here, for every later one from inside the previous item's
        // epilogue, as soon as the exchange area that overlays stage 0 has been read (round 6: the stamps of tests/probe/w43_phases.py put 9-13
        // thousand cycles per item -- 4-5 % -- on the wait for this DMA when it was issued at the item's top, behind the output stores)
        auto first_stage = [&](int cbx, int grpx) {
            if (QUAD) xitem = p.x + (long long)grpx * (4 * 16 * 16) * p.ldx;
            else { offsets(grpx); xitem = p.x + (long long)(grpx / p.items_per_img) * p.H * p.W * p.ldx; }
#pragma unroll
            for (int i = 0; i < NPIECE; ++i) issue_piece(i, 0, 0, cbx);
        };
        int cb = 0, grp = 0, cb_next = 0, grp_next = 0;
        if (QUAD) offsets(0);
        bool have = item_of(0, cb, grp);
        if (have) first_stage(cb, grp);
        for (int n = 0; have; ++n) {
            probe_n = n;
            stamp(0); stamp(12);
            unsigned long long t_vm = 0, t_bar = 0;
            f32x4 acc[18][2];
#pragma unroll
            for (int x = 0; x < 18; ++x) { acc[x][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[x][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
            // the first stage has landed: everything older than this wave's 16 output stores of the previous item (vmcnt counts in issue order;
            // a wave that also wrote statistics waits for one store more)
            if (n == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            lds_barrier();
            stamp(1);
            if (nkt > 1) {
#pragma unroll
                for (int i = 0; i < PAFTER; ++i) issue_piece(i, 1, 1, cb);
            }
            f32x2 R[3][6];
            rowpass(smem + pbase, R, 0, 6);
            stamp(2);
            // K loop.  Tile kt lives in stage kt & 1.  ONE barrier per tile, in front of step SB of its 18 MFMA steps, with the
            // U fragments of steps SB..17 already in registers: behind it stage kt is dead (the patch of tile kt was consumed
            // during tile kt-1) and stage kt+1 has landed, so the remaining three steps run beside the row pass of tile kt+1 --
            // the transform no longer opens every tile with both waves of a SIMD waiting on LDS -- and the DMA of tile kt+2
            // starts into the stage just vacated (pieces 0..2 in steps 15..17, the rest in the first steps of tile kt+1).
            for (int kt = 0; kt < nkt; ++kt) {
                const int st = kt & 1;
                const float* su = smem + st * STAGE + uoff;
                const float* san = smem + (st ^ 1) * STAGE + pbase;
                const bool n1 = kt + 1 < nkt, n2 = kt + 2 < nkt;
                f32x2 Rn[3][6];
                // U fragments: read UPF steps ahead of their MFMAs (ring of UPF + 1), those of steps SB..17 in front of the barrier
                f32x4 uf[UPF + 1], ufl[NAFTER];
#pragma unroll
                for (int e = 0; e < UPF; ++e) uf[e] = *reinterpret_cast<const f32x4*>(su + (18 * HALF + e) * 256);
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    f32x2 V[6];
                    bt_full2<FWD>(R[a], V);
#pragma unroll
                    for (int b = 0; b < 6; ++b) {
                        const int xl = 6 * a + b;
                        if (xl == SB) {
#pragma unroll
                            for (int e = 0; e < NAFTER; ++e) ufl[e] = *reinterpret_cast<const f32x4*>(su + (18 * HALF + SB + e) * 256);
                            if (VD_PROBE_BUILD && p.probe) {        // (probe library: cycles waiting for this wave's DMA / parked at the barrier)
                                const unsigned long long ta = __builtin_amdgcn_s_memtime();
                                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                                const unsigned long long tb = __builtin_amdgcn_s_memtime();
                                __syncthreads();
                                const unsigned long long tc = __builtin_amdgcn_s_memtime();
                                t_vm += tb - ta; t_bar += tc - tb;
                            } else {
                                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                                __syncthreads();
                            }
                        }
                        if (xl + UPF < SB) uf[(xl + UPF) % (UPF + 1)] = *reinterpret_cast<const f32x4*>(su + (18 * HALF + xl + UPF) * 256);
                        // DMA of tile kt+1: pieces PAFTER .. NPIECE-1 in the first steps of this tile (pieces 0 .. PAFTER-1 went out behind
                        // the barrier of tile kt-1); DMA of tile kt+2: pieces 0 .. PAFTER-1 behind this tile's barrier
                        if (xl + PAFTER < NPIECE) { if (n1) issue_piece(xl + PAFTER, kt + 1, st ^ 1, cb); }
                        if (xl >= SB) {
                            if (n2) issue_piece(xl - SB, kt + 2, st, cb);
                            rowpass(san, Rn, (6 * (xl - SB)) / NAFTER, (6 * (xl - SB + 1)) / NAFTER);     // (last tile: a stale stage, result unused)
                        }
                        const f32x4 u = xl < SB ? uf[xl % (UPF + 1)] : ufl[xl < SB ? 0 : xl - SB];
                        acc[xl][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(u[0], V[b][0], acc[xl][0], 0, 0, 0);
                        acc[xl][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(u[2], V[b][0], acc[xl][1], 0, 0, 0);
                        acc[xl][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(u[1], V[b][1], acc[xl][0], 0, 0, 0);
                        acc[xl][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(u[3], V[b][1], acc[xl][1], 0, 0, 0);
                    }
                }
                // (unconditional: a select here would keep both R and Rn alive across the whole tile)
#pragma unroll
                for (int a = 0; a < 3; ++a)
#pragma unroll
                    for (int q = 0; q < 6; ++q) R[a][q] = Rn[a][q];
            }
            stamp(3);
            __syncthreads();                                        // every wave is through its last three steps: the stages are dead
            stamp(4);

            // ---------------- epilogue: dx = A^T M A      A^T = [1 1 1 1 1 0 ; 0 1 -1 2 -2 0 ; 0 1 1 4 4 0 ; 0 1 -1 8 -8 1]
            // this wave holds rows a = 3 HALF + a' of M for both channel blocks: column pass in registers, then its partial row
            // pass PT[u][v]; it hands the channel block it does not finish (1 - HALF) to its partner wave through LDS and
            // finishes block HALF.  lane (li, lq): tile tl, channels 32 cb + 16 blk + 4 lq .. +3
            f32x4* xch = reinterpret_cast<f32x4*>(smem);            // [wave][k = 4u+v][lane]; the stages are dead (last barrier)
            // (forward kernel: the lane coordinates are re-derived here instead of staying live across the K loop -- its dyadic transforms
            //  need the registers; mbcnt = the lane id)
            const int lane_e = FWD ? (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) : lane;
            const int li_e = lane_e & 15, lq_e = lane_e >> 4;
            const int tl_e = 16 * tg + li_e, tyl_e = tl_e >> LGT, tx_e = tl_e & (TWT - 1);
            f32x4 Y[4][4];
            f32x4 RV[FWD ? 4 : 1][FWD ? 4 : 1];                    // forward: the residual of this wave's output block
#pragma unroll
            for (int pass = 0; pass < 2; ++pass) {
                const int blk = pass == 0 ? 1 - HALF : HALF;        // first the block that is given away
                f32x4 S[3][4];
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    const f32x4 m0 = acc[6 * a][blk], m1 = acc[6 * a + 1][blk], m2 = acc[6 * a + 2][blk], m3 = acc[6 * a + 3][blk],
                                m4 = acc[6 * a + 4][blk], m5 = acc[6 * a + 5][blk];
                    const f32x4 p12 = m1 + m2, q12 = m1 - m2, p34 = m3 + m4, q34 = m3 - m4;
                    S[a][0] = m0 + p12 + p34;
                    if (FWD) {                  // A^T of {0, +-3/4, +-3/2, inf}: [1 1 1 1 1 0 ; 0 3/4 -3/4 3/2 -3/2 0 ; 0 9/16 9/16 9/4 9/4 0 ; 0 27/64 -27/64 27/8 -27/8 1]
                        S[a][1] = 0.75f * q12 + 1.5f * q34;
                        S[a][2] = 0.5625f * p12 + 2.25f * p34;
                        S[a][3] = 0.421875f * q12 + 3.375f * q34 + m5;
                    } else {
                        S[a][1] = q12 + 2.f * q34;
                        S[a][2] = p12 + 4.f * p34;
                        S[a][3] = q12 + 8.f * q34 + m5;
                    }
                }
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    if (HALF == 0) {            // rows a = 0, 1, 2
                        const f32x4 s12 = S[1][v] + S[2][v], d12 = S[1][v] - S[2][v];
                        if (FWD) { Y[0][v] = S[0][v] + s12; Y[1][v] = 0.75f * d12; Y[2][v] = 0.5625f * s12; Y[3][v] = 0.421875f * d12; }
                        else { Y[0][v] = S[0][v] + s12; Y[1][v] = d12; Y[2][v] = s12; Y[3][v] = d12; }
                    } else {                    // rows a = 3, 4, 5
                        const f32x4 s34 = S[0][v] + S[1][v], d34 = S[0][v] - S[1][v];
                        if (FWD) { Y[0][v] = s34; Y[1][v] = 1.5f * d34; Y[2][v] = 2.25f * s34; Y[3][v] = 3.375f * d34 + S[2][v]; }
                        else { Y[0][v] = s34; Y[1][v] = 2.f * d34; Y[2][v] = 4.f * s34; Y[3][v] = 8.f * d34 + S[2][v]; }
                    }
                }
                if (pass == 0) {
#pragma unroll
                    for (int u = 0; u < 4; ++u)
#pragma unroll
                        for (int v = 0; v < 4; ++v) xch[(wave * 16 + 4 * u + v) * 64 + lane_e] = Y[u][v];
                    if (FWD) {
                        // the residual of the block this wave finishes: all 16 loads go out here, behind the exchange stores, so that their
                        // latency runs under the barrier and the second transform pass (the accumulators of the block given away are dead)
                        const int img_r = QUAD ? 4 * grp + (tyl_e >> 2) : grp / p.items_per_img;
                        const int part_r = QUAD ? 0 : grp - img_r * p.items_per_img;
                        const int y0r = QUAD ? 4 * (tyl_e & 3) : 4 * (NTR * part_r + tyl_e);
                        // (non-QUAD: the image is uniform over the workgroup, so its 64-bit offset goes into the descriptor base)
                        const unsigned pixr = (unsigned)(((QUAD ? img_r : 0) * p.H + y0r) * p.W + 4 * tx_e);
                        const unsigned n0r = (unsigned)(cb * TN + 16 * HALF + 4 * lq_e);
                        const float* rbase = p.res ? p.res + (QUAD ? 0LL : (long long)img_r * p.H * p.W * p.ldr) : p.y;
                        const __amdgpu_buffer_rsrc_t rrs = make_rsrc(rbase, p.res ? (int)OOB : 0);
#pragma unroll
                        for (int u = 0; u < 4; ++u)
#pragma unroll
                            for (int v = 0; v < 4; ++v) {
                                RV[u][v] = f32x4{0.f, 0.f, 0.f, 0.f};
                                if (p.res) RV[u][v] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                          rrs, (int)(((pixr + (unsigned)(u * p.W + v)) * (unsigned)p.ldr + n0r) * 4u), 0, 0));
                            }
                    }
                    stamp(5);
                    lds_barrier();                                  // (the residual loads stay in flight across it)
                    stamp(6);
                }
            }
            {
                const int partner = wave ^ 4;
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int v = 0; v < 4; ++v) Y[u][v] += xch[(partner * 16 + 4 * u + v) * 64 + lane_e];
            }
            // the exchange area is read: the NEXT item's first stage may overwrite the part of it that overlays stage 0 -- it streams in under
            // the rest of this epilogue (bias / residual / statistics / the 16 output stores) instead of being waited for at the next item's top
            const int cb_cur = cb, grp_cur = grp;
            const bool have_next = item_of(n + 1, cb_next, grp_next);
            lds_barrier();
            stamp(7);
            if (have_next) first_stage(cb_next, grp_next);
            __builtin_amdgcn_sched_barrier(0);                      // (the output stores below stay behind the DMA issue: the next item waits with vmcnt(16))
            asm volatile("" ::: "memory");
            {
                typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                const int img = QUAD ? 4 * grp_cur + (tyl_e >> 2) : grp_cur / p.items_per_img;
                const int part = QUAD ? 0 : grp_cur - img * p.items_per_img;
                const int n0 = cb_cur * TN + 16 * HALF + 4 * lq_e;
                const int y0 = QUAD ? 4 * (tyl_e & 3) : 4 * (NTR * part + tyl_e), x0 = 4 * tx_e;
                const __amdgpu_buffer_rsrc_t yrs = make_rsrc(p.y + (QUAD ? 0LL : (long long)img * p.H * p.W * p.ldy));
                const unsigned ldy_u = (unsigned)p.ldy;
                const unsigned pix00 = (unsigned)(((QUAD ? img : 0) * p.H + y0) * p.W + x0);
                if (FWD) {
                    // + bias (+ residual: the skip path of the residual block, fetched in front of the exchange barrier), then the GroupNorm partial
                    // sums of what is written: per lane over its 16 pixels, over the 16 tiles of the wave by shuffles
                    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
                    if (p.bias) bv = *reinterpret_cast<const f32x4*>(p.bias + n0);
                    f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = s1;
#pragma unroll
                    for (int u = 0; u < 4; ++u)
#pragma unroll
                        for (int v = 0; v < 4; ++v) {
                            const f32x4 val = (Y[u][v] + bv) + RV[FWD ? u : 0][FWD ? v : 0];
                            Y[u][v] = val;
                            s1 += val; s2 += val * val;
                        }
                    if (p.stats) {
#pragma unroll
                        for (int o = 1; o < 16; o <<= 1) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) { s1[j] += __shfl_xor(s1[j], o, 64); s2[j] += __shfl_xor(s2[j], o, 64); }
                        }
                        // [wave][lq][8]: behind the exchange area (the stages are dead; the area is read after the barrier below)
                        float* sarea = smem + 8 * 16 * 64 * 4;
                        if (li_e == 0) {
                            *reinterpret_cast<f32x4*>(sarea + (wave * 4 + lq_e) * 8) = s1;
                            *reinterpret_cast<f32x4*>(sarea + (wave * 4 + lq_e) * 8 + 4) = s2;
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const unsigned vo = ((pix00 + (unsigned)(u * p.W + v)) * ldy_u + (unsigned)n0) * 4u;
                        const f32x4 val = Y[u][v];
                        const u32x4 wv = {__float_as_uint(val[0]), __float_as_uint(val[1]), __float_as_uint(val[2]), __float_as_uint(val[3])};
                        __builtin_amdgcn_raw_buffer_store_b128(wv, yrs, (int)vo, 0, 0);
                    }
            }
            stamp(8);
            if (FWD && p.stats) {
                // partial sums of the item: one chunk per image it holds -- the four tile-group waves of a xi half together (one 32x32
                // image, 16 rows of a 64-wide one) or one wave each (QUAD: a wave's 16 tiles are one 16x16 image).  Fixed order, no atomics.
                // (the area lies in stage 1, which the next item's first stage does not touch; its tile-1 DMA waits behind the next item's first barrier)
                lds_barrier();
                const float* sarea = smem + 8 * 16 * 64 * 4;
                const int tid_e = wave * 64 + (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
                if (tid_e < (QUAD ? 256 : 64)) {
                    const int tgq = QUAD ? tid_e >> 6 : 0, t6 = tid_e & 63;
                    const int st = t6 >> 5, ch = t6 & 31, h = ch >> 4, q4 = (ch >> 2) & 3, j = ch & 3;
                    float v;
                    if (QUAD) v = sarea[((4 * h + tgq) * 4 + q4) * 8 + 4 * st + j];
                    else v = (sarea[((4 * h + 0) * 4 + q4) * 8 + 4 * st + j] + sarea[((4 * h + 1) * 4 + q4) * 8 + 4 * st + j]) +
                             (sarea[((4 * h + 2) * 4 + q4) * 8 + 4 * st + j] + sarea[((4 * h + 3) * 4 + q4) * 8 + 4 * st + j]);
                    const int img = QUAD ? 4 * grp_cur + tgq : grp_cur / p.items_per_img;
                    const int chunk = QUAD ? 0 : grp_cur - img * p.items_per_img;
                    p.stats[(((long long)img * p.chunks_per_img + chunk) * 2 + st) * p.N + cb_cur * TN + ch] = v;
                }
            }
            stamp(9); stamp_val(10, t_vm); stamp_val(11, t_bar);
            cb = cb_next; grp = grp_next; have = have_next;
        }
    };
    if ((wave >> 2) == 0) run(std::integral_constant<int, 0>{});
    else run(std::integral_constant<int, 1>{});
}

// ---- weight transform for the input gradient: U[xi][n = ci][k = co] = (G rot180(w[co][ci]) G^T)[xi] in the packed order
// [n / 32][k / 8][xi][(k % 8) / 2][n % 16][(n % 32) / 16][k % 2].   G = [1/4 0 0 ; -1/6 -1/6 -1/6 ; -1/6 1/6 -1/6 ; 1/24 1/12 1/6 ; 1/24 -1/12 1/6 ; 0 0 1]
// One 256-thread block per (32 n, 8 k) tile: thread t IS position t of the tile's [kq][n][cb][j] order, so every xi is one
// coalesced 1 KB run (thread 16 kq + n's four floats are what MFMA lane 16 kq + n reads with one ds_read_b128).  The transform is evaluated in fp64 and rounded once (its coefficients are not dyadic).
// fwd: the image of the FORWARD pass instead -- GEMM column n = conv output channel, K = conv input channel, kernel not rotated, and the
// G of the points {0, +-3/4, +-3/2, inf}:  G = [64/81 0 0 ; -128/243 -32/81 -8/27 ; -128/243 32/81 -8/27 ; 32/243 16/81 8/27 ; 32/243 -16/81 8/27 ; 0 0 1]
__device__ __forceinline__ void pack43_tile(const float* w, float* U, int Cout, int Cin, int tile, int fwd) {
    const int nkt = (fwd ? Cin : Cout) / KT;
    const int nb = tile / nkt, kt = tile - nb * nkt;
    const int t = threadIdx.x;
    const int kq = t >> 6, n = (t >> 2) & 15, cbk = (t >> 1) & 1, j = t & 1;
    const int cn = nb * 32 + 16 * cbk + n, ck = kt * KT + 2 * kq + j;            // GEMM column / K index
    const int ci = fwd ? ck : cn, co = fwd ? cn : ck;
    const float* src = w + ((long long)co * Cin + ci) * 9;
    double g[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) g[a][b] = (double)(fwd ? src[a * 3 + b] : src[(2 - a) * 3 + (2 - b)]);          // (input gradient: rot180)
    const double Gc[6][3] = {{0.25, 0.0, 0.0}, {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                             {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0.0, 0.0, 1.0}};
    const double Gd[6][3] = {{64.0 / 81, 0.0, 0.0}, {-128.0 / 243, -32.0 / 81, -8.0 / 27}, {-128.0 / 243, 32.0 / 81, -8.0 / 27},
                             {32.0 / 243, 16.0 / 81, 8.0 / 27}, {32.0 / 243, -16.0 / 81, 8.0 / 27}, {0.0, 0.0, 1.0}};
    double Gm[6][3];
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int c = 0; c < 3; ++c) Gm[a][c] = fwd ? Gd[a][c] : Gc[a][c];
    double tmp[6][3];
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int c = 0; c < 3; ++c) tmp[a][c] = Gm[a][0] * g[0][c] + Gm[a][1] * g[1][c] + Gm[a][2] * g[2][c];
    float* dst = U + ((long long)nb * nkt + kt) * U_STAGE + t;
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int b = 0; b < 6; ++b)
            dst[(6 * a + b) * 256] = (float)(tmp[a][0] * Gm[b][0] + tmp[a][1] * Gm[b][1] + tmp[a][2] * Gm[b][2]);
}

__global__ __launch_bounds__(256) void wino43_pack_kernel(const float* w, float* U, int Cout, int Cin, int fwd) {
    pack43_tile(w, U, Cout, Cin, blockIdx.x, fwd);
}

// all tensors of a network in one launch; items: 8 x int64 per tensor {w, U, forward image (1) or input-gradient image (0), Cout, Cin, -, -, first block}
__global__ __launch_bounds__(256) void wino43_pack_batched_kernel(const long long* items, int n) {
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (items[8 * mid + 7] <= (long long)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const long long* it = items + 8 * lo;
    pack43_tile(reinterpret_cast<const float*>(it[0]), reinterpret_cast<float*>(it[1]), (int)it[3], (int)it[4],
                (int)((long long)blockIdx.x - it[7]), (int)it[2]);
}

// ==================================================================================================================
// WEIGHT GRADIENT in the F(4x4,3x3) domain, unfused (round 3):
//   dU[xi][co][ci] = sum_t dM[xi][t][co] V[xi][t][ci],   dM = A dY_t A^T (4x4 -> 6x6),  V = B^T d_t B (6x6 patch),   dw = G^T dU G
// 36 GEMMs whose K index is the 4x4-output tile: 1.78x fewer MFMA cycles than the F(2x2,3x3) weight gradient of wino.hip.  Unlike
// the input gradient, BOTH MFMA operands are transformed per tile here, and a fused kernel cannot amortise those transforms inside
// 160 KB of LDS.  So the transforms run as ONE HBM-bound pass that writes dM and V ([xi][tile][channel], 2.25x
// the size of dY / x each) and the 36 GEMMs run on the plain tile engine as one grouped split-K launch (vd_gemm_grouped_wgrad);
// a last small kernel folds G^T . G and transposes to OIHW.  The bias gradient rides on the GEMMs too: their column sums give
// sum_t dM[xi][t][co], and a fixed combination of those planes is the sum over all pixels of dY (finish kernel).
// ==================================================================================================================
// The weight gradient uses the interpolation points {0, +-3/4, +-3/2, inf} instead of the input gradient's {0, +-1, +-2, inf}: its
// GEMMs accumulate thousands of tile products per fp32 chain, and with the smaller transform coefficients of these points the
// relative L2 error of dw is half of what the classic points give (simulated with sequential fp32 chains and measured: DESIGN.md);
// every coefficient of B^T and A is a dyadic rational (exact in fp32), the /81 and /243 of G appear only in the finish kernel.
//   B^T = [81/64 0 -45/16 0 1 0 ; 0 -27/16 -9/4 3/4 1 0 ; 0 27/16 -9/4 -3/4 1 0 ; 0 -27/32 -9/16 3/2 1 0 ; 0 27/32 -9/16 -3/2 1 0 ; 0 81/64 0 -45/16 0 1]
__device__ __forceinline__ void bt6(const f32x4 (&d)[6], f32x4 (&o)[6]) {
    const f32x4 e1 = d[4] - 2.25f * d[2], f1 = 0.75f * d[3] - 1.6875f * d[1];
    const f32x4 e2 = d[4] - 0.5625f * d[2], f2 = 1.5f * d[3] - 0.84375f * d[1];
    o[0] = 1.265625f * d[0] + (d[4] - 2.8125f * d[2]);
    o[1] = e1 + f1;
    o[2] = e1 - f1;
    o[3] = e2 + f2;
    o[4] = e2 - f2;
    o[5] = 1.265625f * d[1] + (d[5] - 2.8125f * d[3]);
}
// A = (A^T)^T, 6 x 4:  rows [1 0 0 0 ; 1 3/4 9/16 27/64 ; 1 -3/4 9/16 -27/64 ; 1 3/2 9/4 27/8 ; 1 -3/2 9/4 -27/8 ; 0 0 0 1]
__device__ __forceinline__ void a6(const f32x4 (&y)[4], f32x4 (&o)[6]) {
    const f32x4 e1 = y[0] + 0.5625f * y[2], f1 = 0.75f * y[1] + 0.421875f * y[3];
    const f32x4 e2 = y[0] + 2.25f * y[2], f2 = 1.5f * y[1] + 3.375f * y[3];
    o[0] = y[0];
    o[1] = e1 + f1;
    o[2] = e1 - f1;
    o[3] = e2 + f2;
    o[4] = e2 - f2;
    o[5] = y[3];
}

// V / dM layout: K-blocked [T/16][36][16][C]: a transform block's 36 planes land in one contiguous run, the grouped GEMMs step through K
// blocks (vd_gemm_grouped_wgrad_kblk).  (Plane-major [36][T][C] measured slower, round 4.)
// The transform pass writes V / dM once, the grouped GEMMs read them back right after: PLAIN stores (non-temporal ones measured 8-10 % slower,
// 0.164-0.173 vs 0.152 ms at 256 -> 256 @32x32 -- unlike the optimizer pass and the GroupNorm backward, whose streams nobody reads back soon).
__device__ __forceinline__ void st_once(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
struct WgT43 {
    const float* x; long long ldx; const float* dy; long long lddy;
    float* V; float* dM;                       // [T/16][36][16][Cin], [T/16][36][16][Cout]: blocks of 16 tiles, the 36 planes of a block adjacent
    int nimg, H, W, Cin, Cout, TW, TPI, T;     // tiles per row / image / in all
};

// block (64 channel quads, 4 tiles); grid (ceil(max(Cin, Cout) / 256), T / 4, 2): z = 0 transforms the input patches, z = 1 the gradients
__global__ __launch_bounds__(256) void wino43_wgrad_transform_kernel(const WgT43 p) {
    const int quad = blockIdx.y * 64 + threadIdx.x, tile = blockIdx.x * 4 + threadIdx.y;      // (tiles on grid.x: no 65 535 bound)
    if (tile >= p.T) return;
    const int img = tile / p.TPI, tin = tile - img * p.TPI;
    const int ty = tin / p.TW, tx = tin - ty * p.TW;
    const int c4 = 4 * quad;
    if (blockIdx.z == 0) {
        if (c4 >= p.Cin) return;
        f32x4 R[6][6];
#pragma unroll
        for (int pr = 0; pr < 6; ++pr) {
            const int yy = 4 * ty - 1 + pr;
            f32x4 d[6];
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                const int xx = 4 * tx - 1 + q;
                d[q] = f32x4{0.f, 0.f, 0.f, 0.f};
                if ((unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W)
                    d[q] = *reinterpret_cast<const f32x4*>(p.x + ((long long)(img * p.H + yy) * p.W + xx) * p.ldx + c4);
            }
            bt6(d, R[pr]);                                          // along the row: R[pr][b] = sum_q B^T[b][q] d[pr][q]
        }
#pragma unroll
        for (int b = 0; b < 6; ++b) {
            f32x4 col[6], o[6];
#pragma unroll
            for (int pr = 0; pr < 6; ++pr) col[pr] = R[pr][b];
            bt6(col, o);                                            // V[a][b] = sum_p B^T[a][p] R[p][b]
#pragma unroll
            for (int a = 0; a < 6; ++a)
                st_once(p.V + ((long long)((tile >> 4) * 36 + 6 * a + b) * 16 + (tile & 15)) * p.Cin + c4, o[a]);
        }
    } else {
        if (c4 >= p.Cout) return;
        f32x4 R[4][6];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            f32x4 y[4];
#pragma unroll
            for (int v = 0; v < 4; ++v)
                y[v] = *reinterpret_cast<const f32x4*>(p.dy + ((long long)(img * p.H + 4 * ty + u) * p.W + 4 * tx + v) * p.lddy + c4);
            a6(y, R[u]);                                            // R[u][b] = sum_v A[b][v] dY[u][v]
        }
#pragma unroll
        for (int b = 0; b < 6; ++b) {
            const f32x4 col[4] = {R[0][b], R[1][b], R[2][b], R[3][b]};
            f32x4 o[6];
            a6(col, o);                                             // dM[a][b] = sum_u A[a][u] R[u][b]
#pragma unroll
            for (int a = 0; a < 6; ++a)
                st_once(p.dM + ((long long)((tile >> 4) * 36 + 6 * a + b) * 16 + (tile & 15)) * p.Cout + c4, o[a]);
        }
    }
}

// dw[co][ci][3][3] (+)= G^T dU[.][co][ci] G       G^T = [64/81 -128/243 -128/243 32/243 32/243 0 ; 0 -32/81 32/81 16/81 -16/81 0 ; 0 -8/27 -8/27 8/27 8/27 1]
// dbias[co] (+)= sum_ab p_a p_b cs[6a+b][co] with A^T p = (1,1,1,1): p = (-1/9, 8/9, 0, 2/9, 0, -1/8), i.e. sum over ALL pixels of dY
// recovered from the column sums of the dM planes that ride on the GEMMs (fp64: a few hundred values)
__global__ __launch_bounds__(256) void wino43_wgrad_finish_kernel(const float* dU, const float* cs, int Cout, int Cin, int Cout_w, int Cin_w,
                                                                  float* dw, float* dbias, int accumulate) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (dbias && idx < Cout_w) {
        const double pv[6] = {-1.0 / 9, 8.0 / 9, 0.0, 2.0 / 9, 0.0, -1.0 / 8};
        double c = 0.0;
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int b = 0; b < 6; ++b)
                if (pv[a] != 0.0 && pv[b] != 0.0) c += pv[a] * pv[b] * (double)cs[(long long)(6 * a + b) * Cout + idx];
        dbias[idx] = accumulate ? dbias[idx] + (float)c : (float)c;
    }
    if (idx >= (long long)Cout * Cin) return;
    const int co = (int)(idx / Cin), ci = (int)(idx - (long long)co * Cin);
    if (co >= Cout_w || ci >= Cin_w) return;
    float u[6][6];
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int b = 0; b < 6; ++b) u[a][b] = dU[(long long)(6 * a + b) * Cout * Cin + idx];
    constexpr float g0 = 64.f / 81, g12 = 128.f / 243, g34 = 32.f / 243, h12 = 32.f / 81, h34 = 16.f / 81, k = 8.f / 27;
    float t[3][6];
#pragma unroll
    for (int b = 0; b < 6; ++b) {
        const float s12 = u[1][b] + u[2][b], d12 = u[2][b] - u[1][b], s34 = u[3][b] + u[4][b], d34 = u[3][b] - u[4][b];
        t[0][b] = g0 * u[0][b] - g12 * s12 + g34 * s34;
        t[1][b] = h12 * d12 + h34 * d34;
        t[2][b] = k * (s34 - s12) + u[5][b];
    }
    float* o = dw + ((long long)co * Cin_w + ci) * 9;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const float s12 = t[r][1] + t[r][2], d12 = t[r][2] - t[r][1], s34 = t[r][3] + t[r][4], d34 = t[r][3] - t[r][4];
        const float w0 = g0 * t[r][0] - g12 * s12 + g34 * s34;
        const float w1 = h12 * d12 + h34 * d34;
        const float w2 = k * (s34 - s12) + t[r][5];
        o[3 * r] = accumulate ? o[3 * r] + w0 : w0;
        o[3 * r + 1] = accumulate ? o[3 * r + 1] + w1 : w1;
        o[3 * r + 2] = accumulate ? o[3 * r + 2] + w2 : w2;
    }
}

// Wide layers (Cout x Cin >= FUSED_FINISH_MIN = 384 x 384 elements per plane) sum the split-K slabs of the 36 grouped GEMMs in the fold itself (fixed
// order z = 0 .. S-1, fp32: bitwise what reduce_slabs_grouped_kernel into dU followed by the kernel above gives, without writing and
// re-reading dU and one launch shorter): slab z of plane e at slabs + (e * S + z) * Cout * Cin, its column sums at cpart + (e * S + z) * Cout.
// Block (64 elements, 6 columns b): thread (x, b) sums the slabs of the six planes (a, b) of element x -- 6 x S loads in six independent
// chains, lanes = consecutive elements -- and folds them along a; the three rows meet through LDS and threads b < 3 fold row b along
// the columns.  Same-box against the two passes (reduction + fold, ms, B = 128): 768 -> 768 @8x8 0.234 vs 0.296, 1536 -> 768 0.403 vs
// 0.516, 576 -> 576 @16x16 0.577 vs 0.610, 1152 -> 576 0.903 vs 0.959, 384 -> 384 @32x32 0.708 vs 0.726; narrow layers have too few
// elements to hide the load latency this way (256 -> 256: +0.009, 192 -> 192 @64x64 with 24 slabs: +0.06; 16-byte loads on a quarter of the
// threads were slower still) and keep the two passes.
constexpr long long FUSED_FINISH_MIN = 147456;      // 384 x 384 (512 -> 256 measured equal at 32x32 and 0.015 ms slower at 16x16: two passes)
__global__ __launch_bounds__(384) void wino43_wgrad_reduce_finish_kernel(const float* slabs, const float* cpart, int S, int Cout, int Cin,
                                                                         int Cout_w, int Cin_w, float* dw, float* dbias, int accumulate) {
    __shared__ float tsh[3][6][64];
    const int x = threadIdx.x, b = threadIdx.y;
    const long long idx = (long long)blockIdx.x * 64 + x;
    if (dbias && b == 0 && idx < Cout_w) {
        const double pv[6] = {-1.0 / 9, 8.0 / 9, 0.0, 2.0 / 9, 0.0, -1.0 / 8};
        double c = 0.0;
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int bb = 0; bb < 6; ++bb)
                if (pv[a] != 0.0 && pv[bb] != 0.0) {
                    float cs = 0.f;
                    for (int z = 0; z < S; ++z) cs += cpart[((long long)(6 * a + bb) * S + z) * Cout + idx];
                    c += pv[a] * pv[bb] * (double)cs;
                }
        dbias[idx] = accumulate ? dbias[idx] + (float)c : (float)c;
    }
    const long long plane = (long long)Cout * Cin;
    const bool in = idx < plane;
    constexpr float g0 = 64.f / 81, g12 = 128.f / 243, g34 = 32.f / 243, h12 = 32.f / 81, h34 = 16.f / 81, k = 8.f / 27;
    {
        float u[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (in) {
            const float* sl = slabs + (long long)b * S * plane + idx;
            const long long astep = 6LL * S * plane;
            for (int z = 0; z < S; ++z) {
#pragma unroll
                for (int a = 0; a < 6; ++a) u[a] += sl[a * astep + z * plane];
            }
        }
        const float s12 = u[1] + u[2], d12 = u[2] - u[1], s34 = u[3] + u[4], d34 = u[3] - u[4];
        tsh[0][b][x] = g0 * u[0] - g12 * s12 + g34 * s34;
        tsh[1][b][x] = h12 * d12 + h34 * d34;
        tsh[2][b][x] = k * (s34 - s12) + u[5];
    }
    __syncthreads();
    if (!in || b >= 3) return;
    const int co = (int)(idx / Cin), ci = (int)(idx - (long long)co * Cin);
    if (co >= Cout_w || ci >= Cin_w) return;
    const int r = b;
    float t[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) t[c] = tsh[r][c][x];
    const float s12 = t[1] + t[2], d12 = t[2] - t[1], s34 = t[3] + t[4], d34 = t[3] - t[4];
    const float w0 = g0 * t[0] - g12 * s12 + g34 * s34;
    const float w1 = h12 * d12 + h34 * d34;
    const float w2 = k * (s34 - s12) + t[5];
    float* o = dw + ((long long)co * Cin_w + ci) * 9 + 3 * r;
    o[0] = accumulate ? o[0] + w0 : w0;
    o[1] = accumulate ? o[1] + w1 : w1;
    o[2] = accumulate ? o[2] + w2 : w2;
}

struct Wg43Plan { bool ok; int T, S; size_t v_f, m_f, u_f, cs_f, gemm_bytes; };
Wg43Plan wg43_plan(int nimg, int H, int W, int Cin, int Cout) {
    Wg43Plan g = {};
    if (nimg <= 0 || H % 4 || W % 4 || H < 4 || W < 4 || Cin % 4 || Cout % 4) return g;
    g.T = nimg * (H / 4) * (W / 4);
    // split-K slabs: whole residency rounds of the chip (vd_gemm_grouped_wgrad_auto_split), fp32 accumulation chains of at most ~1536
    // tiles (the error of dw grows with the chain length), at most 24 slabs
    g.S = vd_gemm_grouped_wgrad_auto_split(36, Cout, Cin, g.T, (g.T + 1535) / 1536, 24);
    const size_t T16 = ((size_t)g.T + 15) / 16 * 16;       // (V / dM are stored in blocks of 16 tiles)
    g.v_f = (size_t)36 * T16 * Cin; g.m_f = (size_t)36 * T16 * Cout; g.u_f = (size_t)36 * Cout * Cin; g.cs_f = (size_t)36 * Cout;
    g.gemm_bytes = vd_gemm_grouped_wgrad_ws_bytes(36, Cout, Cin, g.S);
    g.ok = true;
    return g;
}

thread_local int g_last43 = 0;
#ifdef VD_PROBES
unsigned long long* g_probe43 = nullptr;   // probe library only (vd_wino43_set_probe)
#else
constexpr unsigned long long* g_probe43 = nullptr;
#endif
thread_local int g_last43w = 0;        // split-K slabs of the calling thread's last vd_conv3x3_wgrad_wino43 launch

}  // namespace

/* 1 when vd_conv3x3_dgrad_wino43 serves the geometry: 32x32 images, 64-wide images with H % 16 == 0, or 16x16 images in multiples of
 * four; Cout % 8 == 0 (GEMM K), Cin % 32 == 0 (output channels), 16-byte aligned rows, tensors below 2 GiB */
extern "C" int vd_conv3x3_dgrad_wino43_supported(int32_t nimg, int32_t H, int32_t W, int32_t Cin, int32_t Cout, int64_t lddy, int64_t lddx) {
    if (nimg <= 0 || Cout % KT || Cin % TN || lddy % 4 || lddx % 4) return 0;
    if (!((W == 32 && H == 32) || (W == 64 && H % 16 == 0 && H >= 16) || (W == 16 && H == 16 && nimg % 4 == 0))) return 0;
    // 32-bit byte offsets: inside ONE image for the 32x32 / 64-wide geometries (the image's 64-bit offset sits in the buffer
    // descriptor), inside the whole tensor for the 16x16 one (a work item spans four images, lanes address different ones)
    const long long px = (W == 16 ? (long long)nimg : 1LL) * H * W, lim = 0x7FFFFFF0LL / 4;
    if (px * lddy >= lim || px * lddx >= lim) return 0;
    return 1;
}

extern "C" size_t vd_wino43_u_floats(int32_t Cout, int32_t Cin) { return (size_t)36 * Cout * Cin; }

/* occupancy rule (round-4 advice): 1 when the F(4x4,3x3) convolution kernel is expected to beat the F(2x2,3x3) one on an (nimg, H, W) batch with
 * N output channels (N = Cout forward, Cin for the input gradient).  Both kernels are persistent, one workgroup per CU, and run in rounds of
 * work items: F(4x4,3x3) has nimg x (H W / 1024) x N / 32 items (16x16: nimg / 4 x N / 32), F(2x2,3x3) four times as many (64 of its 2x2 tiles
 * cover a quarter of the pixels) at 3/8 of the time each -- the ratio measured with every CU busy (256 -> 256 @32x32, B = 128: 0.48 ms in 4
 * rounds against 0.72 ms in 16).  With fewer items than CUs the finer items win: 16 rows of a 256-channel 32x32 layer are 128 F(4x4,3x3)
 * items on half the chip for one long round, or 512 F(2x2,3x3) items in two short ones.  Small-batch sampling is where this decides. */
extern "C" int vd_conv3x3_wino43_preferred(int32_t nimg, int32_t H, int32_t W, int32_t N) {
    if (nimg <= 0 || N <= 0) return 0;
    const long long grp = W == 16 ? (nimg + 3) / 4 : (long long)nimg * (W == 64 ? H / 16 : 1);
    const long long i43 = grp * ((N + TN - 1) / TN), ncu = vd_persistent_cus();
    const long long r43 = (i43 + ncu - 1) / ncu, r23 = (4 * i43 + ncu - 1) / ncu;
    return 8 * r43 < 3 * r23;
}

/* dx[nimg][H][W][Cin] = input gradient of the 3x3 convolution with kernel w[Cout][Cin][3][3] for the output gradient dy[nimg][H][W][Cout];
 * U43 = vd_wino43_pack(w).  Every element of dx[..., :Cin] is written (no accumulation). */
extern "C" int vd_conv3x3_dgrad_wino43(const float* dy, int64_t lddy, const float* U43, float* dx, int64_t lddx, int32_t nimg, int32_t H,
                                       int32_t W, int32_t Cin, int32_t Cout, void* stream) {
    VD_REQUIRE(dy && U43 && dx, "vd_conv3x3_dgrad_wino43: null operand");
    VD_REQUIRE(vd_conv3x3_dgrad_wino43_supported(nimg, H, W, Cin, Cout, lddy, lddx),
               "vd_conv3x3_dgrad_wino43: unsupported geometry nimg=%d H=%d W=%d Cin=%d Cout=%d", nimg, H, W, Cin, Cout);
    VD_REQUIRE(vd_aligned16(dy) && vd_aligned16(U43) && vd_aligned16(dx), "vd_conv3x3_dgrad_wino43: operands must be 16-byte aligned");
    Args43 a = {};
    a.x = dy; a.ldx = lddy; a.U = U43; a.y = dx; a.ldy = lddx;
    a.nimg = nimg; a.H = H; a.W = W; a.K = Cout; a.N = Cin;
    a.items_per_img = W == 64 ? H / 16 : 1;
    a.ngrp = W == 16 ? nimg / 4 : nimg * a.items_per_img;
    a.ncb = Cin / TN;
    const long long items = (long long)a.ngrp * a.ncb;
    VD_REQUIRE(items < (1LL << 30), "vd_conv3x3_dgrad_wino43: too many work items");
    a.nitems = (int)items;
    a.probe = g_probe43;
    const int ncu = vd_persistent_cus();      // (one workgroup per CU, minus the CUs reserved for other streams: vd_set_reserved_cus)
    const dim3 grid((unsigned)(items < ncu ? items : ncu)), blk(THREADS);
    hipStream_t st = (hipStream_t)stream;
    if (W == 16) hipLaunchKernelGGL((wino43_conv_kernel<4, false>), grid, blk, 0, st, a);
    else if (W == 32) hipLaunchKernelGGL((wino43_conv_kernel<8, false>), grid, blk, 0, st, a);
    else hipLaunchKernelGGL((wino43_conv_kernel<16, false>), grid, blk, 0, st, a);
    VD_LAUNCH_CHECK("wino43_conv_kernel<., false>");
    g_last43 = W / 4;
    return 0;
}

/* tiles per row (4 / 8 / 16) of the calling thread's last vd_conv3x3_dgrad_wino43 (positive) or vd_conv3x3_wino43_fwd (negative) launch
 * = the instantiation wino43_conv_kernel<TWT, FWD> */
extern "C" int vd_wino43_last_kernel(void) { return g_last43; }
#ifdef VD_PROBES
extern "C" int vd_wino43_set_probe(unsigned long long* buf) { g_probe43 = buf; return 0; }
#endif

extern "C" int vd_wino43_pack(const float* w_oihw, int32_t Cout, int32_t Cin, float* U43, void* stream) {
    VD_REQUIRE(w_oihw && U43 && Cout % KT == 0 && Cin % TN == 0, "vd_wino43_pack: needs Cout %% 8 == 0 and Cin %% 32 == 0");
    hipLaunchKernelGGL(wino43_pack_kernel, dim3((unsigned)((Cin / TN) * (Cout / KT))), dim3(256), 0, (hipStream_t)stream, w_oihw, U43, Cout, Cin, 0);
    VD_LAUNCH_CHECK("wino43_pack_kernel");
    return 0;
}

/* ---- FORWARD pass through F(4x4,3x3) (interpolation points {0, +-3/4, +-3/2, inf}; see the head of this file for the accuracy budget).
 * Same geometries as the input gradient: 32x32 images, 64-wide images with H % 16 == 0, 16x16 images in multiples of four;
 * Cin % 8 == 0 (GEMM K), Cout % 32 == 0, rows 16-byte aligned, tensors below 2 GiB. */
extern "C" int vd_conv3x3_wino43_fwd_supported(int32_t nimg, int32_t H, int32_t W, int32_t Cin, int32_t Cout, int64_t ldx, int64_t ldy,
                                               int64_t ldres) {
    if (nimg <= 0 || Cin % KT || Cout % TN || ldx % 4 || ldy % 4 || ldres % 4) return 0;
    if (!((W == 32 && H == 32) || (W == 64 && H % 16 == 0 && H >= 16) || (W == 16 && H == 16 && nimg % 4 == 0))) return 0;
    const long long px = (W == 16 ? (long long)nimg : 1LL) * H * W, lim = 0x7FFFFFF0LL / 4;        // (see the input gradient's _supported)
    if (px * ldx >= lim || px * ldy >= lim || (ldres > 0 && px * ldres >= lim)) return 0;
    return 1;
}

/* U43f = forward image of w[Cout][Cin][3][3]: (G w[co][ci] G^T)[36] in the kernel's lane order, vd_wino43_u_floats(Cout, Cin) floats */
extern "C" int vd_wino43_pack_fwd(const float* w_oihw, int32_t Cout, int32_t Cin, float* U43f, void* stream) {
    VD_REQUIRE(w_oihw && U43f && Cin % KT == 0 && Cout % TN == 0, "vd_wino43_pack_fwd: needs Cin %% 8 == 0 and Cout %% 32 == 0");
    hipLaunchKernelGGL(wino43_pack_kernel, dim3((unsigned)((Cout / TN) * (Cin / KT))), dim3(256), 0, (hipStream_t)stream, w_oihw, U43f, Cout, Cin, 1);
    VD_LAUNCH_CHECK("wino43_pack_kernel(fwd)");
    return 0;
}

/* y[nimg][H][W][:Cout] = conv3x3(x[nimg][H][W][:Cin], w) + bias (+ res), U43f = vd_wino43_pack_fwd(w); bias / res may be NULL.
 * stats_part (or NULL): GroupNorm partial sums of y, [nimg][chunks][2][Cout] with one chunk per (image, work item): chunks =
 * vd_conv3x3_wino43_fwd_chunk_rows gives the pixels per chunk (1024 for 32x32 and 64-wide images, 256 for 16x16 ones). */
extern "C" int vd_conv3x3_wino43_fwd_chunk_rows(int32_t H, int32_t W) { return W == 16 ? H * W : (W == 32 ? H * W : 16 * W); }

extern "C" int vd_conv3x3_wino43_fwd(const float* xin, int64_t ldx, const float* U43f, const float* bias, const float* res, int64_t ldres,
                                     float* y, int64_t ldy, int32_t nimg, int32_t H, int32_t W, int32_t Cin, int32_t Cout,
                                     float* stats_part, void* stream) {
    VD_REQUIRE(xin && U43f && y, "vd_conv3x3_wino43_fwd: null operand");
    VD_REQUIRE(vd_conv3x3_wino43_fwd_supported(nimg, H, W, Cin, Cout, ldx, ldy, res ? ldres : 0),
               "vd_conv3x3_wino43_fwd: unsupported geometry nimg=%d H=%d W=%d Cin=%d Cout=%d", nimg, H, W, Cin, Cout);
    VD_REQUIRE(vd_aligned16(xin) && vd_aligned16(U43f) && vd_aligned16(y) && (!res || vd_aligned16(res)) && (!bias || vd_aligned16(bias)),
               "vd_conv3x3_wino43_fwd: operands must be 16-byte aligned");
    Args43 a = {};
    a.x = xin; a.ldx = ldx; a.U = U43f; a.y = y; a.ldy = ldy;
    a.nimg = nimg; a.H = H; a.W = W; a.K = Cin; a.N = Cout;
    a.items_per_img = W == 64 ? H / 16 : 1;
    a.ngrp = W == 16 ? nimg / 4 : nimg * a.items_per_img;
    a.ncb = Cout / TN;
    a.bias = bias; a.res = res; a.ldr = ldres; a.stats = stats_part;
    a.chunks_per_img = (H * W) / vd_conv3x3_wino43_fwd_chunk_rows(H, W);      // ONE statement of the chunk size: what the consuming norm is told
    VD_REQUIRE(a.chunks_per_img == a.items_per_img, "vd_conv3x3_wino43_fwd: statistics chunks (%d per image) do not match the work items (%d)", a.chunks_per_img, a.items_per_img);
    const long long items = (long long)a.ngrp * a.ncb;
    VD_REQUIRE(items < (1LL << 30), "vd_conv3x3_wino43_fwd: too many work items");
    a.nitems = (int)items;
    a.probe = g_probe43;
    const int ncu = vd_persistent_cus();
    const dim3 grid((unsigned)(items < ncu ? items : ncu)), blk(THREADS);
    hipStream_t st = (hipStream_t)stream;
    if (W == 16) hipLaunchKernelGGL((wino43_conv_kernel<4, true>), grid, blk, 0, st, a);
    else if (W == 32) hipLaunchKernelGGL((wino43_conv_kernel<8, true>), grid, blk, 0, st, a);
    else hipLaunchKernelGGL((wino43_conv_kernel<16, true>), grid, blk, 0, st, a);
    VD_LAUNCH_CHECK("wino43_conv_kernel<., true>");
    g_last43 = -(W / 4);
    return 0;
}

/* all tensors in one launch: items_dev = [n][8] int64 {w, U43, fwd, Cout, Cin, 0, 0, first block}; fwd = 0: input-gradient image, (Cin/32)*(Cout/8)
 * blocks; fwd = 1: forward image (vd_wino43_pack_fwd), (Cout/32)*(Cin/8) blocks */
extern "C" int vd_wino43_pack_batched(const int64_t* items_dev, int32_t n, int64_t total_blocks, void* stream) {
    VD_REQUIRE(items_dev && n > 0 && total_blocks > 0 && total_blocks < (1LL << 31), "vd_wino43_pack_batched: bad table");
    hipLaunchKernelGGL(wino43_pack_batched_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const long long*>(items_dev), n);
    VD_LAUNCH_CHECK("wino43_pack_batched_kernel");
    return 0;
}

/* ---- weight (and bias) gradient of the 3x3 convolution through F(4x4,3x3), unfused: same arguments and result as vd_conv3x3_wgrad_wino.
 * Serves H, W multiples of 4 and Cin, Cout multiples of 4; _supported additionally requires at least 512 tiles (the K of the 36 GEMMs:
 * 8x8 images at batch 128; below that the fused F(2x2,3x3) kernel is faster) and tensors below 2 GiB. */
extern "C" int vd_conv3x3_wgrad_wino43_supported(int32_t nimg, int32_t H, int32_t W, int32_t Cin, int32_t Cout, int64_t ldx, int64_t lddy) {
    const Wg43Plan g = wg43_plan(nimg, H, W, Cin, Cout);
    if (!g.ok || ldx % 4 || lddy % 4 || g.T < 512 || g.T % 4) return 0;
    const long long px = (long long)nimg * H * W, lim = 0x7FFFFFF0LL / 4;
    if (px * ldx >= lim || px * lddy >= lim) return 0;
    if ((long long)g.T * (Cin > Cout ? Cin : Cout) * 4 >= (1LL << 40)) return 0;
    return 1;
}

extern "C" size_t vd_conv3x3_wgrad_wino43_ws_bytes(int32_t nimg, int32_t H, int32_t W, int32_t Cin, int32_t Cout) {
    const Wg43Plan g = wg43_plan(nimg, H, W, Cin, Cout);
    if (!g.ok) return 0;
    return (g.v_f + g.m_f + g.u_f + g.cs_f) * sizeof(float) + g.gemm_bytes;
}

/* phases (per-kernel timing): 1 = transforms, 2 = the 36 grouped GEMMs (+ the slab reduction of narrow layers), 4 = G^T . G fold (+ the slab
 * reduction of wide layers: FUSED_FINISH_MIN); 7 = all */
static int wgrad43_impl(const float* xin, int64_t ldx, const float* dy, int64_t lddy, int32_t nimg, int32_t H, int32_t W, int32_t Cin,
                        int32_t Cout, float* dw_oihw, float* dbias, int32_t Cin_w, int32_t Cout_w, int32_t accumulate, float* ws,
                        size_t ws_bytes, void* stream, int phases) {
    VD_REQUIRE(xin && dy && dw_oihw && ws, "vd_conv3x3_wgrad_wino43: null operand");
    VD_REQUIRE(vd_conv3x3_wgrad_wino43_supported(nimg, H, W, Cin, Cout, ldx, lddy),
               "vd_conv3x3_wgrad_wino43: unsupported geometry nimg=%d H=%d W=%d Cin=%d Cout=%d", nimg, H, W, Cin, Cout);
    VD_REQUIRE(Cin_w <= Cin && Cout_w <= Cout, "vd_conv3x3_wgrad_wino43: real dims exceed padded dims");
    VD_REQUIRE(vd_aligned16(xin) && vd_aligned16(dy) && vd_aligned16(ws), "vd_conv3x3_wgrad_wino43: operands must be 16-byte aligned");
    VD_REQUIRE(ws_bytes >= vd_conv3x3_wgrad_wino43_ws_bytes(nimg, H, W, Cin, Cout), "vd_conv3x3_wgrad_wino43: workspace too small");
    const Wg43Plan g = wg43_plan(nimg, H, W, Cin, Cout);
    float* V = ws;
    float* dM = V + g.v_f;
    float* dU = dM + g.m_f;
    float* cs = dU + g.u_f;
    float* gws = cs + g.cs_f;
    hipStream_t st = (hipStream_t)stream;
    if (phases & 1) {
        WgT43 t = {};
        t.x = xin; t.ldx = ldx; t.dy = dy; t.lddy = lddy; t.V = V; t.dM = dM;
        t.nimg = nimg; t.H = H; t.W = W; t.Cin = Cin; t.Cout = Cout; t.TW = W / 4; t.TPI = (H / 4) * (W / 4); t.T = g.T;
        const int cmax = Cin > Cout ? Cin : Cout;
        hipLaunchKernelGGL(wino43_wgrad_transform_kernel, dim3((unsigned)(g.T / 4), (unsigned)((cmax + 255) / 256), 2), dim3(64, 4), 0, st, t);
        VD_LAUNCH_CHECK("wino43_wgrad_transform_kernel");
    }
    if (phases & 2) {
        const float* A[36]; const float* B[36]; float* C[36]; float* colsum[36];
        for (int e = 0; e < 36; ++e) {
            A[e] = dM + (size_t)e * 16 * Cout; B[e] = V + (size_t)e * 16 * Cin;
            C[e] = dU + (size_t)e * Cout * Cin; colsum[e] = cs + (size_t)e * Cout;
        }
        const int rc = vd_gemm_grouped_wgrad_kblk(A, B, C, colsum, 36, Cout, Cin, g.T, Cout, Cin, Cin, g.S, gws, g.gemm_bytes, stream,
                                                  36LL * 16 * Cout, 36LL * 16 * Cin,
                                                  (long long)Cout * Cin >= FUSED_FINISH_MIN ? 1 : 0);
        if (rc) return rc;
        g_last43w = g.S;
    }
    if (phases & 4) {
        const long long tot = (long long)Cout * Cin;
        if (tot >= FUSED_FINISH_MIN) {
            const int used = vd_gemm_grouped_wgrad_used_slabs(36, Cout, Cin, g.T, g.S);
            hipLaunchKernelGGL(wino43_wgrad_reduce_finish_kernel, dim3((unsigned)((tot + 63) / 64)), dim3(64, 6), 0, st, gws,
                               gws + (size_t)36 * used * Cout * Cin, used, Cout, Cin, Cout_w, Cin_w, dw_oihw, dbias, accumulate);
        } else {
            hipLaunchKernelGGL(wino43_wgrad_finish_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, dU, cs, Cout, Cin, Cout_w, Cin_w,
                               dw_oihw, dbias, accumulate);
        }
        VD_LAUNCH_CHECK("wino43_wgrad_finish_kernel");
    }
    return 0;
}

extern "C" int vd_conv3x3_wgrad_wino43(const float* xin, int64_t ldx, const float* dy, int64_t lddy, int32_t nimg, int32_t H, int32_t W,
                                       int32_t Cin, int32_t Cout, float* dw_oihw, float* dbias, int32_t Cin_w, int32_t Cout_w,
                                       int32_t accumulate, float* ws, size_t ws_bytes, void* stream) {
    return wgrad43_impl(xin, ldx, dy, lddy, nimg, H, W, Cin, Cout, dw_oihw, dbias, Cin_w, Cout_w, accumulate, ws, ws_bytes, stream, 7);
}

extern "C" int vd_conv3x3_wgrad_wino43_phase(const float* xin, int64_t ldx, const float* dy, int64_t lddy, int32_t nimg, int32_t H,
                                             int32_t W, int32_t Cin, int32_t Cout, float* dw_oihw, float* dbias, int32_t Cin_w,
                                             int32_t Cout_w, int32_t accumulate, float* ws, size_t ws_bytes, int32_t phase, void* stream) {
    VD_REQUIRE(phase == 1 || phase == 2 || phase == 4, "vd_conv3x3_wgrad_wino43_phase: phase must be 1 (transforms), 2 (GEMMs) or 4 (reduction + finish)");
    return wgrad43_impl(xin, ldx, dy, lddy, nimg, H, W, Cin, Cout, dw_oihw, dbias, Cin_w, Cout_w, accumulate, ws, ws_bytes, stream, phase);
}

/* split-K slabs per xi plane of the calling thread's last vd_conv3x3_wgrad_wino43 launch (0: none yet) -- test / profiling aid */
extern "C" int vd_wino43_wgrad_last_kernel(void) { return g_last43w; }
