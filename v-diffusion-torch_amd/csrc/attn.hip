// attn.hip -- fused scaled-dot-product attention on the fp32 matrix cores: forward with an online softmax, backward with
// recomputation.  Reference op: BaseAttentionBlock.scaled_dot_product (unet.py:55-64): softmax(Q K^T / sqrt(hd)) V per (image, head),
// and its autograd.  The [B, nh, L, L] logits / probabilities never exist in HBM (the three-launch path vd_gemm -> vd_softmax_rows
// -> vd_gemm writes and re-reads them: 16 bytes per logit forward, 28 backward; 8.6 GB per tensor at CelebA's L = 4096 block).
//
// Operands are the rows of the qkv projection: q/k/v[(b L + l) ld + h D + d], D = head dim in {64, 128, 256}.
//
// Layout trick that keeps P in registers: v_mfma_f32_16x16x4_f32 returns D[4 (lane >> 4) + r][lane & 15] in lane's register r.
// Computing the logits transposed, S^T[key][query] = K Q^T, gives lane (li, lq) the logits of ITS query li against keys 4 lq + r:
// the row statistics of the softmax are lane-local (plus two shuffles over lq), and register r is exactly the operand the next
// MFMA wants for k-slot lq if that k-slot is declared to be key 4 lq + r -- the other operand (a V / K / dO / Q fragment read from
// LDS) simply follows the same key numbering.  The head dimension of those second products is numbered the same way
// (row i of block e <-> d = 4 i + e), so that one ds_read_b128 along d feeds four MFMAs and a lane ends up holding four
// consecutive d for its output row (dwordx4 stores).
//
// One workgroup = 4 waves = 64 queries (forward, dQ) or 64 keys (dK/dV); the other sequence dimension streams through LDS in
// tiles of 4096 / D rows, double-buffered with `buffer_load ... lds` (16 pieces of 1 KiB per tile, 4 per wave).  LDS image of a
// tile: row-major [row][D] with the 16-byte chunk index XOR-ed by row & 15 inside each aligned group of 16 chunks -- conflict-free
// both for "16 lanes read the same chunk of 16 rows" (first products) and "16 lanes read the 16 chunks of one row" (second).
// Per 64-key tile and wave (D = 64): 128 MFMAs forward, 192 dQ, 256 dK/dV against ~35 VALU instructions per lane for the softmax.
// Everything is fixed-order (no atomics): bitwise reproducible.
#include "common.h"

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
constexpr int TILE = 4096;                    // floats per LDS tile = (4096 / D) rows of D
constexpr float LOG2E = 1.4426950408889634f;

struct AttnArgs {
    const float* q; const float* k; const float* v; long long ld;
    const float* o; float* ow; long long ldo;
    const float* dout; long long lddo;
    float* dq; float* dk; float* dv; long long ldd;
    float* lse; float* delta;                // [B nh][L]: log2-domain log-sum-exp of the scaled logits; rowsum(dO * O)
    int nh, L; float scale;
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const float* base) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)0x80000000u, 0x00020000);
}

__device__ __forceinline__ int swz(int row, int c) { return (c & ~15) | ((c ^ row) & 15); }

// tile loader: piece q = wave + 4 j of the 16 pieces; lane l of piece q fills LDS chunk q * 64 + l
template <int D>
struct TileLoader {
    unsigned vo[4];
    __device__ __forceinline__ void init(int wave, int lane, long long ld) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ci = (wave + 4 * j) * 64 + lane;
            const int row = ci / (D / 4), pos = ci % (D / 4);
            vo[j] = (unsigned)((row * ld + 4 * swz(row, pos)) * 4);
        }
    }
    __device__ __forceinline__ void issue(const float* base, float* tile, int wave) const {
        const __amdgpu_buffer_rsrc_t rs = rsrc_of(base);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(tile + (wave + 4 * j) * 256), 16, (int)vo[j], 0, 0, 0);
    }
};

// "X" fragment: lane (li, lq) reads chunk c of row r0 + li ; "Y" fragment: chunk c0 + li of row `row` (= r0 + 4 lq + r')
template <int D>
__device__ __forceinline__ f32x4 frag_x(const float* tile, int r0, int li, int c) {
    const int row = r0 + li;
    return *reinterpret_cast<const f32x4*>(tile + row * D + swz(row, c) * 4);
}
template <int D>
__device__ __forceinline__ f32x4 frag_y(const float* tile, int row, int c) {
    return *reinterpret_cast<const f32x4*>(tile + row * D + swz(row, c) * 4);
}

__device__ __forceinline__ float red_lq_max(float v) {
    v = fmaxf(v, __shfl_xor(v, 16));
    return fmaxf(v, __shfl_xor(v, 32));
}
__device__ __forceinline__ float red_lq_sum(float v) {
    v += __shfl_xor(v, 16);
    return v + __shfl_xor(v, 32);
}

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// ------------------------------------------------------------------------------------------------ forward
template <int D>
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(const AttnArgs p) {
    constexpr int BK = TILE / D, NB = BK / 16, NC = D / 16, NG = D / 64;
    __shared__ __attribute__((aligned(1024))) float smem[4 * TILE];      // (K, V) x 2 stages
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lq = lane >> 4;
    const int bh = blockIdx.y, b = bh / p.nh, h = bh - b * p.nh;
    const long long row0 = (long long)b * p.L;
    const float* qb = p.q + row0 * p.ld + h * D;
    const float* kb = p.k + row0 * p.ld + h * D;
    const float* vb = p.v + row0 * p.ld + h * D;
    const int q0 = (blockIdx.x * 4 + wave) * 16;
    const float sl2 = p.scale * LOG2E;
    f32x4 qf[NC];
#pragma unroll
    for (int c4 = 0; c4 < NC; ++c4) qf[c4] = *reinterpret_cast<const f32x4*>(qb + (long long)(q0 + li) * p.ld + 16 * c4 + 4 * lq) * sl2;
    TileLoader<D> ldr;
    ldr.init(wave, lane, p.ld);
    const int nt = p.L / BK;
    f32x4 oacc[NG][4];
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) oacc[g][e] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m = -INFINITY, lsum = 0.f;
    ldr.issue(kb, smem, wave);
    ldr.issue(vb, smem + TILE, wave);
    for (int t = 0; t < nt; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                            // tile t landed; every wave is done with the other stage
        if (t + 1 < nt) {
            float* nx = smem + ((t + 1) & 1) * 2 * TILE;
            ldr.issue(kb + (long long)(t + 1) * BK * p.ld, nx, wave);
            ldr.issue(vb + (long long)(t + 1) * BK * p.ld, nx + TILE, wave);
        }
        const float* sK = smem + (t & 1) * 2 * TILE;
        const float* sV = sK + TILE;
        f32x4 s[NB];
#pragma unroll
        for (int blk = 0; blk < NB; ++blk) s[blk] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c4 = 0; c4 < NC; ++c4)
#pragma unroll
            for (int blk = 0; blk < NB; ++blk) {
                const f32x4 kf = frag_x<D>(sK, 16 * blk, li, 4 * c4 + lq);
#pragma unroll
                for (int j = 0; j < 4; ++j) s[blk] = MFMA(kf[j], qf[c4][j], s[blk]);
            }
        float mx = s[0][0];
#pragma unroll
        for (int blk = 0; blk < NB; ++blk)
#pragma unroll
            for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[blk][r]);
        mx = red_lq_max(mx);
        const float mn = fmaxf(m, mx);
        const float alpha = __builtin_amdgcn_exp2f(m - mn);
        m = mn;
        float ps = 0.f;
#pragma unroll
        for (int blk = 0; blk < NB; ++blk)
#pragma unroll
            for (int r = 0; r < 4; ++r) { s[blk][r] = __builtin_amdgcn_exp2f(s[blk][r] - mn); ps += s[blk][r]; }
        lsum = lsum * alpha + ps;
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) oacc[g][e] *= alpha;
#pragma unroll
        for (int blk = 0; blk < NB; ++blk)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * blk + 4 * lq + r;
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    const f32x4 vf = frag_y<D>(sV, row, 16 * g + li);
#pragma unroll
                    for (int e = 0; e < 4; ++e) oacc[g][e] = MFMA(vf[e], s[blk][r], oacc[g][e]);
                }
            }
    }
    lsum = red_lq_sum(lsum);
    const float inv = 1.f / lsum;
    float* orow = p.ow + (row0 + q0 + li) * p.ldo + h * D;
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const f32x4 out = {oacc[g][0][r] * inv, oacc[g][1][r] * inv, oacc[g][2][r] * inv, oacc[g][3][r] * inv};
            *reinterpret_cast<f32x4*>(orow + 64 * g + 16 * lq + 4 * r) = out;
        }
    if (p.lse && lq == 0) p.lse[(long long)bh * p.L + q0 + li] = m + __log2f(lsum);
}

// ------------------------------------------------------------------------------------------------ backward: dQ (and delta)
// workgroup = 64 queries; keys stream.  dQ[q][d] = scale * sum_j dS[q][j] K[j][d],  dS = P (dP - delta),  dP = dO V^T,
// delta[q] = sum_d dO[q][d] O[q][d] (written for the dK/dV kernel, which runs after this one)
template <int D>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(const AttnArgs p) {
    constexpr int BK = TILE / D, NB = BK / 16, NC = D / 16, NG = D / 64;
    __shared__ __attribute__((aligned(1024))) float smem[4 * TILE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lq = lane >> 4;
    const int bh = blockIdx.y, b = bh / p.nh, h = bh - b * p.nh;
    const long long row0 = (long long)b * p.L;
    const float* qb = p.q + row0 * p.ld + h * D;
    const float* kb = p.k + row0 * p.ld + h * D;
    const float* vb = p.v + row0 * p.ld + h * D;
    const int q0 = (blockIdx.x * 4 + wave) * 16;
    const float sl2 = p.scale * LOG2E;
    f32x4 qf[NC], dof[NC];
    float dl = 0.f;
    {
        const float* orow = p.o + (row0 + q0 + li) * p.ldo + h * D;
        const float* drow = p.dout + (row0 + q0 + li) * p.lddo + h * D;
#pragma unroll
        for (int c4 = 0; c4 < NC; ++c4) {
            qf[c4] = *reinterpret_cast<const f32x4*>(qb + (long long)(q0 + li) * p.ld + 16 * c4 + 4 * lq) * sl2;
            dof[c4] = *reinterpret_cast<const f32x4*>(drow + 16 * c4 + 4 * lq);
            const f32x4 of = *reinterpret_cast<const f32x4*>(orow + 16 * c4 + 4 * lq);
#pragma unroll
            for (int j = 0; j < 4; ++j) dl += of[j] * dof[c4][j];
        }
    }
    dl = red_lq_sum(dl);
    if (lq == 0) p.delta[(long long)bh * p.L + q0 + li] = dl;
    const float lse = p.lse[(long long)bh * p.L + q0 + li];
    TileLoader<D> ldr;
    ldr.init(wave, lane, p.ld);
    const int nt = p.L / BK;
    f32x4 dqacc[NG][4];
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) dqacc[g][e] = f32x4{0.f, 0.f, 0.f, 0.f};
    ldr.issue(kb, smem, wave);
    ldr.issue(vb, smem + TILE, wave);
    for (int t = 0; t < nt; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t + 1 < nt) {
            float* nx = smem + ((t + 1) & 1) * 2 * TILE;
            ldr.issue(kb + (long long)(t + 1) * BK * p.ld, nx, wave);
            ldr.issue(vb + (long long)(t + 1) * BK * p.ld, nx + TILE, wave);
        }
        const float* sK = smem + (t & 1) * 2 * TILE;
        const float* sV = sK + TILE;
        f32x4 s[NB], dp[NB];
#pragma unroll
        for (int blk = 0; blk < NB; ++blk) { s[blk] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[blk] = s[blk]; }
#pragma unroll
        for (int c4 = 0; c4 < NC; ++c4)
#pragma unroll
            for (int blk = 0; blk < NB; ++blk) {
                const f32x4 kf = frag_x<D>(sK, 16 * blk, li, 4 * c4 + lq);
                const f32x4 vf = frag_x<D>(sV, 16 * blk, li, 4 * c4 + lq);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    s[blk] = MFMA(kf[j], qf[c4][j], s[blk]);
                    dp[blk] = MFMA(vf[j], dof[c4][j], dp[blk]);
                }
            }
#pragma unroll
        for (int blk = 0; blk < NB; ++blk)
#pragma unroll
            for (int r = 0; r < 4; ++r) s[blk][r] = __builtin_amdgcn_exp2f(s[blk][r] - lse) * (dp[blk][r] - dl);
#pragma unroll
        for (int blk = 0; blk < NB; ++blk)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * blk + 4 * lq + r;
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    const f32x4 kf = frag_y<D>(sK, row, 16 * g + li);
#pragma unroll
                    for (int e = 0; e < 4; ++e) dqacc[g][e] = MFMA(kf[e], s[blk][r], dqacc[g][e]);
                }
            }
    }
    float* drow = p.dq + (row0 + q0 + li) * p.ldd + h * D;
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const f32x4 out = {dqacc[g][0][r] * p.scale, dqacc[g][1][r] * p.scale, dqacc[g][2][r] * p.scale, dqacc[g][3][r] * p.scale};
            *reinterpret_cast<f32x4*>(drow + 64 * g + 16 * lq + 4 * r) = out;
        }
}

// ------------------------------------------------------------------------------------------------ backward: dK, dV
// workgroup = 64 keys; queries stream (Q and dO tiles).  dV[j][d] = sum_q P[q][j] dO[q][d],  dK[j][d] = scale * sum_q dS[q][j] Q[q][d]
template <int D>
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(const AttnArgs p) {
    constexpr int BQ = TILE / D, NB = BQ / 16, NC = D / 16, NG = D / 64;
    __shared__ __attribute__((aligned(1024))) float smem[4 * TILE];      // (Q, dO) x 2 stages
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lq = lane >> 4;
    const int bh = blockIdx.y, b = bh / p.nh, h = bh - b * p.nh;
    const long long row0 = (long long)b * p.L;
    const float* qb = p.q + row0 * p.ld + h * D;
    const float* kb = p.k + row0 * p.ld + h * D;
    const float* vb = p.v + row0 * p.ld + h * D;
    const float* dob = p.dout + row0 * p.lddo + h * D;
    const float* lseb = p.lse + (long long)bh * p.L;
    const float* dlb = p.delta + (long long)bh * p.L;
    const int k0 = (blockIdx.x * 4 + wave) * 16;
    const float sl2 = p.scale * LOG2E;
    f32x4 kf[NC], vf[NC];
#pragma unroll
    for (int c4 = 0; c4 < NC; ++c4) {
        kf[c4] = *reinterpret_cast<const f32x4*>(kb + (long long)(k0 + li) * p.ld + 16 * c4 + 4 * lq) * sl2;
        vf[c4] = *reinterpret_cast<const f32x4*>(vb + (long long)(k0 + li) * p.ld + 16 * c4 + 4 * lq);
    }
    TileLoader<D> ldq, ldo;
    ldq.init(wave, lane, p.ld);
    ldo.init(wave, lane, p.lddo);
    const int nt = p.L / BQ;
    f32x4 dkacc[NG][4], dvacc[NG][4];
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) { dkacc[g][e] = f32x4{0.f, 0.f, 0.f, 0.f}; dvacc[g][e] = dkacc[g][e]; }
    ldq.issue(qb, smem, wave);
    ldo.issue(dob, smem + TILE, wave);
    for (int t = 0; t < nt; ++t) {
        f32x4 lse4[NB], dl4[NB];
#pragma unroll
        for (int blk = 0; blk < NB; ++blk) {
            lse4[blk] = *reinterpret_cast<const f32x4*>(lseb + t * BQ + 16 * blk + 4 * lq);
            dl4[blk] = *reinterpret_cast<const f32x4*>(dlb + t * BQ + 16 * blk + 4 * lq);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t + 1 < nt) {
            float* nx = smem + ((t + 1) & 1) * 2 * TILE;
            ldq.issue(qb + (long long)(t + 1) * BQ * p.ld, nx, wave);
            ldo.issue(dob + (long long)(t + 1) * BQ * p.lddo, nx + TILE, wave);
        }
        const float* sQ = smem + (t & 1) * 2 * TILE;
        const float* sO = sQ + TILE;
        f32x4 s[NB], dp[NB];
#pragma unroll
        for (int blk = 0; blk < NB; ++blk) { s[blk] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[blk] = s[blk]; }
#pragma unroll
        for (int c4 = 0; c4 < NC; ++c4)
#pragma unroll
            for (int blk = 0; blk < NB; ++blk) {
                const f32x4 qx = frag_x<D>(sQ, 16 * blk, li, 4 * c4 + lq);
                const f32x4 ox = frag_x<D>(sO, 16 * blk, li, 4 * c4 + lq);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    s[blk] = MFMA(qx[j], kf[c4][j], s[blk]);          // rows = queries 4 lq + r, column = key li
                    dp[blk] = MFMA(ox[j], vf[c4][j], dp[blk]);
                }
            }
#pragma unroll
        for (int blk = 0; blk < NB; ++blk)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                s[blk][r] = __builtin_amdgcn_exp2f(s[blk][r] - lse4[blk][r]);     // P
                dp[blk][r] = s[blk][r] * (dp[blk][r] - dl4[blk][r]);               // dS / scale
            }
#pragma unroll
        for (int blk = 0; blk < NB; ++blk)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * blk + 4 * lq + r;
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    const f32x4 oy = frag_y<D>(sO, row, 16 * g + li);
                    const f32x4 qy = frag_y<D>(sQ, row, 16 * g + li);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        dvacc[g][e] = MFMA(oy[e], s[blk][r], dvacc[g][e]);
                        dkacc[g][e] = MFMA(qy[e], dp[blk][r], dkacc[g][e]);
                    }
                }
            }
    }
    float* dkrow = p.dk + (row0 + k0 + li) * p.ldd + h * D;
    float* dvrow = p.dv + (row0 + k0 + li) * p.ldd + h * D;
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const f32x4 ok = {dkacc[g][0][r] * p.scale, dkacc[g][1][r] * p.scale, dkacc[g][2][r] * p.scale, dkacc[g][3][r] * p.scale};
            const f32x4 ov = {dvacc[g][0][r], dvacc[g][1][r], dvacc[g][2][r], dvacc[g][3][r]};
            *reinterpret_cast<f32x4*>(dkrow + 64 * g + 16 * lq + 4 * r) = ok;
            *reinterpret_cast<f32x4*>(dvrow + 64 * g + 16 * lq + 4 * r) = ov;
        }
}

bool shape_ok(int L, int hd, bool bwd) {
    if (L <= 0 || L % 64) return false;
    return hd == 64 || hd == 128 || hd == 256;
}

}  // namespace

extern "C" int vd_attn_supported(int32_t L, int32_t hd, int32_t backward) { return shape_ok(L, hd, backward != 0) ? 1 : 0; }

extern "C" int vd_attn_fwd(const float* q, const float* k, const float* v, int64_t ld, float* o, int64_t ldo, float* lse, int32_t B,
                           int32_t nh, int32_t L, int32_t hd, float scale, void* stream) {
    VD_REQUIRE(shape_ok(L, hd, false), "vd_attn_fwd: L=%d hd=%d not served (L %% 64 == 0, hd in {64,128,256})", L, hd);
    VD_REQUIRE(q && k && v && o && B > 0 && nh > 0, "vd_attn_fwd: null operand / empty batch");
    VD_REQUIRE(ld % 4 == 0 && ldo % 4 == 0 && vd_aligned16(q) && vd_aligned16(k) && vd_aligned16(v) && vd_aligned16(o),
               "vd_attn_fwd: operands must be 16-byte aligned with row pitches that are multiples of 4 floats");
    VD_REQUIRE((long long)L * ld * 4 < (1LL << 31), "vd_attn_fwd: one image's qkv rows must stay below 2 GiB");
    VD_REQUIRE((long long)B * nh <= 65535, "vd_attn_fwd: B * nh = %lld exceeds the grid limit 65535", (long long)B * nh);
    AttnArgs a = {};
    a.q = q; a.k = k; a.v = v; a.ld = ld; a.ow = o; a.ldo = ldo; a.lse = lse; a.nh = nh; a.L = L; a.scale = scale;
    const dim3 grid(L / 64, B * nh);
    hipStream_t st = (hipStream_t)stream;
    if (hd == 64) hipLaunchKernelGGL(attn_fwd_kernel<64>, grid, dim3(256), 0, st, a);
    else if (hd == 128) hipLaunchKernelGGL(attn_fwd_kernel<128>, grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(attn_fwd_kernel<256>, grid, dim3(256), 0, st, a);
    VD_LAUNCH_CHECK("attn_fwd_kernel");
    return 0;
}

namespace {
int attn_bwd_impl(const float* q, const float* k, const float* v, int64_t ld, const float* o, int64_t ldo, const float* dout,
                  int64_t lddo, const float* lse, float* delta, float* dq, float* dk, float* dv, int64_t ldd, int32_t B,
                  int32_t nh, int32_t L, int32_t hd, float scale, int phase, void* stream) {
    VD_REQUIRE(shape_ok(L, hd, true), "vd_attn_bwd: L=%d hd=%d not served (L %% 64 == 0, hd in {64,128,256})", L, hd);
    VD_REQUIRE(q && k && v && o && dout && lse && delta && dq && dk && dv && B > 0 && nh > 0, "vd_attn_bwd: null operand / empty batch");
    VD_REQUIRE(ld % 4 == 0 && ldo % 4 == 0 && lddo % 4 == 0 && ldd % 4 == 0 && vd_aligned16(q) && vd_aligned16(k) && vd_aligned16(v) &&
               vd_aligned16(o) && vd_aligned16(dout) && vd_aligned16(dq) && vd_aligned16(dk) && vd_aligned16(dv) && vd_aligned16(lse) &&
               vd_aligned16(delta), "vd_attn_bwd: operands must be 16-byte aligned with row pitches that are multiples of 4 floats");
    VD_REQUIRE((long long)L * ld * 4 < (1LL << 31) && (long long)L * lddo * 4 < (1LL << 31), "vd_attn_bwd: one image's rows must stay below 2 GiB");
    VD_REQUIRE((long long)B * nh <= 65535, "vd_attn_bwd: B * nh = %lld exceeds the grid limit 65535", (long long)B * nh);
    AttnArgs a = {};
    a.q = q; a.k = k; a.v = v; a.ld = ld; a.o = o; a.ldo = ldo; a.dout = dout; a.lddo = lddo; a.dq = dq; a.dk = dk; a.dv = dv; a.ldd = ldd;
    a.lse = const_cast<float*>(lse); a.delta = delta; a.nh = nh; a.L = L; a.scale = scale;
    const dim3 grid(L / 64, B * nh);
    hipStream_t st = (hipStream_t)stream;
    // phase 1: dQ and delta ; phase 2: dK, dV (needs phase 1's delta) ; 0: both
    if (phase != 2) {
        if (hd == 64) hipLaunchKernelGGL(attn_bwd_dq_kernel<64>, grid, dim3(256), 0, st, a);
        else if (hd == 128) hipLaunchKernelGGL(attn_bwd_dq_kernel<128>, grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL(attn_bwd_dq_kernel<256>, grid, dim3(256), 0, st, a);
        VD_LAUNCH_CHECK("attn_bwd_dq_kernel");
    }
    if (phase != 1) {
        if (hd == 64) hipLaunchKernelGGL(attn_bwd_dkv_kernel<64>, grid, dim3(256), 0, st, a);
        else if (hd == 128) hipLaunchKernelGGL(attn_bwd_dkv_kernel<128>, grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL(attn_bwd_dkv_kernel<256>, grid, dim3(256), 0, st, a);
        VD_LAUNCH_CHECK("attn_bwd_dkv_kernel");
    }
    return 0;
}
}  // namespace

extern "C" int vd_attn_bwd(const float* q, const float* k, const float* v, int64_t ld, const float* o, int64_t ldo, const float* dout,
                           int64_t lddo, const float* lse, float* delta, float* dq, float* dk, float* dv, int64_t ldd, int32_t B,
                           int32_t nh, int32_t L, int32_t hd, float scale, void* stream) {
    return attn_bwd_impl(q, k, v, ld, o, ldo, dout, lddo, lse, delta, dq, dk, dv, ldd, B, nh, L, hd, scale, 0, stream);
}

extern "C" int vd_attn_bwd_phase(const float* q, const float* k, const float* v, int64_t ld, const float* o, int64_t ldo,
                                 const float* dout, int64_t lddo, const float* lse, float* delta, float* dq, float* dk, float* dv,
                                 int64_t ldd, int32_t B, int32_t nh, int32_t L, int32_t hd, float scale, int32_t phase, void* stream) {
    VD_REQUIRE(phase == 1 || phase == 2, "vd_attn_bwd_phase: phase must be 1 (dQ, delta) or 2 (dK, dV)");
    return attn_bwd_impl(q, k, v, ld, o, ldo, dout, lddo, lse, delta, dq, dk, dv, ldd, B, nh, L, hd, scale, phase, stream);
}
