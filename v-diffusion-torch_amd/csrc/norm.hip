// norm.hip -- GroupNorm(32,C) + SiLU + FiLM + dropout + 2x resample, forward and backward, NHWC.
// Reference call sites: nn.GroupNorm unet.py:28-30 (<- :52,119,123,230), nn.SiLU :25, FiLM :145-146,
// nn.Dropout :135,147, nn.AvgPool2d/nn.Upsample :127-132, and their autograd backward.
//
// These kernels are HBM-bound.  Work split:
//   chan_reduce  : per-(image, pixel-chunk, channel) partial sums, float4 over channels so every wave reads
//                  whole contiguous NHWC rows; thread-private accumulators, deterministic LDS tree, no atomics
//   finalize     : tiny kernels turning partials into per-(image,group) statistics / per-(image,channel) tables
//   apply        : float4 elementwise pass reading x once (+ an L1/L2-resident coefficient table)
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int PPC = 128;     // pixels per reduction chunk

// streaming accesses of the single-pass backward kernel: every byte is touched once per launch, so they may bypass the caches' retention
// (non-temporal).  Same-box A/B (tests/probe/r04_pass13.sh): -4 ... -7 % for slabs whose pixel rows are whole 128-byte lines (32-channel
// slabs: 32x32x256 105 -> 98 us, x512 189 -> 178), +19 % for 96-byte rows (24-channel slabs of C = 384), so the launcher instantiates NT only
// for CS % 32 == 0; the apply pass (whole pixel rows) measured +-3 % either way and stays plain.
template <bool NT>
__device__ __forceinline__ f32x4 ld_stream(const float* p) {
    if (NT) return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
    return *reinterpret_cast<const f32x4*>(p);
}
template <bool NT>
__device__ __forceinline__ void st_stream(float* p, f32x4 v) {
    if (NT) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p));
    else *reinterpret_cast<f32x4*>(p) = v;
}

struct ReduceArgs {
    const float* x; long long ldx;        // MODE 0/2: data ; MODE 1: forward input of the norm
    const float* dy; long long lddy;      // MODE 1: upstream gradient (output resolution)
    const float* coef;                    // MODE 1: [nimg][4][C]
    int act; float p_drop; unsigned long long seed; int resample; int has_norm;
    int H, W;                             // input resolution (MODE 1)
    long long HW;                         // pixels per image (rows for MODE 2)
    int C, nimg, chunks;
    float* part;                          // [nimg][chunks][2][C]
};

// upstream gradient g (already gathered at the forward input's resolution) -> gradient wrt the pre-activation z: dropout mask, SiLU'
__device__ __forceinline__ f32x4 dz_finish(const ReduceArgs& p, f32x4 g, f32x4 xv, f32x4 sc, f32x4 of, unsigned long long vec_index) {
    if (p.p_drop > 0.f) g *= vd_dropout_scale4(p.seed, vec_index, p.p_drop);
    if (p.act) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float z = xv[j] * sc[j] + of[j];
            const float s = vd_sigmoid(z);
            g[j] *= s * (1.f + z * (1.f - s));
        }
    }
    return g;
}

// gradient wrt the pre-activation z of one float4 of the forward input (pixel `pix` of an H x W image)
__device__ __forceinline__ f32x4 dz_of(const ReduceArgs& p, const float* dy_img, int pix, int c4, f32x4 xv, f32x4 sc,
                                       f32x4 of, unsigned long long vec_index) {
    f32x4 g;
    if (p.resample == VD_RS_NONE) {
        g = *reinterpret_cast<const f32x4*>(dy_img + (long long)pix * p.lddy + c4);
    } else {
        const int y = pix / p.W, x = pix - y * p.W;
        if (p.resample == VD_RS_DOWN) {
            const int Wo = p.W >> 1;
            g = *reinterpret_cast<const f32x4*>(dy_img + ((long long)(y >> 1) * Wo + (x >> 1)) * p.lddy + c4) * 0.25f;
        } else {
            const int Wo = p.W << 1;
            const float* b0 = dy_img + ((long long)(2 * y) * Wo + 2 * x) * p.lddy + c4;
            g = *reinterpret_cast<const f32x4*>(b0) + *reinterpret_cast<const f32x4*>(b0 + p.lddy) +
                *reinterpret_cast<const f32x4*>(b0 + Wo * p.lddy) + *reinterpret_cast<const f32x4*>(b0 + Wo * p.lddy + p.lddy);
        }
    }
    return dz_finish(p, g, xv, sc, of, vec_index);
}

template <int MODE>
__global__ __launch_bounds__(256) void chan_reduce_kernel(const ReduceArgs p, int Cb) {
    __shared__ float red[256][8];
    const int chunk = blockIdx.x, b = blockIdx.y, c0 = blockIdx.z * Cb;
    const int cb = min(Cb, p.C - c0);
    const int vecs = cb >> 2, rows = 256 / vecs;
    const int tid = threadIdx.x, r = tid / vecs, v = tid % vecs;
    const int c4 = c0 + 4 * v;
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
    const long long pix0 = (long long)chunk * PPC;
    const int np = (int)min((long long)PPC, p.HW - pix0);
    if (r < rows) {
        const float* ximg = p.x + (long long)b * p.HW * p.ldx;
        if (MODE == 1) {
            const float* cf = p.coef + (long long)b * 4 * p.C;
            f32x4 sc = {1.f, 1.f, 1.f, 1.f}, of = {0.f, 0.f, 0.f, 0.f}, nr = sc, nm = of;
            if (p.has_norm) {
                sc = *reinterpret_cast<const f32x4*>(cf + c4);
                of = *reinterpret_cast<const f32x4*>(cf + p.C + c4);
                nr = *reinterpret_cast<const f32x4*>(cf + 2 * p.C + c4);
                nm = *reinterpret_cast<const f32x4*>(cf + 3 * p.C + c4);
            }
            const long long HWo = p.resample == VD_RS_DOWN ? p.HW / 4 : (p.resample == VD_RS_UP ? p.HW * 4 : p.HW);
            const float* dimg = p.dy + (long long)b * HWo * p.lddy;
            if (p.resample == VD_RS_NONE) {
                // four pixels per trip, their eight loads in flight together (one load pair per trip left the pass latency-bound)
                for (int i = r; i < np; i += 4 * rows) {
                    f32x4 xv[4], g[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int pix = (int)pix0 + min(i + u * rows, np - 1);
                        xv[u] = *reinterpret_cast<const f32x4*>(ximg + (long long)pix * p.ldx + c4);
                        g[u] = *reinterpret_cast<const f32x4*>(dimg + (long long)pix * p.lddy + c4);
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (i + u * rows < np) {
                            const int pix = (int)pix0 + i + u * rows;
                            const unsigned long long vi = ((unsigned long long)b * p.HW + pix) * (p.C >> 2) + (c4 >> 2);
                            const f32x4 dz = dz_finish(p, g[u], xv[u], sc, of, vi);
                            a0 += dz;
                            a1 += dz * (xv[u] * nr + nm);
                        }
                }
            } else
            for (int i = r; i < np; i += rows) {
                const int pix = (int)pix0 + i;
                const f32x4 xv = *reinterpret_cast<const f32x4*>(ximg + (long long)pix * p.ldx + c4);
                const unsigned long long vi = ((unsigned long long)b * p.HW + pix) * (p.C >> 2) + (c4 >> 2);
                const f32x4 dz = dz_of(p, dimg, pix, c4, xv, sc, of, vi);
                a0 += dz;
                a1 += dz * (xv * nr + nm);
            }
        } else {
            for (int i = r; i < np; i += rows) {
                const f32x4 xv = *reinterpret_cast<const f32x4*>(ximg + (pix0 + i) * p.ldx + c4);
                a0 += xv;
                if (MODE == 0) a1 += xv * xv;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) { red[tid][j] = a0[j]; red[tid][4 + j] = a1[j]; }
    __syncthreads();
    if (tid < vecs) {
        f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
        for (int q = 0; q < rows; ++q) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { s0[j] += red[q * vecs + tid][j]; s1[j] += red[q * vecs + tid][4 + j]; }
        }
        float* o = p.part + ((long long)(b * p.chunks + chunk) * 2) * p.C + c4;
        *reinterpret_cast<f32x4*>(o) = s0;
        *reinterpret_cast<f32x4*>(o + p.C) = s1;
    }
}

__global__ void gn_stats_finalize_kernel(const float* part, int nimg, int chunks, int C, int G, long long HW, float eps,
                                         float* stats) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nimg * G) return;
    const int b = idx / G, g = idx % G, cg = C / G;
    double s1 = 0.0, s2 = 0.0;
    for (int ch = 0; ch < chunks; ++ch) {
        const float* q = part + ((long long)(b * chunks + ch) * 2) * C + g * cg;
        for (int c = 0; c < cg; ++c) { s1 += (double)q[c]; s2 += (double)q[C + c]; }
    }
    const double n = (double)cg * (double)HW;
    const double mean = s1 / n;
    double var = s2 / n - mean * mean;
    if (var < 0.0) var = 0.0;
    stats[2 * idx] = (float)mean;
    stats[2 * idx + 1] = (float)(1.0 / sqrt(var + (double)eps));
}

// statistics from the partials that producer GEMM epilogues left behind (vd_gemm / vd_conv3x3 `stats`): the normalised
// tensor is the channel concatenation of up to two produced tensors, each with its own chunk size
__global__ __launch_bounds__(256) void gn_stats_from_partials_kernel(const float* p1, int C1, int ch1, const float* p2, int C2,
                                                                     int ch2, int nimg, int G, long long HW, float eps,
                                                                     float* stats, const float* gamma, const float* beta,
                                                                     const float* film, float* coef) {
    // one wave per (image, group): lanes stride over the group's (chunk, channel) partials, fixed-order butterfly in fp64;
    // with `coef` the same wave also writes the group's rows of the apply table (what gn_coef_kernel does from `stats`)
    const int idx = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (idx >= nimg * G) return;
    const int b = idx / G, g = idx % G, C = C1 + C2, cg = C / G;
    double s1 = 0.0, s2 = 0.0;
    const int cbeg = g * cg;
    // a group never straddles the two sources unless C1 is not a multiple of cg: handle the general case per channel
    const int n1 = ch1 * cg, n2 = ch2 * cg;
    for (int e = lane; e < max(n1, n2); e += 64) {
        const int c = cbeg + e % cg, ch = e / cg;
        const bool first = c < C1;
        const int Cs = first ? C1 : C2, cs = first ? c : c - C1, chunks = first ? ch1 : ch2;
        if (ch < chunks) {
            const float* q = (first ? p1 : p2) + ((long long)(b * chunks + ch) * 2) * Cs + cs;
            s1 += (double)q[0]; s2 += (double)q[Cs];
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
    const double n = (double)cg * (double)HW;
    const double mean_d = s1 / n;
    double var = s2 / n - mean_d * mean_d;
    if (var < 0.0) var = 0.0;
    const float mean = (float)mean_d, rstd = (float)(1.0 / sqrt(var + (double)eps));
    if (stats && lane == 0) { stats[2 * idx] = mean; stats[2 * idx + 1] = rstd; }
    if (coef) {
        for (int c = cbeg + lane; c < cbeg + cg; c += 64) {
            float sc = rstd * gamma[c];
            float of = beta[c] - mean * sc;
            if (film) {
                const float shift = film[(long long)b * 2 * C + c], scale = film[(long long)b * 2 * C + C + c];
                sc *= (1.f + scale);
                of = of * (1.f + scale) + shift;
            }
            float* o = coef + (long long)b * 4 * C + c;
            o[0] = sc; o[C] = of; o[2 * C] = rstd; o[3 * C] = -mean * rstd;
        }
    }
}

// coef[b][0][c] = rstd*gamma*(1+scale)   coef[b][1][c] = (beta - mean*rstd*gamma)*(1+scale) + shift
// coef[b][2][c] = rstd                   coef[b][3][c] = -mean*rstd
__global__ void gn_coef_kernel(const float* stats, const float* gamma, const float* beta, const float* film, int nimg,
                               int C, int G, float* coef) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nimg * C) return;
    const int b = idx / C, c = idx % C, g = c / (C / G);
    const float mean = stats[2 * (b * G + g)], rstd = stats[2 * (b * G + g) + 1];
    float sc = rstd * gamma[c];
    float of = beta[c] - mean * sc;
    if (film) {
        const float shift = film[(long long)b * 2 * C + c], scale = film[(long long)b * 2 * C + C + c];
        sc *= (1.f + scale);
        of = of * (1.f + scale) + shift;
    }
    float* o = coef + (long long)b * 4 * C + c;
    o[0] = sc; o[C] = of; o[2 * C] = rstd; o[3 * C] = -mean * rstd;
}

struct ApplyArgs {
    const float* x; long long ldx; const float* coef; int has_norm; int act; float p_drop; unsigned long long seed;
    int resample; float* y; long long ldy; int nimg, H, W, C;
};
// statistics straight from the partial sums a producer epilogue left behind (vd_gn_apply_from_partials): every workgroup of the apply pass
// derives the (mean, rstd) of the groups it touches itself -- a few KB of L2-resident partials per image -- instead of a separate
// 5-microsecond launch per norm in front of it (73 per CIFAR UNet pass); the workgroups of pixel chunk 0 also write the [4][C] table
// the backward pass reads
struct PartArgs {
    const float* p1; int C1, ch1; const float* p2; int C2, ch2; int G; float eps;
    const float* gamma; const float* beta; const float* film; float* coef_out;
};

__device__ __forceinline__ f32x4 fwd_val(const ApplyArgs& p, f32x4 v, int pix, int c4, f32x4 sc, f32x4 of, int b) {
    if (p.has_norm) v = v * sc + of;
    if (p.act) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = v[j] * vd_sigmoid(v[j]);
    }
    if (p.p_drop > 0.f) {
        const unsigned long long vi = ((unsigned long long)b * p.H * p.W + pix) * (p.C >> 2) + (c4 >> 2);
        v *= vd_dropout_scale4(p.seed, vi, p.p_drop);
    }
    return v;
}
__device__ __forceinline__ f32x4 fwd_one(const ApplyArgs& p, const float* ximg, int pix, int c4, f32x4 sc, f32x4 of, int b) {
    return fwd_val(p, *reinterpret_cast<const f32x4*>(ximg + (long long)pix * p.ldx + c4), pix, c4, sc, of, b);
}

constexpr int APIX = 64;     // output pixels per workgroup of the apply kernels

// grid (pixel chunks, images, channel splits): a thread keeps ONE float4 of channels for all its pixels, so the
// coefficient table is read once per thread and the loop has no 64-bit index arithmetic
template <bool FROM_PARTS>
__global__ __launch_bounds__(256) void gn_apply_kernel(const ApplyArgs p, int Cb, const PartArgs q) {
    __shared__ double ps[FROM_PARTS ? 2 : 1][FROM_PARTS ? 1024 : 1];     // per-channel sums of the workgroup's channel range
    __shared__ float gst[FROM_PARTS ? 2 : 1][FROM_PARTS ? 256 : 1];       // (mean, rstd) of its groups
    const int b = blockIdx.y, c0 = blockIdx.z * Cb;
    const int cbw = min(Cb, p.C - c0);
    const int vecs = cbw >> 2, rows = 256 / vecs;
    const int r = threadIdx.x / vecs, c4 = c0 + 4 * (threadIdx.x % vecs);
    const int Ho = p.resample == VD_RS_DOWN ? p.H >> 1 : (p.resample == VD_RS_UP ? p.H << 1 : p.H);
    const int Wo = p.resample == VD_RS_DOWN ? p.W >> 1 : (p.resample == VD_RS_UP ? p.W << 1 : p.W);
    const int p0 = blockIdx.x * APIX, np = min(APIX, Ho * Wo - p0);
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, of = {0.f, 0.f, 0.f, 0.f};
    if (FROM_PARTS) {
        // (the launcher gives a workgroup whole groups: Cb % (C / G) == 0)
        const int C = p.C, cg = C / q.G;
        for (int cl = threadIdx.x; cl < cbw; cl += 256) {
            const int c = c0 + cl;
            const bool first = c < q.C1;
            const int Cs = first ? q.C1 : q.C2, cs = first ? c : c - q.C1, chunks = first ? q.ch1 : q.ch2;
            const float* src = (first ? q.p1 : q.p2) + (long long)b * chunks * 2 * Cs + cs;
            double s1 = 0.0, s2 = 0.0;
            for (int ch = 0; ch < chunks; ++ch) { s1 += (double)src[(long long)ch * 2 * Cs]; s2 += (double)src[(long long)ch * 2 * Cs + Cs]; }
            ps[0][cl] = s1; ps[1][cl] = s2;
        }
        __syncthreads();
        const int ng = cbw / cg;
        for (int g = threadIdx.x; g < ng; g += 256) {
            double s1 = 0.0, s2 = 0.0;
            for (int c = g * cg; c < (g + 1) * cg; ++c) { s1 += ps[0][c]; s2 += ps[1][c]; }
            const double n = (double)cg * (double)p.H * (double)p.W;
            const double mean = s1 / n;
            double var = s2 / n - mean * mean;
            if (var < 0.0) var = 0.0;
            gst[0][g] = (float)mean; gst[1][g] = (float)(1.0 / sqrt(var + (double)q.eps));
        }
        __syncthreads();
        if (r >= rows) return;
        f32x4 nr, nm;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = c4 + j, g = (c - c0) / cg;
            const float mean = gst[0][g], rstd = gst[1][g];
            float s_ = rstd * q.gamma[c];
            float o_ = q.beta[c] - mean * s_;
            if (q.film) {
                const float shift = q.film[(long long)b * 2 * C + c], scale = q.film[(long long)b * 2 * C + C + c];
                s_ *= (1.f + scale);
                o_ = o_ * (1.f + scale) + shift;
            }
            sc[j] = s_; of[j] = o_; nr[j] = rstd; nm[j] = -mean * rstd;
        }
        if (blockIdx.x == 0 && r == 0 && q.coef_out) {
            float* o = q.coef_out + (long long)b * 4 * C + c4;
            *reinterpret_cast<f32x4*>(o) = sc; *reinterpret_cast<f32x4*>(o + C) = of;
            *reinterpret_cast<f32x4*>(o + 2 * C) = nr; *reinterpret_cast<f32x4*>(o + 3 * C) = nm;
        }
    } else {
        if (r >= rows) return;
        if (p.has_norm) {
            const float* cf = p.coef + (long long)b * 4 * p.C;
            sc = *reinterpret_cast<const f32x4*>(cf + c4);
            of = *reinterpret_cast<const f32x4*>(cf + p.C + c4);
        }
    }
    const float* ximg = p.x + (long long)b * p.H * p.W * p.ldx;
    float* yimg = p.y + (long long)b * Ho * Wo * p.ldy;
    if (p.resample == VD_RS_NONE) {
        // four loads in flight per thread before the first value is used (one per iteration kept the kernel at 4.3 TB/s: the loop
        // is latency-bound, not bandwidth-bound, at 8 resident workgroups per CU)
        for (int i = r; i < np; i += 4 * rows) {
            f32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (i + u * rows < np) v[u] = *reinterpret_cast<const f32x4*>(ximg + (long long)(p0 + i + u * rows) * p.ldx + c4);
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (i + u * rows < np) {
                    const int po = p0 + i + u * rows;
                    *reinterpret_cast<f32x4*>(yimg + (long long)po * p.ldy + c4) = fwd_val(p, v[u], po, c4, sc, of, b);
                }
        }
        return;
    }
    for (int i = r; i < np; i += rows) {
        const int po = p0 + i;
        f32x4 out;
        {
            const int yo = po / Wo, xo = po - yo * Wo;
            if (p.resample == VD_RS_UP) out = fwd_one(p, ximg, (yo >> 1) * p.W + (xo >> 1), c4, sc, of, b);
            else {
                const int q = (2 * yo) * p.W + 2 * xo;
                out = (fwd_one(p, ximg, q, c4, sc, of, b) + fwd_one(p, ximg, q + 1, c4, sc, of, b) +
                       fwd_one(p, ximg, q + p.W, c4, sc, of, b) + fwd_one(p, ximg, q + p.W + 1, c4, sc, of, b)) * 0.25f;
            }
        }
        *reinterpret_cast<f32x4*>(yimg + (long long)po * p.ldy + c4) = out;
    }
}

// ---------------------------------------------------------------- backward finalize: one block per image
// in : part [chunks][2][C] partial sums of dz and dz*xhat
// out: q [b][3][C] = {rstd*k, rstd*m1, rstd*m2};  dfilm[b][2C];  pgb [b][2][C] per-image dgamma/dbeta terms
__global__ __launch_bounds__(256) void gn_bwd_finalize_kernel(const float* part, int chunks, const float* coef,
                                                              const float* gamma, const float* beta, const float* film,
                                                              int C, int G, long long HW, float* q, float* dfilm,
                                                              float* pgb) {
    extern __shared__ float sh[];      // [2][C] k*S1, k*S2 ; then [2][G]
    float* k1 = sh; float* k2 = sh + C; float* gm = sh + 2 * C;
    const int b = blockIdx.x, cg = C / G;
    const float* cf = coef + (long long)b * 4 * C;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float s1 = 0.f, s2 = 0.f;
        for (int ch = 0; ch < chunks; ++ch) {
            const float* pp = part + ((long long)(b * chunks + ch) * 2) * C;
            s1 += pp[c]; s2 += pp[C + c];
        }
        float fs = 1.f;
        if (film) {
            fs = 1.f + film[(long long)b * 2 * C + C + c];
            dfilm[(long long)b * 2 * C + c] = s1;                                   // d shift
            dfilm[(long long)b * 2 * C + C + c] = gamma[c] * s2 + beta[c] * s1;     // d scale = sum dz * (gamma*xhat + beta)
        }
        const float k = gamma[c] * fs;
        k1[c] = k * s1; k2[c] = k * s2;
        pgb[((long long)b * 2) * C + c] = fs * s2;          // dgamma term
        pgb[((long long)b * 2 + 1) * C + c] = fs * s1;      // dbeta term
    }
    __syncthreads();
    if (threadIdx.x < G) {
        float m1 = 0.f, m2 = 0.f;
        for (int c = threadIdx.x * cg; c < (threadIdx.x + 1) * cg; ++c) { m1 += k1[c]; m2 += k2[c]; }
        const float inv = 1.f / ((float)cg * (float)HW);
        gm[threadIdx.x] = m1 * inv; gm[G + threadIdx.x] = m2 * inv;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const int g = c / cg;
        const float rstd = cf[2 * C + c];
        const float fs = film ? 1.f + film[(long long)b * 2 * C + C + c] : 1.f;
        float* qq = q + (long long)b * 3 * C + c;
        qq[0] = rstd * gamma[c] * fs; qq[C] = rstd * gm[g]; qq[2 * C] = rstd * gm[G + g];
    }
}

// out[c] (+)= sum_b in[b][c]   (two planes at once: dgamma, dbeta).  Block = 16 channels x 16 lanes over the images,
// fixed-order LDS tree (deterministic).
__global__ __launch_bounds__(256) void sum_over_images_kernel(const float* pgb, int nimg, int C, float* dgamma, float* dbeta,
                                                              int accumulate) {
    __shared__ float sh[2][16][17];
    const int cx = threadIdx.x & 15, ly = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cx;
    float a = 0.f, bsum = 0.f;
    if (c < C)
        for (int b = ly; b < nimg; b += 16) { a += pgb[((long long)b * 2) * C + c]; bsum += pgb[((long long)b * 2 + 1) * C + c]; }
    sh[0][ly][cx] = a; sh[1][ly][cx] = bsum;
    __syncthreads();
    if (ly == 0 && c < C) {
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) { s0 += sh[0][q][cx]; s1 += sh[1][q][cx]; }
        dgamma[c] = accumulate ? dgamma[c] + s0 : s0;
        dbeta[c] = accumulate ? dbeta[c] + s1 : s1;
    }
}

// the same sums for MANY norms in one launch (the per-norm launch is 5 us of latency, 73 times per CIFAR train step):
// items: 8 x int64 per norm {pgb, dgamma, dbeta, nimg, C, accumulate, -, first block}; a norm takes ceil(C / 16) blocks
__global__ __launch_bounds__(256) void sum_over_images_batched_kernel(const long long* items, int n) {
    __shared__ float sh[2][16][17];
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (items[8 * mid + 7] <= (long long)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const long long* it = items + 8 * lo;
    const float* pgb = reinterpret_cast<const float*>(it[0]);
    float* dgamma = reinterpret_cast<float*>(it[1]);
    float* dbeta = reinterpret_cast<float*>(it[2]);
    const int nimg = (int)it[3], C = (int)it[4], accumulate = (int)it[5];
    const int cx = threadIdx.x & 15, ly = threadIdx.x >> 4;
    const int c = (int)((long long)blockIdx.x - it[7]) * 16 + cx;
    float a = 0.f, bsum = 0.f;
    if (c < C)
        for (int b = ly; b < nimg; b += 16) { a += pgb[((long long)b * 2) * C + c]; bsum += pgb[((long long)b * 2 + 1) * C + c]; }
    sh[0][ly][cx] = a; sh[1][ly][cx] = bsum;
    __syncthreads();
    if (ly == 0 && c < C) {
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) { s0 += sh[0][q][cx]; s1 += sh[1][q][cx]; }
        dgamma[c] = accumulate ? dgamma[c] + s0 : s0;
        dbeta[c] = accumulate ? dbeta[c] + s1 : s1;
    }
}

struct BwdApplyArgs {
    ReduceArgs r; const float* q; const float* add; long long ldadd; float* dx; long long lddx; int accumulate_dx;
};

__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const BwdApplyArgs a, int Cb) {
    const ReduceArgs& p = a.r;
    const int b = blockIdx.y, c0 = blockIdx.z * Cb;
    const int vecs = min(Cb, p.C - c0) >> 2, rows = 256 / vecs;
    const int r = threadIdx.x / vecs, c4 = c0 + 4 * (threadIdx.x % vecs);
    if (r >= rows) return;
    const int HW = (int)p.HW;
    const int p0 = blockIdx.x * APIX, np = min(APIX, HW - p0);
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, of = {0.f, 0.f, 0.f, 0.f}, nr = sc, nm = of;
    f32x4 q0 = sc, q1 = of, q2 = of;
    if (p.has_norm) {
        const float* cf = p.coef + (long long)b * 4 * p.C;
        sc = *reinterpret_cast<const f32x4*>(cf + c4);
        of = *reinterpret_cast<const f32x4*>(cf + p.C + c4);
        nr = *reinterpret_cast<const f32x4*>(cf + 2 * p.C + c4);
        nm = *reinterpret_cast<const f32x4*>(cf + 3 * p.C + c4);
        const float* qq = a.q + (long long)b * 3 * p.C;
        q0 = *reinterpret_cast<const f32x4*>(qq + c4);
        q1 = *reinterpret_cast<const f32x4*>(qq + p.C + c4);
        q2 = *reinterpret_cast<const f32x4*>(qq + 2 * p.C + c4);
    }
    const long long HWo = p.resample == VD_RS_DOWN ? p.HW / 4 : (p.resample == VD_RS_UP ? p.HW * 4 : p.HW);
    const float* dimg = p.dy + (long long)b * HWo * p.lddy;
    const float* ximg = p.has_norm ? p.x + (long long)b * p.HW * p.ldx : nullptr;
    const float* aimg = a.add ? a.add + (long long)b * p.HW * a.ldadd : nullptr;
    float* oimg = a.dx + (long long)b * p.HW * a.lddx;
    const int vtot = p.C >> 2;
    if (p.resample == VD_RS_NONE && p.has_norm) {
        // (the second pass of the two-pass form, 64x64 images) four pixels per trip, all their loads in flight together
        for (int i = r; i < np; i += 4 * rows) {
            f32x4 xv[4], g[4], ad[4], od[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int pix = p0 + min(i + u * rows, np - 1);
                xv[u] = *reinterpret_cast<const f32x4*>(ximg + (long long)pix * p.ldx + c4);
                g[u] = *reinterpret_cast<const f32x4*>(dimg + (long long)pix * p.lddy + c4);
                if (aimg) ad[u] = *reinterpret_cast<const f32x4*>(aimg + (long long)pix * a.ldadd + c4);
                if (a.accumulate_dx) od[u] = *reinterpret_cast<const f32x4*>(oimg + (long long)pix * a.lddx + c4);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (i + u * rows < np) {
                    const int pix = p0 + i + u * rows;
                    const unsigned long long vi = ((unsigned long long)b * p.HW + pix) * vtot + (c4 >> 2);
                    f32x4 d = dz_finish(p, g[u], xv[u], sc, of, vi);
                    d = q0 * d - q1 - (xv[u] * nr + nm) * q2;
                    if (aimg) d += ad[u];
                    if (a.accumulate_dx) d += od[u];
                    *reinterpret_cast<f32x4*>(oimg + (long long)pix * a.lddx + c4) = d;
                }
        }
        return;
    }
    for (int i = r; i < np; i += rows) {
        const int pix = p0 + i;
        f32x4 xv = {0.f, 0.f, 0.f, 0.f};
        if (p.has_norm) xv = *reinterpret_cast<const f32x4*>(ximg + (long long)pix * p.ldx + c4);
        const unsigned long long vi = ((unsigned long long)b * p.HW + pix) * vtot + (c4 >> 2);
        f32x4 d = dz_of(p, dimg, pix, c4, xv, sc, of, vi);
        if (p.has_norm) d = q0 * d - q1 - (xv * nr + nm) * q2;
        if (aimg) d += *reinterpret_cast<const f32x4*>(aimg + (long long)pix * a.ldadd + c4);
        float* o = oimg + (long long)pix * a.lddx + c4;
        if (a.accumulate_dx) d += *reinterpret_cast<const f32x4*>(o);
        *reinterpret_cast<f32x4*>(o) = d;
    }
}


// ---- single-pass GroupNorm backward.  One workgroup owns (image b, a slab of CS channels = whole groups, >= 96 B per
// pixel) and ALL pixels of the image: the forward input and the upstream gradient are read ONCE into registers (NPT
// float4 pairs per thread), the per-channel sums are reduced through LDS, the per-group means follow in the same
// workgroup, and dx is written from the registers.  HBM traffic is (dy + x) in, dx out -- the two-pass form below reads
// dy and x twice -- and the dropout mask (Philox) is regenerated once instead of twice.
struct FusedBwdArgs {
    BwdApplyArgs a; const float* gamma; const float* beta; const float* film; float* dfilm; float* pgb; int G, CS;
    // SPLIT form (round 5): an image slab too large for one workgroup (64x64 images) is shared by `nsplit` SIBLING workgroups, HW / nsplit
    // pixels each; they exchange their per-channel partial sums through xbuf [unit][sibling][2][128] and meet at cnt[unit] (zeroed by the
    // launcher).  units = nimg * C / CS, numbered image-major.
    int nsplit; float* xbuf; unsigned* cnt;
    unsigned* spin_max;          // -DVD_PROBES builds only (vd_gn_set_spin_probe): the longest sibling wait of a launch, in spin iterations
};

// SPLIT: the 64x64 layers (CelebA: 13.2 ms per step through the two-pass form, which reads dy and x twice).  The siblings of a unit are
// 8 logical ids apart (with the round-robin XCD assignment of in-order dispatch: on the same XCD); a set of 8 units x nsplit siblings is
// 8 nsplit CONSECUTIVE logical ids.  The exchange uses agent-scope (sc1) stores / loads / atomics, which are coherent across the XCDs'
// L2s, ordered by completion (s_waitcnt vmcnt(0) in front of the counter increment), NOT by release / acquire fences: on this part an
// agent-scope release writes back the whole L2 of the XCD, in the middle of everybody's dx stream (the cooperative-launch form of
// round 4 lost 5x that way).
// Forward progress of the spin rests on the dispatcher, and says so: workgroups are dispatched in id order, round-robin over the 8 XCDs
// (observed behaviour of this part, MI355X_MICROARCH.md, not a HIP guarantee), so the nsplit siblings of a unit -- ids congruent mod 8 --
// land on ONE XCD, next to each other in its queue, and a sibling set (8 units x nsplit workgroups = 8 nsplit consecutive ids, NOT
// "within 24" as the round-5 comment said: up to 248 ids apart at nsplit = 32) fills every XCD with whole units.  The launcher takes this
// form only when every XCD can hold a unit's siblings twice over (8 nsplit x 2 <= resident workgroups of this kernel on the CUs that are not
// reserved: hipOccupancyMaxActiveBlocksPerMultiprocessor; round-5 advice), so spinners cannot occupy every slot of an XCD while a sibling waits
// in the queue; otherwise it falls through to the two-pass form.  Kernels of other streams only delay siblings, they do not wait for us.
// (Round 6, built and dropped: logical ids taken as TICKETS at workgroup start, to be independent of dispatch order.  It DEADLOCKS: start order
//  scatters a unit's siblings over the XCDs, an XCD fills up with spinners whose siblings have not started, and the in-order dispatcher --
//  whose next workgroup is pinned to that XCD by the round-robin -- stops handing out work to the whole device.  The device-wide atomic
//  itself is fine: tests/probe/xcd_atomic.hip, 4096 unique tickets.)  If a sibling never arrives the spin is bounded and the unit's results
// are NaN: loud, not late.  tests/test_multigpu_path_gpu.py soaks it beside side-stream kernels and 1-rank RCCL collectives.
template <int NPT, int TPB, bool NT = false, bool SPLIT = false>
__global__ __launch_bounds__(TPB) void gn_bwd_fused_kernel(const FusedBwdArgs f) {
    __shared__ float red[TPB][8];
    __shared__ float chs[2][128];        // per-channel S1 (sum dz), S2 (sum dz*xhat) ; later k*S1, k*S2
    __shared__ float gm[2][32];
    // 1024 threads x 8 pixels would need 64 registers of payload + temporaries > the 128 a 16-wave workgroup gets: the
    // x_hat values of pixels 4..7 live in LDS instead (64 KB; scratch spills would go to HBM -- PMC showed +75 % traffic)
    constexpr int NREG = (NPT == 8 && TPB == 1024) ? 4 : NPT;
    __shared__ f32x4 xl[NPT - NREG > 0 ? NPT - NREG : 1][NPT - NREG > 0 ? TPB : 1];
    const ReduceArgs& p = f.a.r;
    const int CS = f.CS, C = p.C, tid = threadIdx.x;
    int b = blockIdx.y, slab = blockIdx.x, sib = 0, unit = 0;
    if (SPLIT) {
        const int lid = blockIdx.x, q = lid >> 3;
        sib = q % f.nsplit;
        unit = (q / f.nsplit) * 8 + (lid & 7);
        const int nslab = C / CS;
        if (unit >= p.nimg * nslab) return;                  // (the unit count is padded to a multiple of 8: whole sibling sets leave)
        b = unit / nslab; slab = unit - b * nslab;
    }
    const int c0 = slab * CS;
    const int vecs = CS >> 2, rows = TPB / vecs;
    const int r = tid / vecs, v = tid - r * vecs, c4 = c0 + 4 * v;
    const bool act = r < rows;
    const int HW = SPLIT ? (int)p.HW / f.nsplit : (int)p.HW;     // pixels THIS workgroup holds, starting at pixel `pbeg` of the image
    const int pbeg = SPLIT ? sib * HW : 0;
    const float* cf = p.coef + (long long)b * 4 * C;
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, of = {0.f, 0.f, 0.f, 0.f}, nr = sc, nm = of;
    if (act) {
        sc = *reinterpret_cast<const f32x4*>(cf + c4);
        of = *reinterpret_cast<const f32x4*>(cf + C + c4);
        nr = *reinterpret_cast<const f32x4*>(cf + 2 * C + c4);
        nm = *reinterpret_cast<const f32x4*>(cf + 3 * C + c4);
    }
    const long long HWo = p.resample == VD_RS_DOWN ? p.HW / 4 : (p.resample == VD_RS_UP ? p.HW * 4 : p.HW);
    const float* dimg = p.dy + (long long)b * HWo * p.lddy;
    const float* ximg = p.x + (long long)b * p.HW * p.ldx;
    const int vtot = C >> 2;
    f32x4 xv[NREG], dz[NPT];
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
    if (p.resample == VD_RS_NONE) {
        // the common case (every norm but the four resampling blocks): ALL 2 NPT loads of the thread are issued before the first
        // value is used (the per-pixel form waited on every pair: the resample kind is a run-time value, so the compiler kept each
        // pixel's loads behind its branches).  Measured: no change (107 us at 32x32x256 either way) -- sixteen waves per CU already
        // covered the latency; what holds this kernel at 3.8 TB/s is the 128-byte segment per pixel row of a 32-channel slab (64-byte
        // slabs: -25 %), and starting half of the first round late so that load and store phases of different CUs interleave
        // changes nothing either (-2 % ... +10 %).  Rows beyond the image are clamped (loaded twice, masked below).
        f32x4 xin[NPT];
        if (act) {
#pragma unroll
            for (int i = 0; i < NPT; ++i) {
                const int pc = pbeg + min(r + i * rows, HW - 1);
                xin[i] = ld_stream<NT>(ximg + (long long)pc * p.ldx + c4);
                dz[i] = ld_stream<NT>(dimg + (long long)pc * p.lddy + c4);
            }
        }
#pragma unroll
        for (int i = 0; i < NPT; ++i) {
            const int pix = r + i * rows;
            f32x4 xh = a0 * 0.f;
            if (act && pix < HW) {
                const unsigned long long vi = ((unsigned long long)b * p.HW + pbeg + pix) * vtot + (c4 >> 2);
                dz[i] = dz_finish(p, dz[i], xin[i], sc, of, vi);
                xh = xin[i] * nr + nm;                      // keep x_hat: all the second half needs
                a0 += dz[i];
                a1 += dz[i] * xh;
            } else dz[i] = xh;
            if (i < NREG) xv[i] = xh; else xl[i - NREG][tid] = xh;
        }
    } else {
#pragma unroll
        for (int i = 0; i < NPT; ++i) {
            const int pix = r + i * rows;
            f32x4 xh = a0 * 0.f;
            dz[i] = xh;
            if (act && pix < HW) {
                const f32x4 xin = *reinterpret_cast<const f32x4*>(ximg + (long long)pix * p.ldx + c4);
                const unsigned long long vi = ((unsigned long long)b * p.HW + pix) * vtot + (c4 >> 2);
                dz[i] = dz_of(p, dimg, pix, c4, xin, sc, of, vi);
                xh = xin * nr + nm;
                a0 += dz[i];
                a1 += dz[i] * xh;
            }
            if (i < NREG) xv[i] = xh; else xl[i - NREG][tid] = xh;
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) { red[tid][j] = act ? a0[j] : 0.f; red[tid][4 + j] = act ? a1[j] : 0.f; }
    __syncthreads();
    // two-level fixed-order reduction over the `rows` pixel lanes of every (channel vector, component)
    {
        const int comps = vecs * 8;                       // (v, j) pairs
        const int fan = TPB / comps > 16 ? 16 : (TPB / comps < 1 ? 1 : TPB / comps);   // partial sums per pair
        const int per = (rows + fan - 1) / fan;
        float part = 0.f;
        const int pair = tid / fan, k = tid - pair * fan;
        if (pair < comps) {
            const int vv = pair >> 3, jj = pair & 7;
            for (int q = k * per; q < min(rows, (k + 1) * per); ++q) part += red[q * vecs + vv][jj];
        }
        __syncthreads();
        if (pair < comps) red[tid][0] = part;
        __syncthreads();
        if (tid < comps) {
            float s = 0.f;
            for (int k2 = 0; k2 < fan; ++k2) s += red[tid * fan + k2][0];
            const int vv = tid >> 3, jj = tid & 7;
            chs[jj >> 2][vv * 4 + (jj & 3)] = s;
        }
        __syncthreads();
    }
    if (SPLIT) {
        // this workgroup's sums cover HW / nsplit pixels: leave them for the siblings, wait for theirs, add all nsplit in sibling order
        // (every sibling forms the same sums in the same order -- bitwise the same group terms)
        float* mine = f.xbuf + ((long long)unit * f.nsplit + sib) * 256;
        if (tid < 2 * CS) __hip_atomic_store(mine + (tid / CS) * 128 + (tid % CS), chs[tid / CS][tid % CS], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the stores have reached the coherence point before the counter moves
        __syncthreads();
        if (tid == 0) {
            __hip_atomic_fetch_add(f.cnt + unit, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // (bounded: a sibling that never arrives -- ruled out by in-order dispatch + the launcher's residency check -- must not hang the
            //  device: ~1 s, then the poisoned counter makes every later reader of this unit leave too and the results are visibly wrong, not late)
            unsigned spin = 0;
            for (; __hip_atomic_load(f.cnt + unit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)f.nsplit; ++spin) {
                __builtin_amdgcn_s_sleep(8);
                if (spin > (1u << 22)) { __hip_atomic_fetch_add(f.cnt + unit, 1u << 20, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            }
            if (VD_PROBE_BUILD && f.spin_max) atomicMax(f.spin_max, spin);
        }
        __syncthreads();
        if (tid < 2 * CS) {
            const float* all = f.xbuf + (long long)unit * f.nsplit * 256 + (tid / CS) * 128 + (tid % CS);
            float t = 0.f;
            for (int k = 0; k < f.nsplit; ++k) t += __hip_atomic_load(all + k * 256, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // a poisoned counter (a sibling never arrived within the spin bound) must be LOUD: NaN sums make this unit's dx, dfilm and
            // parameter-gradient terms NaN, which the loss / the clip norm of the next step shows at once
            if (__hip_atomic_load(f.cnt + unit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (1u << 20)) t = __builtin_nanf("");
            chs[tid / CS][tid % CS] = t;
        }
        __syncthreads();
    }
    const int cg = C / f.G;
    if (tid < CS) {
        const int c = c0 + tid;
        const float s1 = chs[0][tid], s2 = chs[1][tid];
        float fs = 1.f;
        if (f.film) {
            fs = 1.f + f.film[(long long)b * 2 * C + C + c];
            if (sib == 0) {
                f.dfilm[(long long)b * 2 * C + c] = s1;
                f.dfilm[(long long)b * 2 * C + C + c] = f.gamma[c] * s2 + f.beta[c] * s1;
            }
        }
        const float k = f.gamma[c] * fs;
        if (sib == 0) {
            f.pgb[((long long)b * 2) * C + c] = fs * s2;
            f.pgb[((long long)b * 2 + 1) * C + c] = fs * s1;
        }
        chs[0][tid] = k * s1; chs[1][tid] = k * s2;
    }
    __syncthreads();
    if (tid < CS / cg) {
        float m1 = 0.f, m2 = 0.f;
        for (int c = tid * cg; c < (tid + 1) * cg; ++c) { m1 += chs[0][c]; m2 += chs[1][c]; }
        const float inv = 1.f / ((float)cg * (float)p.HW);
        gm[0][tid] = m1 * inv; gm[1][tid] = m2 * inv;
    }
    __syncthreads();
    if (!act) return;
    f32x4 q0, q1, q2;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int cl = 4 * v + j, c = c0 + cl, g = cl / cg;
        const float fs = f.film ? 1.f + f.film[(long long)b * 2 * C + C + c] : 1.f;
        q0[j] = nr[j] * f.gamma[c] * fs; q1[j] = nr[j] * gm[0][g]; q2[j] = nr[j] * gm[1][g];
    }
    const float* aimg = f.a.add ? f.a.add + ((long long)b * p.HW + pbeg) * f.a.ldadd : nullptr;
    float* oimg = f.a.dx + ((long long)b * p.HW + pbeg) * f.a.lddx;
    // dx = q0 dz - q1 - x_hat q2 (in place in dz), then the skip-path gradient / the running dx, every load of a kind in flight at once
#pragma unroll
    for (int i = 0; i < NPT; ++i) {
        const f32x4 xh = i < NREG ? xv[i < NREG ? i : 0] : xl[i < NREG ? 0 : i - NREG][tid];
        dz[i] = q0 * dz[i] - q1 - xh * q2;
    }
    if (aimg) {
        f32x4 ad[NPT];
#pragma unroll
        for (int i = 0; i < NPT; ++i) ad[i] = *reinterpret_cast<const f32x4*>(aimg + (long long)min(r + i * rows, HW - 1) * f.a.ldadd + c4);
#pragma unroll
        for (int i = 0; i < NPT; ++i) dz[i] += ad[i];
    }
    if (f.a.accumulate_dx) {
        f32x4 od[NPT];
#pragma unroll
        for (int i = 0; i < NPT; ++i) od[i] = *reinterpret_cast<const f32x4*>(oimg + (long long)min(r + i * rows, HW - 1) * f.a.lddx + c4);
#pragma unroll
        for (int i = 0; i < NPT; ++i) dz[i] += od[i];
    }
#pragma unroll
    for (int i = 0; i < NPT; ++i) {
        const int pix = r + i * rows;
        if (pix < HW) st_stream<NT>(oimg + (long long)pix * f.a.lddx + c4, dz[i]);
    }
}

#ifdef VD_PROBES
unsigned* g_gn_spin_probe = nullptr;     // probe library only (vd_gn_set_spin_probe)
#else
constexpr unsigned* g_gn_spin_probe = nullptr;
#endif
thread_local int g_last_gn_bwd = 0;   // NPT * 10000 + TPB (+ 1000000: non-temporal instantiation) of the last fused launch, -1 = two-pass form, 0 = no norm (plain resample)

template <int TPB>
void launch_fused_bwd(int npt, dim3 grid, hipStream_t st, const FusedBwdArgs& f) {
    const bool nt = f.CS % 32 == 0 && npt > 2;   // whole 128-byte lines per pixel row: non-temporal loads / stores (see ld_stream)
    g_last_gn_bwd = (npt <= 1 ? 1 : (npt <= 2 ? 2 : (npt <= 4 ? 4 : 8))) * 10000 + TPB + (nt ? 1000000 : 0);
    if (npt <= 1) hipLaunchKernelGGL((gn_bwd_fused_kernel<1, TPB>), grid, dim3(TPB), 0, st, f);
    else if (npt <= 2) hipLaunchKernelGGL((gn_bwd_fused_kernel<2, TPB>), grid, dim3(TPB), 0, st, f);
    else if (npt <= 4 && nt) hipLaunchKernelGGL((gn_bwd_fused_kernel<4, TPB, true>), grid, dim3(TPB), 0, st, f);
    else if (npt <= 4) hipLaunchKernelGGL((gn_bwd_fused_kernel<4, TPB>), grid, dim3(TPB), 0, st, f);
    else if (nt) hipLaunchKernelGGL((gn_bwd_fused_kernel<8, TPB, true>), grid, dim3(TPB), 0, st, f);
    else hipLaunchKernelGGL((gn_bwd_fused_kernel<8, TPB>), grid, dim3(TPB), 0, st, f);
}

// channel slab of the fused backward: whole groups, a multiple of 4 channels, at least 24 channels (96 B per pixel row)
inline int fused_slab(int C, int G, int min_cs = 24) {      // measured: 16-channel slabs (64 B rows) run 25 % slower, 8-channel ones 2x slower
    const int cg = C / G;
    for (int k = 1; k * cg <= 128 && k <= G; ++k) {
        const int cs = k * cg;
        if (cs % 4 == 0 && cs >= min_cs && C % cs == 0) return cs;
    }
    return 0;
}

__global__ __launch_bounds__(256) void colsum_finalize_kernel(const float* part, int chunks, int N, float* out, int accumulate) {
    __shared__ float sh[16][17];
    const int cx = threadIdx.x & 15, ly = threadIdx.x >> 4;
    const int n = blockIdx.x * 16 + cx;
    float s = 0.f;
    if (n < N)
        for (int ch = ly; ch < chunks; ch += 16) s += part[(long long)ch * 2 * N + n];
    sh[ly][cx] = s;
    __syncthreads();
    if (ly == 0 && n < N) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) t += sh[q][cx];
        out[n] = accumulate ? out[n] + t : t;
    }
}

inline int pick_cb(int C) {           // channels handled by one reduction block (<= 1024, multiple of 4)
    int split = 1;
    while (split < C && (C % split != 0 || C / split > 1024 || (C / split) % 4 != 0)) ++split;
    return C / split;
}

}  // namespace

#ifdef VD_PROBES
extern "C" int vd_gn_set_spin_probe(unsigned* buf) { g_gn_spin_probe = buf; return 0; }
#endif
extern "C" size_t vd_gn_ws_bytes(int32_t nimg, int32_t HW, int32_t C) {
    const long long chunks = (HW + PPC - 1) / PPC;
    // partials + q table + per-image dgamma/dbeta terms
    return (size_t)(nimg * chunks * 2LL * C + (long long)nimg * 3 * C + (long long)nimg * 2 * C) * sizeof(float);
}

extern "C" int vd_gn_stats(const float* x, int64_t ldx, int32_t nimg, int32_t HW, int32_t C, int32_t G, float eps,
                           float* stats, float* ws, size_t ws_bytes, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    VD_REQUIRE(C % 4 == 0 && C % G == 0 && ldx % 4 == 0, "vd_gn_stats: C=%d G=%d ldx=%lld unsupported", C, G, (long long)ldx);
    VD_REQUIRE(ws_bytes >= vd_gn_ws_bytes(nimg, HW, C), "vd_gn_stats: workspace too small");
    const int Cb = pick_cb(C);
    VD_REQUIRE(C % Cb == 0 && Cb % 4 == 0 && Cb <= 1024, "vd_gn_stats: cannot split C=%d", C);
    ReduceArgs p = {};
    p.x = x; p.ldx = ldx; p.HW = HW; p.C = C; p.nimg = nimg; p.chunks = (HW + PPC - 1) / PPC; p.part = ws;
    hipLaunchKernelGGL(chan_reduce_kernel<0>, dim3(p.chunks, nimg, C / Cb), dim3(256), 0, st, p, Cb);
    VD_LAUNCH_CHECK("chan_reduce_kernel<0>");
    hipLaunchKernelGGL(gn_stats_finalize_kernel, dim3((nimg * G + 127) / 128), dim3(128), 0, st, ws, nimg, p.chunks, C, G,
                       (long long)HW, eps, stats);
    VD_LAUNCH_CHECK("gn_stats_finalize_kernel");
    return 0;
}

extern "C" int vd_gn_stats_from_partials(const float* part1, int32_t C1, int32_t chunks1, const float* part2, int32_t C2,
                                         int32_t chunks2, int32_t nimg, int32_t HW, int32_t G, float eps, float* stats,
                                         void* stream) {
    VD_REQUIRE(part1 && C1 > 0 && chunks1 > 0 && (C1 + C2) % G == 0, "vd_gn_stats_from_partials: bad arguments");
    VD_REQUIRE(C2 == 0 || (part2 && chunks2 > 0), "vd_gn_stats_from_partials: second source incomplete");
    hipLaunchKernelGGL(gn_stats_from_partials_kernel, dim3((nimg * G + 3) / 4), dim3(256), 0, (hipStream_t)stream, part1, C1,
                       chunks1, part2, C2, chunks2, nimg, G, (long long)HW, eps, stats, (const float*)nullptr, (const float*)nullptr,
                       (const float*)nullptr, (float*)nullptr);
    VD_LAUNCH_CHECK("gn_stats_from_partials_kernel");
    return 0;
}

extern "C" int vd_gn_coef_from_partials(const float* part1, int32_t C1, int32_t chunks1, const float* part2, int32_t C2,
                                        int32_t chunks2, int32_t nimg, int32_t HW, int32_t G, float eps, const float* gamma,
                                        const float* beta, const float* film, float* coef, void* stream) {
    VD_REQUIRE(part1 && C1 > 0 && chunks1 > 0 && (C1 + C2) % G == 0, "vd_gn_coef_from_partials: bad arguments");
    VD_REQUIRE(C2 == 0 || (part2 && chunks2 > 0), "vd_gn_coef_from_partials: second source incomplete");
    VD_REQUIRE(gamma && beta && coef, "vd_gn_coef_from_partials: missing norm operands");
    hipLaunchKernelGGL(gn_stats_from_partials_kernel, dim3((nimg * G + 3) / 4), dim3(256), 0, (hipStream_t)stream, part1, C1,
                       chunks1, part2, C2, chunks2, nimg, G, (long long)HW, eps, (float*)nullptr, gamma, beta, film, coef);
    VD_LAUNCH_CHECK("gn_stats_from_partials_kernel");
    return 0;
}

extern "C" int vd_gn_apply(const float* x, int64_t ldx, const float* stats, const float* gamma, const float* beta,
                           const float* film, int32_t act, float p_drop, uint64_t seed, int32_t resample, float* y,
                           int64_t ldy, int32_t nimg, int32_t H, int32_t W, int32_t C, int32_t G, float* coef,
                           void* stream) {
    hipStream_t st = (hipStream_t)stream;
    VD_REQUIRE(C % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0, "vd_gn_apply: C/ld must be multiples of 4");
    VD_REQUIRE(resample != VD_RS_DOWN || (H % 2 == 0 && W % 2 == 0), "vd_gn_apply: odd size cannot be average-pooled");
    const int has_norm = gamma != nullptr;
    if (has_norm && !stats) {
        VD_REQUIRE(coef != nullptr, "vd_gn_apply: neither statistics nor a coefficient table given");   // vd_gn_coef_from_partials ran
    } else if (has_norm) {
        VD_REQUIRE(stats && beta && coef && C % G == 0, "vd_gn_apply: missing norm operands");
        hipLaunchKernelGGL(gn_coef_kernel, dim3((nimg * C + 255) / 256), dim3(256), 0, st, stats, gamma, beta, film, nimg, C,
                           G, coef);
        VD_LAUNCH_CHECK("gn_coef_kernel");
    }
    ApplyArgs p = {x, ldx, coef, has_norm, act, p_drop, seed, resample, y, ldy, nimg, H, W, C};
    const long long Ho = resample == VD_RS_DOWN ? H / 2 : (resample == VD_RS_UP ? H * 2 : H);
    const long long Wo = resample == VD_RS_DOWN ? W / 2 : (resample == VD_RS_UP ? W * 2 : W);
    const int Cba = pick_cb(C);
    VD_REQUIRE(C % Cba == 0 && Cba % 4 == 0 && Cba <= 1024 && nimg <= 65535, "vd_gn_apply: cannot split C=%d / nimg=%d", C, nimg);
    hipLaunchKernelGGL(gn_apply_kernel<false>, dim3((unsigned)((Ho * Wo + APIX - 1) / APIX), nimg, C / Cba), dim3(256), 0, st, p, Cba, PartArgs{});
    VD_LAUNCH_CHECK("gn_apply_kernel");
    return 0;
}

/* vd_gn_coef_from_partials + vd_gn_apply in ONE launch: y = resample(dropout(act(FiLM(GroupNorm(x))))) with the statistics taken from the
 * partial sums of up to two channel-concatenated producers (layout of vd_gemm `stats` / vd_conv3x3 `stats_part`), and the [nimg][4][C]
 * coefficient table written for the backward pass (coef may be NULL at inference).  Reference: nn.GroupNorm unet.py:28-30 + its call sites. */
extern "C" int vd_gn_apply_from_partials(const float* x, int64_t ldx, const float* part1, int32_t C1, int32_t chunks1, const float* part2,
                                         int32_t C2, int32_t chunks2, const float* gamma, const float* beta, const float* film, int32_t act,
                                         float p_drop, uint64_t seed, int32_t resample, float* y, int64_t ldy, int32_t nimg, int32_t H,
                                         int32_t W, int32_t C, int32_t G, float eps, float* coef, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    VD_REQUIRE(x && y && part1 && gamma && beta, "vd_gn_apply_from_partials: null operand");
    VD_REQUIRE(C % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && C % G == 0 && C1 + C2 == C && C1 > 0 && chunks1 > 0,
               "vd_gn_apply_from_partials: bad channel split C=%d C1=%d C2=%d G=%d", C, C1, C2, G);
    VD_REQUIRE(C2 == 0 || (part2 && chunks2 > 0), "vd_gn_apply_from_partials: second source incomplete");
    VD_REQUIRE(resample != VD_RS_DOWN || (H % 2 == 0 && W % 2 == 0), "vd_gn_apply_from_partials: odd size cannot be average-pooled");
    VD_REQUIRE(vd_aligned16(x) && vd_aligned16(y) && (!coef || vd_aligned16(coef)), "vd_gn_apply_from_partials: operands must be 16-byte aligned");
    ApplyArgs p = {x, ldx, coef, 1, act, p_drop, seed, resample, y, ldy, nimg, H, W, C};
    PartArgs q = {part1, C1, chunks1, part2, C2, chunks2, G, eps, gamma, beta, film, coef};
    const long long Ho = resample == VD_RS_DOWN ? H / 2 : (resample == VD_RS_UP ? H * 2 : H);
    const long long Wo = resample == VD_RS_DOWN ? W / 2 : (resample == VD_RS_UP ? W * 2 : W);
    // channel range of a workgroup: whole groups, a multiple of 4 channels, at most 1024
    const int cg = C / G;
    int Cba = pick_cb(C);
    if (Cba % cg) {
        Cba = 0;
        for (int k = G; k >= 1; --k) if (G % k == 0 && (k * cg) % 4 == 0 && k * cg <= 1024) { Cba = k * cg; break; }
    }
    VD_REQUIRE(Cba > 0 && C % Cba == 0 && Cba % 4 == 0 && Cba % cg == 0 && Cba <= 1024 && Cba / cg <= 256 && nimg <= 65535,
               "vd_gn_apply_from_partials: cannot split C=%d into whole-group channel ranges", C);
    hipLaunchKernelGGL(gn_apply_kernel<true>, dim3((unsigned)((Ho * Wo + APIX - 1) / APIX), nimg, C / Cba), dim3(256), 0, st, p, Cba, q);
    VD_LAUNCH_CHECK("gn_apply_kernel<from partials>");
    return 0;
}

static int gn_apply_bwd_impl(const float* dy, int64_t lddy, const float* x, int64_t ldx, const float* coef,
                             const float* gamma, const float* beta, const float* film, int32_t act, float p_drop,
                             uint64_t seed, int32_t resample, const float* add, int64_t ldadd, float* dx, int64_t lddx,
                             int32_t accumulate_dx, float* dfilm, float* dgamma, float* dbeta, int32_t accumulate_params,
                             int32_t nimg, int32_t H, int32_t W, int32_t C, int32_t G, float* ws, size_t ws_bytes,
                             float* pgb_keep, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    VD_REQUIRE(C % 4 == 0 && lddy % 4 == 0 && lddx % 4 == 0, "vd_gn_apply_bwd: C/ld must be multiples of 4");
    const int has_norm = gamma != nullptr;
    const long long HW = (long long)H * W;
    BwdApplyArgs a = {};
    ReduceArgs& p = a.r;
    p.x = x; p.ldx = ldx; p.dy = dy; p.lddy = lddy; p.coef = coef; p.act = act; p.p_drop = p_drop; p.seed = seed;
    p.resample = resample; p.has_norm = has_norm; p.H = H; p.W = W; p.HW = HW; p.C = C; p.nimg = nimg;
    p.chunks = (int)((HW + PPC - 1) / PPC);
    a.add = add; a.ldadd = ldadd; a.dx = dx; a.lddx = lddx; a.accumulate_dx = accumulate_dx;
    g_last_gn_bwd = has_norm ? -1 : 0;
    if (has_norm) {
        VD_REQUIRE(x && coef && beta && ((dgamma && dbeta) || pgb_keep) && C % G == 0 && ldx % 4 == 0, "vd_gn_apply_bwd: missing norm operands");
        VD_REQUIRE(!film || dfilm, "vd_gn_apply_bwd: film given without dfilm");
        VD_REQUIRE(ws && ws_bytes >= vd_gn_ws_bytes(nimg, (int)HW, C), "vd_gn_apply_bwd: workspace too small");
        const int Cb = pick_cb(C);
        VD_REQUIRE(C % Cb == 0 && Cb % 4 == 0 && Cb <= 1024, "vd_gn_apply_bwd: cannot split C=%d", C);
        float* part = ws;
        float* q = part + (long long)nimg * p.chunks * 2 * C;
        // per-image dgamma / dbeta terms: summed over the images right here, or -- pgb_keep -- left in the caller's buffer for ONE
        // vd_gn_param_sums_batched launch over all norms of the backward pass
        float* pgb = pgb_keep ? pgb_keep : q + (long long)nimg * 3 * C;
        p.part = part;
        // single-pass form whenever one workgroup can hold an image's slab in registers (<= 8 float4 pairs per thread)
        static const bool two_pass = getenv("VD_GN_TWO_PASS") != nullptr;        // A/B switch for profiling
        int CS = fused_slab(C, G);
        // slabs whose pixel rows are not whole 128-byte lines (24 channels of C = 192 / 384 / 768, 48 of 1536: 96 / 192-byte rows) stream
        // poorly; where whole groups also form a 96-channel slab (three full lines per row, non-temporal accesses) that slab is taken
        // instead, shared by sibling workgroups where it no longer fits one (SPLIT form).  VD_GN_WIDE=0: the narrow slabs (A/B switch).
        // Same-box, B = 128 (profiles/r05_gn_split.txt, r05_gn_split2.txt, r05_gn_wide.txt): C = 384 @32x32 220 -> 184 us, @64x64 792 (two-pass) -> 708; C = 192 @64x64
        // 403 (two-pass) vs 406: kept two-pass; 32-channel slabs (C = 256 / 512: already whole lines) lose 6-12 % when widened: untouched.
        static const bool wide = !(getenv("VD_GN_WIDE") && atoi(getenv("VD_GN_WIDE")) == 0);
        if (wide && CS > 0 && CS % 32 != 0 && C >= 384) {
            const int cg = C / G;
            for (int k = 1; k * cg <= 128 && k <= G; ++k)
                if ((k * cg) % 32 == 0 && C % (k * cg) == 0) { CS = k * cg; break; }
        }
        if (!two_pass && CS > 0 && nimg <= 65535) {
            // 256-thread workgroups (8+ per CU: their load / reduce / store phases overlap) whenever a slab fits 8 float4 pairs per
            // thread, i.e. up to 2048 pairs (16x16 images at 32-channel slabs); 1024 threads only for the larger slabs
            // (VD_GN_TPB=1024 restores the round-1 choice for A/B runs)
            static const bool big = getenv("VD_GN_TPB") && atoi(getenv("VD_GN_TPB")) == 1024;
            const int TPB = (HW * (CS / 4) > 2048 || (big && HW * (CS / 4) >= 1024)) ? 1024 : 256;
            const int rows = TPB / (CS / 4);
            const int npt = (int)((HW + rows - 1) / rows);
            // larger slabs: nsplit sibling workgroups per (image, slab), HW / nsplit pixels each (SPLIT form of the kernel; VD_GN_SPLIT=0:
            // the two-pass form below, A/B switch).  Not for the resampling norms (their pixel gather spans the image), and only for
            // whole-line slabs: with 96-byte rows the two-pass form (whole pixel rows in both passes) is the faster one (64x64, C = 192:
            // 403 us against 490).
            static const bool no_split = getenv("VD_GN_SPLIT") && atoi(getenv("VD_GN_SPLIT")) == 0;
            if (npt > 8 && !no_split && resample == VD_RS_NONE && CS % 32 == 0) {
                int nsplit = 0;
                for (int k = 2; k <= 32; k *= 2)
                    if (HW % k == 0 && (HW / k + (1024 / (CS / 4)) - 1) / (1024 / (CS / 4)) <= 8) { nsplit = k; break; }
                const long long units = (long long)nimg * (C / CS), upad = (units + 7) / 8 * 8;
                const size_t cnt_bytes = ((size_t)upad * sizeof(unsigned) + 255) / 256 * 256;
                // residency check (round-5 advice): a sibling set (8 units x nsplit workgroups, one unit's siblings per XCD) must fit twice over
                static const int split_occ = [] {
                    int occ = 0;
                    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, gn_bwd_fused_kernel<8, 1024, true, true>, 1024, 0) != hipSuccess) occ = 0;
                    return occ;
                }();
                const long long resident = (long long)split_occ * vd_persistent_cus();
                if (8LL * nsplit * 2 > resident) nsplit = 0;
                const size_t need = cnt_bytes + (size_t)units * nsplit * 256 * sizeof(float);
                const size_t avail = ((size_t)nimg * p.chunks * 2 * C + (size_t)nimg * 3 * C) * sizeof(float);     // (in front of the pgb region)
                if (nsplit && need <= avail && upad * nsplit < (1LL << 31)) {
                    FusedBwdArgs f = {a, gamma, beta, film, dfilm, pgb, G, CS, nsplit, reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + cnt_bytes),
                                      reinterpret_cast<unsigned*>(ws), g_gn_spin_probe};
                    VD_REQUIRE(hipMemsetAsync(ws, 0, cnt_bytes, st) == hipSuccess, "vd_gn_apply_bwd: hipMemsetAsync of the sibling counters failed");
                    const dim3 grid((unsigned)(upad * nsplit));
                    g_last_gn_bwd = 8 * 10000 + 1024 + 1000000 + 100000000 * nsplit;
                    hipLaunchKernelGGL((gn_bwd_fused_kernel<8, 1024, true, true>), grid, dim3(1024), 0, st, f);
                    VD_LAUNCH_CHECK("gn_bwd_fused_kernel<split>");
                    if (!pgb_keep) {
                        hipLaunchKernelGGL(sum_over_images_kernel, dim3((C + 15) / 16), dim3(256), 0, st, pgb, nimg, C, dgamma, dbeta,
                                           accumulate_params);
                        VD_LAUNCH_CHECK("sum_over_images_kernel");
                    }
                    return 0;
                }
            }
            if (npt <= 8) {
                FusedBwdArgs f = {a, gamma, beta, film, dfilm, pgb, G, CS, 1, nullptr, nullptr, nullptr};
                dim3 grid(C / CS, nimg);
                if (TPB == 1024) launch_fused_bwd<1024>(npt, grid, st, f); else launch_fused_bwd<256>(npt, grid, st, f);
                VD_LAUNCH_CHECK("gn_bwd_fused_kernel");
                if (!pgb_keep) {
                    hipLaunchKernelGGL(sum_over_images_kernel, dim3((C + 15) / 16), dim3(256), 0, st, pgb, nimg, C, dgamma, dbeta,
                                       accumulate_params);
                    VD_LAUNCH_CHECK("sum_over_images_kernel");
                }
                return 0;
            }
        }
        hipLaunchKernelGGL(chan_reduce_kernel<1>, dim3(p.chunks, nimg, C / Cb), dim3(256), 0, st, p, Cb);
        VD_LAUNCH_CHECK("chan_reduce_kernel<1>");
        hipLaunchKernelGGL(gn_bwd_finalize_kernel, dim3(nimg), dim3(256), (2 * C + 2 * G) * sizeof(float), st, part, p.chunks,
                           coef, gamma, beta, film, C, G, HW, q, dfilm, pgb);
        VD_LAUNCH_CHECK("gn_bwd_finalize_kernel");
        if (!pgb_keep) {
            hipLaunchKernelGGL(sum_over_images_kernel, dim3((C + 15) / 16), dim3(256), 0, st, pgb, nimg, C, dgamma, dbeta,
                               accumulate_params);
            VD_LAUNCH_CHECK("sum_over_images_kernel");
        }
        a.q = q;
    }
    const int Cba = pick_cb(C);
    VD_REQUIRE(C % Cba == 0 && Cba % 4 == 0 && Cba <= 1024 && nimg <= 65535, "vd_gn_apply_bwd: cannot split C=%d / nimg=%d", C, nimg);
    hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3((unsigned)((HW + APIX - 1) / APIX), nimg, C / Cba), dim3(256), 0, st, a, Cba);
    VD_LAUNCH_CHECK("gn_bwd_apply_kernel");
    return 0;
}

extern "C" int vd_gn_apply_bwd(const float* dy, int64_t lddy, const float* x, int64_t ldx, const float* coef,
                               const float* gamma, const float* beta, const float* film, int32_t act, float p_drop,
                               uint64_t seed, int32_t resample, const float* add, int64_t ldadd, float* dx, int64_t lddx,
                               int32_t accumulate_dx, float* dfilm, float* dgamma, float* dbeta, int32_t accumulate_params,
                               int32_t nimg, int32_t H, int32_t W, int32_t C, int32_t G, float* ws, size_t ws_bytes,
                               void* stream) {
    return gn_apply_bwd_impl(dy, lddy, x, ldx, coef, gamma, beta, film, act, p_drop, seed, resample, add, ldadd, dx, lddx, accumulate_dx,
                             dfilm, dgamma, dbeta, accumulate_params, nimg, H, W, C, G, ws, ws_bytes, nullptr, stream);
}

extern "C" int vd_gn_apply_bwd_keep(const float* dy, int64_t lddy, const float* x, int64_t ldx, const float* coef,
                                    const float* gamma, const float* beta, const float* film, int32_t act, float p_drop,
                                    uint64_t seed, int32_t resample, const float* add, int64_t ldadd, float* dx, int64_t lddx,
                                    int32_t accumulate_dx, float* dfilm, float* pgb_keep, int32_t nimg, int32_t H, int32_t W,
                                    int32_t C, int32_t G, float* ws, size_t ws_bytes, void* stream) {
    VD_REQUIRE(gamma && pgb_keep, "vd_gn_apply_bwd_keep: needs a norm and the [nimg][2][C] buffer for its per-image dgamma / dbeta terms");
    return gn_apply_bwd_impl(dy, lddy, x, ldx, coef, gamma, beta, film, act, p_drop, seed, resample, add, ldadd, dx, lddx, accumulate_dx,
                             dfilm, nullptr, nullptr, 0, nimg, H, W, C, G, ws, ws_bytes, pgb_keep, stream);
}

extern "C" int vd_gn_param_sums_batched(const int64_t* items_dev, int32_t n, int64_t total_blocks, void* stream) {
    VD_REQUIRE(items_dev && n > 0 && total_blocks > 0 && total_blocks < (1LL << 31), "vd_gn_param_sums_batched: bad table");
    hipLaunchKernelGGL(sum_over_images_batched_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const long long*>(items_dev), n);
    VD_LAUNCH_CHECK("sum_over_images_batched_kernel");
    return 0;
}

extern "C" int vd_gn_bwd_last_kernel(void) { return g_last_gn_bwd; }

extern "C" size_t vd_colsum_ws_bytes(int64_t M, int32_t N) {
    return (size_t)(((M + PPC - 1) / PPC) * 2 * N) * sizeof(float);
}

extern "C" int vd_colsum(const float* x, int64_t ldx, int64_t M, int32_t N, float* out, int32_t accumulate, float* ws,
                         size_t ws_bytes, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    VD_REQUIRE(N % 4 == 0 && ldx % 4 == 0, "vd_colsum: N/ld must be multiples of 4 (N=%d)", N);
    VD_REQUIRE(ws && ws_bytes >= vd_colsum_ws_bytes(M, N), "vd_colsum: workspace too small");
    const int Cb = pick_cb(N);
    VD_REQUIRE(N % Cb == 0 && Cb % 4 == 0 && Cb <= 1024, "vd_colsum: cannot split N=%d", N);
    ReduceArgs p = {};
    p.x = x; p.ldx = ldx; p.HW = M; p.C = N; p.nimg = 1; p.chunks = (int)((M + PPC - 1) / PPC); p.part = ws;
    VD_REQUIRE(p.chunks <= 65535 * 32, "vd_colsum: too many rows");
    hipLaunchKernelGGL(chan_reduce_kernel<2>, dim3(p.chunks, 1, N / Cb), dim3(256), 0, st, p, Cb);
    VD_LAUNCH_CHECK("chan_reduce_kernel<2>");
    hipLaunchKernelGGL(colsum_finalize_kernel, dim3((N + 15) / 16), dim3(256), 0, st, ws, p.chunks, N, out, accumulate);
    VD_LAUNCH_CHECK("colsum_finalize_kernel");
    return 0;
}
