// diffusion.hip -- fused elementwise kernels of the diffusion process (HBM / latency bound):
//   q_sample                         reference v_diffusion/diffusion.py:242-245
//   train_loss (mse) forward/backward  diffusion.py:466-490,520-541 ; flat_mean functions.py:102-104
//   one reverse step (p_mean_var + CFG + noise)  diffusion.py:317-392
//   variational-bound terms (KL / discretised decoder NLL, loss_type "kl") forward/backward  diffusion.py:446-464,497-515
// All images here are NCHW, the layout of the reference call surface; the UNet converts at its own boundary.
#include "common.h"

namespace {

enum { OUT_V = 0, OUT_X0 = 1, OUT_EPS = 2, OUT_BOTH = 3 };
enum { RW_CONSTANT = 0, RW_SNR = 1, RW_SNR_TRUNC = 2, RW_SNR_1PLUS = 3 };

// x0_hat = a0*xt + b0x*o[c] + b0e*o[C+c] ; eps_hat = a1*xt + b1x*o[c] + b1e*o[C+c]
struct PredCoef { float a0, b0x, b0e, a1, b1x, b1e; };

__device__ __forceinline__ PredCoef pred_coef(int type, float l) {
    const float s1 = 1.f / (1.f + expf(-l)), s0 = 1.f / (1.f + expf(l));   // sigmoid(l), sigmoid(-l)
    const float sa = sqrtf(s1), ss = sqrtf(s0);
    PredCoef k = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (type == OUT_V) { k.a0 = sa; k.b0x = -ss; k.a1 = ss; k.b1x = sa; }                       // diffusion.py:232-239
    else if (type == OUT_X0) { k.b0x = 1.f; k.a1 = rsqrtf(s0); k.b1x = -expf(0.5f * l); }         // :217-219
    else if (type == OUT_EPS) { k.a0 = rsqrtf(s1); k.b0x = -expf(-0.5f * l); k.b1x = 1.f; }       // :206-208
    else {                                                                                        // :211-214
        const float r = rsqrtf(s1), e = expf(-0.5f * l), E = expf(0.5f * l);
        k.a0 = r * s1; k.b0x = s0; k.b0e = -e * s1;
        k.a1 = rsqrtf(s0) - k.a0 * E; k.b1x = -k.b0x * E; k.b1e = -k.b0e * E;
    }
    return k;
}

__global__ void q_sample_kernel(const float* x0, const float* eps, const float* logsnr, float* xt, long long n,
                                long long CHW) {
    const long long total = n * CHW;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const float l = logsnr[idx / CHW];
        const float sa = sqrtf(1.f / (1.f + expf(-l))), ss = sqrtf(1.f / (1.f + expf(l)));
        xt[idx] = x0[idx] * sa + eps[idx] * ss;
    }
}

__device__ __forceinline__ float block_sum(float v, float* sh) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += sh[i];
    __syncthreads();
    return t;
}

struct LossArgs {
    const float* x0; const float* eps; const float* xt; const float* out;     // NCHW; out has C or 2C channels
    const float* logsnr; int type, rw; int n, C; long long HW;
};

// one workgroup per sample
__global__ __launch_bounds__(256) void loss_fwd_kernel(const LossArgs p, float* loss, float* aux) {
    __shared__ float sh[8];
    const int b = blockIdx.x;
    const float l = p.logsnr[b];
    const PredCoef k = pred_coef(p.type, l);
    const float sa = sqrtf(1.f / (1.f + expf(-l))), ss = sqrtf(1.f / (1.f + expf(l)));
    float e0 = 0.f, e1 = 0.f;
    const long long N = (long long)p.C * p.HW;
    const int Co = p.type == OUT_BOTH ? 2 * p.C : p.C;
    const float* ob = p.out + (long long)b * Co * p.HW;
    for (long long i = threadIdx.x; i < N; i += blockDim.x) {
        const float x0 = p.x0[(long long)b * N + i], ep = p.eps[(long long)b * N + i];
        const float o = ob[i];
        if (p.rw == RW_SNR_TRUNC) {
            const float xt = p.xt[(long long)b * N + i];
            const float oe = p.type == OUT_BOTH ? ob[N + i] : 0.f;
            const float x0h = k.a0 * xt + k.b0x * o + k.b0e * oe;
            const float eph = k.a1 * xt + k.b1x * o + k.b1e * oe;
            e0 += (x0 - x0h) * (x0 - x0h);
            e1 += (ep - eph) * (ep - eph);
        } else {
            const float tgt = p.rw == RW_CONSTANT ? x0 : (p.rw == RW_SNR ? ep : (-x0 * ss + ep * sa));
            e0 += (tgt - o) * (tgt - o);
        }
    }
    e0 = block_sum(e0, sh);
    e1 = block_sum(e1, sh);
    if (threadIdx.x == 0) {
        const float m0 = e0 / (float)N, m1 = e1 / (float)N;
        aux[2 * b] = m0; aux[2 * b + 1] = m1;
        loss[b] = p.rw == RW_SNR_TRUNC ? fmaxf(m0, m1) : m0;
    }
}

__global__ void loss_bwd_kernel(const LossArgs p, const float* aux, const float* gloss, float* dout) {
    const long long N = (long long)p.C * p.HW, total = (long long)p.n * N;
    const int Co = p.type == OUT_BOTH ? 2 * p.C : p.C;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const long long b = idx / N, i = idx % N;
        const float l = p.logsnr[b];
        const PredCoef k = pred_coef(p.type, l);
        const float sa = sqrtf(1.f / (1.f + expf(-l))), ss = sqrtf(1.f / (1.f + expf(l)));
        const float g = gloss[b] * 2.f / (float)N;
        const float* ob = p.out + b * Co * p.HW;
        float* db = dout + b * Co * p.HW;
        const float x0 = p.x0[idx], ep = p.eps[idx], o = ob[i];
        if (p.rw == RW_SNR_TRUNC) {
            const int sel = aux[2 * b] >= aux[2 * b + 1] ? 0 : 1;
            const float xt = p.xt[idx];
            const float oe = p.type == OUT_BOTH ? ob[N + i] : 0.f;
            float r, bx, be;
            if (sel == 0) { r = (k.a0 * xt + k.b0x * o + k.b0e * oe) - x0; bx = k.b0x; be = k.b0e; }
            else          { r = (k.a1 * xt + k.b1x * o + k.b1e * oe) - ep; bx = k.b1x; be = k.b1e; }
            db[i] = g * r * bx;
            if (p.type == OUT_BOTH) db[N + i] = g * r * be;
        } else {
            const float tgt = p.rw == RW_CONSTANT ? x0 : (p.rw == RW_SNR ? ep : (-x0 * ss + ep * sa));
            db[i] = g * (o - tgt);
        }
    }
}


// ---- variational bound terms (diffusion.py:446-464; normal_kl / discretized_gaussian_loglik: functions.py:31-67).
// coef[b][8] = {a0, b0x, b0e, c1, c2, true_logvar, model_logvar, -}: x0_hat = a0*xt + b0x*o (+ b0e*o_eps),
// true_mean = c1*xt + c2*x0, model_mean = c1*xt + c2*x0_hat (the posterior mean weights do not depend on the variance type)
struct BpdArgs {
    const float* x0; const float* xt; const float* out; const float* coef; int type, clip; int n, C; long long HW;
};
constexpr float BPD_PREC = 1.f / 255.f, BPD_CUT = 0.999f, BPD_TOL = 1e-12f, CDF_K = 0.7978845608028654f, CDF_C = 0.044715f;

__device__ __forceinline__ float approx_cdf(float z) { return 0.5f * (1.f + tanhf(CDF_K * (z + CDF_C * z * z * z))); }

// one workgroup per sample: kl[b], nll[b] in bits per dimension, optional x0_hat tensor and its squared error
__global__ __launch_bounds__(256) void bpd_terms_kernel(const BpdArgs p, float* kl, float* nll, float* pred, float* mse) {
    __shared__ float sh[8];
    const int b = blockIdx.x;
    const float* k = p.coef + 8 * b;
    const float a0 = k[0], b0x = k[1], b0e = k[2], c1 = k[3], c2 = k[4], tlv = k[5], mlv = k[6];
    const float d = tlv - mlv, em = expf(-mlv), ed = expf(d), inv = expf(-0.5f * mlv);
    const long long N = (long long)p.C * p.HW;
    const int Co = p.type == OUT_BOTH ? 2 * p.C : p.C;
    const float* ob = p.out + (long long)b * Co * p.HW;
    float s_kl = 0.f, s_nll = 0.f, s_mse = 0.f;
    for (long long i = threadIdx.x; i < N; i += blockDim.x) {
        const float x0 = p.x0[(long long)b * N + i], xt = p.xt[(long long)b * N + i];
        float x0h = a0 * xt + b0x * ob[i] + (p.type == OUT_BOTH ? b0e * ob[N + i] : 0.f);
        if (p.clip) x0h = fminf(fmaxf(x0h, -1.f), 1.f);
        const float tm = c1 * xt + c2 * x0, mm = c1 * xt + c2 * x0h;
        s_kl += 0.5f * ((-1.f - d) + (tm - mm) * (tm - mm) * em + ed);
        const float xc = x0 - x0h;
        const float cu = x0 > BPD_CUT ? 1.f : approx_cdf(inv * (xc + BPD_PREC));
        const float cl = x0 < -BPD_CUT ? 0.f : approx_cdf(inv * (xc - BPD_PREC));
        s_nll -= logf(fmaxf(cu - cl - BPD_TOL, 0.f) + BPD_TOL);
        s_mse += (x0h - x0) * (x0h - x0);
        if (pred) pred[(long long)b * N + i] = x0h;
    }
    s_kl = block_sum(s_kl, sh);
    s_nll = block_sum(s_nll, sh);
    s_mse = block_sum(s_mse, sh);
    if (threadIdx.x == 0) {
        const float sc = 1.f / ((float)N * 0.6931471805599453f);
        kl[b] = s_kl * sc; nll[b] = s_nll * sc;
        if (mse) mse[b] = s_mse / (float)N;
    }
}

// dout = gloss[b] * d (use_kl[b] ? kl_b : nll_b) / d out    (training runs with clip_denoised = False, diffusion.py:514)
__global__ void bpd_bwd_kernel(const BpdArgs p, const float* use_kl, const float* gloss, float* dout) {
    const long long N = (long long)p.C * p.HW, total = (long long)p.n * N;
    const int Co = p.type == OUT_BOTH ? 2 * p.C : p.C;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const long long b = idx / N, i = idx % N;
        const float* k = p.coef + 8 * b;
        const float a0 = k[0], b0x = k[1], b0e = k[2], c1 = k[3], c2 = k[4], mlv = k[6];
        const float* ob = p.out + b * Co * p.HW;
        float* db = dout + b * Co * p.HW;
        const float x0 = p.x0[idx], xt = p.xt[idx];
        float x0h = a0 * xt + b0x * ob[i] + (p.type == OUT_BOTH ? b0e * ob[N + i] : 0.f);
        bool live = true;                                     // clamp passes no gradient outside [-1, 1]
        if (p.clip) { live = x0h >= -1.f && x0h <= 1.f; x0h = fminf(fmaxf(x0h, -1.f), 1.f); }
        float gx;                                             // d term_e / d x0_hat
        if (use_kl[b] != 0.f) {
            const float tm = c1 * xt + c2 * x0, mm = c1 * xt + c2 * x0h;
            gx = -(tm - mm) * expf(-mlv) * c2;
        } else {
            const float inv = expf(-0.5f * mlv), xc = x0 - x0h;
            const float zu = inv * (xc + BPD_PREC), zl = inv * (xc - BPD_PREC);
            const float tu = tanhf(CDF_K * (zu + CDF_C * zu * zu * zu)), tl = tanhf(CDF_K * (zl + CDF_C * zl * zl * zl));
            const float cu = x0 > BPD_CUT ? 1.f : 0.5f * (1.f + tu);
            const float cl = x0 < -BPD_CUT ? 0.f : 0.5f * (1.f + tl);
            const float D = cu - cl - BPD_TOL;
            // d cdf(z) / d x0_hat = 0.5 (1 - tanh^2) K (1 + 3 C z^2) * (-inv)
            const float du = x0 > BPD_CUT ? 0.f : 0.5f * (1.f - tu * tu) * CDF_K * (1.f + 3.f * CDF_C * zu * zu) * (-inv);
            const float dl = x0 < -BPD_CUT ? 0.f : 0.5f * (1.f - tl * tl) * CDF_K * (1.f + 3.f * CDF_C * zl * zl) * (-inv);
            gx = D >= 0.f ? -(du - dl) / (D + BPD_TOL) : 0.f;
        }
        const float g = live ? gloss[b] * gx / ((float)N * 0.6931471805599453f) : 0.f;
        db[i] = g * b0x;
        if (p.type == OUT_BOTH) db[N + i] = g * b0e;
    }
}

struct StepArgs {
    const float* xt; const float* out; const float* noise; float k[8]; const float* kdev;
    int type, cfg, last, clip; float* xn; float* xdup; int n, C; long long HW;
};

// one reverse step for a batch sharing the step index; x0_hat = a0*xt + b0x*o (+ b0e*o_eps), mean = c1*xt + c2*x0_hat
__global__ void sample_step_kernel(const StepArgs p) {
    const long long N = (long long)p.C * p.HW, total = (long long)p.n * N;
    const float* kp = p.kdev ? p.kdev : p.k;       // device-resident coefficients keep the launch HIP-graph replayable
    const float a0 = kp[0], b0x = kp[1], b0e = kp[2], c1 = kp[3], c2 = kp[4], nscale = kp[5], w = kp[6];
    const int mul = 1 + p.cfg, Co = p.type == OUT_BOTH ? 2 * p.C : p.C;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const long long b = idx / N, i = idx % N;
        const float xt = p.xt[idx];
        float mean[2];
        for (int u = 0; u < mul; ++u) {                      // rows interleaved cond, uncond (diffusion.py:369-372)
            const float* ob = p.out + (b * mul + u) * Co * p.HW;
            float x0h = a0 * xt + b0x * ob[i] + (p.type == OUT_BOTH ? b0e * ob[N + i] : 0.f);
            if (p.clip) x0h = fminf(fmaxf(x0h, -1.f), 1.f);
            mean[u] = p.last ? x0h : c1 * xt + c2 * x0h;
        }
        float v = p.cfg ? mean[0] + w * (mean[0] - mean[1]) : mean[0];
        if (p.noise) v += nscale * p.noise[idx];
        p.xn[idx] = v;
        if (p.xdup) { p.xdup[(2 * b) * N + i] = v; p.xdup[(2 * b + 1) * N + i] = v; }
    }
}

inline int grid_for(long long total) {
    long long g = (total + 255) / 256;
    if (g > 4096) g = 4096;
    return (int)(g < 1 ? 1 : g);
}

}  // namespace

extern "C" int vd_q_sample(const float* x0, const float* eps, const float* logsnr, float* xt, int32_t n, int32_t C,
                           int32_t HW, void* stream) {
    const long long chw = (long long)C * HW;
    hipLaunchKernelGGL(q_sample_kernel, dim3(grid_for(n * chw)), dim3(256), 0, (hipStream_t)stream, x0, eps, logsnr, xt,
                       (long long)n, chw);
    VD_LAUNCH_CHECK("q_sample_kernel");
    return 0;
}

extern "C" int vd_loss_fwd(const float* x0, const float* eps, const float* xt, const float* out, const float* logsnr,
                           int32_t type, int32_t rw, float* loss, float* aux, int32_t n, int32_t C, int32_t HW, void* stream) {
    VD_REQUIRE(type >= 0 && type <= 3 && rw >= 0 && rw <= 3, "vd_loss_fwd: bad model_out_type/reweight (%d,%d)", type, rw);
    VD_REQUIRE(!(type == OUT_BOTH && rw != RW_SNR_TRUNC), "vd_loss_fwd: 'both' output needs snr_trunc (reference shape rule)");
    LossArgs p = {x0, eps, xt, out, logsnr, type, rw, n, C, (long long)HW};
    hipLaunchKernelGGL(loss_fwd_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, p, loss, aux);
    VD_LAUNCH_CHECK("loss_fwd_kernel");
    return 0;
}

extern "C" int vd_loss_bwd(const float* x0, const float* eps, const float* xt, const float* out, const float* logsnr,
                           const float* aux, const float* gloss, int32_t type, int32_t rw, float* dout, int32_t n, int32_t C,
                           int32_t HW, void* stream) {
    VD_REQUIRE(type >= 0 && type <= 3 && rw >= 0 && rw <= 3, "vd_loss_bwd: bad model_out_type/reweight");
    LossArgs p = {x0, eps, xt, out, logsnr, type, rw, n, C, (long long)HW};
    hipLaunchKernelGGL(loss_bwd_kernel, dim3(grid_for((long long)n * C * HW)), dim3(256), 0, (hipStream_t)stream, p, aux, gloss,
                       dout);
    VD_LAUNCH_CHECK("loss_bwd_kernel");
    return 0;
}

extern "C" int vd_sample_step(const float* xt, const float* out, const float* noise, const float* k, const float* k_dev,
                              int32_t type, int32_t cfg, int32_t last_step, int32_t clip, float* xn, float* xdup, int32_t n,
                              int32_t C, int32_t HW, void* stream) {
    VD_REQUIRE((k != nullptr) != (k_dev != nullptr), "vd_sample_step: pass the coefficients either as host k or as device k_dev");
    VD_REQUIRE(noise != nullptr || k_dev != nullptr || k[5] == 0.f, "vd_sample_step: noise required when the noise scale is non-zero");
    StepArgs p = {};
    p.xt = xt; p.out = out; p.noise = noise; p.kdev = k_dev;
    for (int i = 0; i < 8; ++i) p.k[i] = k ? k[i] : 0.f;
    p.type = type; p.cfg = cfg; p.last = last_step; p.clip = clip; p.xn = xn; p.xdup = xdup; p.n = n; p.C = C; p.HW = HW;
    hipLaunchKernelGGL(sample_step_kernel, dim3(grid_for((long long)n * C * HW)), dim3(256), 0, (hipStream_t)stream, p);
    VD_LAUNCH_CHECK("sample_step_kernel");
    return 0;
}

extern "C" int vd_bpd_terms(const float* x0, const float* xt, const float* out, const float* coef, int32_t type, int32_t clip,
                            float* kl, float* nll, float* pred, float* mse, int32_t n, int32_t C, int32_t HW, void* stream) {
    VD_REQUIRE(type >= 0 && type <= 3, "vd_bpd_terms: bad model_out_type %d", type);
    VD_REQUIRE(x0 && xt && out && coef && kl && nll, "vd_bpd_terms: null pointer");
    BpdArgs p = {x0, xt, out, coef, type, clip, n, C, (long long)HW};
    hipLaunchKernelGGL(bpd_terms_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, p, kl, nll, pred, mse);
    VD_LAUNCH_CHECK("bpd_terms_kernel");
    return 0;
}

extern "C" int vd_bpd_bwd(const float* x0, const float* xt, const float* out, const float* coef, const float* use_kl,
                          const float* gloss, int32_t type, int32_t clip, float* dout, int32_t n, int32_t C, int32_t HW,
                          void* stream) {
    VD_REQUIRE(type >= 0 && type <= 3, "vd_bpd_bwd: bad model_out_type %d", type);
    VD_REQUIRE(x0 && xt && out && coef && use_kl && gloss && dout, "vd_bpd_bwd: null pointer");
    BpdArgs p = {x0, xt, out, coef, type, clip, n, C, (long long)HW};
    hipLaunchKernelGGL(bpd_bwd_kernel, dim3(grid_for((long long)n * C * HW)), dim3(256), 0, (hipStream_t)stream, p, use_kl, gloss,
                       dout);
    VD_LAUNCH_CHECK("bpd_bwd_kernel");
    return 0;
}
