// diffusion.hip -- fused elementwise kernels of the diffusion process (HBM / latency bound):
//   q_sample                         reference v_diffusion/diffusion.py:242-245
//   train_loss (mse) forward/backward  diffusion.py:466-490,520-541 ; flat_mean functions.py:102-104
//   one reverse step (p_mean_var + CFG + noise)  diffusion.py:317-392
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
