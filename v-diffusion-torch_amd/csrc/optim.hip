// optim.hip -- optimizer tail as multi-tensor passes over FLAT fp32 buffers:
// global-norm clip (nn.utils.clip_grad_norm_, train_utils.py:161) + AdamW (train.py:158) + EMA (utils.py:144-149)
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* g, long long n, float* part) {
    __shared__ float sh[4];
    float s = 0.f;
    const long long n4 = n >> 2;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        const f32x4 v = reinterpret_cast<const f32x4*>(g)[i];
        s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { const float v = g[(n4 << 2) + threadIdx.x]; s += v * v; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}
__global__ void sumsq_final_kernel(const float* part, int nb, float* out) {
    __shared__ double sh[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < nb; i += blockDim.x) s += (double)part[i];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) out[0] = (float)sh[0];
}

// [r_lo, r_hi): an index range with its own treatment this step (the class-embedding tensors, reference unet.py:207-215):
//   r_mode 1: the range received NO gradient (class-conditional network called with y = None: the reference leaves .grad None
//             and torch.optim.AdamW skips the parameter -- no moment decay, no weight decay, no update, step not advanced);
//             p, m, v stay untouched, the EMA shadow still follows p (utils.py:144-149 walks every parameter);
//   r_mode 2: the range is updated with its own bias corrections r_bc1 / r_bc2 (its per-parameter step count lags the rest).
//   r_mode 3: decided ON THE DEVICE (vd_adamw_ema_flagged): r_flag[0] > 0 -> as r_mode 2 with the bias corrections of step count
//             r_steps[0] + 1 (1 - beta^k evaluated here in double), else as r_mode 1; cls_step_advance_kernel then advances r_steps[0].
//             In data-parallel runs the flag is one extra slot behind the flat gradient buffer, summed over ranks by the LAST bucket's
//             all-reduce (trainer.FlatState): whether the class embedding received a gradient is one decision for all replicas, made
//             without a collective of its own and without any host reading it.
// every operand of the update is streamed once: non-temporal accesses (VD_OPT_NT=0: plain; same-box A/B tests/probe/r04_pass15.sh:
// 404-409 vs 439-442 us for the 243 MB CIFAR buffers, 5.4 vs 5.0 TB/s)
#ifndef VD_OPT_NT
#define VD_OPT_NT 1
#endif
__device__ __forceinline__ float ld1(const float* p) {
#if VD_OPT_NT
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}
__device__ __forceinline__ void st1(float* p, float v) {
#if VD_OPT_NT
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}
__global__ void adamw_ema_kernel(float* p, const float* g, float* m, float* v, float* ema, long long n, const float* gnorm_sq,
                                 float max_norm, float lr, float b1, float b2, float eps, float wd, float bc1, float bc2,
                                 float ema_decay, long long r_lo, long long r_hi, int r_mode, float r_bc1, float r_bc2,
                                 const float* r_flag, const int* r_steps, double r_beta1, double r_beta2) {
    if (r_mode == 3) {
        if (r_flag[0] > 0.f) {
            const double kc = (double)(r_steps[0] + 1);
            r_bc1 = (float)(1.0 - pow(r_beta1, kc));
            r_bc2 = (float)(1.0 - pow(r_beta2, kc));
            r_mode = 2;
        } else {
            r_mode = 1;
        }
    }
    float clip = 1.f;
    if (gnorm_sq && max_norm > 0.f) {
        const float c = max_norm / (sqrtf(gnorm_sq[0]) + 1e-6f);          // torch clip_grad_norm_
        clip = c < 1.f ? c : 1.f;
    }
    const float step = lr / bc1, rs2 = 1.f / sqrtf(bc2);
    const float r_step = lr / r_bc1, r_rs2 = 1.f / sqrtf(r_bc2);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const bool in_r = r_mode != 0 && i >= r_lo && i < r_hi;
        float pi = ld1(p + i);
        if (!(in_r && r_mode == 1)) {
            const float gi = ld1(g + i) * clip;
            pi *= 1.f - lr * wd;
            const float mi = b1 * ld1(m + i) + (1.f - b1) * gi;
            const float vi = b2 * ld1(v + i) + (1.f - b2) * gi * gi;
            pi -= (in_r ? r_step : step) * mi / (sqrtf(vi) * (in_r ? r_rs2 : rs2) + eps);
            st1(p + i, pi); st1(m + i, mi); st1(v + i, vi);
        }
        if (ema) { const float e = ld1(ema + i); st1(ema + i, e + (1.f - ema_decay) * (pi - e)); }
    }
}

__global__ void cls_step_advance_kernel(const float* r_flag, int* r_steps) {
    if (r_flag[0] > 0.f) r_steps[0] += 1;
}

constexpr int SUMSQ_BLOCKS = 1024;
}  // namespace

extern "C" size_t vd_sumsq_ws_bytes(int64_t) { return SUMSQ_BLOCKS * sizeof(float); }

extern "C" int vd_sumsq(const float* g, int64_t n, float* out1, float* ws, size_t ws_bytes, void* stream) {
    VD_REQUIRE(ws && ws_bytes >= SUMSQ_BLOCKS * sizeof(float), "vd_sumsq: workspace too small");
    VD_REQUIRE(vd_aligned16(g), "vd_sumsq: buffer must be 16-byte aligned");
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3(SUMSQ_BLOCKS), dim3(256), 0, (hipStream_t)stream, g, (long long)n, ws);
    VD_LAUNCH_CHECK("sumsq_partial_kernel");
    hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, ws, SUMSQ_BLOCKS, out1);
    VD_LAUNCH_CHECK("sumsq_final_kernel");
    return 0;
}

extern "C" int vd_adamw_ema(float* p, const float* g, float* m, float* v, float* ema, int64_t n, const float* gnorm_sq,
                            float max_norm, float lr, float beta1, float beta2, float eps, float wd, float bc1, float bc2,
                            float ema_decay, int64_t r_lo, int64_t r_hi, int32_t r_mode, float r_bc1, float r_bc2, void* stream) {
    VD_REQUIRE(r_mode >= 0 && r_mode <= 2, "vd_adamw_ema: r_mode must be 0, 1 or 2");
    VD_REQUIRE(r_mode == 0 || (0 <= r_lo && r_lo <= r_hi && r_hi <= n), "vd_adamw_ema: bad range [%lld, %lld)", (long long)r_lo, (long long)r_hi);
    VD_REQUIRE(r_mode != 2 || (r_bc1 > 0.f && r_bc2 > 0.f), "vd_adamw_ema: r_mode 2 needs positive bias corrections");
    long long grid = (n + 255) / 256;
    if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(adamw_ema_kernel, dim3((int)grid), dim3(256), 0, (hipStream_t)stream, p, g, m, v, ema, (long long)n,
                       gnorm_sq, max_norm, lr, beta1, beta2, eps, wd, bc1, bc2, ema_decay, (long long)r_lo, (long long)r_hi, (int)r_mode,
                       r_mode == 2 ? r_bc1 : 1.f, r_mode == 2 ? r_bc2 : 1.f, (const float*)nullptr, (const int*)nullptr, 0.0, 0.0);
    VD_LAUNCH_CHECK("adamw_ema_kernel");
    return 0;
}

extern "C" int vd_adamw_ema_flagged(float* p, const float* g, float* m, float* v, float* ema, int64_t n, const float* gnorm_sq,
                                    float max_norm, float lr, float beta1, float beta2, float eps, float wd, float bc1, float bc2,
                                    float ema_decay, int64_t r_lo, int64_t r_hi, const float* r_flag, int32_t* r_steps,
                                    double r_beta1, double r_beta2, void* stream) {
    VD_REQUIRE(r_flag && r_steps, "vd_adamw_ema_flagged: needs the device flag and the device step counter");
    VD_REQUIRE(0 <= r_lo && r_lo <= r_hi && r_hi <= n, "vd_adamw_ema_flagged: bad range [%lld, %lld)", (long long)r_lo, (long long)r_hi);
    VD_REQUIRE(r_beta1 >= 0.0 && r_beta1 < 1.0 && r_beta2 >= 0.0 && r_beta2 < 1.0, "vd_adamw_ema_flagged: betas must lie in [0, 1)");
    long long grid = (n + 255) / 256;
    if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(adamw_ema_kernel, dim3((int)grid), dim3(256), 0, (hipStream_t)stream, p, g, m, v, ema, (long long)n,
                       gnorm_sq, max_norm, lr, beta1, beta2, eps, wd, bc1, bc2, ema_decay, (long long)r_lo, (long long)r_hi, 3,
                       1.f, 1.f, r_flag, (const int*)r_steps, r_beta1, r_beta2);
    VD_LAUNCH_CHECK("adamw_ema_kernel");
    hipLaunchKernelGGL(cls_step_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, r_flag, (int*)r_steps);
    VD_LAUNCH_CHECK("cls_step_advance_kernel");
    return 0;
}
