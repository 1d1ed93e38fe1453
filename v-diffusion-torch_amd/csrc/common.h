// common.h -- shared helpers for libvdiff_hip (gfx950 only)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/vdiff_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

typedef float f32x2 __attribute__((ext_vector_type(2)));

// -DVD_PROBES builds libvdiff_hip_probe.so (tests/probe/: in-kernel timestamps, timing-only kernel variants that compute WRONG
// results).  The product library is built without it: every probe branch below is guarded by this constant and folds away, and
// no environment variable can reach a wrong-result path.
#ifdef VD_PROBES
constexpr bool VD_PROBE_BUILD = true;
#else
constexpr bool VD_PROBE_BUILD = false;
#endif

void vd_set_error(const char* fmt, ...);
int vd_cu_count(void);                       // compute units of the calling thread's current device (cached per device id)
int vd_persistent_cus(void);                 // vd_cu_count() minus the CUs reserved for other streams' kernels (vd_set_reserved_cus)
int vd_gemm_grouped_wgrad_kblk(const float* const* A, const float* const* B, float* const* C, float* const* colsum, int32_t count, int32_t M,
                               int32_t N, int32_t K, int64_t lda, int64_t ldb, int64_t ldc, int32_t splitk, float* ws, size_t ws_bytes,
                               void* stream, int64_t a_kblk, int64_t b_kblk, int32_t slabs_only);   // gemm.hip: grouped weight-gradient GEMMs on K-blocked operands
int vd_gemm_grouped_wgrad_used_slabs(int32_t count, int32_t M, int32_t N, int32_t K, int32_t splitk);   // slabs such a launch fills
extern thread_local int vd_g_last_tile;      // code of the calling thread's last matmul-shaped launch (vd_gemm_last_tile)

#define VD_REQUIRE(cond, ...)                                   \
    do { if (!(cond)) { vd_set_error(__VA_ARGS__); return 1; } } while (0)

#define VD_LAUNCH_CHECK(name)                                                            \
    do { hipError_t e_ = hipGetLastError();                                              \
         if (e_ != hipSuccess) { vd_set_error("%s: %s", name, hipGetErrorString(e_)); return 2; } } while (0)

static inline bool vd_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// (v_rcp_f32, <= 1 ulp, instead of the IEEE division's ten instructions: the GroupNorm kernels run their arithmetic between their load and
// store phases, not beside them)
__device__ __forceinline__ float vd_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

// Philox4x32-10 counter-based generator: 4 uniform words for (seed, 64-bit counter)
__device__ __forceinline__ void vd_philox4(uint64_t seed, uint64_t ctr, uint32_t out[4]) {
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    uint32_t c0 = (uint32_t)ctr, c1 = (uint32_t)(ctr >> 32), c2 = 0x243F6A88u, c3 = 0x85A308D3u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
// keep-mask scale for dropout element group `vec_index` (4 consecutive channels): 0 or 1/(1-p)
__device__ __forceinline__ f32x4 vd_dropout_scale4(uint64_t seed, uint64_t vec_index, float p) {
    uint32_t r[4];
    vd_philox4(seed, vec_index, r);
    const float inv = 1.0f / (1.0f - p);
    f32x4 m;
#pragma unroll
    for (int j = 0; j < 4; ++j) m[j] = ((r[j] >> 8) * (1.0f / 16777216.0f) >= p) ? inv : 0.0f;
    return m;
}
