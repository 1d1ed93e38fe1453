#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* a, float* o) {
    __shared__ __attribute__((aligned(16))) float lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = 7.0f;
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)a, 0, 0x80000000u, 0x00020000);
    int vo = (threadIdx.x & 1) ? (int)0x80000000 : (int)(threadIdx.x * 16);   // odd lanes out of range
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds, 16, vo, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += 64) o[i] = lds[i];
}
int main() {
    float *a, *o; float h[256], ha[1024];
    for (int i = 0; i < 1024; ++i) ha[i] = 100 + i;
    hipMalloc(&a, 4096); hipMalloc(&o, 1024);
    hipMemcpy(a, ha, 4096, hipMemcpyHostToDevice);
    k<<<1, 64>>>(a, o);
    hipMemcpy(h, o, 1024, hipMemcpyDeviceToHost);
    for (int i = 0; i < 24; ++i) printf("%g ", h[i]);
    printf("\n");
    return 0;
}
