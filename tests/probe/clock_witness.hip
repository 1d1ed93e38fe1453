// clock_witness.hip -- the shader clock the chip HOLDS while some other kernel runs, read from inside the GPU without touching that kernel
// (MI355X_MICROARCH.md "DVFS give-back" item 6: in-kernel clock = delta s_memtime / delta s_memrealtime x 100 MHz; board power and the
// sysfs sclk are not the test).  One single-lane workgroup per XCD (workgroup w lands on XCD w % 8) sits beside the kernel under test --
// no LDS, one wave slot -- and stamps (s_memtime, s_memrealtime) every `period` ticks of the 100 MHz constant clock.  The product
// kernels carry no stamp: this is its own tiny library (tests/probe/libclock_witness.so), loaded only by tests/probe/clock_by_kernel.py.
#include <hip/hip_runtime.h>

__global__ void clock_witness_kernel(unsigned long long* out, int nsamp, unsigned long long period) {
    if (threadIdx.x != 0) return;
    unsigned long long* o = out + (size_t)blockIdx.x * 2 * nsamp;
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < nsamp; ++i) {
        while (__builtin_amdgcn_s_memrealtime() - r0 < (unsigned long long)i * period) __builtin_amdgcn_s_sleep(64);
        const unsigned long long c = __builtin_amdgcn_s_memtime(), r = __builtin_amdgcn_s_memrealtime();
        o[2 * i] = c; o[2 * i + 1] = r;
    }
}

extern "C" int launch_clock_witness(unsigned long long* out, int nwg, int nsamp, unsigned long long period, void* stream) {
    hipLaunchKernelGGL(clock_witness_kernel, dim3(nwg), dim3(64), 0, (hipStream_t)stream, out, nsamp, period);
    return (int)hipGetLastError();
}
