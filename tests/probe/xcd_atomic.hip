// probe (round 6): is an agent-scope atomic RMW on ordinary device memory ONE counter for the whole device, or one per XCD L2?
// Every workgroup of a 4096-workgroup grid takes a ticket; tickets must be a permutation of 0..4095.  Also with system scope.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
template <int SCOPE>
__global__ void k(unsigned* ticket, unsigned* seen, unsigned* xcc) {
    if (threadIdx.x == 0) {
        const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, SCOPE);
        if (t < 8192) atomicAdd(seen + t, 1u);
        unsigned id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        xcc[blockIdx.x] = id & 0xf;
    }
}
int main() {
    unsigned *ticket, *seen, *xcc;
    hipMalloc(&ticket, 4); hipMalloc(&seen, 8192 * 4); hipMalloc(&xcc, 4096 * 4);
    for (int scope = 0; scope < 2; ++scope) {
        for (int rep = 0; rep < 3; ++rep) {
            hipMemset(ticket, 0, 4); hipMemset(seen, 0, 8192 * 4);
            if (scope == 0) k<__HIP_MEMORY_SCOPE_AGENT><<<4096, 64>>>(ticket, seen, xcc);
            else k<__HIP_MEMORY_SCOPE_SYSTEM><<<4096, 64>>>(ticket, seen, xcc);
            hipDeviceSynchronize();
            std::vector<unsigned> h(8192), x(4096); unsigned t;
            hipMemcpy(h.data(), seen, 8192 * 4, hipMemcpyDeviceToHost); hipMemcpy(&t, ticket, 4, hipMemcpyDeviceToHost);
            hipMemcpy(x.data(), xcc, 4096 * 4, hipMemcpyDeviceToHost);
            int dup = 0, missing = 0;
            for (int i = 0; i < 4096; ++i) { if (h[i] > 1) ++dup; if (h[i] == 0) ++missing; }
            printf("%s scope: final counter %u, tickets handed out twice %d, never %d; XCC of blocks 0..15:", scope ? "system" : "agent", t, dup, missing);
            for (int i = 0; i < 16; ++i) printf(" %u", x[i]);
            printf("\n");
        }
    }
    return 0;
}
