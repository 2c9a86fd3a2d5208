// Issue rate of the 32-bit integer multiplies on gfx950: v_mul_lo_u32 (full 32 x 32 -> low 32, what a 32-bit hash such as
// the counter RNG's mix32 uses) against v_mul_u32_u24 (24 x 24 bits) and, for scale, v_fma_f32 and v_rcp_f32.
// Build: hipcc --offload-arch=gfx950 -O3 -o int_mul_rate int_mul_rate.hip ; run: ./int_mul_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define BODY(OP)                                                                                                             \
    for (int i = 0; i < iters; ++i) {                                                                                        \
        _Pragma("unroll 8") for (int k = 0; k < 8; ++k) {                                                                     \
            asm volatile(OP " %0, %0, %8\n " OP " %1, %1, %8\n " OP " %2, %2, %8\n " OP " %3, %3, %8\n " OP " %4, %4, %8\n " OP  \
                            " %5, %5, %8\n " OP " %6, %6, %8\n " OP " %7, %7, %8"                                             \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));          \
        }                                                                                                                    \
    }

template <int WHICH> __global__ void __launch_bounds__(256) k(uint32_t *out, int iters) {
    uint32_t a0 = threadIdx.x | 1u, a1 = a0 + 2, a2 = a0 + 4, a3 = a0 + 6, a4 = a0 + 8, a5 = a0 + 10, a6 = a0 + 12, a7 = a0 + 14;
    const uint32_t m = WHICH == 2 ? 0x3f800001u : 0x7feb352du;
    if (WHICH == 0) { BODY("v_mul_lo_u32") }
    if (WHICH == 1) { BODY("v_mul_u32_u24") }
    if (WHICH == 2) { BODY("v_mul_f32") }
    if (WHICH == 3) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll 8
            for (int k2 = 0; k2 < 8; ++k2)
                asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = ((a0 ^ a1) ^ (a2 ^ a3)) ^ ((a4 ^ a5) ^ (a6 ^ a7));
}

int main() {
    const int blocks = 256 * 8, threads = 256, iters = 2000;
    uint32_t *out;
    if (hipMalloc(&out, sizeof(uint32_t) * blocks * threads) != hipSuccess) return 1;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const char *names[] = {"v_mul_lo_u32", "v_mul_u32_u24", "v_mul_f32", "v_rcp_f32"};
    for (int which = 0; which < 4; ++which) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (which == 0) k<0><<<blocks, threads>>>(out, iters);
            if (which == 1) k<1><<<blocks, threads>>>(out, iters);
            if (which == 2) k<2><<<blocks, threads>>>(out, iters);
            if (which == 3) k<3><<<blocks, threads>>>(out, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
        }
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double wave_insts = (double)blocks * (threads / 64) * iters * 64.0;
        printf("%-14s %8.3f ms  %7.1f G wave-instructions/s\n", names[which], ms, wave_insts / (ms * 1e-3) / 1e9);
    }
    hipFree(out);
    return 0;
}
