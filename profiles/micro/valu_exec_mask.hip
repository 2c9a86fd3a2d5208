// Does a VALU instruction whose EXEC mask leaves whole 16-lane groups idle issue faster on gfx950?
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_exec_mask valu_exec_mask.hip ; run: ./valu_exec_mask
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ void __launch_bounds__(256) k_fma(float *out, int iters, uint64_t lane_mask) {
    const int lane = threadIdx.x & 63;
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float m = 1.0000001f, c = 0.5f;
    if ((lane_mask >> lane) & 1ull) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll 8
            for (int k = 0; k < 8; ++k) {
                a0 = __builtin_fmaf(a0, m, c); a1 = __builtin_fmaf(a1, m, c); a2 = __builtin_fmaf(a2, m, c); a3 = __builtin_fmaf(a3, m, c);
                a4 = __builtin_fmaf(a4, m, c); a5 = __builtin_fmaf(a5, m, c); a6 = __builtin_fmaf(a6, m, c); a7 = __builtin_fmaf(a7, m, c);
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
}

// the same with instructions the compiler cannot pair into v_pk_fma_f32, and in binary64
__global__ void __launch_bounds__(256) k_fma_single(float *out, int iters) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float m = 1.0000001f, c = 0.5f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll 8
        for (int k = 0; k < 8; ++k) {
            asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
}
__global__ void __launch_bounds__(256) k_fma_double(float *out, int iters) {
    double a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const double m = 1.0000001, c = 0.5;
    for (int i = 0; i < iters; ++i) {
#pragma unroll 8
        for (int k = 0; k < 8; ++k) {
            a0 = __builtin_fma(a0, m, c); a1 = __builtin_fma(a1, m, c); a2 = __builtin_fma(a2, m, c); a3 = __builtin_fma(a3, m, c);
            a4 = __builtin_fma(a4, m, c); a5 = __builtin_fma(a5, m, c); a6 = __builtin_fma(a6, m, c); a7 = __builtin_fma(a7, m, c);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (float)(((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7)));
}

int main() {
    const int blocks = 256 * 8, threads = 256, iters = 4000;
    float *out;
    if (hipMalloc(&out, sizeof(float) * blocks * threads) != hipSuccess) return 1;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    struct { const char *name; uint64_t mask; } cases[] = {
        {"all 64 lanes", ~0ull}, {"lanes 0-31", 0xFFFFFFFFull}, {"lanes 0-15", 0xFFFFull},
        {"every 4th lane (16 lanes over all groups)", 0x1111111111111111ull}, {"lanes 0-15 and 48-63", 0xFFFF00000000FFFFull},
    };
    for (auto &cs : cases) {
        k_fma<<<blocks, threads>>>(out, 10, cs.mask);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        k_fma<<<blocks, threads>>>(out, iters, cs.mask);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        double wave_insts = double(blocks) * (threads / 64) * iters * 64.0;
        printf("%-45s %8.3f ms  %7.1f G wave-FMA/s\n", cs.name, ms, wave_insts / (ms * 1e-3) / 1e9);
    }
    for (int which = 0; which < 2; ++which) {
        if (which == 0) k_fma_single<<<blocks, threads>>>(out, 10); else k_fma_double<<<blocks, threads>>>(out, 10);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        if (which == 0) k_fma_single<<<blocks, threads>>>(out, iters); else k_fma_double<<<blocks, threads>>>(out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        double wave_insts = double(blocks) * (threads / 64) * iters * 64.0;
        printf("%-45s %8.3f ms  %7.1f G wave-FMA/s\n", which == 0 ? "v_fma_f32 (unpaired), all lanes" : "v_fma_f64, all lanes", ms, wave_insts / (ms * 1e-3) / 1e9);
    }
    hipFree(out);
    return 0;
}
