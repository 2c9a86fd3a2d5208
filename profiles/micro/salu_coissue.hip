// Do scalar instructions take issue slots away from vector instructions on gfx950?  Each loop iteration holds 8
// v_fma_f32 and K scalar adds (K = 0, 4, 8, 16), all independent; 8 waves per SIMD are resident, so the scheduler can
// pair one wave's scalar work with another wave's vector work if the hardware allows it.
// Build: hipcc --offload-arch=gfx950 -O3 -o salu_coissue salu_coissue.hip
#include <hip/hip_runtime.h>
#include <cstdio>

template <int K>
__global__ void __launch_bounds__(256) k_mix(float *out, int iters) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float m = 1.0000001f, c = 0.5f;
    int s0 = iters, s1 = iters + 1, s2 = iters + 2, s3 = iters + 3;
    for (int i = 0; i < iters; ++i) {
#pragma unroll 8
        for (int k = 0; k < 8; ++k) {
            asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
#pragma unroll
            for (int j = 0; j < K / 4; ++j)
                asm volatile("s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 3\n s_add_u32 %2, %2, 5\n s_add_u32 %3, %3, 7"
                             : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc");
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7)) + (float)(s0 ^ s1 ^ s2 ^ s3);
}

template <int K>
static void run(float *out, int blocks, int iters) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k_mix<K><<<blocks, 256>>>(out, 10);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k_mix<K><<<blocks, 256>>>(out, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) printf("error: %s\n", hipGetErrorString(err));
    double v = double(blocks) * 4 * iters * 64.0;
    printf("8 v_fma_f32 + %2d s_add_u32 per group: %7.3f ms  %7.1f G VALU/s  %7.1f G SALU/s\n", K, ms, v / (ms * 1e-3) / 1e9,
           v * K / 8.0 / (ms * 1e-3) / 1e9);
}

int main() {
    const int blocks = 256 * 8, iters = 4000;
    float *out;
    if (hipMalloc(&out, sizeof(float) * blocks * 256) != hipSuccess) return 1;
    run<0>(out, blocks, iters); run<4>(out, blocks, iters); run<8>(out, blocks, iters); run<16>(out, blocks, iters);
    (void)hipFree(out);
    return 0;
}
