// What does a global_load_dwordx4 cost when its 64 lanes touch 64 different 128-byte lines, against 8 lines (8 lanes
// per line, the line read whole) - both from an L2-resident table, addresses pseudo-random per iteration?
// Build: hipcc --offload-arch=gfx950 -O3 -o gather_lines gather_lines.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

// MODE 0: every lane its own line, LOADS chunks of it (LOADS load instructions, 64 lines each)
// MODE 1: 8 lanes share a line and read its 8 chunks (one instruction = 8 whole lines); LOADS instructions fetch LOADS*8 lines
template <int MODE, int LOADS>
__global__ void __launch_bounds__(256) k_gather(const float4 *table, uint32_t n_lines, int iters, float *out) {
    const uint32_t lane = threadIdx.x & 63u, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    float acc = 0.0f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < LOADS; ++k) {
            uint32_t line, chunk;
            if (MODE == 0) { line = mix(wave * 64u + lane + 0x9E3779B9u * (uint32_t)it) % n_lines; chunk = (uint32_t)k & 7u; }
            else { line = mix(wave * 64u + (lane >> 3) + 8u * (uint32_t)k + 0x9E3779B9u * (uint32_t)it) % n_lines; chunk = lane & 7u; }
            float4 v = table[(size_t)line * 8u + chunk];
            acc += v.x + v.w;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int MODE, int LOADS>
static void run(const float4 *table, uint32_t n_lines, float *out, const char *name) {
    const int blocks = 256 * 8, iters = 2000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k_gather<MODE, LOADS><<<blocks, 256>>>(table, n_lines, 10, out);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k_gather<MODE, LOADS><<<blocks, 256>>>(table, n_lines, iters, out);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    double instr = double(blocks) * 4 * iters * LOADS;
    double lines = instr * (MODE == 0 ? 64.0 : 8.0);
    printf("%-46s table %6.1f MB: %7.3f ms  %6.2f G load instr/s  %7.1f G lines/s  %7.1f GB/s useful\n", name, n_lines * 128.0 / 1e6, ms,
           instr / (ms * 1e-3) / 1e9, lines / (ms * 1e-3) / 1e9, instr * 1024.0 / (ms * 1e-3) / 1e9);
}

int main() {
    for (uint32_t n_lines : {1u << 14, 1u << 20}) {   // 2 MB (L2-resident), 128 MB (Infinity-Cache / HBM)
        float4 *table; float *out;
        if (hipMalloc(&table, (size_t)n_lines * 128) != hipSuccess || hipMalloc(&out, sizeof(float) * 256 * 8 * 256) != hipSuccess) return 1;
        (void)hipMemset(table, 0, (size_t)n_lines * 128);
        run<0, 8>(table, n_lines, out, "64 lines per instruction, 8 chunks of each");
        run<0, 4>(table, n_lines, out, "64 lines per instruction, 4 chunks of each");
        run<0, 1>(table, n_lines, out, "64 lines per instruction, 1 chunk of each");
        run<1, 8>(table, n_lines, out, "8 whole lines per instruction, 8 instructions");
        (void)hipFree(table); (void)hipFree(out);
    }
    return 0;
}
