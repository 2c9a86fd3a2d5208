// What does the FETCH_SIZE counter report for the access patterns of this library, against byte counts known by
// construction?  A 2 GiB table (eight times the 256 MiB Infinity Cache) is read exactly once per kernel, every
// 128-byte line at most once, so the bytes that must come from HBM are known:
//   k_stream        16 bytes per lane, consecutive lanes consecutive addresses (the queue records)     -> the whole table
//   k_gather<7>     every lane 7 x 16 bytes of its own, pseudo-randomly placed line (a BVH node fetch)  -> 128 B per line if lines
//                                                                                                         are fetched whole, else 2 x 64
//   k_gather<4>     every lane the first 64 bytes of its own line (a leaf primitive / a shading row)     -> 64 or 128 B per line
//   k_gather<1>     every lane 16 bytes of its own line                                                  -> 64 or 128 B per line
// Run each under `rocprofv3 --pmc FETCH_SIZE` (profiles/calibrate_fetch.sh) and compare per kernel.
// Build: hipcc --offload-arch=gfx950 -O3 -o fetch_calibration fetch_calibration.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

__global__ void __launch_bounds__(256) k_stream(const float4 *table, size_t n_vec, float *out) {
    float acc = 0.0f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n_vec; i += (size_t)gridDim.x * blockDim.x) {
        float4 v = table[i];
        acc += v.x + v.w;
    }
    if (acc == 12345.678f) out[0] = acc;
}

// line(i) = (i * odd) mod n_lines is a bijection for n_lines a power of two: every line is visited exactly once
template <int CHUNKS>
__global__ void __launch_bounds__(256) k_gather(const float4 *table, uint32_t n_lines, uint32_t n_touch, float *out) {
    float acc = 0.0f;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_touch; i += gridDim.x * blockDim.x) {
        const uint32_t line = (i * 2654435761u) & (n_lines - 1u);
        const float4 *p = table + (size_t)line * 8u;
#pragma unroll
        for (int k = 0; k < CHUNKS; ++k) { float4 v = p[k]; acc += v.x + v.w; }
    }
    if (acc == 12345.678f) out[0] = acc;
}

int main() {
    const uint32_t n_lines = 1u << 24;                 // 2 GiB of 128-byte lines
    const size_t bytes = (size_t)n_lines * 128;
    float4 *table; float *out;
    if (hipMalloc(&table, bytes) != hipSuccess || hipMalloc(&out, 256) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMemset(table, 0, bytes);
    (void)hipDeviceSynchronize();
    const uint32_t n_touch = n_lines / 2;              // the gathers touch half of the lines: 8M lines = 1 GiB if whole
    k_stream<<<8192, 256>>>(table, bytes / 16, out);
    k_gather<7><<<8192, 256>>>(table, n_lines, n_touch, out);
    k_gather<4><<<8192, 256>>>(table, n_lines, n_touch, out);
    k_gather<1><<<8192, 256>>>(table, n_lines, n_touch, out);
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 2; }
    printf("expected bytes: k_stream %zu; k_gather lines touched %u (x128 = %zu, x64 = %zu)\n", bytes, n_touch, (size_t)n_touch * 128, (size_t)n_touch * 64);
    (void)hipFree(table); (void)hipFree(out);
    return 0;
}
