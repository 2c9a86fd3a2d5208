// How fast do the display bytes of a frame reach page-locked host memory?  hipMemcpyAsync (the runtime's DMA engine)
// against a kernel that stores into the host buffer through its device address, for one 1080p frame (6.2 MB) and one
// 1024x1024 frame (3.1 MB), one copy at a time and two at once on two streams.
// Build: hipcc --offload-arch=gfx950 -O3 -o d2h_copy d2h_copy.hip ; run: ./d2h_copy
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void k_copy(const uint4 *src, uint4 *dst, size_t n16) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

int main() {
    for (size_t bytes : {(size_t)1920 * 1080 * 3, (size_t)1024 * 1024 * 3}) {
        unsigned char *d[2], *h[2], *hd[2];
        hipStream_t st[2];
        for (int k = 0; k < 2; ++k) {
            CHK(hipMalloc((void **)&d[k], bytes));
            CHK(hipHostMalloc((void **)&h[k], bytes, hipHostMallocDefault));
            CHK(hipHostGetDevicePointer((void **)&hd[k], h[k], 0));
            CHK(hipStreamCreateWithFlags(&st[k], hipStreamNonBlocking));
            CHK(hipMemset(d[k], k + 1, bytes));
        }
        auto run = [&](const char *name, int streams, int blocks) {
            const int reps = 200;
            for (int warm = 0; warm < 2; ++warm) {
                auto t0 = std::chrono::steady_clock::now();
                for (int r = 0; r < reps; ++r)
                    for (int k = 0; k < streams; ++k) {
                        if (blocks == 0) CHK(hipMemcpyAsync(h[k], d[k], bytes, hipMemcpyDeviceToHost, st[k]));
                        else k_copy<<<blocks, 256, 0, st[k]>>>((const uint4 *)d[k], (uint4 *)hd[k], bytes / 16);
                    }
                for (int k = 0; k < streams; ++k) CHK(hipStreamSynchronize(st[k]));
                double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                if (warm) printf("%8zu bytes  %-28s streams %d: %7.1f us per copy, %6.1f GB/s total\n", bytes, name, streams, 1e6 * s / reps, bytes * (double)reps * streams / s / 1e9);
            }
        };
        run("hipMemcpyAsync", 1, 0);
        run("hipMemcpyAsync", 2, 0);
        for (int blocks : {4, 16, 64, 256}) {
            char name[64];
            snprintf(name, sizeof name, "kernel, %d blocks", blocks);
            run(name, 1, blocks);
        }
        run("kernel, 16 blocks", 2, 16);
        if (h[0][bytes - 1] != 1 || h[1][bytes - 1] != 2) printf("copy check failed\n");
    }
    return 0;
}
