// Micro-probe (diagnostic, not part of the library): do two PROCESSES that share one MI355X see each other's data through a
// cache when they use the SAME virtual addresses?  Each process fills a buffer with its own tag and launches kernels whose
// waves re-read it through (a) scalar loads (s_load: the scalar data cache) and (b) vector loads (global_load: TCP / L2),
// counting words that are not the process's tag.  Run two copies at once with different tags:
//     ./va_alias_probe 1 20 &  ./va_alias_probe 2 20
// build: hipcc --offload-arch=gfx950 -O3 -o scripts/micro/va_alias_probe scripts/micro/va_alias_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ __launch_bounds__(256) void probe(const int *__restrict__ buf, int n_words, int tag, int iters,
                                             unsigned long long *__restrict__ bad /* [0] scalar, [1] vector */) {
    unsigned long long bs = 0, bv = 0;
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    unsigned idx = (unsigned)wave * 977u;
    for (int k = 0; k < iters; ++k) {
        idx = idx * 1664525u + 1013904223u;
        const unsigned base = __builtin_amdgcn_readfirstlane(idx % (unsigned)(n_words - 64));   // wave-uniform -> s_load
        const int s = buf[base];
        const int v = buf[base + lane];
        bs += (s != tag);
        bv += (v != tag);
    }
    if (bs) atomicAdd(&bad[0], bs);
    if (bv) atomicAdd(&bad[1], bv);
}

__global__ void fill(int *buf, int n, int tag) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) buf[i] = tag;
}

int main(int argc, char **argv) {
    const int tag = argc > 1 ? atoi(argv[1]) : 1;
    const double secs = argc > 2 ? atof(argv[2]) : 10.0;
    const int n = argc > 3 ? atoi(argv[3]) : (1 << 16);        // words: small, so the lines stay in the caches
    int *buf;
    unsigned long long *bad;
    if (hipMalloc(&buf, (size_t)n * 4) != hipSuccess || hipMalloc(&bad, 16) != hipSuccess) return 2;
    (void)hipMemset(bad, 0, 16);
    fill<<<256, 256>>>(buf, n, tag);
    (void)hipDeviceSynchronize();
    printf("tag %d: buffer at %p (%d words)\n", tag, (void *)buf, n);
    fflush(stdout);
    const auto t0 = std::chrono::steady_clock::now();
    long launches = 0;
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs) {
        for (int i = 0; i < 50; ++i) probe<<<2048, 256>>>(buf, n, tag, 2000, bad);
        (void)hipDeviceSynchronize();
        launches += 50;
    }
    unsigned long long h[2];
    (void)hipMemcpy(h, bad, 16, hipMemcpyDeviceToHost);
    printf("tag %d: %ld launches, words that were not the tag: scalar loads %llu, vector loads %llu (of %.3g each)\n", tag, launches,
           h[0], h[1], (double)launches * 2048 * 4 * 2000);
    return 0;
}
