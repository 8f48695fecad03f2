// Micro-reproducer (diagnostic): does a kernel that runs bf16 matrix instructions on one HIP stream change the results of
// cross-lane / gather instructions of a kernel running AT THE SAME TIME on another stream of the same process?
// Victim kernels each exercise one instruction class in a loop and compare with a value computed without that class
// (analytic expectation), counting mismatches; they are launched back-to-back on stream B while a disturber kernel loops on
// stream A.  usage: mfma_xlane_interference <disturber 0..4> <seconds>
//   disturbers: 0 none, 1 v_mfma_f32_16x16x32_bf16 loop, 2 v_mfma_f32_16x16x4_f32 loop, 3 f32 -> bf16 conversion loop,
//               4 LDS read loop (ds_read_b128)
// build: hipcc --offload-arch=gfx950 -O3 -o scripts/micro/mfma_xlane_interference scripts/micro/mfma_xlane_interference.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned mix(unsigned a, unsigned b) {
    unsigned x = a * 2654435761u ^ (b + 0x9e3779b9u + (a << 6) + (a >> 2));
    x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
    return x;
}

// ---- disturbers (one 256-thread block per CU-ish slot, modest registers so that victims fit beside them) ----
__global__ __launch_bounds__(256) void d_mfma_bf16(float *sink, int iters) {
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.001f * (threadIdx.x + j)); b[j] = (__bf16)(0.002f * (threadIdx.x ^ j)); }
    f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    for (int i = 0; i < iters; ++i) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, a, acc1, 0, 0, 0);
    }
    if (acc0[0] + acc1[1] == 12345.678f) sink[0] = acc0[0];
}
__global__ __launch_bounds__(256) void d_mfma_f32(float *sink, int iters) {
    float a = 0.001f * threadIdx.x, b = 0.002f * (threadIdx.x ^ 5);
    f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    for (int i = 0; i < iters; ++i) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, acc1, 0, 0, 0);
    }
    if (acc0[0] + acc1[1] == 12345.678f) sink[0] = acc0[0];
}
__global__ __launch_bounds__(256) void d_cvt(float *sink, int iters) {
    float x = 0.37f * threadIdx.x, s = 0.f;
    for (int i = 0; i < iters; ++i) {
        const __bf16 h = (__bf16)x;
        const float r = x - (float)h;
        const __bf16 h2 = (__bf16)r;
        s += (float)h2;
        x = x * 1.0001f + 0.5f;
    }
    if (s == 12345.678f) sink[0] = s;
}
__global__ __launch_bounds__(256) void d_lds(float *sink, int iters) {
    __shared__ float4 buf[2048];
    for (int i = threadIdx.x; i < 2048; i += 256) buf[i] = make_float4(i, 1, 2, 3);
    __syncthreads();
    float s = 0.f;
    unsigned k = threadIdx.x;
    for (int i = 0; i < iters; ++i) {
        const float4 v = buf[k & 2047];
        s += v.x + v.w;
        k = k * 5 + 1;
    }
    if (s == 12345.678f) sink[0] = s;
}

// ---- victims: count[c] += lanes whose result differs from the analytic expectation ----
__global__ __launch_bounds__(256) void v_dpp(unsigned long long *count, int iters, unsigned seed) {
    const unsigned gl = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63;
    unsigned long long bad = 0;
    for (int i = 0; i < iters; ++i) {
        const unsigned x = mix(gl, seed + i);
        // quad broadcast of sub-lane 2 (the so_team_bcast form), then a row rotate-free quad permute [1,0,3,2]
        const unsigned y = (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 2 * 0x55, 0xf, 0xf, true);
        const unsigned z = (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xf, 0xf, true);
        const unsigned ey = mix((gl & ~3u) | 2u, seed + i), ez = mix(gl ^ 1u, seed + i);
        bad += (y != ey) + (z != ez);
    }
    if (bad) atomicAdd(&count[0], bad);
    (void)lane;
}
__global__ __launch_bounds__(256) void v_bperm(unsigned long long *count, int iters, unsigned seed) {
    const unsigned gl = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63;
    unsigned long long bad = 0;
    for (int i = 0; i < iters; ++i) {
        const unsigned x = mix(gl, seed + i);
        const unsigned src = (lane ^ (16u + (i & 15))) & 63u;
        const unsigned y = (unsigned)__builtin_amdgcn_ds_bpermute((int)(src << 2), (int)x);
        const unsigned ey = mix((gl & ~63u) | src, seed + i);
        const unsigned dst = (lane * 5u + 3u) & 63u;                          // a bijection of the lanes
        const unsigned z = (unsigned)__builtin_amdgcn_ds_permute((int)(dst << 2), (int)x);
        // lane l receives from the lane s with (5 s + 3) % 64 == l  ->  s = 13 (l - 3) % 64   (5 * 13 = 65 = 1 mod 64)
        const unsigned s = (13u * (lane - 3u)) & 63u;
        const unsigned ez = mix((gl & ~63u) | s, seed + i);
        bad += (y != ey) + (z != ez);
    }
    if (bad) atomicAdd(&count[1], bad);
}
__global__ __launch_bounds__(256) void v_gather(unsigned long long *count, const uint4 *__restrict__ table, unsigned n_rows,
                                                int iters, unsigned seed) {
    const unsigned gl = blockIdx.x * 256 + threadIdx.x;
    unsigned long long bad = 0;
    for (int i = 0; i < iters; ++i) {
        const unsigned r = mix(gl >> 2, seed + i) % n_rows;                 // 4 lanes fetch the 4 quarters of one 64-byte row
        const uint4 v = table[(size_t)r * 4 + (gl & 3u)];
        const unsigned e = mix(r * 4 + (gl & 3u), 77u);
        bad += (v.x != e) + (v.y != e + 1) + (v.z != e + 2) + (v.w != e + 3);
    }
    if (bad) atomicAdd(&count[2], bad);
}
__global__ __launch_bounds__(256) void v_trans(unsigned long long *count, int iters, unsigned seed) {
    const unsigned gl = blockIdx.x * 256 + threadIdx.x;
    unsigned long long bad = 0;
    for (int i = 0; i < iters; ++i) {
        const float x = (float)(mix(gl, seed + i) & 0xffff) * (1.0f / 4096.0f) - 8.0f;
        float x1 = x, x2 = x;
        asm volatile("" : "+v"(x1));
        asm volatile("" : "+v"(x2));
        const float a = __expf(x1) * __builtin_amdgcn_rcpf(1.0f + x1 * x1), b = __expf(x2) * __builtin_amdgcn_rcpf(1.0f + x2 * x2);
        bad += (__float_as_uint(a) != __float_as_uint(b));
    }
    if (bad) atomicAdd(&count[3], bad);
}
__global__ void fill_table(uint4 *t, unsigned n) {
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const unsigned e = mix(i, 77u);
        t[i] = make_uint4(e, e + 1, e + 2, e + 3);
    }
}

int main(int argc, char **argv) {
    const int dist = argc > 1 ? atoi(argv[1]) : 1;
    const double secs = argc > 2 ? atof(argv[2]) : 5.0;
    hipStream_t sa, sb;
    (void)hipStreamCreate(&sa);
    (void)hipStreamCreate(&sb);
    unsigned long long *count;
    float *sink;
    uint4 *table;
    const unsigned n_rows = 1u << 20;                      // 64 MB table
    (void)hipMalloc(&count, 64);
    (void)hipMalloc(&sink, 64);
    (void)hipMalloc(&table, (size_t)n_rows * 64);
    (void)hipMemset(count, 0, 64);
    fill_table<<<1024, 256>>>(table, n_rows * 4);
    (void)hipDeviceSynchronize();
    const auto t0 = std::chrono::steady_clock::now();
    long rounds = 0;
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs) {
        for (int r = 0; r < 8; ++r) {
            switch (dist) {     // ~1-2 ms of disturber per launch, 512 blocks: two per CU at most, the rest of the CU is free
                case 1: d_mfma_bf16<<<512, 256, 0, sa>>>(sink, 40000); break;
                case 2: d_mfma_f32<<<512, 256, 0, sa>>>(sink, 20000); break;
                case 3: d_cvt<<<512, 256, 0, sa>>>(sink, 60000); break;
                case 4: d_lds<<<512, 256, 0, sa>>>(sink, 60000); break;
                default: break;
            }
            const unsigned seed = (unsigned)(rounds * 8 + r) * 1000003u;
            v_dpp<<<4096, 256, 0, sb>>>(count, 400, seed);
            v_bperm<<<4096, 256, 0, sb>>>(count, 400, seed);
            v_gather<<<4096, 256, 0, sb>>>(count, table, n_rows, 200, seed);
            v_trans<<<4096, 256, 0, sb>>>(count, 400, seed);
        }
        (void)hipDeviceSynchronize();
        ++rounds;
    }
    unsigned long long h[4];
    (void)hipMemcpy(h, count, 32, hipMemcpyDeviceToHost);
    const char *names[] = {"none", "mfma_f32_16x16x32_bf16", "mfma_f32_16x16x4_f32", "f32->bf16 cvt", "ds_read_b128"};
    printf("disturber %-24s %5ld rounds: wrong results  dpp %llu  ds_bpermute/ds_permute %llu  gather(dwordx4) %llu  exp/rcp %llu\n",
           names[dist], rounds, h[0], h[1], h[2], h[3]);
    return 0;
}
