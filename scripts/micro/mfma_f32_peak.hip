// dev micro-benchmark: what f32 MFMA rate does the chip sustain on RANDOM data, as a function of resident waves per SIMD,
// instruction shape, an LDS operand stream and an epilogue of stores?  (hipcc --offload-arch=gfx950 -O3 mfma_f32_peak.hip)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>   // 0: 16x16x4 regs only, 1: 32x32x2 regs only, 2: 16x16x4 + ds_read_b128 stream, 3: mode 2 + 24 stores per 144 MFMAs
__global__ __launch_bounds__(256) void k(const float *__restrict__ in, float *__restrict__ out, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[96 * 100];
    const int lane = threadIdx.x & 63;
    for (int e = threadIdx.x; e < 96 * 100; e += 256) lds[e] = in[e % 4096];
    __syncthreads();
    float a[24];
#pragma unroll
    for (int j = 0; j < 24; ++j) a[j] = in[(threadIdx.x * 24 + j) % 4096];
    float sink = 0.0f;
    if (MODE == 1) {
        f32x16 acc0 = {0}, acc1 = {0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 24; j += 2) {      // 72 MFMAs of 32x32x2 = the flops of 144 of 16x16x4
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], a[(j + r + 1) % 24], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j + 1], a[(j + r + 2) % 24], acc1, 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int v = 0; v < 16; ++v) sink += acc0[v] + acc1[v];
    } else {
        f32x4 acc[6] = {{0}, {0}, {0}, {0}, {0}, {0}};
        const float *bbase = lds + (lane & 15) * 100 + (lane >> 4) * 24;
        for (int it = 0; it < iters; ++it) {
            float4 b0, b1;
            if (MODE >= 2) { b0 = *(const float4 *)bbase; b1 = *(const float4 *)(bbase + 1600); }
            else { b0 = make_float4(a[1], a[2], a[3], a[4]); b1 = make_float4(a[5], a[6], a[7], a[8]); }
#pragma unroll
            for (int s = 0; s < 18; ++s) {
                const int tp = s / 6, q = s % 6;
                float4 nb0 = b0, nb1 = b1;
                if (MODE >= 2 && s + 1 < 18) {
                    nb0 = *(const float4 *)(bbase + 3200 * ((s + 1) / 6) + 4 * ((s + 1) % 6));
                    nb1 = *(const float4 *)(bbase + 3200 * ((s + 1) / 6) + 1600 + 4 * ((s + 1) % 6));
                }
                __builtin_amdgcn_sched_barrier(0);
                acc[2 * tp] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4 * q], b0.x, acc[2 * tp], 0, 0, 0);
                acc[2 * tp + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4 * q], b1.x, acc[2 * tp + 1], 0, 0, 0);
                acc[2 * tp] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4 * q + 1], b0.y, acc[2 * tp], 0, 0, 0);
                acc[2 * tp + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4 * q + 1], b1.y, acc[2 * tp + 1], 0, 0, 0);
                acc[2 * tp] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4 * q + 2], b0.z, acc[2 * tp], 0, 0, 0);
                acc[2 * tp + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4 * q + 2], b1.z, acc[2 * tp + 1], 0, 0, 0);
                acc[2 * tp] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4 * q + 3], b0.w, acc[2 * tp], 0, 0, 0);
                acc[2 * tp + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4 * q + 3], b1.w, acc[2 * tp + 1], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                b0 = nb0; b1 = nb1;
            }
            if (MODE == 4) {     // the same 6 KB as six dwordx4 stores: lane (row = lane & 15, kq) writes columns 16 t + 4 kq .. + 3
                float *o = out + ((size_t)(blockIdx.x * 4 + (threadIdx.x >> 6)) * iters + it) % 200000 * 1536 + (lane & 15) * 96 + (lane >> 4) * 4;
#pragma unroll
                for (int t = 0; t < 6; ++t) {
                    *(float4 *)(o + 16 * t) = make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]);
                    acc[t][0] = acc[t][1] = acc[t][2] = acc[t][3] = 0.0f;
                }
            }
            if (MODE == 5) {     // through LDS: full 128-byte lines per store instruction (8 lines per dwordx4 store)
                float *stage = lds + 96 * 100 - 0;   // (aliases nothing that is read again in this micro-benchmark)
                (void)stage;
            }
            if (MODE == 3) {
                float *o = out + ((size_t)(blockIdx.x * 4 + (threadIdx.x >> 6)) * iters + it) % 200000 * 1536 + (lane >> 4) * 4 * 96 + (lane & 15);
#pragma unroll
                for (int t = 0; t < 6; ++t)
#pragma unroll
                    for (int j = 0; j < 4; ++j) { o[j * 96 + 16 * t] = acc[t][j]; acc[t][j] = 0.0f; }
            }
        }
#pragma unroll
        for (int t = 0; t < 6; ++t) sink += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    }
    if (sink == 12345.678f) out[threadIdx.x] = sink;
}

template <int MODE>
void run(const char *name, const float *in, float *out, int blocks_per_cu, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * blocks_per_cu;
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, in, out, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, in, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    const double flops = (double)blocks * 4 * iters * 144 * 2048.0;      // 144 x (16x16x4x2) per iteration (mode 1: 72 x 4096)
    printf("%-44s waves/SIMD %d: %7.3f ms  %6.1f TFLOP/s\n", name, blocks_per_cu, ms, flops / ms / 1e9);
}

int main() {
    float *in, *out;
    hipMalloc(&in, 4096 * 4); hipMalloc(&out, (size_t)200000 * 1536 * 4 + 4096);
    std::vector<float> h(4096);
    srand(1);
    for (auto &v : h) v = (float)rand() / RAND_MAX - 0.5f;
    hipMemcpy(in, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    for (int w = 1; w <= 4; ++w) {
        run<0>("16x16x4, operands in registers", in, out, w, 400 / w);
        run<1>("32x32x2, operands in registers", in, out, w, 400 / w);
        run<2>("16x16x4, B from LDS (b128, double-buffered)", in, out, w, 400 / w);
        run<3>("  + 24 dword stores per 144 MFMAs", in, out, w, 400 / w);
        run<4>("  + 6 dwordx4 stores per 144 MFMAs", in, out, w, 400 / w);
    }
    // zero data: the DVFS give-back
    hipMemset(in, 0, 4096 * 4);
    run<0>("16x16x4, registers, ZERO data", in, out, 2, 200);
    run<2>("16x16x4, LDS stream, ZERO data", in, out, 2, 200);
    return 0;
}
