// Micro-benchmark (round 5): the MSDA grad_value scatter as a TILE GEMM on the bf16 matrix pipe — the INNER LOOP ONLY,
// on synthetic entries that are already grouped by output tile and perfectly balanced (no sort, no list building, no tails).
//   g_value tile [16 px (2 rows x 8) x 16 ch] += A^T[16 px x 32 entries] . B[32 entries x 16 ch]     per k-step
//   A[m][e] = bilinear hat weight of entry e on pixel m (x attention weight), built in registers from the entry's 16-byte record
//   B[e][n] = g_out row of the entry (a 64-byte gather from a (query, head) table in random order, as in the real list)
//   exact three-way bf16 split of both operands, six v_mfma_f32_16x16x32_bf16 per step, one non-atomic tile store at the end.
// Sizes = one hw-plane band launch of the training iteration: 30 M list entries (profiles/r5_c_train_bwd_pmc.txt), each touching
// on average ~2.2 such tiles (a bilinear footprint straddles tile borders) -> 66 M tile incidences; today's band kernel: 344 us.
// build: hipcc --offload-arch=gfx950 -O3 -o scripts/micro/band_tile_gemm scripts/micro/band_tile_gemm.hip ; run: ./band_tile_gemm
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstring>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split3(const float (&x)[8], bf16x8 &a1, bf16x8 &a2, bf16x8 &a3) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const __bf16 b1 = (__bf16)x[j];
        const float r1 = x[j] - (float)b1;
        const __bf16 b2 = (__bf16)r1;
        a1[j] = b1; a2[j] = b2; a3[j] = (__bf16)(r1 - (float)b2);
    }
}

// one wave per tile; `steps` k-steps of 32 entries each
template <bool WITH_B_GATHER>
__global__ __launch_bounds__(256) void tile_gemm(const float4 *__restrict__ recs, const int *__restrict__ rows,
                                                 const float *__restrict__ g_out, float *__restrict__ g_value, int steps) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long tile = (long long)blockIdx.x * 4 + wave;
    const int m = lane & 15, kb = lane >> 4;
    const int py = m >> 3, px = m & 7;                       // this lane's pixel of the 2 x 8 tile (A operand row)
    const float4 *rp = recs + tile * steps * 32 + 8 * kb;
    const int *ip = rows + tile * steps * 32 + 8 * kb;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < steps; ++s) {
        float a[8], b[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float4 r = rp[s * 32 + j];                 // (lh, lw, aw, packed dy / dx): the 16 lanes of a k-block read the same record
            const int pk = __float_as_int(r.w);
            const int dy = (pk >> 16) - 1, dx = (pk & 0xffff) - 1;             // low corner relative to the tile: dy in -1..1, dx in -1..7
            const float wy = (dy == py) ? 1.0f - r.x : ((dy + 1 == py) ? r.x : 0.0f);
            const float wx = (dx == px) ? 1.0f - r.y : ((dx + 1 == px) ? r.y : 0.0f);
            a[j] = wy * wx * r.z;
            if constexpr (WITH_B_GATHER) b[j] = g_out[(size_t)ip[s * 32 + j] * 16 + m];
            else b[j] = r.z + (float)m;
        }
        bf16x8 a1, a2, a3, b1, b2, b3;
        split3(a, a1, a2, a3);
        split3(b, b1, b2, b3);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3, b1, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b3, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, b2, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, b1, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b2, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b1, acc, 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) g_value[(tile * 16 + 4 * kb + j) * 16 + m] = acc[j];
}

int main() {
    const long long n_tiles = 57376;                 // 6 cameras x 6 heads x 25 500 pixels / 16
    const int steps = 36;                            // 1 152 incidences per tile -> 66.1 M incidences = 30 M entries x 2.2
    const long long n_inc = n_tiles * steps * 32;
    const int n_rows = 66049 * 6;                    // (query, head) rows of g_out
    std::vector<float4> h_rec(n_inc);
    std::vector<int> h_row(n_inc);
    unsigned st = 12345u;
    auto rnd = [&]() { st = st * 1664525u + 1013904223u; return st >> 8; };
    for (long long i = 0; i < n_inc; ++i) {
        const int dy = (int)(rnd() % 3), dx = (int)(rnd() % 9);
        float pk;
        const int v = (dy << 16) | dx;
        std::memcpy(&pk, &v, 4);
        h_rec[i] = make_float4((rnd() & 1023) / 1024.0f, (rnd() & 1023) / 1024.0f, (rnd() & 1023) / 4096.0f, pk);
        h_row[i] = (int)(rnd() % (unsigned)n_rows);          // list order is query order per band, but per TILE the rows are scattered
    }
    float4 *d_rec; int *d_row; float *d_go, *d_gv;
    (void)hipMalloc(&d_rec, n_inc * 16); (void)hipMalloc(&d_row, n_inc * 4);
    (void)hipMalloc(&d_go, (size_t)n_rows * 64); (void)hipMalloc(&d_gv, n_tiles * 16 * 64);
    (void)hipMemcpy(d_rec, h_rec.data(), n_inc * 16, hipMemcpyHostToDevice);
    (void)hipMemcpy(d_row, h_row.data(), n_inc * 4, hipMemcpyHostToDevice);
    (void)hipMemset(d_go, 0x3c, (size_t)n_rows * 64);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int variant = 0; variant < 2; ++variant) {
        auto launch = [&]() {
            if (variant == 0) tile_gemm<true><<<(unsigned)(n_tiles / 4), 256>>>(d_rec, d_row, d_go, d_gv, steps);
            else tile_gemm<false><<<(unsigned)(n_tiles / 4), 256>>>(d_rec, d_row, d_go, d_gv, steps);
        };
        for (int i = 0; i < 3; ++i) launch();
        (void)hipEventRecord(e0);
        for (int i = 0; i < 20; ++i) launch();
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%-46s %8.1f us per launch   (%.1f M tile incidences, %.2f ns each; band kernel today: 344 us for the 30 M entries behind them)\n",
               variant == 0 ? "tile GEMM inner loop, g_out rows gathered" : "tile GEMM inner loop, B operand synthetic (no gather)",
               ms / 20 * 1e3, n_inc / 1e6, ms / 20 * 1e6 / n_inc);
    }
    return 0;
}
