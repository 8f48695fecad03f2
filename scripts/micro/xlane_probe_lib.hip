// Shared-library form of the micro-reproducer (diagnostic; loaded by scripts/diag/two_stream_race.py through ctypes)
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


__global__ __launch_bounds__(256) void v_pkfma(unsigned long long *count, int iters, unsigned seed) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const unsigned gl = blockIdx.x * 256 + threadIdx.x;
    unsigned long long bad = 0;
    for (int i = 0; i < iters; ++i) {
        const unsigned h = mix(gl, seed + i);
        f32x2 w = {(float)(h & 0xff) * 0.01f, (float)((h >> 8) & 0xff) * 0.01f}, t = {(float)((h >> 16) & 0xff) - 100.f, (float)(h >> 24) * 0.5f};
        f32x2 acc = {0.25f, -0.5f};
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(w), "v"(t));
        const float e0 = __builtin_fmaf(w[0], t[0], 0.25f), e1 = __builtin_fmaf(w[1], t[1], -0.5f);
        bad += (__float_as_uint(acc[0]) != __float_as_uint(e0)) + (__float_as_uint(acc[1]) != __float_as_uint(e1));
    }
    if (bad) atomicAdd(&count[4], bad);
}
__global__ __launch_bounds__(256) void v_div(unsigned long long *count, int iters, unsigned seed) {
    const unsigned gl = blockIdx.x * 256 + threadIdx.x;
    unsigned long long bad = 0;
    for (int i = 0; i < iters; ++i) {
        const unsigned h = mix(gl, seed + i);
        float a = (float)(h & 0xffff) - 30000.0f, b = (float)((h >> 16) | 1u);
        float a2 = a, b2 = b;
        asm volatile("" : "+v"(a), "+v"(b));
        asm volatile("" : "+v"(a2), "+v"(b2));
        const float q = a / b, q2 = a2 / b2;                       // v_div_scale / v_div_fmas / v_div_fixup twice
        bad += (__float_as_uint(q) != __float_as_uint(q2)) + (fabsf(q * b - a) > 1e-3f * fabsf(a) + 1e-3f);
    }
    if (bad) atomicAdd(&count[5], bad);
}
__global__ __launch_bounds__(256) void v_addr64(unsigned long long *count, int iters, unsigned seed) {
    const unsigned gl = blockIdx.x * 256 + threadIdx.x;
    unsigned long long bad = 0;
    for (int i = 0; i < iters; ++i) {
        const unsigned h = mix(gl, seed + i);
        unsigned long long base = 0x7f0000000000ull + ((unsigned long long)h << 4);
        long long idx = (int)(mix(h, 5u) >> 3);
        asm volatile("" : "+v"(base), "+v"(idx));
        const unsigned long long p = base + ((unsigned long long)idx << 2);            // v_lshl_add_u64
        const unsigned long long m = (unsigned long long)(unsigned)h * (unsigned)(h >> 7) + base;   // v_mad_u64_u32
        unsigned lo = (unsigned)base, hi = (unsigned)(base >> 32);
        const unsigned sl = (unsigned)idx << 2, sh = (unsigned)((unsigned long long)idx >> 30);
        const unsigned rl = lo + sl, rh = hi + sh + (rl < lo);
        const unsigned long long pm = (unsigned long long)(unsigned)h * (unsigned long long)(unsigned)(h >> 7);
        const unsigned ml = lo + (unsigned)pm, mh = hi + (unsigned)(pm >> 32) + (ml < lo);
        bad += (p != (((unsigned long long)rh << 32) | rl)) + (m != (((unsigned long long)mh << 32) | ml));
    }
    if (bad) atomicAdd(&count[6], bad);
}
// a disturber shaped like linear_fwd_b3_kernel: 60 KB of dynamic LDS holding three bf16 planes, ds_read_b128 B operands, six
// bf16 MFMAs per step, f32 -> bf16 splits of the A operand in registers, global loads of A and global stores of the result.
// `skip` bits remove one ingredient each: 1 MFMA (v_pk adds instead), 2 LDS reads, 4 global stores, 8 f32->bf16 splits,
// 16 global loads
__global__ __launch_bounds__(256) void d_b3like(const float *__restrict__ x, float *__restrict__ y, int rows, int iters, int skip) {
    extern __shared__ __attribute__((aligned(16))) __bf16 wb[];     // [3][96][104]
    for (int i = threadIdx.x; i < 3 * 96 * 104; i += 256) wb[i] = (__bf16)(0.001f * (i % 97));
    __syncthreads();
    const int lane = threadIdx.x & 63, n = lane & 15, kb = lane >> 4;
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, nw = gridDim.x * 4;
    for (int it = 0; it < iters; ++it) {
        for (int wt = wave; wt * 16 < rows; wt += nw) {
            f32x4 acc[6] = {};
            for (int ks = 0; ks < 3; ++ks) {
                const float *xb = x + (size_t)(wt * 16 + n) * 96 + 32 * ks + 8 * kb;
                float4 lo = make_float4(0.1f * lane, 0.2f, 0.3f * wt, 0.4f), hi = make_float4(0.5f, 0.6f * ks, 0.7f, 0.8f);
                if (!(skip & 16)) { lo = *(const float4 *)xb; hi = *(const float4 *)(xb + 4); }
                const float xr[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                bf16x8 a1, a2, a3;
                if (!(skip & 8)) {
                    for (int j = 0; j < 8; ++j) {
                        const __bf16 b1 = (__bf16)xr[j];
                        const float r1 = xr[j] - (float)b1;
                        const __bf16 b2 = (__bf16)r1;
                        a1[j] = b1; a2[j] = b2; a3[j] = (__bf16)(r1 - (float)b2);
                    }
                } else {
                    union { float f[4]; bf16x8 v; } u1, u2, u3;
                    for (int j = 0; j < 4; ++j) { u1.f[j] = xr[j]; u2.f[j] = xr[j + 4]; u3.f[j] = xr[j] + xr[j + 4]; }
                    a1 = u1.v; a2 = u2.v; a3 = u3.v;
                }
                for (int t = 0; t < 6; ++t) {
                    bf16x8 b1 = a1, b2 = a2, b3 = a3;
                    if (!(skip & 2)) {
                        const __bf16 *bp = wb + (size_t)(16 * t + n) * 104 + 32 * ks + 8 * kb;
                        b1 = *(const bf16x8 *)bp; b2 = *(const bf16x8 *)(bp + 96 * 104); b3 = *(const bf16x8 *)(bp + 2 * 96 * 104);
                    }
                    if (!(skip & 1)) {
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3, b1, acc[t], 0, 0, 0);
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b3, acc[t], 0, 0, 0);
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, b2, acc[t], 0, 0, 0);
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, b1, acc[t], 0, 0, 0);
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b2, acc[t], 0, 0, 0);
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b1, acc[t], 0, 0, 0);
                    } else {
                        union { bf16x8 v; f32x4 f; } p, q, r;
                        p.v = b1; q.v = b2; r.v = b3;
                        for (int rep = 0; rep < 6; ++rep) acc[t] = acc[t] * 1.0001f + p.f * q.f + r.f;
                    }
                }
            }
            if (!(skip & 4)) {
                for (int t = 0; t < 6; ++t)
                    for (int j = 0; j < 4; ++j) y[(size_t)(wt * 16 + 4 * kb + j) * 96 + 16 * t + n] = acc[t][j];
            } else {
                float s = 0.f;
                for (int t = 0; t < 6; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
                if (s == 12345.678f) y[0] = s;
            }
        }
    }
}

// packed-FP32 forms, slots 9..16 of count[] (the form that went wrong in the MSDA bilinear setup is pkmul_swz_fresh: two v_cvt_f32_i32 into a register
// pair, then v_pk_mul_f32 with op_sel:[0,1] op_sel_hi:[1,0]).  slots 9..13 of count[]
template <int FORM>
__global__ __launch_bounds__(256) void v_pkforms(unsigned long long *count, int iters, unsigned seed, const int2 *__restrict__ tab) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const unsigned gl = blockIdx.x * 256 + threadIdx.x;
    unsigned long long bad = 0;
    const unsigned h0 = mix(gl, seed ^ 0x55u);
    f32x2 bold = {(float)(int)(1 + (h0 & 0xff)), (float)(int)(1 + ((h0 >> 8) & 0xff))};
    asm volatile("" : "+v"(bold));
    for (int i = 0; i < iters; ++i) {
        const unsigned h = mix(gl, seed + i);
        const int2 hw = tab[h & 1023];                                   // (H, W) pairs like the level table, from memory
        int hl = hw.x, wl = hw.y;
        f32x2 a = {(float)(h & 0xffff) * (1.0f / 65536.0f), (float)(h >> 16) * (1.0f / 65536.0f)};
        asm volatile("" : "+v"(a));
        f32x2 r;
        float e0, e1;
        if constexpr (FORM == 0) {            // swizzled, operands fresh from v_cvt_f32_i32
            f32x2 b = {(float)hl, (float)wl};
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(r) : "v"(a), "v"(b));
            int h2 = hl, w2 = wl; float ax = a[0], ay = a[1];
            asm volatile("" : "+v"(h2), "+v"(w2), "+v"(ax), "+v"(ay));
            e0 = ax * (float)w2; e1 = ay * (float)h2;
        } else if constexpr (FORM == 1) {     // not swizzled, operands fresh from v_cvt_f32_i32
            f32x2 b = {(float)hl, (float)wl};
            asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
            int h2 = hl, w2 = wl; float ax = a[0], ay = a[1];
            asm volatile("" : "+v"(h2), "+v"(w2), "+v"(ax), "+v"(ay));
            e0 = ax * (float)h2; e1 = ay * (float)w2;
        } else if constexpr (FORM == 2) {     // swizzled, operands written long ago
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(r) : "v"(a), "v"(bold));
            float ax = a[0], ay = a[1], b0 = bold[0], b1 = bold[1];
            asm volatile("" : "+v"(b0), "+v"(b1), "+v"(ax), "+v"(ay));
            e0 = ax * b1; e1 = ay * b0;
        } else if constexpr (FORM == 3) {     // v_pk_fma_f32 with a broadcast multiplier (op_sel_hi 0 on src0)
            f32x2 acc = {0.25f, -0.5f};
            f32x2 w = {(float)wl * 0.01f, 123.0f};
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(w), "v"(a));
            r = acc;
            float wx = w[0], ax = a[0], ay = a[1];
            asm volatile("" : "+v"(wx), "+v"(ax), "+v"(ay));
            e0 = __builtin_fmaf(wx, ax, 0.25f); e1 = __builtin_fmaf(wx, ay, -0.5f);
        } else if constexpr (FORM >= 5) {     // the other low-half-broadcast forms the library contains
            f32x2 acc = {0.25f, -0.5f};
            f32x2 w = {(float)wl * 0.01f, 123.0f};
            if constexpr (FORM == 5) {        // multiplier broadcast from src1: op_sel_hi:[1,0,1]
                asm volatile("v_pk_fma_f32 %0, %2, %1, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(w), "v"(a));
            } else if constexpr (FORM == 6) { // src0 broadcast, addend = inline 0: op_sel_hi:[0,1,0]
                asm volatile("v_pk_fma_f32 %0, %1, %2, 0 op_sel_hi:[0,1,0]" : "=v"(acc) : "v"(w), "v"(a));
            } else {                          // src1 broadcast, addend = inline 0: op_sel_hi:[1,0,0]
                asm volatile("v_pk_fma_f32 %0, %2, %1, 0 op_sel_hi:[1,0,0]" : "=v"(acc) : "v"(w), "v"(a));
            }
            r = acc;
            float wx = w[0], ax = a[0], ay = a[1];
            asm volatile("" : "+v"(wx), "+v"(ax), "+v"(ay));
            if constexpr (FORM == 5) { e0 = __builtin_fmaf(ax, wx, 0.25f); e1 = __builtin_fmaf(ay, wx, -0.5f); }
            else if constexpr (FORM == 6) { e0 = __builtin_fmaf(wx, ax, 0.0f); e1 = __builtin_fmaf(wx, ay, 0.0f); }
            else { e0 = __builtin_fmaf(ax, wx, 0.0f); e1 = __builtin_fmaf(ay, wx, 0.0f); }
        } else {                              // v_pk_add_f32 with an inline constant
            f32x2 b = {(float)hl, (float)wl};
            f32x2 t = a * b;
            asm volatile("v_pk_add_f32 %0, %1, -0.5 op_sel_hi:[1,0]" : "=v"(r) : "v"(t));
            float t0 = t[0], t1 = t[1];
            asm volatile("" : "+v"(t0), "+v"(t1));
            e0 = t0 - 0.5f; e1 = t1 - 0.5f;
        }
        bad += (__float_as_uint(r[0]) != __float_as_uint(e0)) + (__float_as_uint(r[1]) != __float_as_uint(e1));
    }
    if (bad) atomicAdd(&count[FORM <= 4 ? 9 + FORM : 9 + FORM], bad);
}

// many gathers in flight per wave (the msda kernels keep up to 64 dwordx4 loads outstanding): NL independent loads issued
// back to back, checked afterwards
template <int NL>
__global__ __launch_bounds__(256) void v_gather_deep(unsigned long long *count, const uint4 *__restrict__ table, unsigned n_rows,
                                                     int iters, unsigned seed, int slot) {
    const unsigned gl = blockIdx.x * 256 + threadIdx.x;
    unsigned long long bad = 0;
    for (int i = 0; i < iters; ++i) {
        uint4 v[NL];
        unsigned r[NL];
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            r[j] = mix(gl >> 2, seed + i * NL + j) % n_rows;
            v[j] = table[(size_t)r[j] * 4 + (gl & 3u)];
        }
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            const unsigned e = mix(r[j] * 4 + (gl & 3u), 77u);
            bad += (v[j].x != e) + (v[j].y != e + 1) + (v[j].z != e + 2) + (v[j].w != e + 3);
        }
    }
    if (bad) atomicAdd(&count[slot], bad);
}

extern "C" int probe_victims(void *stream, unsigned long long *count, unsigned seed, const void *table, unsigned n_rows) {
    hipStream_t sb = (hipStream_t)stream;
    v_dpp<<<4096, 256, 0, sb>>>(count, 400, seed);
    v_bperm<<<4096, 256, 0, sb>>>(count, 400, seed);
    v_gather<<<4096, 256, 0, sb>>>(count, (const uint4 *)table, n_rows, 200, seed);
    v_trans<<<4096, 256, 0, sb>>>(count, 400, seed);
    v_pkfma<<<4096, 256, 0, sb>>>(count, 400, seed);
    v_div<<<4096, 256, 0, sb>>>(count, 200, seed);
    v_addr64<<<4096, 256, 0, sb>>>(count, 400, seed);
    v_gather_deep<16><<<8192, 256, 0, sb>>>(count, (const uint4 *)table, n_rows, 8, seed, 7);
    v_gather_deep<32><<<8192, 256, 0, sb>>>(count, (const uint4 *)table, n_rows, 4, seed, 8);
    v_pkforms<0><<<4096, 256, 0, sb>>>(count, 400, seed, (const int2 *)table);
    v_pkforms<1><<<4096, 256, 0, sb>>>(count, 400, seed, (const int2 *)table);
    v_pkforms<2><<<4096, 256, 0, sb>>>(count, 400, seed, (const int2 *)table);
    v_pkforms<3><<<4096, 256, 0, sb>>>(count, 400, seed, (const int2 *)table);
    v_pkforms<4><<<4096, 256, 0, sb>>>(count, 400, seed, (const int2 *)table);
    v_pkforms<5><<<4096, 256, 0, sb>>>(count, 400, seed, (const int2 *)table);
    v_pkforms<6><<<4096, 256, 0, sb>>>(count, 400, seed, (const int2 *)table);
    v_pkforms<7><<<4096, 256, 0, sb>>>(count, 400, seed, (const int2 *)table);
    return (int)hipGetLastError();
}
extern "C" int probe_fill_table(void *stream, void *table, unsigned n_words4) {
    fill_table<<<1024, 256, 0, (hipStream_t)stream>>>((uint4 *)table, n_words4);
    return (int)hipGetLastError();
}
extern "C" int probe_disturber(int kind, void *stream, float *sink, int iters, const float *x, float *y, int rows) {
    hipStream_t sa = (hipStream_t)stream;
    switch (kind) {
        case 1: d_mfma_bf16<<<512, 256, 0, sa>>>(sink, iters); break;
        case 2: d_mfma_f32<<<512, 256, 0, sa>>>(sink, iters); break;
        case 3: d_cvt<<<512, 256, 0, sa>>>(sink, iters); break;
        case 4: d_lds<<<512, 256, 0, sa>>>(sink, iters); break;
        case 5: d_b3like<<<512, 256, 3 * 96 * 104 * 2, sa>>>(x, y, rows, iters & 0xff, iters >> 8); break;
        default: break;
    }
    return (int)hipGetLastError();
}
