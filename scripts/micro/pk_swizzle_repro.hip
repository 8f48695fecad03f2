// Stand-alone reproducer (no dependency on this repository's library, no Python): half-swapping packed-FP32 instructions beside
// bf16 MFMA waves on gfx950.  One process, two HIP streams: stream A loops a disturber kernel, stream B launches the victim
// kernels of xlane_probe_lib.hip, which compare every result bit for bit with the same value computed by unpacked instructions.
//   build: hipcc --offload-arch=gfx950 -O3 -o scripts/micro/pk_swizzle_repro scripts/micro/pk_swizzle_repro.hip
//   run:   scripts/micro/pk_swizzle_repro [seconds per disturber, default 6]
// Output (MI355X, ROCm 7.2, profiles/r5_b_packed_fp32_mfma.txt): only `v_pk_mul_f32 ... op_sel:[0,1] op_sel_hi:[1,0]` ever
// differs, and only while a kernel issuing v_mfma_f32_16x16x32_bf16 runs beside it.  With THESE synthetic disturbers the event is
// rare (16 wrong results of 1e11 in two sessions out of three, 0 in the third); beside the library's real bf16x3 GEMM kernel
// (scripts/diag/two_stream_race.py with DIAG_MICRO_VICTIMS=1 DIAG_DISTURB=bricks.linear_fwd) the same victims show 15 247.
#include "xlane_probe_lib.hip"
#include <vector>

int main(int argc, char **argv) {
    const double secs = argc > 1 ? atof(argv[1]) : 6.0;
    hipStream_t sa, sb;
    (void)hipStreamCreate(&sa);
    (void)hipStreamCreate(&sb);
    unsigned long long *count;
    float *sink, *x, *y;
    uint4 *table;
    const unsigned n_rows = 1u << 20;
    const int rows = 78896;
    (void)hipMalloc(&count, 24 * 8);
    (void)hipMalloc(&sink, 64);
    (void)hipMalloc(&table, (size_t)n_rows * 64);
    (void)hipMalloc(&x, (size_t)rows * 96 * 4);
    (void)hipMalloc(&y, (size_t)rows * 96 * 4);
    (void)hipMemset(x, 0x3c, (size_t)rows * 96 * 4);
    probe_fill_table(nullptr, table, n_rows * 4);
    (void)hipDeviceSynchronize();
    const char *names[17] = {"dpp", "ds_bpermute/ds_permute", "gather dwordx4", "exp/rcp", "v_pk_fma_f32 plain", "f32 division",
                             "64-bit address math", "16 gathers in flight", "32 gathers in flight",
                             "v_pk_mul_f32 op_sel:[0,1] op_sel_hi:[1,0] (fresh cvt)", "v_pk_mul_f32 straight (fresh cvt)",
                             "v_pk_mul_f32 op_sel:[0,1] op_sel_hi:[1,0] (old regs)", "v_pk_fma_f32 op_sel_hi:[0,1,1] (broadcast)",
                             "v_pk_add_f32 inline constant", "v_pk_fma_f32 op_sel_hi:[1,0,1]", "v_pk_fma_f32 .., 0 op_sel_hi:[0,1,0]",
                             "v_pk_fma_f32 .., 0 op_sel_hi:[1,0,0]"};
    struct Dist { const char *name; int kind, iters; };
    const Dist dists[] = {{"none", 0, 0}, {"v_mfma_f32_16x16x32_bf16 loop", 1, 40000}, {"v_mfma_f32_16x16x4_f32 loop", 2, 20000},
                          {"bf16x3 GEMM-shaped kernel (LDS planes + bf16 MFMA + global loads / stores), 30 passes", 5, 30}};
    for (const Dist &d : dists) {
        (void)hipMemset(count, 0, 24 * 8);
        const auto t0 = std::chrono::steady_clock::now();
        long rounds = 0;
        while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs) {
            for (int r = 0; r < 4; ++r) {
                if (d.kind) probe_disturber(d.kind, sa, sink, d.iters, x, y, rows);
                probe_victims(sb, count, (unsigned)(rounds * 4 + r) * 1000003u, table, n_rows);
            }
            (void)hipDeviceSynchronize();
            ++rounds;
        }
        unsigned long long h[24];
        (void)hipMemcpy(h, count, 24 * 8, hipMemcpyDeviceToHost);
        printf("disturber: %s   (%ld victim launches of each kind)\n", d.name, rounds * 4);
        for (int i = 0; i < 17; ++i)
            if (h[i] || i >= 9) printf("    %-58s wrong results: %llu\n", names[i], h[i]);
    }
    return 0;
}
