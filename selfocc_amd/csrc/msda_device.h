// msda_device.h — device-side building blocks of msda.hip (included inside an anonymous
// namespace by each translation unit): value addressing, the bilinear point set-up, channel-team gathers,
// inside-point compaction and the group reduce-scatter.  See the header of msda.hip for the hardware mapping.
#pragma once

struct MsdaDims {
    int bs, nv, nq, heads, L, P;
    int go_shared;   // band kernel: g_out has one row per (query, head) shared by all batch items (camera loop)
    int vs;          // camera-loop forward: floats between consecutive pixels of `value` (0 = dense: heads * D)
    int hm;          // value / grad_value layout: 0 = (bs, nv, heads, D) (mmcv), 1 = head-major (bs, heads, nv, D)
    int off_ld, lg_ld;   // floats between consecutive QUERY rows of the raw offsets / attention logits (and of their gradients).
                         // Dense tensors: heads * L * P * 2 and heads * L * P.  One merged projection row per query
                         // [heads * L * P * 2 offsets | heads * L * P logits] (round 6): 3 * heads * L * P for both.
};

// float index of point pt's raw offset pair / attention logit of (query row bq, head h); LP = L * P
SO_DEVFN size_t so_off_index(const MsdaDims &dm, long long bq, int h, int LP, int pt) {
    return (size_t)bq * dm.off_ld + ((size_t)h * LP + pt) * 2;
}
SO_DEVFN size_t so_lg_index(const MsdaDims &dm, long long bq, int h, int LP, int pt) {
    return (size_t)bq * dm.lg_ld + (size_t)h * LP + pt;
}

// Addressing of `value` / `grad_value` in either layout.  Head-major puts the D channels of horizontally adjacent
// pixels of ONE head next to each other (64-byte segments back to back), so that the two x-corners of a bilinear
// footprint usually share a 128-byte cache line and neighbouring sampling points re-use lines; in the mmcv layout a
// line holds the same pixel for two DIFFERENT heads, whose sampling points go elsewhere — half of every line fill is
// wasted (measured on the hw-plane cross-attention shape: 0.50 ms vs 0.34 ms for the same points).
SO_DEVFN int so_pix_stride(const MsdaDims &dm, int D) { return dm.hm ? D : (dm.vs ? dm.vs : dm.heads * D); }
SO_DEVFN long long so_value_base(const MsdaDims &dm, int D, long long b, int h, long long first_pix) {
    if (dm.hm) return ((b * dm.heads + h) * dm.nv + first_pix) * D;
    return (b * dm.nv + first_pix) * (long long)so_pix_stride(dm, D) + (long long)h * D;
}

struct Bilin {
    bool any;          // sample inside the (-1, H) x (-1, W) window
    int off[4];        // element offsets of the 4 corners (pixel * heads * D), clamped
    float w[4];        // bilinear weights, 0 for out-of-map corners
    float lh, lw, hh, hw;
    bool valid[4];
    int h_low, w_low;  // floor(h_im), floor(w_im) (may be -1)
};

SO_DEVFN Bilin so_bilinear_setup(float lx, float ly, int Hl, int Wl, int pix_stride) {
    Bilin r;
    const float h_im = ly * (float)Hl - 0.5f;
    const float w_im = lx * (float)Wl - 0.5f;
    r.any = (h_im > -1.0f) && (w_im > -1.0f) && (h_im < (float)Hl) && (w_im < (float)Wl);
    const float fh = floorf(h_im), fw = floorf(w_im);
    const int h_low = (int)fh, w_low = (int)fw;
    const int h_high = h_low + 1, w_high = w_low + 1;
    r.h_low = h_low; r.w_low = w_low;
    r.lh = h_im - fh; r.lw = w_im - fw;
    r.hh = 1.0f - r.lh; r.hw = 1.0f - r.lw;
    r.valid[0] = r.any && h_low >= 0 && w_low >= 0;
    r.valid[1] = r.any && h_low >= 0 && w_high <= Wl - 1;
    r.valid[2] = r.any && h_high <= Hl - 1 && w_low >= 0;
    r.valid[3] = r.any && h_high <= Hl - 1 && w_high <= Wl - 1;
    const int hl = min(max(h_low, 0), Hl - 1), hh_ = min(max(h_high, 0), Hl - 1);
    const int wl = min(max(w_low, 0), Wl - 1), wh_ = min(max(w_high, 0), Wl - 1);
    r.off[0] = (hl * Wl + wl) * pix_stride;
    r.off[1] = (hl * Wl + wh_) * pix_stride;
    r.off[2] = (hh_ * Wl + wl) * pix_stride;
    r.off[3] = (hh_ * Wl + wh_) * pix_stride;
    r.w[0] = r.valid[0] ? r.hh * r.hw : 0.0f;
    r.w[1] = r.valid[1] ? r.hh * r.lw : 0.0f;
    r.w[2] = r.valid[2] ? r.lh * r.hw : 0.0f;
    r.w[3] = r.valid[3] ? r.lh * r.lw : 0.0f;
    return r;
}

SO_DEVFN int so_level_of(int pt, int P, int L) {
    int l = 0;
    for (int k = 1; k < L; ++k) l += (pt >= k * P);
    return l;
}


// (b, q, h) of a (batch, query, head) group index.  n_groups is wave-uniform, so the common case
// takes 32-bit divisions (the 64-bit expansion costs ~120 VALU instructions per division).
SO_DEVFN void so_split_group(long long gq, long long n_groups, int nq, int heads, int &h, int &b,
                             long long &bq) {
    if (n_groups < (1LL << 31)) {
        const unsigned g = (unsigned)gq;
        const unsigned q = g / (unsigned)heads;
        h = (int)(g - q * (unsigned)heads);
        b = (int)(q / (unsigned)nq);
        bq = q;
    } else {
        bq = gq / heads;
        h = (int)(gq - bq * heads);
        b = (int)(bq / nq);
    }
}

// Head-outer group order of the fused / camera-loop kernels: group g -> head g / (bs * nq), query g % (bs * nq), so
// the groups of a block (and of an XCD's eighth of the grid) are CONSECUTIVE QUERIES OF ONE HEAD.  Neighbouring
// queries sample neighbouring pixels, and only groups of the same head can share a 64-byte corner segment in L1;
// in (query, head) order a block holds one or two queries x all heads.  Measured: camera-loop forward 0.87 -> 0.74 ms,
// fused forward 0.50 -> 0.47 ms (scripts/bench_msda.py), eval encoder -1.3 %.  gq = (b, q, h) index (tensor order).
SO_DEVFN void so_split_group_head_outer(long long g, const MsdaDims &dm, int &h, int &b, long long &bq, long long &gq) {
    const long long nbq = (long long)dm.bs * dm.nq;
    if (nbq * dm.heads < (1LL << 31)) {
        const unsigned hh = (unsigned)g / (unsigned)nbq;
        const unsigned q = (unsigned)g - hh * (unsigned)nbq;
        h = (int)hh;
        bq = q;
        b = (int)(q / (unsigned)dm.nq);
    } else {
        h = (int)(g / nbq);
        bq = g - h * nbq;
        b = (int)(bq / dm.nq);
    }
    gq = bq * dm.heads + h;
}

// XCD-aware block order.  Workgroup b is observed to run on XCD b % 8, each XCD with its own 4 MB L2.  With the
// plain order consecutive queries — which sample neighbouring pixels — are dealt round-robin to the 8 XCDs, so every
// L2 sees the whole 10-60 MB value map; here XCD x works on ONE contiguous eighth of the (query, head) groups and
// its L2 only has to hold the pixels that eighth looks at.  A bijection of [0, gridDim.x) for any grid size; the
// mapping affects speed only (placement is not guaranteed by HIP).
SO_DEVFN unsigned so_xcd_block() {
#ifdef SO_MSDA_NO_XCD_SWIZZLE
    return blockIdx.x;
#else
    const unsigned nb = gridDim.x, b = blockIdx.x;
    const unsigned x = b & 7u, k = b >> 3;
    const unsigned q = nb >> 3, r = nb & 7u;
    return x * q + (x < r ? x : r) + k;
#endif
}

// ---------------------------------------------------------------------------------------
// forward building blocks.
//
// A lane OWNS one sampling point (coalesced loc / attw / logits reads, one bilinear setup per point,
// no redundant coordinate math) but the gather is done by CHANNEL TEAMS: QL = D / 4 adjacent lanes
// walk through their QL points together, each lane fetching its own 16-byte quarter of every corner.
// One wave-level load then touches 64 / QL distinct 64-byte corner segments instead of 64 — the
// per-point layout (every lane issuing D / 4 dwordx4 loads into its private segment) is bound by the
// L1 tag rate, not by bytes (measured 0.92 ms vs 0.17 ms of TA issue time at the nuscenes_occ
// hw-plane shape).  The point record travels through the team with DPP quad permutes (no LDS).
// ---------------------------------------------------------------------------------------
constexpr int so_ilog2(int v) { return v <= 1 ? 0 : 1 + so_ilog2(v >> 1); }
// fused kernels: points a lane may own (rounds of its group): small groups take more rounds (L * P = 36 on 8 lanes x 5
// rounds keeps 90 % of the lanes busy; 16 x 3: 75 %)
constexpr int so_maxr(int logg) { return logg >= 4 ? 4 : 8; }

struct MsdaPoint {
    int off[4];   // corner element offsets relative to the (b, head) base, level start included
    float w[4];   // bilinear corner weights (0: outside the map / no point)
    float aw;     // attention weight
};

SO_DEVFN MsdaPoint so_point_none() {
    MsdaPoint p;
#pragma unroll
    for (int k = 0; k < 4; ++k) { p.off[k] = 0; p.w[k] = 0.0f; }
    p.aw = 0.0f;
    return p;
}

SO_DEVFN MsdaPoint so_point_setup(float lx, float ly, float aw, int Hl, int Wl, int level_off,
                                  int pix_stride) {
    const Bilin bl = so_bilinear_setup(lx, ly, Hl, Wl, pix_stride);
    MsdaPoint p;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        p.off[k] = bl.any ? level_off + bl.off[k] : 0;
        p.w[k] = bl.any ? bl.w[k] : 0.0f;
    }
    p.aw = bl.any ? aw : 0.0f;
    return p;
}

// x of sub-lane I of this lane's QL-lane team (QL <= 4: DPP quad permute, one VALU move)
template <int QL, int I>
SO_DEVFN int so_team_bcast(int x) {
    if constexpr (QL == 1) {
        return x;
    } else if constexpr (QL == 2) {
        return __builtin_amdgcn_update_dpp(0, x, I | (I << 2) | ((2 + I) << 4) | ((2 + I) << 6), 0xf, 0xf, true);
    } else if constexpr (QL == 4) {
        return __builtin_amdgcn_update_dpp(0, x, I * 0x55, 0xf, 0xf, true);
    } else {
        return __shfl(x, (int)(((threadIdx.x & 63) & ~(QL - 1)) | I), 64);
    }
}

template <int QL, int I>
SO_DEVFN float so_team_bcastf(float x) {
    return __int_as_float(so_team_bcast<QL, I>(__float_as_int(x)));
}

// 4 consecutive channels of `value`: float32, or bfloat16 storage (the bits, as uint16_t) widened exactly.  The bf16
// option of the fused / camera-loop entry points halves the corner segments the gathers move (64 -> 32 bytes); the
// arithmetic stays float32 on the widened values.
SO_DEVFN float4 so_ld4(const float *p) { return *(const float4 *)p; }
SO_DEVFN float4 so_ld4(const uint16_t *p) {
    const uint2 t = *(const uint2 *)p;
    return make_float4(__uint_as_float(t.x << 16), __uint_as_float(t.x & 0xffff0000u), __uint_as_float(t.y << 16),
                       __uint_as_float(t.y & 0xffff0000u));
}

// the team adds the point owned by its sub-lane I: acc[0..3] are this lane's 4 channels.
// The 4-channel FMAs are written as two explicit 2-wide ones (v_pk_fma_f32 with the scalar weight broadcast from the low half,
// op_sel_hi:[0,1,1]): the library is built WITHOUT the compiler's vectorizers (csrc/build.sh: the half-swapping op_sel forms
// they produce are not safe beside bf16 MFMA waves on gfx950), and this loop is the one place where the packed rate is worth
// having back (msda_cross_fwd 0.52 -> 0.41 ms at the shipped hw-plane size).  Only the broadcast / straight forms are used
// here; tests/test_isa_lint.py checks the built library for any other.
#ifndef SO_TEAM_PK_PLAIN
#define SO_TEAM_PK_PLAIN 0      // 2-wide FMAs also in the plain / fused kernels (so_team_gather)?  Measured neutral (fused <16,3>
                                // 0.266 vs 0.259 ms, eval frame 7.98 vs 7.97): off; the camera-loop kernels always have them (0.52 -> 0.41 ms)
#endif
template <int D, int I, bool PK = true, typename VT>
SO_DEVFN void so_team_step(const VT *vb, const MsdaPoint &mp, float (&acc)[4]) {
    constexpr int QL = D / 4;
    const float aw = so_team_bcastf<QL, I>(mp.aw);
    if constexpr (PK) {
        so_f32x2 v01 = {0.0f, 0.0f}, v23 = {0.0f, 0.0f};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int off = so_team_bcast<QL, I>(mp.off[k]);
            const float w = so_team_bcastf<QL, I>(mp.w[k]);
            const float4 t = so_ld4(vb + off);
            const so_f32x2 ww = {w, w}, t01 = {t.x, t.y}, t23 = {t.z, t.w};
            v01 = __builtin_elementwise_fma(ww, t01, v01);
            v23 = __builtin_elementwise_fma(ww, t23, v23);
        }
        const so_f32x2 aa = {aw, aw};
        so_f32x2 a01 = {acc[0], acc[1]}, a23 = {acc[2], acc[3]};
        a01 = __builtin_elementwise_fma(aa, v01, a01);
        a23 = __builtin_elementwise_fma(aa, v23, a23);
        acc[0] = a01[0]; acc[1] = a01[1]; acc[2] = a23[0]; acc[3] = a23[1];
    } else {
        float val[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int off = so_team_bcast<QL, I>(mp.off[k]);
            const float w = so_team_bcastf<QL, I>(mp.w[k]);
            const float4 t = so_ld4(vb + off);
            val[0] = fmaf(w, t.x, val[0]);
            val[1] = fmaf(w, t.y, val[1]);
            val[2] = fmaf(w, t.z, val[2]);
            val[3] = fmaf(w, t.w, val[3]);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = fmaf(aw, val[c], acc[c]);
    }
}

// Whole-wave / half-wave groups: move the points that touch the map to the front of their group, dealt round-robin over
// its teams, and return how many team steps they need.  The pillar of a zh / wz query runs across the whole scene, so a camera that
// sees the query sees only a fraction of its 48 points (the rest sample the zero padding: exactly 0); uncompacted every
// point costs a gather slot.  One ballot + 9 ds_permute per round; skipped when >= 3/4 of the lanes are inside.
template <int D, int LOGG>
SO_DEVFN int so_compact_plan(bool ins, int &dst, bool &moved) {
    // -> team steps needed; moved: the caller permutes its per-point values with ds_permute(dst, .) (and brings results
    // back with ds_bpermute(dst, .)).  Points stay inside their group (whole wave or half wave).  Wave-uniform result.
    constexpr int G = 1 << LOGG, QL = D / 4, TEAMS = G / QL;
    static_assert(LOGG == 5 || LOGG == 6, "whole-wave or half-wave groups");
    const unsigned long long m = __ballot(ins);
    const int lane = threadIdx.x & 63;
    moved = false;
    dst = 0;
    int cnt, below, most;
    if constexpr (LOGG == 6) {
        cnt = most = __popcll(m);
        below = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
    } else {
        const unsigned lo = (unsigned)m, hi = (unsigned)(m >> 32);
        const int c0 = __popc(lo), c1 = __popc(hi);
        most = max(c0, c1);
        cnt = lane < 32 ? c0 : c1;
        below = __popc((lane < 32 ? lo : hi) & ((1u << (lane & 31)) - 1u));
    }
    if (most > G - TEAMS) return QL;              // (nearly) every team step is needed anyway
    if (most == 0) return 0;
    const int gl = lane & (G - 1);
    const int k = ins ? below : cnt + (gl - below);                            // stable partition: inside points first
    dst = ((lane & ~(G - 1)) | ((k % TEAMS) * QL + k / TEAMS)) << 2;           // point k -> team k % TEAMS, sub-lane k / TEAMS
    moved = true;
    return (most + TEAMS - 1) / TEAMS;
}

template <int D, int LOGG>
SO_DEVFN int so_compact_points(MsdaPoint &mp) {
    int dst;
    bool moved;
    const int steps = so_compact_plan<D, LOGG>(mp.aw != 0.0f, dst, moved);
    if (moved) {   // wave-uniform
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            mp.off[c] = __builtin_amdgcn_ds_permute(dst, mp.off[c]);
            mp.w[c] = __int_as_float(__builtin_amdgcn_ds_permute(dst, __float_as_int(mp.w[c])));
        }
        mp.aw = __int_as_float(__builtin_amdgcn_ds_permute(dst, __float_as_int(mp.aw)));
    }
    return steps;
}

// the first `steps` team steps only (wave-uniform; after so_compact_points)
template <int D, typename VT>
SO_DEVFN void so_team_gather_steps(const VT *vb, const MsdaPoint &mp, float (&acc)[4], int steps) {
    constexpr int QL = D / 4;
    if (steps > 0) so_team_step<D, 0>(vb, mp, acc);
    if constexpr (QL > 1) { if (steps > 1) so_team_step<D, 1>(vb, mp, acc); }
    if constexpr (QL > 2) {
        if (steps > 2) so_team_step<D, 2>(vb, mp, acc);
        if (steps > 3) so_team_step<D, 3>(vb, mp, acc);
    }
    if constexpr (QL > 4) {
        if (steps > 4) so_team_step<D, 4>(vb, mp, acc);
        if (steps > 5) so_team_step<D, 5>(vb, mp, acc);
        if (steps > 6) so_team_step<D, 6>(vb, mp, acc);
        if (steps > 7) so_team_step<D, 7>(vb, mp, acc);
    }
}

template <int D, typename VT>
SO_DEVFN void so_team_gather(const VT *vb, const MsdaPoint &mp, float (&acc)[4]) {
    constexpr int QL = D / 4;
    so_team_step<D, 0, SO_TEAM_PK_PLAIN != 0>(vb, mp, acc);
    if constexpr (QL > 1) so_team_step<D, 1, SO_TEAM_PK_PLAIN != 0>(vb, mp, acc);
    if constexpr (QL > 2) {
        so_team_step<D, 2, SO_TEAM_PK_PLAIN != 0>(vb, mp, acc);
        so_team_step<D, 3, SO_TEAM_PK_PLAIN != 0>(vb, mp, acc);
    }
    if constexpr (QL > 4) {
        so_team_step<D, 4, SO_TEAM_PK_PLAIN != 0>(vb, mp, acc);
        so_team_step<D, 5, SO_TEAM_PK_PLAIN != 0>(vb, mp, acc);
        so_team_step<D, 6, SO_TEAM_PK_PLAIN != 0>(vb, mp, acc);
        so_team_step<D, 7, SO_TEAM_PK_PLAIN != 0>(vb, mp, acc);
    }
}

// Sum acc[4] over the 2^NJ teams of a group (team index j = lane bits SHIFT .. SHIFT + NJ - 1) and
// store: a reduce-scatter (the first two exchanges halve the vector, bit i of j choosing the half a
// lane keeps), then plain butterflies.  Lanes with (j >> STEPS) == 0 end up with 4 >> STEPS channels.

template <int NJ, int SHIFT>
SO_DEVFN void so_group_reduce_store(float (&acc)[4], int j, bool live, float *o4) {
    constexpr int STEPS = NJ < 2 ? NJ : 2;
    int base = 0;
#pragma unroll
    for (int i = 0; i < STEPS; ++i) {
        const int hn = 4 >> (i + 1);
        const bool upper = (j & (1 << i)) != 0;
#pragma unroll
        for (int c = 0; c < hn; ++c) {
            // opaque copies: otherwise LLVM folds select(load, load) into a dynamically indexed
            // load of acc[] and lowers that to a v_cndmask chain per element
            float lo = acc[c], hi = acc[c + hn];
            asm("" : "+v"(lo), "+v"(hi));
            const float send = upper ? lo : hi;
            const float keep = upper ? hi : lo;
            acc[c] = keep + __shfl_xor(send, 1 << (i + SHIFT), 64);
        }
        if (upper) base += hn;
    }
    constexpr int N = 4 >> STEPS;
#pragma unroll
    for (int i = STEPS; i < NJ; ++i) {
#pragma unroll
        for (int c = 0; c < N; ++c) acc[c] += __shfl_xor(acc[c], 1 << (i + SHIFT), 64);
    }
    if (live && (j >> STEPS) == 0) {
#pragma unroll
        for (int c = 0; c < N; ++c) o4[base + c] = acc[c];
    }
}


// 4 waves / SIMD (<= 128 VGPRs) for the shipped head width: the camera-loop kernel otherwise takes 144 VGPRs (3 waves);
// measured -1.5 % on the eval encoder, 5 / 6 waves spill (+20 % / +39 %)
#define SO_MSDA_FWD_WAVES(D) ((D) <= 16 ? 4 : 1)
// the camera loop at 16 channels with <= 16 lanes per (query, head) group (L * P <= 16: no shipped config) keeps 4+ groups'
// state per lane round: 128 registers spilled 36 - 108 bytes; three waves per SIMD (170 registers) hold it
#define SO_MSDA_CROSS_WAVES(D, LOGG) (((D) == 16 && (LOGG) <= 4) ? 3 : SO_MSDA_FWD_WAVES(D))
