// psh_device.h -- device code shared by the translation units of libpsh_hip.so (psh_scan.hip, psh_embed.hip,
// psh_select.hip): small helpers, the per-query preparation, the per-lane window arithmetic of the scans, the
// staging of a segment, the deferred candidate append.  Internal; see psh_scan.hip for the design overview.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "psh_kernels.h"

namespace psh {

// ----------------------------------------------------------------------------------
// small helpers
// ----------------------------------------------------------------------------------
typedef const __attribute__((address_space(4))) float* const_f32p;  // scalar (SGPR) loads
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define PSH_INF_BITS 0x7f800000u
// relative margin put on every threshold derived from a bin edge or a sample value: a
// window at or above tau then has a strictly larger DISTANCE than anything counted
// below it (sqrt and the division compress a few ulps, 2^-16 is ~250 ulps)
#define PSH_TAU_MARGIN (1.0f + 1.0f / 65536.0f)

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63u); }

// u / d without the ~40-instruction scalar division sequence (two of them per segment
// sat on every wave's critical path: ~1000 cycles): magic = floor(2^32 / d) gives
// umulhi(u, magic) in {u/d - 1, u/d} for every u < 2^32 (the product falls short of u/d
// by u * frac(2^32/d) / 2^32 < 1), one compare-and-fix makes it exact.
__device__ __forceinline__ unsigned fast_div(unsigned u, unsigned magic, unsigned d) {
    if (d == 1u) return u;
    unsigned q = __umulhi(u, magic);
    q += (u - q * d >= d) ? 1u : 0u;
    return q;
}

// LDS tile layout: logical float p lives at p + 4*(p/64): one 16-byte pad slot after
// every 16 slots.  Lanes read 16-byte slots at a stride of 4 slots (16 windows); the
// pad makes the 16 lanes of every ds_read_b128 service group hit 16 distinct slots.
__device__ __forceinline__ int lds_pad(int p) { return p + ((p >> 6) << 2); }

__device__ __forceinline__ void wave_lds_fence() {
    // orders this wave's LDS writes before its later LDS reads of OTHER lanes' data;
    // LDS operations of one wave execute in issue order, the fence stops the compiler
    // from moving the (provably non-aliasing per lane) reads above the writes.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// IEEE correctly rounded, denormal-preserving: the same results as the host's
// sqrtf / division the reference's CPU path goes through.
__device__ __forceinline__ float dist_from_acc(float acc, float xn) {
    // plain sqrtf and '/' : hipcc's default code generation for both is the correctly
    // rounded, denormal-preserving sequence (-fhip-fp32-correctly-rounded-divide-sqrt);
    // the __fsqrt_rn intrinsic is NOT (it maps to the approximate native sqrt).
    return __builtin_sqrtf(acc) / xn;
}

// sum of squares in the order of ATen's contiguous last-dim norm reduce (the oracle's
// sumsq8 documents the probe): 8 lanes of fma over whole blocks of 8, lanes added left
// to right, tail: groups of 4 as rounded products added one by one, then a scalar fma
// chain for the last < 4.
template <typename F>
__device__ inline float sumsq8(F get, int W) {
    float lane[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int nb = W / 8;
    for (int b = 0; b < nb; ++b) {
#pragma unroll
        for (int i = 0; i < 8; ++i) { const float v = get(8 * b + i); lane[i] = __builtin_fmaf(v, v, lane[i]); }
    }
    float s = lane[0];
#pragma unroll
    for (int i = 1; i < 8; ++i) s = __fadd_rn(s, lane[i]);
    int j = 8 * nb;
    for (; j + 4 <= W; j += 4) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { const float v = get(j + i); s = __fadd_rn(s, __fmul_rn(v, v)); }
    }
    for (; j < W; ++j) { const float v = get(j); s = __builtin_fmaf(v, v, s); }
    return s;
}

// ----------------------------------------------------------------------------------
// K0: per-query preparation -- ||x||, state reset
// ----------------------------------------------------------------------------------
__device__ __forceinline__ void prep_query(const PrepArgs& a, int b) {
    {
        const float* x = a.queries + (int64_t)b * a.W;
        const float s = sumsq8([&](int j) { return x[j]; }, a.W);
        QueryState q;
        q.xn = a.qnorm_in ? a.qnorm_in[b] : __builtin_sqrtf(s);
        q.tau_bits = PSH_INF_BITS;       // +inf until the bootstrap lowers it
        q.n_valid = 0;
        q.nx = s;
        q.thr_base = __uint_as_float(PSH_INF_BITS);   // rejects nothing until the threshold kernel sets it
        q.mx_scale = 0.0f;                            // the matrix-core filter is off until the threshold kernel arms it
        q.mx_thr = __uint_as_float(PSH_INF_BITS);
        q.tau2_bits = PSH_INF_BITS;                   // = tau until the threshold kernel estimates it
        q.mx_thr2 = __uint_as_float(PSH_INF_BITS);
        q.mx8_P = q.mx8_L = q.mx8_k1 = 0.0f;         // (the 8-bit test likewise)
        a.qstate[b] = q;
        a.total[b] = 0;
        if (a.status) a.status[b] = PSH_STATUS_OK_;
    }
}


// ----------------------------------------------------------------------------------
// the sliding-window scan
// ----------------------------------------------------------------------------------
// Per-lane accumulation of the L=16 consecutive windows starting at logical tile
// index 16*lane.  win[s] holds y[16*lane + m] for the newest m = s (mod 16); at step
// j window i reads slot (i + j) & 15 and slot j & 15 is then refilled with y[.. + j + 16].
// The chain over j is strictly sequential per window: the reference's order.
// One step j of the 16 chains of a lane: D_i = x_j - y_{i+j}; acc_i = fma(D_i, D_i, acc_i)
// -- the reference's two roundings per term, in its order.  Written as two blocks of
// 8 v_sub_f32 followed by their 8 v_fmac_f32: left to itself hipcc emits every FMA right
// behind the subtraction it depends on (one temporary register), and a back-to-back
// dependent VALU pair issues at about half rate unless the SIMD has other waves to
// switch to (measured: 45 vs 63 T lane-ops/s at 4 waves/SIMD).  Plain VALU RAW hazards
// are interlocked in hardware, so nothing inside the block needs a wait state.
__device__ __forceinline__ void step8(float xj, float w0, float w1, float w2, float w3, float w4, float w5,
                                      float w6, float w7, float& a0, float& a1, float& a2, float& a3,
                                      float& a4, float& a5, float& a6, float& a7) {
    float t0, t1, t2, t3, t4, t5, t6, t7;
    asm volatile(
        "v_sub_f32 %8, %16, %17\n\t"
        "v_sub_f32 %9, %16, %18\n\t"
        "v_sub_f32 %10, %16, %19\n\t"
        "v_sub_f32 %11, %16, %20\n\t"
        "v_sub_f32 %12, %16, %21\n\t"
        "v_sub_f32 %13, %16, %22\n\t"
        "v_sub_f32 %14, %16, %23\n\t"
        "v_sub_f32 %15, %16, %24\n\t"
        "v_fmac_f32 %0, %8, %8\n\t"
        "v_fmac_f32 %1, %9, %9\n\t"
        "v_fmac_f32 %2, %10, %10\n\t"
        "v_fmac_f32 %3, %11, %11\n\t"
        "v_fmac_f32 %4, %12, %12\n\t"
        "v_fmac_f32 %5, %13, %13\n\t"
        "v_fmac_f32 %6, %14, %14\n\t"
        "v_fmac_f32 %7, %15, %15"
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7),
          "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7)
        : "s"(xj), "v"(w0), "v"(w1), "v"(w2), "v"(w3), "v"(w4), "v"(w5), "v"(w6), "v"(w7));
}

__device__ __forceinline__ void step16(float xj, const float (&win)[PSH_L], int jj, float (&acc)[PSH_L]) {
    step8(xj, win[(0 + jj) & 15], win[(1 + jj) & 15], win[(2 + jj) & 15], win[(3 + jj) & 15],
          win[(4 + jj) & 15], win[(5 + jj) & 15], win[(6 + jj) & 15], win[(7 + jj) & 15],
          acc[0], acc[1], acc[2], acc[3], acc[4], acc[5], acc[6], acc[7]);
    step8(xj, win[(8 + jj) & 15], win[(9 + jj) & 15], win[(10 + jj) & 15], win[(11 + jj) & 15],
          win[(12 + jj) & 15], win[(13 + jj) & 15], win[(14 + jj) & 15], win[(15 + jj) & 15],
          acc[8], acc[9], acc[10], acc[11], acc[12], acc[13], acc[14], acc[15]);
}

// PAD = false: the tile is stored unpadded (stage_store<false>) -- every address is the lane's base plus a constant, ONE
// address register instead of nine (the one-wave sample kernel of psh_stream.hip, which lives in 64 VGPRs; bank
// conflicts do not matter there)
template <bool PAD> __device__ __forceinline__ int lds_idx(int p) { return PAD ? lds_pad(p) : p; }
template <int WT, bool PAD = true>
__device__ __forceinline__ void accumulate16(const float* tile, int lane, const_f32p x, int W,
                                             float (&acc)[PSH_L]) {
    float win[PSH_L];
    const int base = PSH_L * lane;
#pragma unroll
    for (int c = 0; c < PSH_L / 4; ++c) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(tile + lds_idx<PAD>(base + 4 * c));
        win[4 * c + 0] = v[0]; win[4 * c + 1] = v[1]; win[4 * c + 2] = v[2]; win[4 * c + 3] = v[3];
    }
#pragma unroll
    for (int i = 0; i < PSH_L; ++i) acc[i] = 0.0f;

    const int Wc = WT > 0 ? WT : W;
    int j0 = 0;
    // whole blocks of 16 steps (fully unrolled when W is a compile-time constant)
    auto block16 = [&](int jb) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 nx = *reinterpret_cast<const f32x4*>(tile + lds_idx<PAD>(base + jb + PSH_L + 4 * g));
            const float nv[4] = {nx[0], nx[1], nx[2], nx[3]};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int jj = 4 * g + q;
                const float xj = x[jb + jj];
                step16(xj, win, jj, acc);
                win[jj] = nv[q];
            }
        }
    };
    if constexpr (WT > 0) {
#pragma unroll
        for (int blk = 0; blk < WT / PSH_L; ++blk) { block16(j0); j0 += PSH_L; }
    } else {
        for (; j0 + PSH_L <= Wc; j0 += PSH_L) block16(j0);
    }
    // remainder: Wc - j0 in [0, 16) steps
    const int rem = Wc - j0;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        if (4 * g < rem) {
            const f32x4 nx = *reinterpret_cast<const f32x4*>(tile + lds_idx<PAD>(base + j0 + PSH_L + 4 * g));
            const float nv[4] = {nx[0], nx[1], nx[2], nx[3]};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int jj = 4 * g + q;
                if (jj < rem) {
                    const float xj = x[j0 + jj];
                    step16(xj, win, jj, acc);
                    win[jj] = nv[q];
                }
            }
        }
    }
}

// 8 correlation chains, one v_fmac_f32 each, with the query tap in a VGPR.  Measured issue
// cost per wave64 instruction on this part (tools/ubench_dot2.hip, 4 waves/SIMD):
//   v_sub/v_fmac with <= 2 distinct VGPR sources 1.12 ns,  v_fmac c, x(VGPR), w 1.39 ns,
//   v_fmac c, x(SGPR), w 1.98 ns,  v_dot2c_f32_bf16 2.0 ns,  v_cvt_pk_bf16_f32 3.0 ns
// -- so the tap is kept in a VGPR although it is wave-uniform.
__device__ __forceinline__ void corr8(float xj, float w0, float w1, float w2, float w3, float w4, float w5,
                                      float w6, float w7, float& c0, float& c1, float& c2, float& c3,
                                      float& c4, float& c5, float& c6, float& c7) {
    asm volatile(
        "v_fmac_f32 %0, %8, %9\n\t"
        "v_fmac_f32 %1, %8, %10\n\t"
        "v_fmac_f32 %2, %8, %11\n\t"
        "v_fmac_f32 %3, %8, %12\n\t"
        "v_fmac_f32 %4, %8, %13\n\t"
        "v_fmac_f32 %5, %8, %14\n\t"
        "v_fmac_f32 %6, %8, %15\n\t"
        "v_fmac_f32 %7, %8, %16"
        : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7)
        : "v"(xj), "v"(w0), "v"(w1), "v"(w2), "v"(w3), "v"(w4), "v"(w5), "v"(w6), "v"(w7));
}

// ---- bound-then-verify: the cheap test of the full scan -------------------------------
// The exact chain costs 2 VALU operations per term (subtract, fma) and that, not HBM, is
// what bounds the scan: 41 lane-operations per window against ~64 T lane-ops/s is
// 83 us for the 1.3e8 windows of one query, HBM needs ~85.  But only ~1e-4 of the windows
// can be admitted, so the scan first evaluates   S = nx + ny - 2c   (c: correlation with
// the query, 1 fma per term; ny: window energy from a running prefix sum, ~3 operations
// per window) -- 25 operations per window -- with a rigorous rounding-error bound, rejects
// every window that provably cannot satisfy acc < tau, and re-evaluates the survivors
// with the exact chain.  Ranking only ever sees exact values.
//   t_i  = ny_i - 2 c_i  (computed),   |t_i - (ny_i - 2c_i)| <= 2^-17 (nx + NY)
//   NY   = energy of the lane's W+15 values (bounds every prefix-sum error)
// Compile-time W >= 17 only (the prefix differences P_{i+W} - P_i are taken while the
// values stream through the 16-register window).
template <int WT>
__device__ __forceinline__ void approx16(const float* tile, int lane, const float (&xv)[WT], float (&t)[PSH_L], float& NY) {
    static_assert(WT >= 17 && WT <= 32, "approx16 streams W in [17, 32]");
    float win[PSH_L], c[PSH_L], Ps[PSH_L];
    const int base = PSH_L * lane;
#pragma unroll
    for (int q = 0; q < PSH_L / 4; ++q) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(tile + lds_pad(base + 4 * q));
        win[4 * q + 0] = v[0]; win[4 * q + 1] = v[1]; win[4 * q + 2] = v[2]; win[4 * q + 3] = v[3];
    }
    float P = 0.0f;                                   // P_m = sum_{n<m} y_n^2
#pragma unroll
    for (int m = 0; m < PSH_L; ++m) { Ps[m] = P; P = __builtin_fmaf(win[m], win[m], P); c[m] = 0.0f; }
    // steps j = 0 .. WT-1; after step j the value y_{j+16} replaces y_j in slot j & 15
#pragma unroll
    for (int g = 0; g < (WT + 3) / 4; ++g) {
        f32x4 nx4 = {0.f, 0.f, 0.f, 0.f};
        if (4 * g + 16 <= WT + 14) nx4 = *reinterpret_cast<const f32x4*>(tile + lds_pad(base + PSH_L + 4 * g));
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int j = 4 * g + q;
            if (j < WT) {
                const float xj = xv[j];
                corr8(xj, win[(0 + j) & 15], win[(1 + j) & 15], win[(2 + j) & 15], win[(3 + j) & 15],
                      win[(4 + j) & 15], win[(5 + j) & 15], win[(6 + j) & 15], win[(7 + j) & 15],
                      c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7]);
                corr8(xj, win[(8 + j) & 15], win[(9 + j) & 15], win[(10 + j) & 15], win[(11 + j) & 15],
                      win[(12 + j) & 15], win[(13 + j) & 15], win[(14 + j) & 15], win[(15 + j) & 15],
                      c[8], c[9], c[10], c[11], c[12], c[13], c[14], c[15]);
                if (j + 16 <= WT + 14) {              // y_{j+16} is still needed by some window
                    const float v = nx4[q];
                    win[j & 15] = v;
                    P = __builtin_fmaf(v, v, P);        // P_{j+17}
                    if (j + 17 >= WT) Ps[j + 17 - WT] = P - Ps[j + 17 - WT];   // ny_i, i = j + 17 - W
                }
            }
        }
    }
    NY = P;
#pragma unroll
    for (int i = 0; i < PSH_L; ++i) t[i] = __builtin_fmaf(-2.0f, c[i], Ps[i]);
}

// the exact chain of ONE window (survivors of the cheap test): tile index p = first value
template <int WT>
__device__ __forceinline__ float exact_one(const float* tile, int p, const_f32p x) {
    float y[WT];
#pragma unroll
    for (int j = 0; j < WT; ++j) y[j] = tile[lds_pad(p + j)];
    float a = 0.0f;
#pragma unroll
    for (int j = 0; j < WT; ++j) { const float D = __fsub_rn(x[j], y[j]); a = __builtin_fmaf(D, D, a); }
    return a;
}

// the same with a run-time window length
__device__ __forceinline__ float exact_one_rt(const float* tile, int p, const_f32p x, int W) {
    float a = 0.0f;
    for (int j = 0; j < W; ++j) { const float D = __fsub_rn(x[j], tile[lds_pad(p + j)]); a = __builtin_fmaf(D, D, a); }
    return a;
}

// One-window-per-row edge case (T == W + h): the reference's numerator uses the
// 8-lane order instead of the sequential chain (probed; see the oracle).  Only window 0
// of segment 0 exists; exhaustive path only.
__device__ inline float acc_single_window(const float* tile, int lane, const_f32p x, int W) {
    const int base = PSH_L * lane;
    return sumsq8([&](int j) { return __fsub_rn(x[j], tile[lds_pad(base + j)]); }, W);
}

// minimum of the 16 accumulators of an MFMA tile in 8 v_min3_f32.  (fminf() makes the compiler quiet possible signalling
// NaNs first -- two v_max x, x per tile in the hottest loop of the batched scan; v_min3 returns the non-NaN operands'
// minimum just the same: a NaN accumulator is ignored, which is what the callers want -- its window can never be admitted.)
__device__ __forceinline__ float min3f(float a, float b, float c) {
    // nested minnum: the compiler forms v_min3_f32 and -- unlike with inline assembly -- knows an MFMA result is being
    // read: the wait states after the MFMA are its to insert (an asm version of this read accumulators the scheduler had
    // moved right behind their MFMA)
    return __builtin_fminf(__builtin_fminf(a, b), c);
}
// x^ of the 8-bit batched scan: a query sample on the batch's step (inv_s0 = 127 / max|x| of the batch) -- ONE expression for
// mq_prep_kernel (the residues' norms) and threshold_kernel (the table, ||x^||_1): the bound needs them to agree bit for bit
__device__ __forceinline__ int mq8_quant(float xv, float inv_s0) {
    const int q = (int)rintf(xv * inv_s0);
    return q > 127 ? 127 : (q < -127 ? -127 : q);
}
// The maximum of a non-negative value over the wave, in every lane (uniform): four DPP steps inside the rows of 16 lanes, then the
// four rows through SGPRs -- no trip through the LDS crossbar (six dependent ds_bpermute are ~100+ cycles each beside a busy LDS).
__device__ __forceinline__ float wave_max_nonneg(float x) {
    // (as integers: non-negative floats order like their bit patterns, and an integer maximum needs no quieting of its operands
    //  -- fmaxf() cost a v_max x, x per operand and kept the DPP move a separate instruction; old = 0 with bound_ctrl lets the
    //  compiler fold the move into the v_max_i32 itself)
    int v = __float_as_int(x);
    auto step = [&](int moved) { v = v > moved ? v : moved; };
    step(__builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true));     // quad_perm [1, 0, 3, 2]
    step(__builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true));     // quad_perm [2, 3, 0, 1]
    step(__builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true));    // row_half_mirror
    step(__builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true));    // row_mirror
    const int r0 = __builtin_amdgcn_readlane(v, 0), r1 = __builtin_amdgcn_readlane(v, 16);
    const int r2 = __builtin_amdgcn_readlane(v, 32), r3 = __builtin_amdgcn_readlane(v, 48);
    const int m01 = r0 > r1 ? r0 : r1, m23 = r2 > r3 ? r2 : r3;
    return __int_as_float(m01 > m23 ? m01 : m23);
}
// The sum of a value over the wave, in every lane (uniform), the same way: pairs, quads, half rows, rows by DPP, the four rows
// through SGPRs.  (The order of the additions is not the lane order: callers that bound a rounding error allow for any order.)
__device__ __forceinline__ float wave_sum_dpp(float x) {
    int v = __float_as_int(x);
    auto step = [&](int moved) { v = __float_as_int(__int_as_float(v) + __int_as_float(moved)); };
    step(__builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true));     // quad_perm [1, 0, 3, 2]
    step(__builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true));     // quad_perm [2, 3, 0, 1]
    step(__builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true));    // row_half_mirror
    step(__builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true));    // row_mirror
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(v, 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(v, 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(v, 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(v, 48));
    return (r0 + r1) + (r2 + r3);
}
__device__ __forceinline__ float max3f(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }   // v_max3_f32, as above
typedef float f32x16_t __attribute__((ext_vector_type(16)));
__device__ __forceinline__ float tile_min16(const f32x16_t& t) {
    // a tree of depth 3 (five independent v_min3 first), not a chain of 8: the chain's latency was the longest stretch of
    // a query group's epilogue in the batched scan.  Every read of an accumulator register is a compiler-visible
    // operation: the wait states between an MFMA and the first read of its result are the compiler's to insert.
    const float m0 = min3f(t[0], t[1], t[2]), m1 = min3f(t[3], t[4], t[5]), m2 = min3f(t[6], t[7], t[8]);
    const float m3 = min3f(t[9], t[10], t[11]), m4 = min3f(t[12], t[13], t[14]);
    return min3f(min3f(m0, m1, m2), min3f(m3, m4, t[15]), __uint_as_float(PSH_INF_BITS));
}

__device__ __forceinline__ float min16(const float (&a)[PSH_L]) {
    float m = fminf(fminf(a[0], a[1]), a[2]);
#pragma unroll
    for (int i = 3; i + 1 < PSH_L; i += 2) m = fminf(fminf(m, a[i]), a[i + 1]);
    return fminf(m, a[PSH_L - 1]);
}

struct Stage {  // one segment in flight from HBM, 5 x 16 bytes per lane
    f32x4 v[PSH_NSTAGE];
};

// one of the PSH_NSTAGE 16-byte loads of a segment (q is a compile-time index at every
// call site).  row: first float of the row; floats [seg_start, seg_start + nfloat) are
// wanted, clamped to the row (the clamped tail only feeds inadmissible windows).
template <bool ALIGNED>
__device__ __forceinline__ void stage_load_one(Stage& st, int q, const float* __restrict__ row, int64_t T,
                                               int seg_start, int nfloat, int lane) {
    if (ALIGNED) {
        const f32x4* src = reinterpret_cast<const f32x4*>(row + seg_start);
        const int last = (int)((T - seg_start) >> 2) - 1;  // last float4 inside the row
        const int nq = (nfloat + 3) >> 2;                   // 256 <= nq <= 320
        int m = lane + 64 * q;
        if (q < PSH_NSTAGE - 1 || m < nq) {
            m = m > last ? last : m;
            st.v[q] = __builtin_nontemporal_load(src + m);
        }
    } else {
        const int lastf = (int)(T - seg_start) - 1;
        float e[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            int p = 4 * (lane + 64 * q) + c;
            p = p > lastf ? lastf : p;
            e[c] = (4 * (lane + 64 * q) < nfloat) ? row[seg_start + p] : 0.0f;
        }
        st.v[q] = f32x4{e[0], e[1], e[2], e[3]};
    }
}

template <bool ALIGNED>
__device__ __forceinline__ void stage_load(Stage& st, const float* __restrict__ row, int64_t T,
                                           int seg_start, int nfloat, int lane) {
#pragma unroll
    for (int q = 0; q < PSH_NSTAGE; ++q) stage_load_one<ALIGNED>(st, q, row, T, seg_start, nfloat, lane);
}

template <bool PAD = true>
__device__ __forceinline__ void stage_store(const Stage& st, float* tile, int nfloat, int lane) {
    const int nq = (nfloat + 3) >> 2;
#pragma unroll
    for (int q = 0; q < PSH_NSTAGE; ++q) {
        const int m = lane + 64 * q;
        if (q < PSH_NSTAGE - 1 || m < nq) *reinterpret_cast<f32x4*>(tile + lds_idx<PAD>(4 * m)) = st.v[q];
    }
}

// ---- matrix-core rejection test: f16 staging layout (scan_mx_kernel, scan_mq_kernel, scan_fused_kernel) ---------
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define PSH_MX_SLOTS 144                      // 16-byte slots per f16 array: 32*31 + 64 values, whole groups of 16 slots
#define PSH_MX_NHALF (PSH_MX_SLOTS * 8)
#define PSH_MX_PEND 64                        // >= 64: one ballot can admit a whole wave

// logical f16 index -> LDS index.  A-fragment reads of the 32 rows sit 64 bytes apart
// (4 slots): rotating the slot inside its group of 16 by the group number spreads 16
// consecutive rows over 16 distinct slots without any padding.
__device__ __forceinline__ int mx_half(int idx) {
    const int slot = idx >> 3;
    return (((slot & ~15) | ((slot + (slot >> 4)) & 15)) << 3) | (idx & 7);
}

// A call that reports PSH_STATUS_RETRY leaves NO plausible numbers behind: NaN distances and (-1, -1) indices instead of an
// earlier call's results (the protocol says "invalid"; a caller that forgot to look at the status sees it at once).
__device__ __forceinline__ void poison_results(float* out_d, int32_t* out_idx, int k, int tid, int nthreads) {
    for (int i = tid; i < k; i += nthreads) {
        out_d[i] = __uint_as_float(0x7fc00000u);
        out_idx[2 * i + 0] = -1;
        out_idx[2 * i + 1] = -1;
    }
}

// ---- deferred candidate append -----------------------------------------------------
// vmcnt retires in order: a global store issued by the (rare) admission path would be
// YOUNGER than the prefetch of the next segment, so anything that later waits for that
// store -- including the compiler's conservative wait at the loop head -- would also wait
// for the prefetch and serialise HBM latency with compute.  Admitted windows therefore go
// to a wave-private LDS buffer first (LDS traffic only inside the hot loop) and are
// written out at the top of the next iteration, BEFORE the next prefetch is issued.
#define PSH_PEND 64                       // entries per wave: one flush lane each
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void pend_flush(const u32x4* pend, int npend, int* lcount, const ScanArgs& a, int lane) {
    wave_lds_fence();                                  // other lanes' entries
    if (lane < npend) {
        const u32x4 e = pend[lane];                    // {acc bits, r, t, query}
        const int b = (int)e[3];
        const int pos = atomicAdd(&lcount[b], 1);      // LDS: this block's cursor for query b
        if (pos < a.slice) {
            // (2-D grids: blockIdx.y picks a chunk of queries, so the blocks of one column never write the same query)
            const int64_t o = (int64_t)b * a.cap + (int64_t)blockIdx.x * a.slice + pos;
            a.cand_d[o] = dist_from_acc(__uint_as_float(e[0]), a.qstate[b].xn);
            a.cand_rt[o] = make_int2((int)e[1], (int)e[2]);
        }
    }
}


// what one query keeps of the 16 accumulators of a lane (the three modes of scan_kernel)
template <int MODE>
__device__ __forceinline__ void emit16(const ScanArgs& a, int b, const float (&acc)[PSH_L], int nvalid, int lane,
                                       unsigned rs, int r_global, int t_lane, float tau, float xn,
                                       u32x4* pend, int& npend, int* lcount) {
    if (MODE == PSH_MODE_BOOT) {
        float m = __uint_as_float(PSH_INF_BITS);
#pragma unroll
        for (int i = 0; i < PSH_L; ++i) m = (i < nvalid) ? fminf(m, acc[i]) : m;
        if (a.boot_per_wave) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) m = fminf(m, __shfl_xor(m, off, 64));
            if (lane == 0) a.minbuf[(int64_t)b * a.min_stride + (int64_t)rs] = m;
        } else {
            a.minbuf[(int64_t)b * a.min_stride + (int64_t)rs * 64 + lane] = m;
        }
    } else if (MODE == PSH_MODE_ALL) {
        const int64_t base = (int64_t)b * a.cap + (int64_t)rs * PSH_SEG + PSH_L * lane;
#pragma unroll
        for (int i = 0; i < PSH_L; ++i) {
            const bool ok = i < nvalid;
            a.cand_d[base + i] = ok ? dist_from_acc(acc[i], xn) : __uint_as_float(PSH_INF_BITS);
            a.cand_rt[base + i] = ok ? make_int2(r_global, t_lane + i) : make_int2(-1, -1);
        }
    } else {
        if (!__any(min16(acc) < tau)) return;
        unsigned hm = 0u;
#pragma unroll
        for (int i = 0; i < PSH_L; ++i) hm |= ((i < nvalid) && (acc[i] < tau)) ? (1u << i) : 0u;
#pragma unroll 1
        for (int i = 0; i < PSH_L; ++i) {
            const bool hit = ((hm >> i) & 1u) != 0u;
            const unsigned long long mask = __ballot(hit);
            if (!mask) continue;
            float v = acc[0];
#pragma unroll
            for (int j = 1; j < PSH_L; ++j) v = (i == j) ? acc[j] : v;   // i is wave-uniform
            const int nh = __popcll(mask);
            if (npend + nh > PSH_PEND) {
                pend_flush(pend, npend, lcount, a, lane);
                npend = 0;
                wave_lds_fence();                            // the flush has read pend before it is refilled
            }
            if (hit) {
                const int slot = npend + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32),
                                             __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
                pend[slot] = u32x4{__float_as_uint(v), (unsigned)r_global, (unsigned)(t_lane + i), (unsigned)b};
            }
            npend += nh;
        }
    }
}

}  // namespace psh
