// psh_kernels.hip -- gfx950 (MI355X, CDNA4) kernels of the k-nearest-path scan.
//
// What is computed (reference RudyMorel/shadowing, shadowing/path_shadowing/):
//   path_embedding.py:129-139  Identity embedding == the window y[r, t:t+W] itself
//   path_distance.py:62-65     RelativeMSE  d = ||x - y_win|| / ||x||, evaluated by
//                              the reference as the sequential fp32 chain
//                                 D_j = x_j - y_{t+j};  acc = fma(D_j, D_j, acc)
//                                 d   = fl(fl(sqrt(acc)) / xn)
//   path_shadowing.py:149-173  top-k over all windows of all rows (+ running merge)
//   path_shadowing.py:43-58    flat index -> (row, t)
//
// Design (DESIGN.md has the long form):
//   * one WAVE (64 lanes) owns a segment of 1024 consecutive windows of one row: it streams
//     the 4 KB (+ W-1 halo) with coalesced non-temporal 16-byte loads and stages them in
//     wave-private LDS (no block barrier in any scan loop); each dataset element is fetched
//     from HBM once.
//   * ranking only ever sees the reference's exact value: the per-window chain is kept in the
//     reference's order (bit-exact distances are what make indices bit-exact) -- no
//     tree/shuffle reduction.
//   * exact fp32 is 41 VALU operations per 4 bytes -- more than the vector ALUs issue at
//     8 TB/s -- so the scans are bound-then-verify: a cheap quantity with a RIGOROUS error
//     bound rejects what cannot be admitted, the ~1e-4 survivors get the exact chain.  The
//     cheap quantity is a banded f16 product on the matrix cores (scan_mx_kernel, one query;
//     scan_mq_kernel / boot_mq_kernel, batches) or an fp32 correlation + prefix sums on the
//     VALU (scan_kernel, every other window length; PSH_FILTER=valu).
//   * an admission threshold tau (a provable upper bound of the k-th smallest acc: the k-th
//     smallest over ANY subset of the windows bounds the global one) keeps all but ~1e4 of
//     the 1e8 windows out of the candidate lists; it comes from a bootstrap pass over 1/16
//     of the rows.  Survivors are appended to per-block slices with LDS cursors -- no global
//     atomics anywhere (device-scope atomics on one line cost ~25 ns each on this 8-XCD
//     part; 5e4 of them were 5x the whole scan).  A one-block radix select then picks the
//     k best and orders them by (d, r, t).
//   * the embedded scan (embed_scan_kernel) runs the same pipeline behind a linear embedding.
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
        q.pad[0] = q.pad[1] = q.pad[2] = 0;
        a.qstate[b] = q;
        a.total[b] = 0;
        if (a.status) a.status[b] = PSH_STATUS_OK_;
    }
}

__global__ void prep_kernel(PrepArgs a) {
    if (threadIdx.x == 0) prep_query(a, (int)blockIdx.x);
}

__global__ void qnorm_kernel(const float* queries, int B, int W, float* out) {
    const int b = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (b >= B) return;
    const float* x = queries + (int64_t)b * W;
    out[b] = __builtin_sqrtf(sumsq8([&](int j) { return x[j]; }, W));
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

template <int WT>
__device__ __forceinline__ void accumulate16(const float* tile, int lane, const_f32p x, int W,
                                             float (&acc)[PSH_L]) {
    float win[PSH_L];
    const int base = PSH_L * lane;
#pragma unroll
    for (int c = 0; c < PSH_L / 4; ++c) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(tile + lds_pad(base + 4 * c));
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
            const f32x4 nx = *reinterpret_cast<const f32x4*>(tile + lds_pad(base + jb + PSH_L + 4 * g));
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
            const f32x4 nx = *reinterpret_cast<const f32x4*>(tile + lds_pad(base + j0 + PSH_L + 4 * g));
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
    float r;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
typedef float f32x16_t __attribute__((ext_vector_type(16)));
__device__ __forceinline__ float tile_min16(const f32x16_t& t) {
    float m = min3f(t[0], t[1], t[2]);
    m = min3f(m, t[3], t[4]);
    m = min3f(m, t[5], t[6]);
    m = min3f(m, t[7], t[8]);
    m = min3f(m, t[9], t[10]);
    m = min3f(m, t[11], t[12]);
    m = min3f(m, t[13], t[14]);
    return min3f(m, t[15], t[15]);
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

__device__ __forceinline__ void stage_store(const Stage& st, float* tile, int nfloat, int lane) {
    const int nq = (nfloat + 3) >> 2;
#pragma unroll
    for (int q = 0; q < PSH_NSTAGE; ++q) {
        const int m = lane + 64 * q;
        if (q < PSH_NSTAGE - 1 || m < nq) *reinterpret_cast<f32x4*>(tile + lds_pad(4 * m)) = st.v[q];
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

// MODE_BOOT  : minimum over the admissible windows of each lane (or of the whole wave
//              segment) -> minbuf: a subset of the windows, so its k-th smallest bounds
//              the global k-th smallest from above
// MODE_FILTER: append windows with acc < tau to this block's slice
// MODE_ALL   : every admissible window to its own slot (exhaustive path)
template <int WT, bool ALIGNED, int MODE>
__global__ __launch_bounds__(PSH_SCAN_THREADS) void scan_kernel(ScanArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = lane_id();
    const int wave_in_block = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // provably uniform
    float* tile = smem + (size_t)wave_in_block * a.tile_floats;
    // FILTER: this block's append cursor per query lives in LDS -- same-address global
    // atomics are served at the memory side of the 8 non-coherent XCD L2s (tens of ns
    // each, serialised), so the candidate list is written in per-block slices instead
    int* lcount = reinterpret_cast<int*>(smem + (size_t)(PSH_SCAN_THREADS / 64) * a.tile_floats);
    u32x4* pend = reinterpret_cast<u32x4*>(lcount + ((a.B + 3) & ~3) + 4) + (size_t)wave_in_block * PSH_PEND;
    int npend = 0;                                       // wave-uniform
    // lcount[B .. ]: the block's work cursor.  Waves of one SIMD are served oldest first,
    // so with a static split the young waves of every SIMD finish up to 2x later than the
    // old ones and the tail of the launch runs at a fraction of the occupancy (measured:
    // waves end between 69 and 149 us).  All 16 waves of the block therefore pull
    // segments from one LDS counter; the block's own share of the units is static.
    int* next_unit = lcount + ((a.B + 3) & ~3);
    if (threadIdx.x == 0) { next_unit[0] = 0; next_unit[1] = 0; }
    if (MODE == PSH_MODE_FILTER)
        for (int q = (int)threadIdx.x; q < a.B; q += PSH_SCAN_THREADS) lcount[q] = 0;
    __syncthreads();
    float wmax = 0.0f;     // BOOT: largest |y| this lane has seen (-> f16 scale of the matrix-core filter)

    const int W = WT > 0 ? WT : a.W;
    const int nfloat = PSH_SEG + W - 1;
    const unsigned n_rs = (unsigned)a.n_rows * (unsigned)a.nseg;      // (row, segment) units
    const unsigned n_units = n_rs * (unsigned)a.n_qgroups;             // host guarantees < 2^31
    const unsigned u_lo = (unsigned)(((unsigned long long)n_units * blockIdx.x) / gridDim.x);
    const unsigned u_hi = (unsigned)(((unsigned long long)n_units * (blockIdx.x + 1)) / gridDim.x);
    const unsigned gw = blockIdx.x * (PSH_SCAN_THREADS / 64) + (unsigned)wave_in_block;
    const const_f32p xq = (const_f32p)a.queries;
    // per-query state through the scalar cache: no VGPR destination, no vmcnt
    typedef const __attribute__((address_space(4))) QueryState* const_qsp;
    const const_qsp qstate_k = (const_qsp)a.qstate;

    auto grab = [&]() -> unsigned {   // next unit of this block (wave-uniform), >= u_hi when exhausted
        int v = 0;
        if (lane == 0) v = atomicAdd(next_unit, 1);
        return u_lo + (unsigned)__builtin_amdgcn_readfirstlane(v);
    };

    if (a.dbg_times && lane == 0) a.dbg_times[2 * gw] = wall_clock64();
#ifdef PSH_PHASE_TIMING
    unsigned long long ph[5] = {0, 0, 0, 0, 0};
#endif
    Stage st;
    unsigned u = grab();
    // unit -> (query group, row index, segment)
    auto decode = [&](unsigned uu, unsigned& rs, unsigned& ri, unsigned& sg, unsigned& qgi) {
        qgi = fast_div(uu, a.magic_nrs, n_rs);
        rs = uu - qgi * n_rs;
        ri = fast_div(rs, a.magic_nseg, (unsigned)a.nseg);
        sg = rs - ri * (unsigned)a.nseg;
    };
    unsigned rs, ri, sg, qgi;
    if (u < u_hi) {
        decode(u, rs, ri, sg, qgi);
        const int64_t row = a.row0 + (int64_t)ri * a.row_stride;
        stage_load<ALIGNED>(st, a.dataset + row * a.T, a.T, (int)sg * PSH_SEG, nfloat, lane);
    }
    while (u < u_hi) {
        decode(u, rs, ri, sg, qgi);
        const int qg = (int)qgi;
        const int64_t row = a.row0 + (int64_t)ri * a.row_stride;
        const int seg_start = (int)sg * PSH_SEG;

#ifdef PSH_PHASE_TIMING
        const unsigned long long tp0 = __builtin_readcyclecounter();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long tp1 = __builtin_readcyclecounter();
#endif
        stage_store(st, tile, nfloat, lane);
        if (MODE == PSH_MODE_BOOT && a.blockmax) {
            const int nq = (nfloat + 3) >> 2;
#pragma unroll
            for (int q = 0; q < PSH_NSTAGE; ++q)
                if (q < PSH_NSTAGE - 1 || lane + 64 * q < nq)
                    wmax = fmaxf(fmaxf(wmax, fmaxf(fabsf(st.v[q][0]), fabsf(st.v[q][1]))),
                                 fmaxf(fabsf(st.v[q][2]), fabsf(st.v[q][3])));
        }
        wave_lds_fence();
#ifdef PSH_PHASE_TIMING
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const unsigned long long tp2 = __builtin_readcyclecounter();
#endif
        if (MODE == PSH_MODE_FILTER && npend > 0) {   // last iteration's admissions, ahead of the prefetch
            pend_flush(pend, npend, lcount, a, lane);
            npend = 0;
        }
        const unsigned un = grab();
        {   // prefetch the next unit of this wave while this one is computed.  (Spreading
            // these five loads over the arithmetic through a hook was tried: +6 % time --
            // the extra live ranges cost a spill at the 128-VGPR cap.)
#if defined(PSH_ABL) && (PSH_ABL == 1)
            if (false) {                                   // ablation 1: no HBM traffic after the first segment
#else
            if (un < u_hi) {
#endif
                unsigned rsn, rin, sgn, qgn;
                decode(un, rsn, rin, sgn, qgn);
                const int64_t rown = a.row0 + (int64_t)rin * a.row_stride;
                stage_load<ALIGNED>(st, a.dataset + rown * a.T, a.T, (int)sgn * PSH_SEG, nfloat, lane);
            }
        }

#ifdef PSH_PHASE_TIMING
        const unsigned long long tp3 = __builtin_readcyclecounter();
#endif
        const int t_lane = seg_start + PSH_L * lane;           // first window of this lane
        int nvalid = a.Tp - t_lane;                             // admissible windows of this lane
        nvalid = nvalid < 0 ? 0 : (nvalid > PSH_L ? PSH_L : nvalid);
        const int r_global = (int)(row + a.r_offset);

        const int q_begin = qg * a.q_per_group;
        const int q_end = (q_begin + a.q_per_group) < a.B ? (q_begin + a.q_per_group) : a.B;
        for (int b = q_begin; b < q_end; ++b) {
            const const_f32p x = xq + (int64_t)b * W;
            const float tau = (MODE == PSH_MODE_FILTER) ? __uint_as_float(qstate_k[b].tau_bits) : 0.0f;
            const float xn = (MODE != PSH_MODE_BOOT) ? qstate_k[b].xn : 0.0f;

            // FILTER: the cheap quantity rejects; BOOT: the same quantity plus its error bound is an UPPER
            // bound of acc, and upper bounds are all the threshold needs (25 lane-operations per window
            // instead of 41 for the exact chain)
            constexpr bool CHEAP = (MODE == PSH_MODE_FILTER || MODE == PSH_MODE_BOOT) && (WT >= 17) && (WT <= 32);
            float acc[PSH_L];
            float thr = 0.0f;
            if (CHEAP) {
                float NY;
                constexpr int WX = CHEAP ? WT : 20;
                float xv[WX];                              // the query taps as VGPRs (see corr8)
#pragma unroll
                for (int j = 0; j < WX; ++j) { xv[j] = x[j]; asm volatile("" : "+v"(xv[j])); }
#if defined(PSH_ABL) && (PSH_ABL == 2)
                NY = tile[lds_pad(PSH_L * lane)];                          // ablation 2: no arithmetic
#pragma unroll
                for (int i = 0; i < PSH_L; ++i) acc[i] = 1e30f;
#else
                approx16<WX>(tile, lane, xv, acc, NY);     // acc[] holds t_i = ny_i - 2 c_i here
#endif
                if (MODE == PSH_MODE_BOOT) {
                    // acc_i <= (nx + t_i + 2^-17 (nx + NY)) (1 + 2^-19): add nx (1 + 2^-16) + 2^-16 NY, both rounded
                    // up generously; the query state is not set up yet (that happens in the threshold kernel)
                    float nx = 0.0f;
#pragma unroll
                    for (int j = 0; j < WX; ++j) nx = __builtin_fmaf(xv[j], xv[j], nx);
                    const float add = __builtin_fmaf(NY, 1.0f / 65536.0f, nx * (1.0f + 1.0f / 32768.0f));
#pragma unroll
                    for (int i = 0; i < PSH_L; ++i) acc[i] = (acc[i] + add) * (1.0f + 1.0f / 65536.0f);
                } else {
                    thr = __builtin_fmaf(1.0f / 65536.0f, NY, qstate_k[b].thr_base);
                }
            } else if (MODE == PSH_MODE_ALL && a.Tp == 1) {
#pragma unroll
                for (int i = 0; i < PSH_L; ++i) acc[i] = 0.0f;
                acc[0] = acc_single_window(tile, lane, x, W);
            } else {
                accumulate16<WT>(tile, lane, x, W, acc);
            }

            if (MODE == PSH_MODE_BOOT) {
                float m;
                if (__all(nvalid == PSH_L)) {
                    m = min16(acc);
                } else {
                    m = __uint_as_float(PSH_INF_BITS);
#pragma unroll
                    for (int i = 0; i < PSH_L; ++i) m = (i < nvalid) ? fminf(m, acc[i]) : m;
                }
                if (a.boot_per_wave == 2) {                 // one minimum per half segment (lanes 0-31 / 32-63)
#pragma unroll
                    for (int off = 16; off > 0; off >>= 1) m = fminf(m, __shfl_xor(m, off, 64));
                    if ((lane & 31) == 0) a.minbuf[(int64_t)b * a.min_stride + 2 * (int64_t)rs + (lane >> 5)] = m;
                } else if (a.boot_per_wave) {
#pragma unroll
                    for (int off = 32; off > 0; off >>= 1) m = fminf(m, __shfl_xor(m, off, 64));
                    if (lane == 0) a.minbuf[(int64_t)b * a.min_stride + (int64_t)rs] = m;
                } else {
                    a.minbuf[(int64_t)b * a.min_stride + (int64_t)rs * 64 + lane] = m;
                }
            } else if (MODE == PSH_MODE_ALL) {
                // every window has its own slot: no cursor, no atomics; inadmissible -> r = -1
                const int64_t base = (int64_t)b * a.cap + (int64_t)rs * PSH_SEG + PSH_L * lane;
#pragma unroll
                for (int i = 0; i < PSH_L; ++i) {
                    const bool ok = i < nvalid;
                    a.cand_d[base + i] = ok ? dist_from_acc(acc[i], xn) : __uint_as_float(PSH_INF_BITS);
                    a.cand_rt[base + i] = ok ? make_int2(r_global, t_lane + i) : make_int2(-1, -1);
                }
            } else {
                // CHEAP: keep unless provably rejected (NaN-safe: !(t > thr)); else the exact test
                const bool wave_hit = CHEAP ? __any(!(min16(acc) > thr)) : __any(min16(acc) < tau);
                if (wave_hit) {  // rare: ~1e-4 of the windows survive
                    // kept small on purpose (a rolled loop, one flush site): unrolling this
                    // path 16x costs the hot loop ~30 VGPRs and a wave of occupancy
                    unsigned hm = 0u;                            // bit i: window i admitted
#pragma unroll
                    for (int i = 0; i < PSH_L; ++i)
                        hm |= ((i < nvalid) && (CHEAP ? !(acc[i] > thr) : (acc[i] < tau))) ? (1u << i) : 0u;
#pragma unroll 1
                    for (int i = 0; i < PSH_L; ++i) {
                        bool hit = ((hm >> i) & 1u) != 0u;
                        if (!__ballot(hit)) continue;
                        float v;
                        if (CHEAP) {      // survivors of the cheap test: the exact chain decides
                            v = hit ? exact_one<(CHEAP ? WT : 20)>(tile, PSH_L * lane + i, x) : 0.0f;
                            hit = hit && (v < tau);
                        } else {
                            v = acc[0];
#pragma unroll
                            for (int j = 1; j < PSH_L; ++j) v = (i == j) ? acc[j] : v;   // i is wave-uniform
                        }
                        const unsigned long long mask = __ballot(hit);
                        if (!mask) continue;
                        const int nh = __popcll(mask);
                        if (npend + nh > PSH_PEND) {           // buffer full: write it out now
                            pend_flush(pend, npend, lcount, a, lane);
                            npend = 0;
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
        }
        wave_lds_fence();  // all lanes done with the tile before it is overwritten
#ifdef PSH_PHASE_TIMING
        {
            const unsigned long long tp4 = __builtin_readcyclecounter();
            ph[0] += tp1 - tp0; ph[1] += tp2 - tp1; ph[2] += tp3 - tp2; ph[3] += tp4 - tp3; ph[4] += 1;
        }
#endif
        u = un;
    }
#ifdef PSH_PHASE_TIMING
    if (a.dbg_times && lane == 0) for (int i = 0; i < 5; ++i) a.dbg_times[2 * 8192 + 5 * gw + i] = ph[i];
#endif
    if (a.dbg_times && lane == 0) a.dbg_times[2 * gw + 1] = wall_clock64();
    if (MODE == PSH_MODE_FILTER) {
        if (npend > 0) pend_flush(pend, npend, lcount, a, lane);
        __syncthreads();
        for (int q = (int)threadIdx.x; q < a.B; q += PSH_SCAN_THREADS)
            a.bcount[(int64_t)q * PSH_MAX_BLOCKS + blockIdx.x] = lcount[q];
    }
    if (MODE == PSH_MODE_BOOT && a.blockmax) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) wmax = fmaxf(wmax, __shfl_xor(wmax, off, 64));
        if (lane == 0) atomicMax(reinterpret_cast<unsigned*>(next_unit + 1), __float_as_uint(wmax));   // >= 0: bits order as values
        __syncthreads();
        if (threadIdx.x == 0) a.blockmax[blockIdx.x] = __uint_as_float((unsigned)next_unit[1]);
    }
}

// ----------------------------------------------------------------------------------
// the cheap test on the matrix cores (single query, compile-time W <= 33)
// ----------------------------------------------------------------------------------
// Exact fp32 costs 41 VALU lane-operations per window against 4 bytes of HBM traffic:
// at 8 TB/s that is 82 T lane-ops/s, more than the vector ALUs deliver, so the scan is
// co-limited by the VALU even with the 25-operation bound-then-verify test above (measured:
// 71 us of VALU issue against 67 us of HBM time, 107 us together).  The REJECTION test does
// not need fp32: any rigorous lower bound of acc will do, and the survivors (~1e-4 of the
// windows) are re-evaluated with the exact chain anyway.  So the bound is evaluated where
// the chip has 16x the arithmetic: as a banded (Toeplitz) product on the MFMA units, on
// f16 copies of the data scaled by a power of two s (exact) into f16 range:
//     t^ = sum_j (y~_j^2)^ * 1  +  sum_j y^_j * (-2 x^_j)         y~ = 2^s y,  x~ = 2^s x
// One v_mfma_f32_32x32x16_f16 group covers the 1024 windows of a wave segment: row m of A
// is the 64 consecutive values y^[32m .. 32m+63], column n of B is the query shifted down
// by n (B[k][n] = -2 x^[k-n] for 0 <= k-n < W), so C[m][n] belongs to window 32m + n.  The
// window energy comes from the same instruction with A = (y~^2)^ and B = the band of ones.
// 8 MFMAs per segment, 256 matrix-core cycles; the VALU only converts (60 instructions per
// lane and segment instead of 430).
//
// Error bound (u = 2^-11 f16 round-to-nearest, eta = 2^-25 half the smallest f16
// subnormal -- MFMA keeps subnormal inputs, products are exact in the fp32 accumulator,
// <= 128 fp32 additions): with real ny~ = sum y~^2, nx~ = sum x~^2, t~ = ny~ - 2 sum x~ y~
//     |t^ - t~| <= (3u + 2^-15)(nx~ + ny~) + 41 * 2^-24  <=  a (nx~ + ny~) + b,
//     a = 2^-9, b = 2^-18        (tools/ubench_mfma_filter.hip measures 0.19 of it)
// and ny~ <= 2 (acc~ + nx~), so  acc~ (1 + 2a) >= nx~ (1 - 3a) + t^ - b:  a window with
//     t^ > mx_thr := tau~ (1 + 2^-17)(1 + 2a) - nx~ (1 - 3a) + b
// has a real acc above tau (1 + 2^-17), hence an fp32 chain value >= tau: it could not be
// admitted and is skipped.  Everything else is handed to exact_one().
// Values beyond f16 range need no special path.  The scale puts the largest |value| of the
// bootstrap rows and of the query into [4, 8), and tau is an acc of a bootstrap window, so
// tau~ <= 20 (8 + 8)^2 = 5120.  An unsampled outlier with y~^2 >= 65520 (|y~| > 255)
// converts to +inf; a window that contains it has acc~ >= (255 - 8)^2 > tau~ and is
// rightly rejected when its t^ comes out +inf, and is kept for the exact recheck when it
// comes out NaN (inf * 0 from the zero part of the band, inf - inf): both are correct.
// ----------------------------------------------------------------------------------
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

template <int WT, bool ALIGNED>
__global__ __launch_bounds__(PSH_SCAN_THREADS) void scan_mx_kernel(ScanArgs a) {
    static_assert(WT >= 0 && WT <= 33, "the shifted-query band must fit K = 64 (WT = 0: run-time W <= 33)");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NW = PSH_SCAN_THREADS / 64;
    const int lane = lane_id();
    const int wave_in_block = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    float* tile = smem + (size_t)wave_in_block * a.tile_floats;          // fp32 values: the exact recheck reads these
    int* lcount = reinterpret_cast<int*>(smem + (size_t)NW * a.tile_floats);
    int* next_unit = lcount + 4;                                          // B == 1: lcount[0] is the only cursor
    u32x4* pend0 = reinterpret_cast<u32x4*>(lcount + 8);
    u32x4* pend = pend0 + (size_t)wave_in_block * PSH_MX_PEND;
    _Float16* ah = reinterpret_cast<_Float16*>(pend0 + (size_t)NW * PSH_MX_PEND) + (size_t)wave_in_block * 2 * PSH_MX_NHALF;
    _Float16* a1 = ah;                                                    // y^
    _Float16* a2 = ah + PSH_MX_NHALF;                                     // (y~^2)^
    int npend = 0;
    const int gw_dbg = (int)blockIdx.x * NW + wave_in_block;
    if (a.dbg_times && lane == 0) a.dbg_times[2 * gw_dbg] = wall_clock64();   // tuning aid (tools/wave_times.py)
    if (threadIdx.x == 0) { *next_unit = 0; lcount[0] = 0; lcount[1] = 0; }
    {   // the tail slots no segment ever writes must hold finite values (0 * NaN poisons a row)
        unsigned* z = reinterpret_cast<unsigned*>(ah);
        for (int i = lane; i < PSH_MX_NHALF; i += 64) z[i] = 0u;          // 2 arrays x NHALF halves = NHALF dwords
    }
    __syncthreads();

    const int W = WT > 0 ? WT : a.W;
    const int nfloat = PSH_SEG + W - 1;
    const unsigned n_rs = (unsigned)a.n_rows * (unsigned)a.nseg;
    const unsigned u_lo = (unsigned)(((unsigned long long)n_rs * blockIdx.x) / gridDim.x);
    const unsigned u_hi = (unsigned)(((unsigned long long)n_rs * (blockIdx.x + 1)) / gridDim.x);
    const const_f32p x = (const_f32p)a.queries;
    typedef const __attribute__((address_space(4))) QueryState* const_qsp;
    const const_qsp qs = (const_qsp)a.qstate;
    const float tau = __uint_as_float(qs[0].tau_bits);
    const float tau2 = a.bcount2 ? __uint_as_float(qs[0].tau2_bits) : tau;   // no second class without its counters
    const float scale = qs[0].mx_scale;
    const float thr = qs[0].mx_thr;
    const float thr2 = a.bcount2 ? qs[0].mx_thr2 : thr;                      // <= thr: what cannot be below tau2
    // (if the threshold kernel could not arm the filter -- absurd magnitudes -- scale is 0 and
    // thr +inf: nothing is rejected, every window goes through exact_one: slow, still exact)

    // B fragments: lane (n = lane & 31, hk = lane >> 5) holds k = 16 s + 8 hk + i, i < 8
    f16x8 bx[4], bo[4];
    {
        const int n = lane & 31, hk = lane >> 5;
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int j = 16 * s + 8 * hk + i - n;
                const bool in = j >= 0 && j < W;
                const float xv = x[in ? j : 0];
                bx[s][i] = (_Float16)(in ? -2.0f * (xv * scale) : 0.0f);
                bo[s][i] = (_Float16)(in ? 1.0f : 0.0f);
            }
    }

    auto grab = [&]() -> unsigned {
        int v = 0;
        if (lane == 0) v = atomicAdd(next_unit, 1);
        return u_lo + (unsigned)__builtin_amdgcn_readfirstlane(v);
    };
    auto decode = [&](unsigned uu, unsigned& ri, unsigned& sg) {
        ri = fast_div(uu, a.magic_nseg, (unsigned)a.nseg);
        sg = uu - ri * (unsigned)a.nseg;
    };

    const float xn = qs[0].xn;
    // candidate append of this kernel: one query, its norm in an SGPR -- stores only, so that
    // nothing issued here ever has to be waited for together with a prefetch (vmcnt is in order)
    auto flush = [&]() {
        wave_lds_fence();
        if (lane < npend) {
            const u32x4 e = pend[lane];
            // below tau2 (where the k-th smallest is expected, times two): front of the slice; the rest of
            // what tau admits: back of the slice, read only if the front lists hold fewer than k
            // (a window the cheap test could only place above tau2 arrives unverified -- marker instead of
            // its acc: the exact chain is spent on it only if the selection ever has to read the back lists)
            const bool verified = e[0] != PSH_UNVERIFIED_BITS;
            const bool front = verified && __uint_as_float(e[0]) < tau2;
            const int pos = atomicAdd(&lcount[front ? 0 : 1], 1);
            if (pos < a.slice) {
                const int64_t o = (int64_t)blockIdx.x * a.slice + (front ? pos : a.slice - 1 - pos);
                a.cand_d[o] = verified ? dist_from_acc(__uint_as_float(e[0]), xn) : __uint_as_float(PSH_UNVERIFIED_BITS);
                a.cand_rt[o] = make_int2((int)e[1], (int)e[2]);
            }
        }
        npend = 0;
    };
    auto load_unit = [&](Stage& sx, unsigned uu) {
        unsigned ri, sg;
        decode(uu, ri, sg);
        stage_load<ALIGNED>(sx, a.dataset + (a.row0 + (int64_t)ri * a.row_stride) * a.T, a.T, (int)sg * PSH_SEG, nfloat, lane);
    };
    // one segment: `cur` holds its values; returns the unit whose load now occupies `cur`
    auto process = [&](Stage& cur, unsigned ucur) -> unsigned {
        unsigned ri, sg;
        decode(ucur, ri, sg);
        const int64_t row = a.row0 + (int64_t)ri * a.row_stride;
        const int seg_start = (int)sg * PSH_SEG;

        stage_store(cur, tile, nfloat, lane);
        {   // the f16 copies: y^ and (y~^2)^, 4 values = one 8-byte store per array and chunk
            const int nq = (nfloat + 3) >> 2;
#pragma unroll
            for (int q = 0; q < PSH_NSTAGE; ++q) {
                const int m = lane + 64 * q;
                if (q < PSH_NSTAGE - 1 || m < nq) {
                    const f32x4 v = cur.v[q] * scale;
                    const f32x4 v2 = v * v;
                    *reinterpret_cast<f16x4*>(a1 + mx_half(4 * m)) = __builtin_convertvector(v, f16x4);
                    *reinterpret_cast<f16x4*>(a2 + mx_half(4 * m)) = __builtin_convertvector(v2, f16x4);
                }
            }
        }
        wave_lds_fence();
        if (npend > 0) flush();   // last segment's admissions, ahead of the prefetch
        const unsigned un = grab();
        if (un < u_hi) load_unit(cur, un);

        const int r_global = (int)(row + a.r_offset);
        auto push = [&](bool hit, float v, int t) {      // wave-uniform control flow
            const unsigned long long mask = __ballot(hit);
            if (!mask) return;
            const int nh = __popcll(mask);
            if (npend + nh > PSH_MX_PEND) {
                flush();
                wave_lds_fence();
            }
            if (hit) {
                const int slot = npend + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32),
                                             __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
                pend[slot] = u32x4{__float_as_uint(v), (unsigned)r_global, (unsigned)t, 0u};
            }
            npend += nh;
        };

        {
            const int m = lane & 31, hk = lane >> 5;
            f16x8 fa[4];                                   // four A fragments per LDS round trip
            f32x16 acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
#pragma unroll
            for (int s = 0; s < 4; ++s) fa[s] = *reinterpret_cast<const f16x8*>(a2 + mx_half(32 * m + 16 * s + 8 * hk));
#pragma unroll
            for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[s], bo[s], acc, 0, 0, 0);
#pragma unroll
            for (int s = 0; s < 4; ++s) fa[s] = *reinterpret_cast<const f16x8*>(a1 + mx_half(32 * m + 16 * s + 8 * hk));
#pragma unroll
            for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[s], bx[s], acc, 0, 0, 0);
            bool keep = false;                             // NaN-safe: !(t^ > thr)
#pragma unroll
            for (int r = 0; r < 16; ++r) keep = keep || !(acc[r] > thr);
            if (__any(keep)) {                             // about one segment in four
                unsigned hm = 0u, hm2 = 0u;                // bit r: window of accumulator r survives tau / tau2
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    hm |= !(acc[r] > thr) ? (1u << r) : 0u;
                    hm2 |= !(acc[r] > thr2) ? (1u << r) : 0u;
                }
#pragma unroll 1
                for (int r = 0; r < 16; ++r) {
                    const int p = 32 * ((r & 3) + 8 * (r >> 2) + 4 * hk) + m;      // C layout: row -> window
                    const bool hit = (((hm >> r) & 1u) != 0u) && (seg_start + p < a.Tp);
                    if (!__ballot(hit)) continue;
                    // the exact chain only where the window may still be below tau2; the others are admitted
                    // unverified (they can only matter if the front lists end up short of k)
                    const bool may2 = hit && (((hm2 >> r) & 1u) != 0u);
                    float v = __uint_as_float(PSH_UNVERIFIED_BITS);
                    if (__ballot(may2)) {
                        if (may2) { if constexpr (WT > 0) v = exact_one<(WT > 0 ? WT : 20)>(tile, p, x); else v = exact_one_rt(tile, p, x, W); }
                    }
                    push(hit && (!may2 || v < tau), v, seg_start + p);
                }
            }
        }
        wave_lds_fence();  // all lanes done with the tile before it is overwritten
        return un;
    };

    // one segment in flight per wave besides the one being processed.  (Two in flight --
    // a second Stage, consumed one iteration later -- was measured: 89.6 us against 82.9.)
    Stage st;
    unsigned u = grab();
    if (u < u_hi) load_unit(st, u);
    while (u < u_hi) u = process(st, u);
    if (a.dbg_times && lane == 0) a.dbg_times[2 * gw_dbg + 1] = wall_clock64();
    if (npend > 0) flush();
    __syncthreads();
    if (threadIdx.x == 0) {
        a.bcount[blockIdx.x] = lcount[0];
        if (a.bcount2) a.bcount2[blockIdx.x] = lcount[1];
    }
}

// ----------------------------------------------------------------------------------
// the matrix-core rejection test for BATCHED queries (BASELINE configs[2]: W <= 25)
// ----------------------------------------------------------------------------------
// Same bound, same exact recheck as scan_mx_kernel; the banded product is laid out for
// several queries: the N dimension holds 4 queries x 8 shifts, row m of A is
// y^[256 g + 8 m .. + 31] (g = 0..3 covers the segment), K = 32.  The window energies (A =
// y~^2, B = band of ones) are computed once per segment into 4 accumulator tiles that seed
// the 2 MFMAs per tile of every query group: 2 MFMAs and ~25 VALU instructions per query
// and segment against 430 VALU instructions for the test on the vector ALUs.
// Blocks of 8 waves (2 per SIMD: the tiles and the A fragments of a segment stay in
// registers, 256 VGPRs); blockIdx.y selects a chunk of PSH_MQ_CHUNK queries whose B
// fragments (built by the threshold kernel, one common power-of-two scale) and thresholds
// sit in LDS.  A segment holding a value beyond f16 range keeps everything (exact path).
template <int MODE>
__device__ __forceinline__ void emit16(const ScanArgs& a, int b, const float (&acc)[PSH_L], int nvalid, int lane,
                                       unsigned rs, int r_global, int t_lane, float tau, float xn,
                                       u32x4* pend, int& npend, int* lcount);      // defined with the embedded scan below

#define PSH_MQ_THREADS 512
#define PSH_MQ_CHUNK 112          // queries per block pass (their fragments, thresholds and values sit in LDS)
#define PSH_MQ_QCAP 192           // survivors of the cheap test queued per wave before a dense exact pass

template <int WT, bool ALIGNED>
__global__ __launch_bounds__(PSH_MQ_THREADS) void scan_mq_kernel(ScanArgs a) {
    static_assert(WT >= 0 && WT <= 25, "query + 7 shifts must fit K = 32 (WT = 0: run-time W <= 25)");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NW = PSH_MQ_THREADS / 64;
    const int lane = lane_id();
    const int wave_in_block = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    float* tile = smem + (size_t)wave_in_block * a.tile_floats;
    int* lcount = reinterpret_cast<int*>(smem + (size_t)NW * a.tile_floats);
    int* next_unit = lcount + ((a.B + 3) & ~3);
    u32x4* pend0 = reinterpret_cast<u32x4*>(next_unit + 4);
    u32x4* pend = pend0 + (size_t)wave_in_block * PSH_PEND;
    _Float16* hbase = reinterpret_cast<_Float16*>(pend0 + (size_t)NW * PSH_PEND);
    _Float16* a1 = hbase + (size_t)wave_in_block * 2 * PSH_MX_NHALF;      // y^
    _Float16* a2 = a1 + PSH_MX_NHALF;                                      // (y~^2)^
    _Float16* fragL = hbase + (size_t)NW * 2 * PSH_MX_NHALF;               // [group][K-step][lane] x 8 halves
    float* thrL = reinterpret_cast<float*>(fragL + (size_t)(PSH_MQ_CHUNK / 4) * 2 * 64 * 8);
    float* tauL = thrL + PSH_MQ_CHUNK;
    float* xL = tauL + PSH_MQ_CHUNK;                                       // the chunk's queries: the exact recheck reads them
    unsigned* sq = reinterpret_cast<unsigned*>(xL + PSH_MQ_CHUNK * 25) + (size_t)wave_in_block * PSH_MQ_QCAP;   // survivor queue
    const int W = WT > 0 ? WT : a.W;
    int npend = 0;
    int nsq = 0;

    const int q0 = (int)blockIdx.y * PSH_MQ_CHUNK;                         // this block's queries: [q0, q0 + nq)
    const int nq = (a.B - q0) < PSH_MQ_CHUNK ? (a.B - q0) : PSH_MQ_CHUNK;
    const int ngroups = (nq + 3) >> 2;
    if (threadIdx.x == 0) *next_unit = 0;
    for (int q = (int)threadIdx.x; q < a.B; q += PSH_MQ_THREADS) lcount[q] = 0;
    {
        unsigned* z = reinterpret_cast<unsigned*>(a1);
        for (int i = lane; i < PSH_MX_NHALF; i += 64) z[i] = 0u;
        const f32x4* src = reinterpret_cast<const f32x4*>(a.mq_frag) + (size_t)(q0 >> 2) * 2 * 64;   // 16 bytes = 8 halves
        f32x4* dst = reinterpret_cast<f32x4*>(fragL);
        for (int i = (int)threadIdx.x; i < ngroups * 2 * 64; i += PSH_MQ_THREADS) {
            // lanes of queries past the end of the batch: zero fragments
            const int ln = i & 63, qq = 4 * (i >> 7) + ((ln & 31) >> 3);
            dst[i] = qq < nq ? src[i] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        for (int i = (int)threadIdx.x; i < PSH_MQ_CHUNK; i += PSH_MQ_THREADS) {
            thrL[i] = i < nq ? a.qstate[q0 + i].mx_thr : -__uint_as_float(PSH_INF_BITS);   // -inf: reject everything
            tauL[i] = i < nq ? __uint_as_float(a.qstate[q0 + i].tau_bits) : 0.0f;
        }
        for (int i = (int)threadIdx.x; i < nq * W; i += PSH_MQ_THREADS) xL[i] = a.queries[(int64_t)q0 * W + i];
    }
    __syncthreads();

    const int nfloat = PSH_SEG + W - 1;
    const unsigned n_rs = (unsigned)a.n_rows * (unsigned)a.nseg;
    const unsigned u_lo = (unsigned)(((unsigned long long)n_rs * blockIdx.x) / gridDim.x);
    const unsigned u_hi = (unsigned)(((unsigned long long)n_rs * (blockIdx.x + 1)) / gridDim.x);
    typedef const __attribute__((address_space(4))) QueryState* const_qsp;
    const float scale = ((const_qsp)a.qstate)[0].mx_scale;                // one scale for the whole batch
    const int n = lane & 31, hk = lane >> 5, qsub = n >> 3, shift = n & 7;

    f16x8 bo[2];                                                           // the band of ones
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int j = 16 * s + 8 * hk + i - shift;
            bo[s][i] = (_Float16)((j >= 0 && j < W) ? 1.0f : 0.0f);
        }

    auto grab = [&]() -> unsigned {
        int v = 0;
        if (lane == 0) v = atomicAdd(next_unit, 1);
        return u_lo + (unsigned)__builtin_amdgcn_readfirstlane(v);
    };
    auto load_unit = [&](Stage& sx, unsigned uu) {
        const unsigned ri = fast_div(uu, a.magic_nseg, (unsigned)a.nseg);
        const unsigned sg = uu - ri * (unsigned)a.nseg;
        stage_load<ALIGNED>(sx, a.dataset + (a.row0 + (int64_t)ri * a.row_stride) * a.T, a.T, (int)sg * PSH_SEG, nfloat, lane);
    };

    Stage st;
    unsigned u = grab();
    if (u < u_hi) load_unit(st, u);
    while (u < u_hi) {
        const unsigned ri = fast_div(u, a.magic_nseg, (unsigned)a.nseg);
        const unsigned sg = u - ri * (unsigned)a.nseg;
        const int64_t row = a.row0 + (int64_t)ri * a.row_stride;
        const int seg_start = (int)sg * PSH_SEG;
        const int r_global = (int)(row + a.r_offset);

        stage_store(st, tile, nfloat, lane);
        float lmax = 0.0f;
        {
            const int nqd = (nfloat + 3) >> 2;
#pragma unroll
            for (int q = 0; q < PSH_NSTAGE; ++q) {
                const int m = lane + 64 * q;
                if (q < PSH_NSTAGE - 1 || m < nqd) {
                    const f32x4 v = st.v[q] * scale;
                    const f32x4 v2 = v * v;
                    lmax = fmaxf(fmaxf(lmax, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
                    *reinterpret_cast<f16x4*>(a1 + mx_half(4 * m)) = __builtin_convertvector(v, f16x4);
                    *reinterpret_cast<f16x4*>(a2 + mx_half(4 * m)) = __builtin_convertvector(v2, f16x4);
                }
            }
        }
        wave_lds_fence();
        if (npend > 0) { pend_flush(pend, npend, lcount, a, lane); npend = 0; }
        const unsigned un = grab();
        if (un < u_hi) load_unit(st, un);
        // a value beyond f16 range (or no armed filter): nothing may be rejected in this segment
        const bool keep_all = __any(!(lmax <= 128.0f)) || !(scale > 0.0f);

        // window energies of the 4 row groups, and the y^ fragments, once per segment
        f32x16 ny[4];
        f16x8 fy[4][2];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f16x8 e0 = *reinterpret_cast<const f16x8*>(a2 + mx_half(256 * g + 8 * n + 8 * hk));
            const f16x8 e1 = *reinterpret_cast<const f16x8*>(a2 + mx_half(256 * g + 8 * n + 16 + 8 * hk));
            fy[g][0] = *reinterpret_cast<const f16x8*>(a1 + mx_half(256 * g + 8 * n + 8 * hk));
            fy[g][1] = *reinterpret_cast<const f16x8*>(a1 + mx_half(256 * g + 8 * n + 16 + 8 * hk));
#pragma unroll
            for (int i = 0; i < 16; ++i) ny[g][i] = 0.0f;
            ny[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(e0, bo[0], ny[g], 0, 0, 0);
            ny[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(e1, bo[1], ny[g], 0, 0, 0);
        }

        // the exact chain for the queued survivors, one per lane
        auto drain = [&]() {
            wave_lds_fence();                                              // other lanes' queue entries
            while (nsq > 0) {
                const int m = nsq < 64 ? nsq : 64;
                nsq -= m;
                bool hit = lane < m;
                const unsigned e = hit ? sq[nsq + lane] : 0u;
                const int p = (int)(e & 0xffffu), ql2 = (int)(e >> 16);
                hit = hit && (seg_start + p < a.Tp);
                float v = 0.0f;
                if (hit) {
                    const float* xq = xL + ql2 * W;
#pragma unroll
                    for (int j2 = 0; j2 < W; ++j2) {
                        const float D = __fsub_rn(xq[j2], tile[lds_pad(p + j2)]);
                        v = __builtin_fmaf(D, D, v);
                    }
                    hit = v < tauL[ql2];
                }
                const unsigned long long mask = __ballot(hit);
                if (!mask) continue;
                const int nh = __popcll(mask);
                if (npend + nh > PSH_PEND) {
                    pend_flush(pend, npend, lcount, a, lane);
                    npend = 0;
                    wave_lds_fence();
                }
                if (hit) {
                    const int slot = npend + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32),
                                                 __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
                    pend[slot] = u32x4{__float_as_uint(v), (unsigned)r_global, (unsigned)(seg_start + p), (unsigned)(q0 + ql2)};
                }
                npend += nh;
            }
            wave_lds_fence();                                              // queue slots are reused
        };

#pragma unroll 1
        for (int G = 0; G < ngroups; ++G) {
            const f16x8 b0 = *reinterpret_cast<const f16x8*>(fragL + ((size_t)(2 * G + 0) * 64 + lane) * 8);
            const f16x8 b1 = *reinterpret_cast<const f16x8*>(fragL + ((size_t)(2 * G + 1) * 64 + lane) * 8);
            const int ql = 4 * G + qsub;                                   // this lane's query within the chunk
            const float thr = keep_all ? __uint_as_float(PSH_INF_BITS) : thrL[ql];
            // all 8 MFMAs of the group first (4 independent accumulator tiles), then the tests:
            // a test-and-branch per tile serialises MFMA latency, min tree and branch 4 times
            f32x16 acc[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fy[g][0], b0, ny[g], 0, 0, 0);
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fy[g][1], b1, acc[g], 0, 0, 0);
            float mn[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                mn[g] = tile_min16(acc[g]);
            }
            // values are finite here unless keep_all (then thr = +inf keeps NaN too)
            if (!__any(!(min3f(min3f(mn[0], mn[1], mn[2]), mn[3], mn[3]) > thr))) continue;
            // survivors are only QUEUED here (window, query): a lane-by-lane exact chain would run ~140
            // instructions for the one or two lanes that hold a survivor; the queue is drained 64 at a time
            const bool lane_ok = ql < nq;
            const int nsq0 = nsq;
            bool full = false;                                             // wave-uniform
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (!__any(!(mn[g] > thr))) continue;
                // the survivors of this tile as a per-lane bit mask (pure VALU), then one queue round per
                // survivor of the busiest lane (usually one): a ballot per accumulator register would put 16
                // VALU -> SALU round trips on every tile that holds a survivor
                unsigned hm = 0u;
#pragma unroll
                for (int r = 0; r < 16; ++r) hm |= !(acc[g][r] > thr) ? (1u << r) : 0u;
                if (!lane_ok) hm = 0u;
                for (;;) {
                    const bool act = hm != 0u;
                    const unsigned long long M = __ballot(act);
                    if (!M) break;
                    const int nh = __popcll(M);
                    if (nsq + nh > PSH_MQ_QCAP) { full = true; break; }
                    if (act) {
                        const int r = (int)__builtin_ctz(hm);
                        hm &= hm - 1u;
                        const int p = 256 * g + 8 * ((r & 3) + 8 * (r >> 2) + 4 * hk) + shift;
                        const int slot = nsq + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(M >> 32),
                                                     __builtin_amdgcn_mbcnt_lo((unsigned)M, 0u));
                        sq[slot] = ((unsigned)ql << 16) | (unsigned)p;
                    }
                    nsq += nh;
                }
            }
            if (full) {
                // more survivors in one group than the queue holds (massive near-ties, or nothing may be
                // rejected in this segment): forget the group's entries and run its queries exactly
                nsq = nsq0;
                const int t_lane = seg_start + PSH_L * lane;
                int nvalid = a.Tp - t_lane;
                nvalid = nvalid < 0 ? 0 : (nvalid > PSH_L ? PSH_L : nvalid);
#pragma unroll 1
                for (int c = 0; c < 4; ++c) {
                    const int ql2 = 4 * G + c;
                    if (ql2 >= nq) break;
                    float accv[PSH_L];
                    accumulate16<WT>(tile, lane, (const_f32p)a.queries + (int64_t)(q0 + ql2) * W, W, accv);
                    emit16<PSH_MODE_FILTER>(a, q0 + ql2, accv, nvalid, lane, u, r_global, t_lane, tauL[ql2], 0.0f, pend, npend, lcount);
                }
            }
            if (nsq >= 64) drain();                                        // the ONE in-loop call site (the body is ~200 instructions)
        }
        if (nsq > 0) drain();
        wave_lds_fence();  // all lanes done with the tile before it is overwritten
        u = un;
    }
    if (npend > 0) pend_flush(pend, npend, lcount, a, lane);
    __syncthreads();
    for (int q = q0 + (int)threadIdx.x; q < q0 + nq; q += PSH_MQ_THREADS)          // this block's queries only
        a.bcount[(int64_t)q * PSH_MAX_BLOCKS + blockIdx.x] = lcount[q];
}

// ----------------------------------------------------------------------------------
// the bootstrap on the matrix cores: upper bounds instead of exact minima
// ----------------------------------------------------------------------------------
// tau only has to be an upper bound of the k-th smallest acc, and the f16 product that
// rejects windows in the full scan bounds acc from ABOVE just as rigorously:
//     acc~ (1 - 2a) <= nx~ (1 + 3a) + t^ + b
// so the minimum of that bound over a segment is an acc-or-more of one particular window of
// the segment, and the k-th smallest of those minima still has k windows at or below it.
// Same layout as scan_mq_kernel (4 queries x 8 shifts; a single query rides in a group of
// its own), the scale comes from the queries alone (the bootstrap runs before anything is
// known about the data): a segment holding |y~| > 128 falls back to the exact chain.
// Also records the largest |y| per block for the scale of the full scan.
template <int WT, bool ALIGNED>
__global__ __launch_bounds__(PSH_MQ_THREADS) void boot_mq_kernel(ScanArgs a) {
    static_assert(WT >= 0 && WT <= 25, "query + 7 shifts must fit K = 32 (WT = 0: run-time W <= 25)");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NW = PSH_MQ_THREADS / 64;
    const int lane = lane_id();
    const int wave_in_block = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    float* tile = smem + (size_t)wave_in_block * a.tile_floats;
    int* next_unit = reinterpret_cast<int*>(smem + (size_t)NW * a.tile_floats);      // [0] cursor, [1] max bits, [2] query max bits
    _Float16* hbase = reinterpret_cast<_Float16*>(next_unit + 4);
    _Float16* a1 = hbase + (size_t)wave_in_block * 2 * PSH_MX_NHALF;
    _Float16* a2 = a1 + PSH_MX_NHALF;
    _Float16* fragL = hbase + (size_t)NW * 2 * PSH_MX_NHALF;
    float* nxL = reinterpret_cast<float*>(fragL + (size_t)(PSH_MQ_CHUNK / 4) * 2 * 64 * 8);   // nx~ per query of the chunk

    const int W = WT > 0 ? WT : a.W;
    const int q0 = (int)blockIdx.y * PSH_MQ_CHUNK;
    const int nq = (a.B - q0) < PSH_MQ_CHUNK ? (a.B - q0) : PSH_MQ_CHUNK;
    const int ngroups = (nq + 3) >> 2;
    if (threadIdx.x == 0) { next_unit[0] = 0; next_unit[1] = 0; next_unit[2] = 0; }
    {
        unsigned* z = reinterpret_cast<unsigned*>(a1);
        for (int i = lane; i < PSH_MX_NHALF; i += 64) z[i] = 0u;
    }
    __syncthreads();
    {   // scale: the largest |x| of the whole batch into [4, 8)
        unsigned mb = 0u;
        for (int64_t j = threadIdx.x; j < (int64_t)a.B * W; j += PSH_MQ_THREADS) mb = max(mb, __float_as_uint(fabsf(a.queries[j])));
        if (mb) atomicMax(reinterpret_cast<unsigned*>(next_unit + 2), mb);
    }
    __syncthreads();
    const unsigned qmaxbits = (unsigned)next_unit[2];
    const int sexp = 3 - ((int)((qmaxbits >> 23) & 255u) - 126);
    const bool sane = sexp <= 60 && sexp >= -60 && qmaxbits >= 0x00800000u && qmaxbits < PSH_INF_BITS;
    const float scale = sane ? __uint_as_float((unsigned)(127 + sexp) << 23) : 0.0f;     // 0: exact chain everywhere
    const float unscale2 = sane ? __uint_as_float((unsigned)(127 - 2 * sexp) << 23) : 0.0f;
    for (int i = (int)threadIdx.x; i < ngroups * 2 * 64 * 8; i += PSH_MQ_THREADS) {
        // fragment table entry i = ((2 G + s) * 64 + lane) * 8 + e
        const int e = i & 7, ln = (i >> 3) & 63, s2 = (i >> 9) & 1, G = i >> 10;
        const int hk = ln >> 5, qsub = (ln & 31) >> 3, shift = ln & 7;
        const int j = 16 * s2 + 8 * hk + e - shift, ql = 4 * G + qsub;
        const bool in = j >= 0 && j < W && ql < nq;
        const float xv = in ? a.queries[(int64_t)(q0 + ql) * W + j] : 0.0f;
        fragL[i] = (_Float16)(in ? -2.0f * (xv * scale) : 0.0f);
    }
    for (int i = (int)threadIdx.x; i < PSH_MQ_CHUNK; i += PSH_MQ_THREADS) {
        float s = 0.0f;
        if (i < nq)
            for (int j = 0; j < W; ++j) { const float v = a.queries[(int64_t)(q0 + i) * W + j] * scale; s = __builtin_fmaf(v, v, s); }
        nxL[i] = s;
    }
    __syncthreads();

    const int nfloat = PSH_SEG + W - 1;
    const unsigned n_rs = (unsigned)a.n_rows * (unsigned)a.nseg;
    const unsigned u_lo = (unsigned)(((unsigned long long)n_rs * blockIdx.x) / gridDim.x);
    const unsigned u_hi = (unsigned)(((unsigned long long)n_rs * (blockIdx.x + 1)) / gridDim.x);
    const int n = lane & 31, hk = lane >> 5, qsub = n >> 3, shift = n & 7;
    const const_f32p xk = (const_f32p)a.queries;
    // acc~ <= (nx~ (1 + 3a) + t^ + b) / (1 - 2a), a = 2^-9, b = 2^-18; constants rounded up, fp32 slack included
    const float C1 = 1.0f + 3.0f / 512.0f + 1.0f / 65536.0f, C2 = (1.0f / (1.0f - 2.0f / 512.0f)) * (1.0f + 1.0f / 32768.0f);
    const float BB = 1.0f / 262144.0f;

    f16x8 bo[2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int j = 16 * s + 8 * hk + i - shift;
            bo[s][i] = (_Float16)((j >= 0 && j < W) ? 1.0f : 0.0f);
        }
    auto grab = [&]() -> unsigned {
        int v = 0;
        if (lane == 0) v = atomicAdd(next_unit, 1);
        return u_lo + (unsigned)__builtin_amdgcn_readfirstlane(v);
    };
    auto load_unit = [&](Stage& sx, unsigned uu) {
        const unsigned ri = fast_div(uu, a.magic_nseg, (unsigned)a.nseg);
        const unsigned sg = uu - ri * (unsigned)a.nseg;
        stage_load<ALIGNED>(sx, a.dataset + (a.row0 + (int64_t)ri * a.row_stride) * a.T, a.T, (int)sg * PSH_SEG, nfloat, lane);
    };

    float wmax = 0.0f;
    Stage st;
    unsigned u = grab();
    if (u < u_hi) load_unit(st, u);
    while (u < u_hi) {
        const unsigned ri = fast_div(u, a.magic_nseg, (unsigned)a.nseg);
        const unsigned sg = u - ri * (unsigned)a.nseg;
        const int seg_start = (int)sg * PSH_SEG;
        const bool ragged = seg_start + PSH_SEG > a.Tp;        // some windows of this segment are not admissible

        float lmax = 0.0f;
        {
            const int nqd = (nfloat + 3) >> 2;
#pragma unroll
            for (int q = 0; q < PSH_NSTAGE; ++q) {
                const int m = lane + 64 * q;
                if (q < PSH_NSTAGE - 1 || m < nqd) {
                    lmax = fmaxf(fmaxf(lmax, fmaxf(fabsf(st.v[q][0]), fabsf(st.v[q][1]))), fmaxf(fabsf(st.v[q][2]), fabsf(st.v[q][3])));
                    const f32x4 v = st.v[q] * scale;
                    const f32x4 v2 = v * v;
                    *reinterpret_cast<f16x4*>(a1 + mx_half(4 * m)) = __builtin_convertvector(v, f16x4);
                    *reinterpret_cast<f16x4*>(a2 + mx_half(4 * m)) = __builtin_convertvector(v2, f16x4);
                }
            }
        }
        wmax = fmaxf(wmax, lmax);
        const bool exact = __any(!(lmax * scale <= 128.0f)) || !(scale > 0.0f);   // beyond f16 range: exact chain
        if (exact) stage_store(st, tile, nfloat, lane);
        wave_lds_fence();
        const unsigned un = grab();
        if (un < u_hi) load_unit(st, un);

        if (!exact) {
            f32x16 ny[4];
            f16x8 fy[4][2];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f16x8 e0 = *reinterpret_cast<const f16x8*>(a2 + mx_half(256 * g + 8 * n + 8 * hk));
                const f16x8 e1 = *reinterpret_cast<const f16x8*>(a2 + mx_half(256 * g + 8 * n + 16 + 8 * hk));
                fy[g][0] = *reinterpret_cast<const f16x8*>(a1 + mx_half(256 * g + 8 * n + 8 * hk));
                fy[g][1] = *reinterpret_cast<const f16x8*>(a1 + mx_half(256 * g + 8 * n + 16 + 8 * hk));
#pragma unroll
                for (int i = 0; i < 16; ++i) ny[g][i] = 0.0f;
                ny[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(e0, bo[0], ny[g], 0, 0, 0);
                ny[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(e1, bo[1], ny[g], 0, 0, 0);
            }
#pragma unroll 1
            for (int G = 0; G < ngroups; ++G) {
                const f16x8 b0 = *reinterpret_cast<const f16x8*>(fragL + ((size_t)(2 * G + 0) * 64 + lane) * 8);
                const f16x8 b1 = *reinterpret_cast<const f16x8*>(fragL + ((size_t)(2 * G + 1) * 64 + lane) * 8);
                const int ql = 4 * G + qsub;
                float mn = __uint_as_float(PSH_INF_BITS);
                f32x16 acc[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fy[g][0], b0, ny[g], 0, 0, 0);
#pragma unroll
                for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fy[g][1], b1, acc[g], 0, 0, 0);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    if (ragged) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int p = 256 * g + 8 * ((r & 3) + 8 * (r >> 2) + 4 * hk) + shift;
                            acc[g][r] = (seg_start + p < a.Tp) ? acc[g][r] : __uint_as_float(PSH_INF_BITS);
                        }
                    }
                    float m2 = fminf(fminf(acc[g][0], acc[g][1]), acc[g][2]);
#pragma unroll
                    for (int i = 3; i + 1 < 16; i += 2) m2 = fminf(fminf(m2, acc[g][i]), acc[g][i + 1]);
                    mn = fminf(mn, fminf(m2, acc[g][15]));
                }
                // lanes of one query: 8 shifts x 2 halves
                mn = fminf(mn, __shfl_xor(mn, 1, 64));
                mn = fminf(mn, __shfl_xor(mn, 2, 64));
                mn = fminf(mn, __shfl_xor(mn, 4, 64));
                mn = fminf(mn, __shfl_xor(mn, 32, 64));
                if (shift == 0 && hk == 0 && ql < nq) {
                    const float ub = (__builtin_fmaf(nxL[ql], C1, mn) + BB) * C2;      // scaled units, >= acc~
                    a.minbuf[(int64_t)(q0 + ql) * a.min_stride + (int64_t)u] = ub * unscale2;
                }
            }
        } else {
            const int t_lane = seg_start + PSH_L * lane;
            int nvalid = a.Tp - t_lane;
            nvalid = nvalid < 0 ? 0 : (nvalid > PSH_L ? PSH_L : nvalid);
#pragma unroll 1
            for (int ql = 0; ql < nq; ++ql) {
                float acc[PSH_L];
                accumulate16<WT>(tile, lane, xk + (int64_t)(q0 + ql) * W, W, acc);
                float m = __uint_as_float(PSH_INF_BITS);
#pragma unroll
                for (int i = 0; i < PSH_L; ++i) m = (i < nvalid) ? fminf(m, acc[i]) : m;
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) m = fminf(m, __shfl_xor(m, off, 64));
                if (lane == 0) a.minbuf[(int64_t)(q0 + ql) * a.min_stride + (int64_t)u] = m;
            }
        }
        wave_lds_fence();
        u = un;
    }
    if (a.blockmax) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) wmax = fmaxf(wmax, __shfl_xor(wmax, off, 64));
        if (lane == 0) atomicMax(reinterpret_cast<unsigned*>(next_unit + 1), __float_as_uint(wmax));
        __syncthreads();
        if (threadIdx.x == 0) a.blockmax[blockIdx.y * gridDim.x + blockIdx.x] = __uint_as_float((unsigned)next_unit[1]);
    }
}

// ----------------------------------------------------------------------------------
// the embedded scan: a general linear embedding (Foveal, user kernels) in front of the
// distance -- reference path_embedding.py:117-132 (conv1d with a (d,1,K) kernel) feeding
// path_distance.py:62-65, i.e. for every window t of every row
//     hy_i = sum_j ker[i][j] * y[t + j]      (fma chain, increasing j)
//     acc  = sum_i (hx_i - hy_i)^2           (D = hx_i - hy_i rounded, fma chain, increasing i)
//     d    = sqrt(acc) / ||hx||
// The reference evaluates these sums in library-chosen orders (MKL-DNN / MIOpen conv1d,
// vectorised norm), so parity with it is a tolerance (1e-5 relative), not bit equality;
// the order above is the oracle's (oracle/psh_oracle.c: psh_oracle_scan_topk_embedded) and
// the kernel reproduces THAT bit for bit.
//
// Same skeleton as scan_kernel (one 16-wave block per CU, LDS work queue, wave-private
// padded tile, per-block candidate slices), but VALU-bound by a wide margin (d*K fma per
// window against 4 bytes), so segments are loaded synchronously and the registers go to
// the accumulators of PSH_EMB_BG queries that share one evaluation of the embedding.
// ----------------------------------------------------------------------------------
#define PSH_EMB_BG 3

__device__ __forceinline__ void corr16(float tap, const float (&win)[PSH_L], int jj, float (&c)[PSH_L]) {
    corr8(tap, win[(0 + jj) & 15], win[(1 + jj) & 15], win[(2 + jj) & 15], win[(3 + jj) & 15],
          win[(4 + jj) & 15], win[(5 + jj) & 15], win[(6 + jj) & 15], win[(7 + jj) & 15],
          c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7]);
    corr8(tap, win[(8 + jj) & 15], win[(9 + jj) & 15], win[(10 + jj) & 15], win[(11 + jj) & 15],
          win[(12 + jj) & 15], win[(13 + jj) & 15], win[(14 + jj) & 15], win[(15 + jj) & 15],
          c[8], c[9], c[10], c[11], c[12], c[13], c[14], c[15]);
}

// c_w = sum_{j < n} taps[j] * tile[base + w + j] for the 16 windows w of a lane.  taps: LDS,
// 16-byte aligned, readable (zero padded) up to the next multiple of 4 past n; base % 4 == 0.
// Every tile slot this reads was written by stage_store or by the zero fill at kernel
// start, so a zero tap never meets a non-finite stale value.
__device__ __forceinline__ void correlate16(const float* tile, int base, const float* taps, int n,
                                            float (&c)[PSH_L]) {
    float win[PSH_L];
#pragma unroll
    for (int q = 0; q < PSH_L / 4; ++q) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(tile + lds_pad(base + 4 * q));
        win[4 * q + 0] = v[0]; win[4 * q + 1] = v[1]; win[4 * q + 2] = v[2]; win[4 * q + 3] = v[3];
    }
#pragma unroll
    for (int i = 0; i < PSH_L; ++i) c[i] = 0.0f;
#pragma unroll 1
    for (int jb = 0; jb < n; jb += PSH_L) {
        const int rem = n - jb;                              // wave-uniform
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (4 * g < rem) {
                const f32x4 nx = *reinterpret_cast<const f32x4*>(tile + lds_pad(base + jb + PSH_L + 4 * g));
                const f32x4 tp = *reinterpret_cast<const f32x4*>(taps + jb + 4 * g);   // broadcast read
                const float nv[4] = {nx[0], nx[1], nx[2], nx[3]};
                const float tv[4] = {tp[0], tp[1], tp[2], tp[3]};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int jj = 4 * g + q;
                    corr16(tv[q], win, jj, c);
                    win[jj] = nv[q];
                }
            }
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

// ---- suffix rows (Foveal, reference path_embedding.py:142-172) -------------------------
// A Foveal kernel row is c_i on the LAST n_i taps and zero elsewhere (n_i = 1, 1, .., 2, .., 115
// for max_context 126): 865 fma per window through the dense chain above, but only 115 DISTINCT
// partial sums -- row i is c_i times the running sum of the window's newest n_i samples.  The
// kernel recognises the general form of that structure by itself, from the matrix it was given:
//     every row is one constant on  U & [a_i, K)  -- U the union of all supports --
// (an ImputationContext's gap only removes taps from U, so padded kernels qualify too), and then
// scans bound-then-verify like the Identity path:
//   cheap  : S <- running sum over the taps of U, newest sample first, ONE pass for all rows;
//            when the pass reaches a_i:  e_i = hx_i - c_i S;  acc^ += e_i^2   (~120 VALU ops/window)
//   bound  : both the cheap h^_i = c_i S and the exact chain's h_i carry at most
//            n_i u |c_i| sum|y| of rounding error (u = 2^-24, any summation order), so
//            |h^_i - h_i| <= 2 u |c_i| n_i^2 ymax  and, in the embedding space,
//            sqrt(acc) >= sqrt(acc^) - ymax * cerr   with cerr = 2u sqrt(sum_i (c_i (n_i+1)^2)^2)
//            (ymax = max |y| over the segment);  the sums of d squares add (d + 3) u relative
//   verify : a window survives unless  acc^ > (sqrt(tau)(1 + 2^-15) + ymax cerr)^2 (1 + 2^-14);
//            survivors (a few per million) get the exact dense chain, and only exact values are
//            ever ranked.  The bootstrap uses the same bound the other way round (upper bounds).
// Non-finite data needs no special path: NaN fails the '>' and is kept, an infinite ymax makes
// the threshold infinite (everything is verified exactly).
#define PSH_NEST_BG 2                // queries sharing one pass of running sums (register budget: 128 VGPRs)
#define PSH_NEST_MAX_K 256           // support masks are 4 x 64 bits, 16 blocks of 16 taps
struct NestHdr { int ok; int nops; float cerr; int n_empty; unsigned blk[16]; int ncl[PSH_NEST_MAX_K]; };   // blk: active taps | closing taps << 16; ncl: rows closing at a tap

// The running sums of a lane's 16 windows.  The window registers are addressed by DATA index:
// y[base + m] lives in slot m & 15, so at tap j (PH = j & 15) window w reads slot (w + PH) & 15,
// and the sample that enters for tap j - 1 replaces the one that leaves, in slot (PH - 1) & 15:
// walking the taps downwards in blocks of 16 makes every register index a compile-time constant.
// Slots s and s + 8 share a 64-bit register pair (W2[s & 7]), windows w and w + 8 likewise
// (S2[w]): the two windows of a pair always read the two slots of one pair, in order or swapped,
// which is what v_pk_add_f32's op_sel expresses -- 8 packed adds per tap.  Entering samples
// arrive four at a time (aligned 16-byte LDS reads, one per 4 taps, issued 4 taps ahead) in two
// alternating quads: group G = (j - 1) >> 2 sits in Q[G & 1].
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int PH>
__device__ __forceinline__ void nest_add(f32x2 (&S2)[8], const f32x2 (&W2)[8]) {
#pragma unroll
    for (int w = 0; w < 8; ++w) {
        if (((w + PH) & 15) < 8)
            asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(S2[w]) : "v"(W2[(w + PH) & 7]));
        else
            asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0]" : "+v"(S2[w]) : "v"(W2[(w + PH) & 7]));
    }
}

template <int PH>
__device__ __forceinline__ void nest_shift(f32x2 (&W2)[8], f32x4 (&Q)[2], const float* tile, int base, int j) {
    constexpr int GP = (PH % 4 == 0) ? (((PH >> 2) + 3) & 1) : ((PH >> 2) & 1);
    constexpr int SE = (PH + 15) & 15;                         // slot of the entering sample
    W2[SE & 7][SE >> 3] = Q[GP][(PH + 3) & 3];
    asm volatile("" : "+v"(W2[SE & 7]));                      // materialise the pair now: one v_mov into its half, not a re-assembly per use
    if (PH % 4 == 0) {                                         // group G - 1 for the four taps after the next three
        int jq = j - 8;
        jq = jq < 0 ? 0 : jq;                                  // (a clamped quad is never consumed)
        Q[GP ^ 1] = *reinterpret_cast<const f32x4*>(tile + lds_pad(base + jq));
    }
}

// correlate16 on packed fp32: the 16 chains of a lane as 8 register pairs (windows w and w + 8), the window slots
// paired the same way (slot s with s + 8, as in nest_add), so one tap is 8 v_pk_fma_f32 -- each half an IEEE fma of its
// own: the same bits as 16 v_fmac_f32, at half the issue slots.  The tap comes straight out of the 16-byte LDS read
// (op_sel picks its half of the pair), the window pair is read in order or swapped.
template <int JJ, int Q>
__device__ __forceinline__ void corr16_pk(const f32x4& tp, const f32x2 (&W2)[8], f32x2 (&C2)[8]) {
    const f32x2 tpair = (Q < 2) ? f32x2{tp[0], tp[1]} : f32x2{tp[2], tp[3]};
#pragma unroll
    for (int w = 0; w < 8; ++w) {
        const bool swapped = ((w + JJ) & 15) >= 8;
        if ((Q & 1) == 0) {
            if (!swapped) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(C2[w]) : "v"(tpair), "v"(W2[(w + JJ) & 7]));
            else          asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[0,0,1]" : "+v"(C2[w]) : "v"(tpair), "v"(W2[(w + JJ) & 7]));
        } else {
            if (!swapped) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(C2[w]) : "v"(tpair), "v"(W2[(w + JJ) & 7]));
            else          asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1]" : "+v"(C2[w]) : "v"(tpair), "v"(W2[(w + JJ) & 7]));
        }
    }
}

template <int G4>
__device__ __forceinline__ void corr16_pk_group(const f32x4& tp, const f32x4& nx, f32x2 (&W2)[8], f32x2 (&C2)[8]) {
    corr16_pk<4 * G4 + 0, 0>(tp, W2, C2);
    W2[(4 * G4 + 0) & 7][(4 * G4 + 0) >> 3] = nx[0];
    asm volatile("" : "+v"(W2[(4 * G4 + 0) & 7]));
    corr16_pk<4 * G4 + 1, 1>(tp, W2, C2);
    W2[(4 * G4 + 1) & 7][(4 * G4 + 1) >> 3] = nx[1];
    asm volatile("" : "+v"(W2[(4 * G4 + 1) & 7]));
    corr16_pk<4 * G4 + 2, 2>(tp, W2, C2);
    W2[(4 * G4 + 2) & 7][(4 * G4 + 2) >> 3] = nx[2];
    asm volatile("" : "+v"(W2[(4 * G4 + 2) & 7]));
    corr16_pk<4 * G4 + 3, 3>(tp, W2, C2);
    W2[(4 * G4 + 3) & 7][(4 * G4 + 3) >> 3] = nx[3];
    asm volatile("" : "+v"(W2[(4 * G4 + 3) & 7]));
}

// c_w = sum_{j < n} taps[j] * tile[base + w + j], windows w and w + 8 in C2[w] -- same contract as correlate16
__device__ __forceinline__ void correlate16_pk(const float* tile, int base, const float* taps, int n, f32x2 (&C2)[8]) {
    f32x2 W2[8];
#pragma unroll
    for (int q = 0; q < PSH_L / 4; ++q) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(tile + lds_pad(base + 4 * q));
#pragma unroll
        for (int e = 0; e < 4; ++e) W2[(4 * q + e) & 7][(4 * q + e) >> 3] = v[e];
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) C2[i] = f32x2{0.0f, 0.0f};
#pragma unroll 1
    for (int jb = 0; jb < n; jb += PSH_L) {
        const int rem = n - jb;                              // wave-uniform
        if (0 < rem) {
            const f32x4 nx = *reinterpret_cast<const f32x4*>(tile + lds_pad(base + jb + PSH_L + 0));
            const f32x4 tp = *reinterpret_cast<const f32x4*>(taps + jb + 0);
            corr16_pk_group<0>(tp, nx, W2, C2);
        }
        if (4 < rem) {
            const f32x4 nx = *reinterpret_cast<const f32x4*>(tile + lds_pad(base + jb + PSH_L + 4));
            const f32x4 tp = *reinterpret_cast<const f32x4*>(taps + jb + 4);
            corr16_pk_group<1>(tp, nx, W2, C2);
        }
        if (8 < rem) {
            const f32x4 nx = *reinterpret_cast<const f32x4*>(tile + lds_pad(base + jb + PSH_L + 8));
            const f32x4 tp = *reinterpret_cast<const f32x4*>(taps + jb + 8);
            corr16_pk_group<2>(tp, nx, W2, C2);
        }
        if (12 < rem) {
            const f32x4 nx = *reinterpret_cast<const f32x4*>(tile + lds_pad(base + jb + PSH_L + 12));
            const f32x4 tp = *reinterpret_cast<const f32x4*>(taps + jb + 12);
            corr16_pk_group<3>(tp, nx, W2, C2);
        }
    }
}

// THREADS / BG / NBG: 1024 threads (4 waves per SIMD, 128 VGPRs) with 3 (dense) or 2 (suffix rows) queries per evaluation of
// the embedding, or -- batches of 7 and more -- 512 threads (2 waves per SIMD, 256 VGPRs) with 10 or 6: the embedding is the
// cost, and a wave that carries 4x the accumulators evaluates it 4x less often.
template <bool ALIGNED, int MODE, int THREADS, int BG, int NBG>
__global__ __launch_bounds__(THREADS) void embed_scan_kernel(ScanArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = lane_id();
    const int wave_in_block = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    constexpr int NW = THREADS / 64;
    float* tile = smem + (size_t)wave_in_block * a.tile_floats;
    int* lcount = reinterpret_cast<int*>(smem + (size_t)NW * a.tile_floats);
    int* next_unit = lcount + ((a.B + 3) & ~3);
    u32x4* pend0 = reinterpret_cast<u32x4*>(lcount + ((a.B + 3) & ~3) + 4);
    u32x4* pend = pend0 + (size_t)wave_in_block * PSH_PEND;
    const int K = a.W, Kp = (K + 3) & ~3, d = a.emb_d;
    float* kerL = reinterpret_cast<float*>(pend0 + (size_t)NW * PSH_PEND);   // d x Kp, rows zero padded
    int2* rng = reinterpret_cast<int2*>(kerL + (size_t)d * Kp);              // per row: {first tap & ~3, taps to visit}
    int npend = 0;
    // suffix-rows fast path (BOOT / FILTER): header, rows in closing order, analysis scratch
    NestHdr* nh = reinterpret_cast<NestHdr*>(rng + ((d + 1) & ~1));          // 16-byte aligned
    int4* prog = reinterpret_cast<int4*>(nh + 1);                            // d x {closing tap a_i, row, c bits, n_i}
    unsigned long long* rmask = reinterpret_cast<unsigned long long*>(prog + d);   // d x 4 support masks, then U
    int* sl = reinterpret_cast<int*>(rmask + (size_t)4 * (d + 1)) + (size_t)wave_in_block * 192;   // wave-private: 64 survivors,
    float* Dl = reinterpret_cast<float*>(sl + 64);                                                     //   128 row differences

    if (threadIdx.x == 0) *next_unit = 0;
    if (MODE == PSH_MODE_FILTER)
        for (int q = (int)threadIdx.x; q < a.B; q += THREADS) lcount[q] = 0;
    for (int e = (int)threadIdx.x; e < d * Kp; e += THREADS) {
        const int i = e / Kp, j = e - i * Kp;
        kerL[e] = j < K ? a.ker[(int64_t)i * K + j] : 0.0f;
    }
    for (int p = lane; p < a.tile_floats; p += 64) tile[p] = 0.0f;   // no slot is ever read uninitialised
    __syncthreads();
    if ((int)threadIdx.x < d) {
        const float* row = kerL + (size_t)threadIdx.x * Kp;
        int lo = K, hi = 0;
        for (int j = 0; j < K; ++j)
            if (row[j] != 0.0f) { lo = j < lo ? j : lo; hi = j + 1; }
        if (hi == 0) lo = 0;
        lo &= ~3;
        rng[threadIdx.x] = make_int2(lo, hi - lo);
    }
    if (threadIdx.x == 0) {
        nh->ok = (MODE != PSH_MODE_ALL && K <= PSH_NEST_MAX_K && !a.emb_dense) ? 1 : 0;
        nh->nops = d; nh->cerr = 0.0f; nh->n_empty = 0;
        for (int q = 0; q < 16; ++q) nh->blk[q] = 0u;
    }
    if ((int)threadIdx.x < PSH_NEST_MAX_K) nh->ncl[threadIdx.x] = 0;
    if (threadIdx.x == 0) {
        for (int q = 0; q < 4; ++q) rmask[4 * d + q] = 0ull;
    }
    __syncthreads();
    if (MODE != PSH_MODE_ALL && K <= PSH_NEST_MAX_K) {       // every block repeats the (tiny) analysis of the matrix
        const int tid = (int)threadIdx.x;
        unsigned long long m[4] = {0ull, 0ull, 0ull, 0ull};
        int n = 0, lowest = 1 << 20;                         // empty rows close before the first tap
        float c = 0.0f;
        if (tid < d) {                                       // support mask, size, constant of row tid
            const float* row = kerL + (size_t)tid * Kp;
            bool okc = true;
            for (int j = K - 1; j >= 0; --j) {
                const float v = row[j];
                if (v != 0.0f) {                             // (NaN included: it then fails v == v)
                    if (n == 0) c = v;
                    okc = okc && (v == v) && (__float_as_uint(v) == __float_as_uint(c));
                    m[j >> 6] |= 1ull << (j & 63);
                    lowest = j;
                    ++n;
                }
            }
            okc = okc && (fabsf(c) <= 3.0e38f);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                rmask[4 * tid + q] = m[q];
                if (m[q]) atomicOr(&rmask[4 * d + q], m[q]);
            }
            prog[tid] = make_int4(lowest, tid, (int)__float_as_uint(c), n);   // (unsorted: read back below)
            if (!okc) atomicAnd(&nh->ok, 0);
        }
        __syncthreads();
        int rk = 0;
        if (tid < d) {                                       // the row must be all of U from its lowest tap up; rank by closing tap
            bool oks = true;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int lo_bit = lowest - 64 * q;          // taps of word q at or above `lowest`
                const unsigned long long keep = lo_bit <= 0 ? ~0ull : (lo_bit >= 64 ? 0ull : (~0ull << lo_bit));
                oks = oks && (m[q] == (rmask[4 * d + q] & keep));
            }
            if (n == 0) oks = true;
            if (!oks) atomicAnd(&nh->ok, 0);
            for (int i2 = 0; i2 < d; ++i2) { const int l2 = prog[i2].x; rk += (l2 > lowest || (l2 == lowest && i2 < tid)) ? 1 : 0; }
        }
        __syncthreads();
        if (tid < d) {
            prog[rk] = make_int4(lowest, tid, (int)__float_as_uint(c), n);
            if (n == 0) atomicAdd(&nh->n_empty, 1);
            else { atomicOr(&nh->blk[lowest >> 4], 0x10000u << (lowest & 15)); atomicAdd(&nh->ncl[lowest], 1); }
        }
        if (tid < 16) {
            const unsigned long long uw = rmask[4 * d + (tid >> 2)];
            atomicOr(&nh->blk[tid], (unsigned)((uw >> (16 * (tid & 3))) & 0xffffull));
        }
        __syncthreads();
        if (tid == 0) {
            float e2 = 0.0f;
            for (int i = 0; i < d; ++i) {
                const int4 o = prog[i];
                const float n1 = (float)(o.w + 1);
                const float t = fabsf(__uint_as_float((unsigned)o.z)) * n1 * n1;
                e2 = __builtin_fmaf(t, t, e2);
            }
            nh->cerr = 1.05f * 2.0f * 5.9604645e-8f * __builtin_sqrtf(e2);   // 2u sqrt(sum (c_i (n_i+1)^2)^2), margin for its own rounding
            if (!(e2 < 3.0e38f)) nh->ok = 0;
        }
        __syncthreads();
    }
    const bool nested = (MODE != PSH_MODE_ALL) && (__builtin_amdgcn_readfirstlane(nh->ok) != 0);
    const int n_empty = __builtin_amdgcn_readfirstlane(nh->n_empty);
    const float cerr = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(nh->cerr)));
    // rows in closing order, one per lane (two registers: d <= 128): what a closing row needs comes by v_readlane
    int ctab[2] = {0, 0}, rtab[2] = {0, 0};                  // -c_i bits, row index
    if (nested) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            if (lane + 64 * q < d) {
                const int4 o = prog[lane + 64 * q];
                ctab[q] = (int)(__float_as_uint(__uint_as_float((unsigned)o.z)) ^ 0x80000000u);
                rtab[q] = o.y;
            }
        }
    }

    const int nfloat = PSH_SEG + K - 1;
    const unsigned n_rs = (unsigned)a.n_rows * (unsigned)a.nseg;
    const unsigned n_units = n_rs * (unsigned)a.n_qgroups;
    const unsigned u_lo = (unsigned)(((unsigned long long)n_units * blockIdx.x) / gridDim.x);
    const unsigned u_hi = (unsigned)(((unsigned long long)n_units * (blockIdx.x + 1)) / gridDim.x);
    const const_f32p hxk = (const_f32p)a.hx;
    typedef const __attribute__((address_space(4))) QueryState* const_qsp;
    const const_qsp qstate_k = (const_qsp)a.qstate;

    for (;;) {
        int v = 0;
        if (lane == 0) v = atomicAdd(next_unit, 1);
        const unsigned u = u_lo + (unsigned)__builtin_amdgcn_readfirstlane(v);
        if (u >= u_hi) break;
        const unsigned qgi = fast_div(u, a.magic_nrs, n_rs);
        const unsigned rs = u - qgi * n_rs;
        const unsigned ri = fast_div(rs, a.magic_nseg, (unsigned)a.nseg);
        const unsigned sg = rs - ri * (unsigned)a.nseg;
        const int64_t row = a.row0 + (int64_t)ri * a.row_stride;
        const int seg_start = (int)sg * PSH_SEG;

        if (MODE == PSH_MODE_FILTER && npend > 0) {   // stores ahead of the loads: vmcnt retires in order
            pend_flush(pend, npend, lcount, a, lane);
            npend = 0;
        }
        float ymax = 0.0f;
        {
            Stage st;
            stage_load<ALIGNED>(st, a.dataset + row * a.T, a.T, seg_start, nfloat, lane);
            if (MODE != PSH_MODE_ALL && nested) {            // max |y| of everything this segment reads (NaN ignored: see above)
#pragma unroll
                for (int q = 0; q < PSH_NSTAGE; ++q) {
                    if (q < PSH_NSTAGE - 1 || lane + 64 * q < ((nfloat + 3) >> 2)) {
                        ymax = fmaxf(ymax, fmaxf(fmaxf(fabsf(st.v[q][0]), fabsf(st.v[q][1])),
                                                 fmaxf(fabsf(st.v[q][2]), fabsf(st.v[q][3]))));
                    }
                }
            }
            stage_store(st, tile, nfloat, lane);
        }
        wave_lds_fence();

        const int t_lane = seg_start + PSH_L * lane;
        int nvalid = a.Tp - t_lane;
        nvalid = nvalid < 0 ? 0 : (nvalid > PSH_L ? PSH_L : nvalid);
        const int r_global = (int)(row + a.r_offset);
        const int q_begin = (int)qgi * a.q_per_group;
        const int q_end = (q_begin + a.q_per_group) < a.B ? (q_begin + a.q_per_group) : a.B;

        if (MODE != PSH_MODE_ALL && nested) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) ymax = fmaxf(ymax, __shfl_xor(ymax, off, 64));
            const float err = ymax * cerr;                    // radius of the cheap embedding around the exact one
            const int base = PSH_L * lane;
            int ns = 0;                                       // survivors waiting in sl (wave-uniform)
            // Exact verification of the listed survivors (window index | query << 12), rows across the lanes:
            // lane (el, l) runs the chains of the l-th shortest and then the l-th longest row of survivor el
            // (equal work per lane), 64 / ceil(d/2) survivors per pass; one lane per survivor then adds the d
            // squares in row order.  A row's taps need no matrix: c_i on U & [a_i, K), zero elsewhere (the zero
            // taps the dense chain visits are visited too: fma(0, y, .) matters for non-finite y).
            auto verify_list = [&]() {
                wave_lds_fence();
                const int H = (d + 1) >> 1, EPP = 64 / H;
                const int el = lane / H, l = lane - el * H;
                const int4 oA = prog[l];
                const int sB = d - 1 - l;
                const bool hasB = sB > l;
                const int4 oB = prog[hasB ? sB : l];
                const int2 gA = rng[oA.y], gB = rng[oB.y];
                const float cA = __uint_as_float((unsigned)oA.z), cB = __uint_as_float((unsigned)oB.z);
#pragma unroll 1
                for (int e0 = 0; e0 < ns; e0 += EPP) {
                    const bool lv = el < EPP && e0 + el < ns;
                    const int ent = lv ? sl[e0 + el] : 0;
                    const int pwin = ent & 4095, b = ent >> 12;
                    const int nA4 = lv ? ((gA.y + 3) & ~3) : 0, nB4 = (lv && hasB) ? ((gB.y + 3) & ~3) : 0;
                    auto chain = [&](int lo, int n4, int ath, float c) -> float {
                        int lm = n4;
#pragma unroll
                        for (int off = 32; off > 0; off >>= 1) { const int o2 = __shfl_xor(lm, off, 64); lm = o2 > lm ? o2 : lm; }
                        lm = __builtin_amdgcn_readfirstlane(lm);
                        float hy = 0.0f;
#pragma unroll 1
                        for (int it = 0; it < lm; it += 4) {             // lo, n4 are multiples of 4: a group is all or nothing
                            const bool act = it < n4;
                            const int j0 = act ? lo + it : 0;
                            float y[4];
#pragma unroll
                            for (int q = 0; q < 4; ++q) y[q] = tile[lds_pad(pwin + j0 + q)];
                            const unsigned ub = nh->blk[(j0 >> 4) & 15] >> (j0 & 15);
                            float t = hy;
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const bool on = (j0 + q >= ath) && (((ub >> q) & 1u) != 0u);
                                t = __builtin_fmaf(on ? c : 0.0f, y[q], t);
                            }
                            hy = act ? t : hy;
                        }
                        return hy;
                    };
                    const float hyA = chain(gA.x, nA4, oA.x, cA);
                    const float hyB = chain(gB.x, nB4, oB.x, cB);
                    if (lv) {
                        const float* hxb = a.hx + (int64_t)b * d;
                        Dl[el * d + oA.y] = __fsub_rn(hxb[oA.y], hyA);
                        if (hasB) Dl[el * d + oB.y] = __fsub_rn(hxb[oB.y], hyB);
                    }
                    wave_lds_fence();
                    float ea = __uint_as_float(PSH_INF_BITS);
                    bool hit = false;
                    if (lv && l == 0) {
                        ea = 0.0f;
                        for (int i = 0; i < d; ++i) { const float D = Dl[el * d + i]; ea = __builtin_fmaf(D, D, ea); }
                        hit = ea < __uint_as_float(a.qstate[b].tau2_bits);
                    }
                    const unsigned long long mask = __ballot(hit);
                    wave_lds_fence();                        // Dl is rewritten by the next pass
                    if (!mask) continue;
                    const int nh2 = __popcll(mask);
                    if (npend + nh2 > PSH_PEND) {
                        pend_flush(pend, npend, lcount, a, lane);
                        npend = 0;
                        wave_lds_fence();
                    }
                    if (hit) {
                        const int slot = npend + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32),
                                                     __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
                        pend[slot] = u32x4{__float_as_uint(ea), (unsigned)r_global, (unsigned)(seg_start + pwin), (unsigned)b};
                    }
                    npend += nh2;
                }
                wave_lds_fence();                            // sl is refilled afterwards
            };
            for (int b0 = q_begin; b0 < q_end; b0 += NBG) {
                const int nq = (q_end - b0) < NBG ? (q_end - b0) : NBG;
                f32x2 acc[NBG][8], S[8], win[8];          // element x: window w / slot s, element y: w + 8 / s + 8
                f32x4 Q[2];
#pragma unroll
                for (int w = 0; w < 8; ++w) { S[w] = f32x2{0.f, 0.f}; win[w] = f32x2{0.f, 0.f}; }
#pragma unroll
                for (int g = 0; g < NBG; ++g)
#pragma unroll
                    for (int w = 0; w < 8; ++w) acc[g][w] = f32x2{0.f, 0.f};
                Q[0] = f32x4{0.f, 0.f, 0.f, 0.f};
                Q[1] = Q[0];
                // the group's embedded queries, permuted into closing order across the lanes
                int pc = 0;
                int hxt[NBG][2];
#pragma unroll
                for (int g = 0; g < NBG; ++g)
#pragma unroll
                    for (int q = 0; q < 2; ++q)
                        hxt[g][q] = (g < nq && lane + 64 * q < d) ? (int)__float_as_uint(a.hx[(int64_t)(b0 + g) * d + rtab[q]]) : 0;
                auto close_row = [&]() {                     // e = hx - c S;  acc += e^2  for the row at the head of the list
                    const int sl2 = pc & 63;
                    const bool lo64 = pc < 64;
                    const int c0 = __builtin_amdgcn_readlane(ctab[0], sl2), c1 = __builtin_amdgcn_readlane(ctab[1], sl2);
                    const float nc = __uint_as_float((unsigned)(lo64 ? c0 : c1));
                    const f32x2 nc2 = f32x2{nc, nc};
#pragma unroll
                    for (int g = 0; g < NBG; ++g) {
                        if (g < nq) {                        // wave-uniform
                            const int h0 = __builtin_amdgcn_readlane(hxt[g][0], sl2), h1 = __builtin_amdgcn_readlane(hxt[g][1], sl2);
                            const float hv = __uint_as_float((unsigned)(lo64 ? h0 : h1));
                            const f32x2 hx2 = f32x2{hv, hv};
#pragma unroll
                            for (int w = 0; w < 8; ++w) {
                                const f32x2 e = __builtin_elementwise_fma(nc2, S[w], hx2);
                                acc[g][w] = __builtin_elementwise_fma(e, e, acc[g][w]);
                            }
                        }
                    }
                    ++pc;
                };
                for (int e0 = 0; e0 < n_empty; ++e0) close_row();            // all-zero rows: h = 0
                {
                    // window of tap jtop = 16 q + 15 >= K - 1: slot 15 <- y[base + jtop], slots 0..14 <- the 15 samples above
                    const int qb = (K - 1) >> 4;
                    const f32x4 v3 = *reinterpret_cast<const f32x4*>(tile + lds_pad(base + 16 * qb + 12));
                    win[7][1] = v3[3];
#pragma unroll
                    for (int sq = 0; sq < 4; ++sq) {
                        const f32x4 v = *reinterpret_cast<const f32x4*>(tile + lds_pad(base + 16 * qb + 16 + 4 * sq));
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (4 * sq + e < 15) win[(4 * sq + e) & 7][(4 * sq + e) >> 3] = v[e];
                    }
                    Q[1] = v3;                                                   // group 4 qb + 3 (taps 15..13 take its samples 2..0)
                    Q[0] = *reinterpret_cast<const f32x4*>(tile + lds_pad(base + 16 * qb + 8));   // group 4 qb + 2
#pragma unroll 1
                    for (int jb = 16 * qb; jb >= 0 && pc < d; jb -= 16) {
                        const unsigned bm = (unsigned)__builtin_amdgcn_readfirstlane((int)nh->blk[jb >> 4]);
#define PSH_NEST_TAP(PH)                                                                        \
                        if (bm & (1u << (PH))) nest_add<PH>(S, win);                            \
                        nest_shift<PH>(win, Q, tile, base, jb + (PH));                          \
                        if (bm & (0x10000u << (PH))) {                                          \
                            for (int cl = __builtin_amdgcn_readfirstlane(nh->ncl[jb + (PH)]); cl > 0; --cl) close_row(); \
                        }
                        PSH_NEST_TAP(15) PSH_NEST_TAP(14) PSH_NEST_TAP(13) PSH_NEST_TAP(12)
                        PSH_NEST_TAP(11) PSH_NEST_TAP(10) PSH_NEST_TAP(9) PSH_NEST_TAP(8)
                        PSH_NEST_TAP(7) PSH_NEST_TAP(6) PSH_NEST_TAP(5) PSH_NEST_TAP(4)
                        PSH_NEST_TAP(3) PSH_NEST_TAP(2) PSH_NEST_TAP(1) PSH_NEST_TAP(0)
#undef PSH_NEST_TAP
                    }
                }
#pragma unroll
                for (int g = 0; g < NBG; ++g) {
                    const int b = b0 + g;
                    if (g >= nq) continue;
                    if (MODE == PSH_MODE_BOOT) {
                        // upper bound of the exact acc of the lane's (wave's) best window
                        float m = __uint_as_float(PSH_INF_BITS);
#pragma unroll
                        for (int w = 0; w < PSH_L; ++w) m = (w < nvalid) ? fminf(m, acc[g][w & 7][w >> 3]) : m;
                        if (a.boot_per_wave) {
#pragma unroll
                            for (int off = 32; off > 0; off >>= 1) m = fminf(m, __shfl_xor(m, off, 64));
                        }
                        const float su = __builtin_sqrtf(m) * (1.0f + 1.0f / 32768.0f) + err;
                        const float ub = su * su * (1.0f + 1.0f / 16384.0f);
                        if (a.boot_per_wave) {
                            if (lane == 0) a.minbuf[(int64_t)b * a.min_stride + (int64_t)rs] = ub;
                        } else {
                            a.minbuf[(int64_t)b * a.min_stride + (int64_t)rs * 64 + lane] = ub;
                        }
                    } else {
                        const float tau = __uint_as_float(qstate_k[b].tau2_bits);
                        const float st = __builtin_sqrtf(tau) * (1.0f + 1.0f / 32768.0f) + err;
                        const float thr = st * st * (1.0f + 1.0f / 16384.0f);
                        unsigned hm = 0u;
#pragma unroll
                        for (int w = 0; w < PSH_L; ++w) hm |= ((w < nvalid) && !(acc[g][w & 7][w >> 3] > thr)) ? (1u << w) : 0u;
                        // survivors go to the wave's list; the whole wave verifies them together (verify_list)
                        while (__any(hm != 0u)) {
                            const bool has = hm != 0u;
                            const int w = has ? (int)__builtin_ctz(hm) : 0;
                            hm &= hm - 1u;
                            const unsigned long long sm = __ballot(has);
                            const int ne = __popcll(sm);
                            if (ns + ne > 64) { verify_list(); ns = 0; }
                            if (has) sl[ns + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(sm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)sm, 0u))] =
                                (base + w) | (b << 12);
                            ns += ne;
                        }
                    }
                }
            }
            if (MODE == PSH_MODE_FILTER && ns > 0) { verify_list(); ns = 0; }   // before the tile is overwritten
        } else
        for (int b0 = q_begin; b0 < q_end; b0 += BG) {
            f32x2 acc2[BG][8];
#pragma unroll
            for (int g = 0; g < BG; ++g)
#pragma unroll
                for (int w = 0; w < 8; ++w) acc2[g][w] = f32x2{0.0f, 0.0f};
#pragma unroll 1
            for (int i = 0; i < d; ++i) {
                const int2 rg = rng[i];
                const int jlo = __builtin_amdgcn_readfirstlane(rg.x);
                const int n = __builtin_amdgcn_readfirstlane(rg.y);
                f32x2 c2[8];                                 // windows w (x) and w + 8 (y)
                correlate16_pk(tile, PSH_L * lane + jlo, kerL + (size_t)i * Kp + jlo, n, c2);
#pragma unroll
                for (int g = 0; g < BG; ++g) {
                    if (b0 + g < q_end) {                    // wave-uniform
                        const float hxv = hxk[(int64_t)(b0 + g) * d + i];
                        const f32x2 hx2 = f32x2{hxv, hxv};
#pragma unroll
                        for (int w = 0; w < 8; ++w) {        // D = hx - c (one rounding), acc = fma(D, D, acc): per half, IEEE
                            const f32x2 D = hx2 - c2[w];
                            acc2[g][w] = __builtin_elementwise_fma(D, D, acc2[g][w]);
                        }
                    }
                }
            }
            float acc[BG][PSH_L];
#pragma unroll
            for (int g = 0; g < BG; ++g)
#pragma unroll
                for (int w = 0; w < PSH_L; ++w) acc[g][w] = acc2[g][w & 7][w >> 3];
#pragma unroll
            for (int g = 0; g < BG; ++g) {
                const int b = b0 + g;
                if (b < q_end) {
                    const float tau = (MODE == PSH_MODE_FILTER) ? __uint_as_float(qstate_k[b].tau2_bits) : 0.0f;   // see below: the estimate, when there is one
                    const float xn = (MODE == PSH_MODE_ALL) ? qstate_k[b].xn : 0.0f;
                    emit16<MODE>(a, b, acc[g], nvalid, lane, rs, r_global, t_lane, tau, xn, pend, npend, lcount);
                }
            }
        }
        wave_lds_fence();  // all lanes done with the tile before it is overwritten
    }
    if (MODE == PSH_MODE_FILTER) {
        if (npend > 0) pend_flush(pend, npend, lcount, a, lane);
        __syncthreads();
        for (int q = (int)threadIdx.x; q < a.B; q += THREADS)
            a.bcount[(int64_t)q * PSH_MAX_BLOCKS + blockIdx.x] = lcount[q];
    }
}

// ----------------------------------------------------------------------------------
// one-window rows (T == W + h): the ensemble is N points of W samples -- what
// PathDistance.forward_topk scans (a pre-embedded y, reference path_distance.py:10-49), and
// shadow() on paths exactly one window long.  The reference's numerator is then a
// CONTIGUOUS reduce (8-lane order: sumsq8), one per row, and a wave takes 64 rows at a time,
// a row per lane:
//   FILTER : the 64 rows are 64*T contiguous floats -- coalesced 16-byte loads, scattered into
//            LDS at an odd row stride (every lane then walks its own row without bank
//            conflicts); admits acc < tau into the block's slice like the other scans
//   BOOT   : the sampled rows are `row_stride` apart: every lane reads its own row from
//            memory; one exact acc per sampled row -> minbuf
// HBM-bound for a handful of queries (2.3 W VALU operations per row and query against 4 T bytes).
// ----------------------------------------------------------------------------------
#define PSH_ROWS_THREADS 128

// staging of the 64 rows of a chunk (FILTER / ALL), chosen by the launcher:
//   PSH_ROWS_FLAT  : gcd(T, 64) <= 2 -- the chunk is copied as it lies (16-byte LDS writes, no index arithmetic); a lane
//                    then walks its row at stride T with at most a 2-way bank conflict
//   PSH_ROWS_QUADS : T % 4 == 0 -- one row/column split per float4 (rows start on float4 boundaries), odd LDS row stride
//   PSH_ROWS_SPLIT : anything else (long rows with a horizon tail, unaligned ensembles): one split per element, only
//                    the first W samples of a row are kept
#define PSH_ROWS_SPLIT 0
#define PSH_ROWS_FLAT 1
#define PSH_ROWS_QUADS 2

template <int MODE>
__global__ __launch_bounds__(PSH_ROWS_THREADS) void rows_kernel(ScanArgs a, int aligned16, int staging) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    constexpr int NW = PSH_ROWS_THREADS / 64;
    const int ds = a.tile_floats;                            // LDS floats per row, odd
    float* tile = smem + (size_t)wave * 64 * ds;
    int* lcount = reinterpret_cast<int*>(smem + (size_t)NW * 64 * ds);
    u32x4* pend = reinterpret_cast<u32x4*>(lcount + ((a.B + 3) & ~3)) + (size_t)wave * PSH_PEND;
    int npend = 0;
    if (MODE == PSH_MODE_FILTER) {
        for (int q = (int)threadIdx.x; q < a.B; q += PSH_ROWS_THREADS) lcount[q] = 0;
        __syncthreads();
    }
    const int W = a.W;
    const int64_t T = a.T;
    const int n_chunks = (a.n_rows + 63) >> 6;
    const int c_lo = (int)(((int64_t)n_chunks * blockIdx.x) / gridDim.x);
    const int c_hi = (int)(((int64_t)n_chunks * (blockIdx.x + 1)) / gridDim.x);
    const const_f32p qk = (const_f32p)a.queries;
    typedef const __attribute__((address_space(4))) QueryState* const_qsp;
    const const_qsp qstate_k = (const_qsp)a.qstate;
    const bool staged = (MODE != PSH_MODE_BOOT) && a.row_stride == 1;

    for (int c = c_lo + wave; c < c_hi; c += NW) {
        const int i = 64 * c + lane;                         // this lane's row of the launch
        const bool valid = i < a.n_rows;
        const int64_t row = a.row0 + (int64_t)(valid ? i : a.n_rows - 1) * a.row_stride;
        const float* yrow = a.dataset + row * T;
        if (MODE == PSH_MODE_FILTER && npend > 0) { pend_flush(pend, npend, lcount, a, lane); npend = 0; }
        if (staged) {
            const int nr = (a.n_rows - 64 * c) < 64 ? (a.n_rows - 64 * c) : 64;
            const int64_t nfl = (int64_t)nr * T;
            const float* src = a.dataset + (a.row0 + (int64_t)64 * c) * T;
            if (staging == PSH_ROWS_FLAT) {
                for (int64_t e4 = lane; 4 * e4 < nfl; e4 += 64) {
                    if (4 * e4 + 3 < nfl) {
                        *reinterpret_cast<f32x4*>(tile + 4 * e4) = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src) + e4);
                    } else {
                        for (int k2 = 0; 4 * e4 + k2 < nfl; ++k2) tile[4 * e4 + k2] = src[4 * e4 + k2];
                    }
                }
            } else if (staging == PSH_ROWS_QUADS) {
                const unsigned T4 = (unsigned)(T >> 2);
                const unsigned magic4 = (unsigned)((1ull << 32) / (unsigned long long)T4);
                for (int64_t e4 = lane; 4 * e4 < nfl; e4 += 64) {
                    const f32x4 q4 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src) + e4);
                    const unsigned r = fast_div((unsigned)e4, magic4, T4);
                    const unsigned c = 4u * ((unsigned)e4 - r * T4);
                    if (c < (unsigned)W) {
                        float* dstp = tile + r * ds + c;
                        dstp[0] = q4[0]; dstp[1] = q4[1]; dstp[2] = q4[2]; dstp[3] = q4[3];   // (columns >= W of the last quad: unused slots of the row)
                    }
                }
            } else {
            const unsigned magic = (unsigned)((1ull << 32) / (unsigned long long)T);
            for (int64_t e4 = lane; 4 * e4 < nfl; e4 += 64) {
                float v[4];
                if (aligned16 && 4 * e4 + 3 < nfl) {
                    const f32x4 q4 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src) + e4);
                    v[0] = q4[0]; v[1] = q4[1]; v[2] = q4[2]; v[3] = q4[3];
                } else {
#pragma unroll
                    for (int k2 = 0; k2 < 4; ++k2) v[k2] = (4 * e4 + k2 < nfl) ? src[4 * e4 + k2] : 0.0f;
                }
#pragma unroll
                for (int k2 = 0; k2 < 4; ++k2) {
                    const unsigned e = (unsigned)(4 * e4 + k2);
                    const unsigned r = fast_div(e, magic, (unsigned)T);
                    const unsigned col = e - r * (unsigned)T;
                    if (col < (unsigned)W && (int64_t)e < nfl) tile[r * ds + col] = v[k2];
                }
            }
            }
            wave_lds_fence();
        }
        for (int b = 0; b < a.B; ++b) {
            const const_f32p x = qk + (int64_t)b * W;
            float acc;
            if (staged) acc = sumsq8([&](int j) { return __fsub_rn(x[j], tile[lane * ds + j]); }, W);
            else        acc = sumsq8([&](int j) { return __fsub_rn(x[j], yrow[j]); }, W);
            if (MODE == PSH_MODE_BOOT) {
                if (valid) a.minbuf[(int64_t)b * a.min_stride + i] = acc;
            } else if (MODE == PSH_MODE_ALL) {               // exhaustive path: one slot per row of the chunk
                if (valid) {
                    a.cand_d[(int64_t)b * a.cap + i] = dist_from_acc(acc, qstate_k[b].xn);
                    a.cand_rt[(int64_t)b * a.cap + i] = make_int2((int)(row + a.r_offset), 0);
                }
            } else {
                const float tau = __uint_as_float(qstate_k[b].tau2_bits);   // the estimate when there is one (psh_capi.hip), else tau
                const bool hit = valid && (acc < tau);
                const unsigned long long mask = __ballot(hit);
                if (!mask) continue;
                const int nh = __popcll(mask);
                if (npend + nh > PSH_PEND) {
                    pend_flush(pend, npend, lcount, a, lane);
                    npend = 0;
                    wave_lds_fence();
                }
                if (hit) {
                    const int slot = npend + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32),
                                                 __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
                    pend[slot] = u32x4{__float_as_uint(acc), (unsigned)(int)(row + a.r_offset), 0u, (unsigned)b};
                }
                npend += nh;
            }
        }
        if (staged) wave_lds_fence();                        // all lanes done with the tile before it is overwritten
    }
    if (MODE == PSH_MODE_FILTER) {
        if (npend > 0) pend_flush(pend, npend, lcount, a, lane);
        __syncthreads();
        for (int q = (int)threadIdx.x; q < a.B; q += PSH_ROWS_THREADS)
            a.bcount[(int64_t)q * PSH_MAX_BLOCKS + blockIdx.x] = lcount[q];
    }
}

// ----------------------------------------------------------------------------------
// one-block selection machinery (threshold of the bootstrap sample, final top-k, merge)
// ----------------------------------------------------------------------------------
#define PSH_RB 11                          // radix-select digit width: 2048 counters per pass
struct SelectShared {
    unsigned hist[1 << PSH_RB];
    uint64_t prefix, kmin, kmax;
    uint64_t prefix_b;                 // radix_select64(rank_b): first-pass bucket of a second rank (an estimate)
    int remaining, done, nsel, cnt, overflow;
    int offs[PSH_MAX_BLOCKS + 1];
    int cnt_front[PSH_MAX_BLOCKS];     // two-class slices: entries at the front of each slice
};

// min / max of the live 64-bit keys over the block (kmin > kmax when nothing is live)
// (`walk(body)` calls body(key) once per live candidate, every thread its share)
template <typename WalkFn>
__device__ inline void block_minmax64(WalkFn walk, SelectShared* sm, uint64_t* out_min, uint64_t* out_max) {
    const int tid = (int)threadIdx.x;
    __syncthreads();
    if (tid == 0) { sm->kmin = ~0ull; sm->kmax = 0ull; }
    __syncthreads();
    uint64_t lo = ~0ull, hi = 0ull;
    walk([&](uint64_t k) {
        lo = k < lo ? k : lo;
        hi = k > hi ? k : hi;
    });
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const uint64_t l2 = __shfl_xor(lo, off, 64), h2 = __shfl_xor(hi, off, 64);
        lo = l2 < lo ? l2 : lo;
        hi = h2 > hi ? h2 : hi;
    }
    if ((tid & 63) == 0) { atomicMin((unsigned long long*)&sm->kmin, (unsigned long long)lo); atomicMax((unsigned long long*)&sm->kmax, (unsigned long long)hi); }
    __syncthreads();
    *out_min = sm->kmin;
    *out_max = sm->kmax;
    __syncthreads();
}

// Rank-`rank` (1-based) smallest 64-bit key among the candidates i with live(i); only
// key bits >= sh_floor are examined.  MSB-first, PSH_RB bits per pass (two passes cover
// the ~20 bits in which distance keys differ), starting at the first bit in which the
// keys differ at all.  [kmin, kmax] may be passed in (have_minmax) when the caller
// already knows them.  On return, in every thread: the live candidates with
// (key >> sh) <= (prefix >> sh) are exactly the `rank` smallest -- unless keys tie down to
// sh_floor (*exact false): then more may match and *remaining of the ones equal to
// prefix at sh_floor are still wanted.
// The candidates are visited through `walk(body)` (body(key) once per live candidate).
template <typename WalkFn>
__device__ inline void radix_select64_walk(WalkFn walk, int rank, int sh_floor,
                                           SelectShared* sm, uint64_t* out_prefix, int* out_sh, bool* out_exact,
                                           int* out_remaining, bool have_minmax = false, uint64_t kmin_in = 0,
                                           uint64_t kmax_in = 0, int good_enough_sh = -1, int rank_b = 0) {
    const int tid = (int)threadIdx.x;
    constexpr unsigned NB = 1u << PSH_RB;
    uint64_t kmin = kmin_in, kmax = kmax_in;
    if (!have_minmax) block_minmax64(walk, sm, &kmin, &kmax);
    const uint64_t diff = (kmin ^ kmax) >> sh_floor;
    if (kmin > kmax || diff == 0ull) {       // nothing live, or every key equal above the floor
        *out_prefix = (kmin > kmax) ? 0ull : ((kmin >> sh_floor) << sh_floor);
        *out_sh = sh_floor;
        *out_exact = false;
        *out_remaining = rank;
        return;
    }
    const int top_bit = 63 - __clzll((unsigned long long)(diff << sh_floor));   // highest differing bit
    // the first digit's MSB is the highest differing bit, so its 2^PSH_RB counters spread
    // over [kmin, kmax]; a digit grid fixed to sh_floor can leave the first pass two or
    // three live counters and 1e4 LDS atomics serialised on them (15 us of a 20 us select)
    int bits = (top_bit - sh_floor + 1) < PSH_RB ? (top_bit - sh_floor + 1) : PSH_RB;
    int sh = top_bit + 1 - bits;
    __syncthreads();
    if (tid == 0) {
        sm->prefix = (top_bit + 1 >= 64) ? 0ull : ((kmin >> (top_bit + 1)) << (top_bit + 1));   // shared high bits
        sm->remaining = rank;
        sm->done = 0;
    }
    __syncthreads();
    int sh_done = sh;
    bool first = true;
    for (;;) {
        for (unsigned i = (unsigned)tid; i < NB; i += PSH_SELECT_THREADS) sm->hist[i] = 0u;
        __syncthreads();
        const uint64_t prefix = sm->prefix;
        const int shp = sh + bits;
        const unsigned dmask = (1u << bits) - 1u;
        walk([&](uint64_t key) {
            const bool match = first || shp >= 64 || ((key >> shp) == (prefix >> shp));
            if (match) atomicAdd(&sm->hist[(unsigned)(key >> sh) & dmask], 1u);
        });
        __syncthreads();
        if (tid < 64) {
            // bucket holding the rank: wave-wide prefix over the counters, NB/64 per lane
            // (a serial walk by one thread is NB dependent LDS round trips)
            constexpr int PER = (int)(NB / 64);
            const int rem = sm->remaining;
            unsigned h[PER];
            unsigned sl = 0;
#pragma unroll
            for (int q = 0; q < PER; ++q) { h[q] = sm->hist[PER * tid + q]; sl += h[q]; }
            unsigned inc = sl;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const unsigned v = __shfl_up(inc, off, 64);
                if (tid >= off) inc += v;
            }
            unsigned cum = inc - sl;
            if (first && rank_b > 0 && cum < (unsigned)rank_b && inc >= (unsigned)rank_b) {   // the second rank's bucket
                unsigned c2 = cum;
                int bucket_b = PER * tid;
#pragma unroll
                for (int q = 0; q < PER; ++q) {
                    if (c2 < (unsigned)rank_b && c2 + h[q] >= (unsigned)rank_b) bucket_b = PER * tid + q;
                    c2 += h[q];
                }
                sm->prefix_b = prefix | ((uint64_t)(unsigned)bucket_b << sh) | ((sh > 0) ? ((1ull << sh) - 1ull) : 0ull);   // upper edge
            }
            if (cum < (unsigned)rem && inc >= (unsigned)rem) {      // exactly one lane
                int bucket = PER * tid;
                unsigned before = cum, hb = 0;
#pragma unroll
                for (int q = 0; q < PER; ++q) {
                    if (cum < (unsigned)rem && cum + h[q] >= (unsigned)rem) { bucket = PER * tid + q; before = cum; hb = h[q]; }
                    cum += h[q];
                }
                const int r2 = rem - (int)before;
                sm->remaining = r2;
                sm->prefix = prefix | ((uint64_t)(unsigned)bucket << sh);
                sm->done = ((int)hb == r2) ? 1 : 0;
            }
        }
        __syncthreads();
        sh_done = sh;
        first = false;
        if (sm->done || sh <= sh_floor) break;
        if (sh <= good_enough_sh) break;      // the caller only needs a bound of the rank-th key: bucket edge at 2^sh
        bits = (sh - sh_floor) < PSH_RB ? (sh - sh_floor) : PSH_RB;
        sh -= bits;
    }
    *out_prefix = sm->prefix;
    *out_sh = sh_done;
    *out_exact = sm->done != 0;
    *out_remaining = sm->remaining;
    __syncthreads();
}

// the same over an index range: candidate i (live(i)) has key key_of(i)
template <typename KeyFn, typename LiveFn>
__device__ inline void radix_select64(KeyFn key_of, LiveFn live, int n, int rank, int sh_floor,
                                      SelectShared* sm, uint64_t* out_prefix, int* out_sh, bool* out_exact,
                                      int* out_remaining, bool have_minmax = false, uint64_t kmin_in = 0,
                                      uint64_t kmax_in = 0, int good_enough_sh = -1, int rank_b = 0) {
    radix_select64_walk([&](auto&& body) {
                            for (int i = (int)threadIdx.x; i < n; i += PSH_SELECT_THREADS)
                                if (live(i)) body(key_of(i));
                        },
                        rank, sh_floor, sm, out_prefix, out_sh, out_exact, out_remaining, have_minmax, kmin_in, kmax_in,
                        good_enough_sh, rank_b);
}

// bootstrap threshold: tau = k-th smallest of the sampled minima (+ margin).  The sample
// is staged in LDS once; every selection pass then runs at LDS latency.
__global__ __launch_bounds__(PSH_SELECT_THREADS) void threshold_kernel(ThresholdArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned tkeys[];   // n_entries (or nothing)
    __shared__ SelectShared sm;
    const int b = (int)blockIdx.x;
    const int tid = (int)threadIdx.x;
    const float* v = a.minbuf + (int64_t)b * a.min_stride;
    const int n = a.n_entries;
    __shared__ unsigned s_maxbits;                         // largest |value| among the sampled data and this query
    if (tid == 0) { prep_query(a.prep, b); s_maxbits = 0u; sm.prefix_b = ~0ull; }   // ||x||, sum of squares, state reset
    __syncthreads();                                       // (block-scope visibility of qstate[b] for thread 0 below)
    if (a.blockmax) {
        unsigned mb = 0u;                                  // non-negative floats order as their bit patterns
        for (int i = tid; i < a.n_blockmax; i += PSH_SELECT_THREADS) mb = max(mb, __float_as_uint(a.blockmax[i]));
        // batched matrix-core scan: ONE scale for all queries (they share the f16 copy of the data)
        const int64_t xlo = a.mq_frag ? 0 : (int64_t)b * a.prep.W;
        const int64_t xhi = a.mq_frag ? (int64_t)a.prep.B * a.prep.W : xlo + a.prep.W;
        for (int64_t j = xlo + tid; j < xhi; j += PSH_SELECT_THREADS)
            mb = max(mb, __float_as_uint(fabsf(a.prep.queries[j])));
        if (mb) atomicMax(&s_maxbits, mb);
        __syncthreads();
    }
    if (n < a.k) return;                                   // tau stays +inf (host avoids this)
    const bool in_lds = a.keys_in_lds != 0;
    if (in_lds) {
        // the keys' min / max fall out of the staging pass (the selection would otherwise re-read all of them)
        unsigned kmin32 = 0xffffffffu, kmax32 = 0u;
        if (tid == 0) { sm.kmin = ~0ull; sm.kmax = 0ull; }
        __syncthreads();
#pragma unroll 4
        for (int i = tid; i < n; i += PSH_SELECT_THREADS) {
            const unsigned kb = __float_as_uint(v[i]);
            tkeys[i] = kb;
            kmin32 = kb < kmin32 ? kb : kmin32;
            kmax32 = kb > kmax32 ? kb : kmax32;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const unsigned l2 = __shfl_xor(kmin32, off, 64), h2 = __shfl_xor(kmax32, off, 64);
            kmin32 = l2 < kmin32 ? l2 : kmin32;
            kmax32 = h2 > kmax32 ? h2 : kmax32;
        }
        if ((tid & 63) == 0) {
            atomicMin((unsigned long long*)&sm.kmin, (unsigned long long)kmin32 << 32);
            atomicMax((unsigned long long*)&sm.kmax, (unsigned long long)kmax32 << 32);
        }
        __syncthreads();
    }
    const uint64_t kmin64 = in_lds ? sm.kmin : 0ull, kmax64 = in_lds ? sm.kmax : 0ull;
    uint64_t prefix;
    int sh, rem;
    bool exact;
    // tau only has to bound the k-th smallest minimum from above: once the digits examined pin it to
    // 2^15 ulps (0.4 %) the bucket's upper edge serves -- usually one pass instead of three
    radix_select64([&](int i) { return (uint64_t)(in_lds ? tkeys[i] : __float_as_uint(v[i])) << 32; },
                   [](int) { return true; }, n, a.k, 32, &sm, &prefix, &sh, &exact, &rem, in_lds, kmin64, kmax64, 32 + 15, a.rank2);
    if (tid == 0) {
        // every sampled value whose bits >> (sh-32) are <= the prefix's is among the k
        // smallest: the largest float with that truncated prefix bounds them all
        const unsigned hi_bits = (unsigned)(prefix >> 32) | ((sh > 32) ? ((1u << (sh - 32)) - 1u) : 0u);
        if (hi_bits < PSH_INF_BITS) {
            const float tau0 = __uint_as_float(hi_bits) * PSH_TAU_MARGIN;   // strictly above the k-th value
            if (tau0 < __uint_as_float(PSH_INF_BITS) && tau0 > 0.0f) {
                QueryState* qs = a.qstate + b;
                qs->tau_bits = __float_as_uint(tau0);
                qs->tau2_bits = __float_as_uint(tau0);
                // bound-then-verify filter (see approx16): with S = nx + ny - 2c the real
                // value of a window's sum, a window the exact fp32 chain would admit
                // (acc < tau) satisfies  ny - 2c < tau(1+23u) - nx, and the computed
                // t = ny^ - 2c^ is within 2^-17 (nx + NY) of ny - 2c.  Everything rounded
                // towards "keep": in double, then up to the next float.
                const double e16 = 1.0 / 65536.0;
                const double nx = (double)qs->nx;
                const double A = (double)tau0 * (1.0 + e16) - nx * (1.0 - e16) + (nx * (1.0 + e16)) / 65536.0;
                float Af = (float)A;
                if ((double)Af < A) Af = __uint_as_float(Af >= 0.0f ? __float_as_uint(Af) + 1u : __float_as_uint(Af) - 1u);
                qs->thr_base = Af;
                if (a.blockmax && s_maxbits < PSH_INF_BITS) {
                    // matrix-core filter (scan_mx_kernel): scale = 2^s puts the largest sampled
                    // |value| into [4, 8) -- f16 keeps 11 bits down to 2^-14 and y~^2 stays below
                    // 65504 up to |y~| = 255 -- and mx_thr is the bound derived there, evaluated in
                    // double and rounded up (towards "keep")
                    int e = (int)((s_maxbits >> 23) & 255u) - 126;          // max in [2^(e-1), 2^e)
                    const int sexp = 3 - e;
                    const bool sane = sexp <= 60 && sexp >= -60 && s_maxbits >= 0x00800000u;   // normal, squares stay in fp32 range
                    const float sc = __uint_as_float((unsigned)(127 + sexp) << 23);
                    const float* xq = a.prep.queries + (int64_t)b * a.prep.W;
                    double nxs = 0.0;
                    for (int j = 0; j < a.prep.W; ++j) { const double v = (double)xq[j] * (double)sc; nxs += v * v; }
                    const double am = 1.0 / 512.0, bm = 1.0 / 262144.0;
                    const double taus = (double)tau0 * (double)sc * (double)sc;
                    const double T = taus * (1.0 + 1.0 / 131072.0) * (1.0 + 2.0 * am) - nxs * (1.0 - 3.0 * am) * (1.0 - 1e-12) + bm;
                    float Tf = (float)T;
                    if ((double)Tf < T) Tf = __uint_as_float(Tf >= 0.0f ? __float_as_uint(Tf) + 1u : __float_as_uint(Tf) - 1u);
                    if (sane && Tf == Tf && fabsf(Tf) < __uint_as_float(PSH_INF_BITS)) {
                        qs->mx_thr = Tf;
                        qs->mx_thr2 = Tf;
                        qs->mx_scale = sc;
                    }
                }
            }
        }
    }
    if (a.rank2 > 0 && a.rank2 < a.k && tid == 0) {
        // tau2: where the k-th smallest acc of the WHOLE ensemble is expected, with a 2x margin -- the
        // rank2-th smallest sampled minimum (rank2 = 2 k * sampled rows / rows), read off the first
        // histogram pass of the selection above (its bucket's upper edge).  Only an estimate: the scan
        // admits with tau as before but files what is below tau2 separately, and the selection falls back
        // to everything when fewer than k candidates are below tau2.
        QueryState* qs = a.qstate + b;
        const unsigned hi2 = (unsigned)(sm.prefix_b >> 32), t1 = qs->tau_bits;
        qs->tau2_bits = (hi2 < t1) ? hi2 : t1;           // positive floats: bit order = value order
        if (qs->mx_scale > 0.0f && hi2 < t1) {
            // the rejection threshold for tau2: same bound, same rounding towards "keep" as mx_thr
            const double sc = (double)qs->mx_scale;
            const float* xq = a.prep.queries + (int64_t)b * a.prep.W;
            double nxs = 0.0;
            for (int j = 0; j < a.prep.W; ++j) { const double v = (double)xq[j] * sc; nxs += v * v; }
            const double am = 1.0 / 512.0, bm = 1.0 / 262144.0;
            const double taus = (double)__uint_as_float(hi2) * sc * sc;
            const double T2 = taus * (1.0 + 1.0 / 131072.0) * (1.0 + 2.0 * am) - nxs * (1.0 - 3.0 * am) * (1.0 - 1e-12) + bm;
            float Tf = (float)T2;
            if ((double)Tf < T2) Tf = __uint_as_float(Tf >= 0.0f ? __float_as_uint(Tf) + 1u : __float_as_uint(Tf) - 1u);
            if (Tf == Tf && Tf < qs->mx_thr) qs->mx_thr2 = Tf;
        }
    }
    if (a.mq_frag) {
        // this query's share of scan_mq_kernel's B-fragment table: group b / 4, K-step s, lane
        // 32 hk + 8 (b & 3) + shift, element i holds -2 x~[k - shift] for k = 16 s + 8 hk + i in the band
        __syncthreads();
        const float sc = a.qstate[b].mx_scale;             // thread 0 above; 0 = filter not armed
        if (tid < 256) {
            const int i = tid & 7, shift = (tid >> 3) & 7, hk = (tid >> 6) & 1, s2 = tid >> 7;
            const int j = 16 * s2 + 8 * hk + i - shift;
            const bool in = j >= 0 && j < a.prep.W;
            const float xv = in ? a.prep.queries[(int64_t)b * a.prep.W + j] : 0.0f;
            const int ln = 32 * hk + 8 * (b & 3) + shift;
            reinterpret_cast<_Float16*>(a.mq_frag)[(((int64_t)(b >> 2) * 2 + s2) * 64 + ln) * 8 + i] =
                (_Float16)(in ? -2.0f * (xv * sc) : 0.0f);
        }
    }
}

// survivors -> k best by (d, r, t): radix select on the distance bits, ties at the k-th
// VALUE broken by a second radix select on (r, t), then a bitonic sort of the k selected
__device__ __forceinline__ bool item_less(uint64_t x, uint64_t y, const int2* rt) {
    const unsigned dx = (unsigned)(x >> 32), dy = (unsigned)(y >> 32);
    if (dx != dy) return dx < dy;
    const unsigned sx = (unsigned)x, sy = (unsigned)y;
    if (sx == sy) return false;
    if (dx == 0xffffffffu) return sx < sy; // both padding (made distinct by their position): any strict order
    if (sx == 0xffffffffu) return false;   // padding sorts last
    if (sy == 0xffffffffu) return true;
    const int2 a = rt[sx], b = rt[sy];
    if (a.x != b.x) return (unsigned)a.x < (unsigned)b.x;
    return (unsigned)a.y < (unsigned)b.y;
}

// the lower-bound comparison of the merge sort by ranking: distance bits alone -- the (almost always empty) range of
// equal distances is then stepped over with item_less -- except for PADDING items (d = 0xffffffff, kpad - k of them,
// made distinct by their low word, ordered by it): there the whole 64-bit key is the order, and stepping over
// thousands of "equal" padding entries one by one was 1 ms per level at k = 10000
__device__ __forceinline__ bool sort_key_less(uint64_t sib, uint64_t mine) {
    return ((unsigned)(mine >> 32) == 0xffffffffu) ? (sib < mine) : ((unsigned)(sib >> 32) < (unsigned)(mine >> 32));
}

__global__ __launch_bounds__(PSH_SELECT_THREADS) void select_kernel(SelectArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint64_t items[];   // kpad entries, then key_cap u32 keys
    __shared__ SelectShared sm;
    unsigned* keys = reinterpret_cast<unsigned*>(items + a.kpad);

    const int b = (int)blockIdx.x;
    const int tid = (int)threadIdx.x;
    int dbg_i = 0;
    auto mark = [&]() { if (a.dbg_times && b == 0 && tid == 0) a.dbg_times[dbg_i] = wall_clock64(); ++dbg_i; };
    mark();                                              // 0: start
    const float* cd = a.cand_d + (int64_t)b * a.cand_stride;
    const int2* crt = a.cand_rt + (int64_t)b * a.cand_stride;
    const bool slices = a.bcount != nullptr;
    int n;
    if (slices) {
        // ---- the scan left one slice per block: offs[] = exclusive prefix of their sizes
        const int* bc = a.bcount + (int64_t)b * PSH_MAX_BLOCKS;
        const int* bc2 = a.bcount2 ? a.bcount2 + (int64_t)b * PSH_MAX_BLOCKS : nullptr;
        if (tid == 0) { sm.overflow = 0; sm.offs[0] = 0; }
        __syncthreads();
        // two classes per slice (scan_mx_kernel): acc < tau2 at the front, [tau2, tau) at the back.  tau2 is
        // where the k-th smallest was EXPECTED (x2): when the front lists alone hold k candidates -- the
        // normal case, ~2k of them instead of ~17k -- the back lists are never read
        for (int pass = 0; pass < 2; ++pass) {
            const bool with_back = pass == 1;
            for (int i = tid; i < a.nblk; i += PSH_SELECT_THREADS) {
                const int cf = bc[i], cb = bc2 ? bc2[i] : 0;
                if (cf + cb > a.slice) sm.overflow = 1;              // the two ends met: entries were lost or overwritten
                if (bc2) sm.cnt_front[i] = cf < a.slice ? cf : a.slice;
                int c = cf + (with_back ? cb : 0);
                if (c > a.slice) c = a.slice;
                sm.offs[i + 1] = c;
            }
            __syncthreads();
            for (int off = 1; off < a.nblk; off <<= 1) {            // inclusive scan of offs[1..nblk]
                int v[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int i = tid + e * PSH_SELECT_THREADS + 1;
                    v[e] = (i <= a.nblk && i - off >= 1) ? sm.offs[i - off] : 0;
                }
                __syncthreads();
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int i = tid + e * PSH_SELECT_THREADS + 1;
                    if (i <= a.nblk) sm.offs[i] += v[e];
                }
                __syncthreads();
            }
            if (!bc2 || sm.offs[a.nblk] >= a.k) break;                // (uniform) enough candidates without the back lists
            if (pass == 0 && a.dataset) {
                // the back lists are needed after all: their unverified entries get their exact distance now
                // (the scan's arithmetic: sequential fp32 chain, correctly rounded sqrt and division)
                float* cdw = const_cast<float*>(cd);
                const float xn = a.qstate[b].xn;
                const float* xq = a.queries + (int64_t)b * a.W;
                const int tps2 = (a.nblk >= PSH_SELECT_THREADS) ? 1 : PSH_SELECT_THREADS / a.nblk;
                for (int sl = tid / tps2; sl < a.nblk; sl += PSH_SELECT_THREADS / tps2) {
                    int cb = bc2[sl];
                    if (cb > a.slice) cb = a.slice;
                    for (int j = tid % tps2; j < cb; j += tps2) {
                        const int64_t o = (int64_t)sl * a.slice + (a.slice - 1 - j);
                        if (__float_as_uint(cd[o]) != PSH_UNVERIFIED_BITS) continue;
                        const int2 rt = crt[o];
                        const float* y = a.dataset + ((int64_t)rt.x - a.r_offset) * a.T + rt.y;
                        float acc = 0.0f;
                        for (int jj = 0; jj < a.W; ++jj) { const float D = __fsub_rn(xq[jj], y[jj]); acc = __builtin_fmaf(D, D, acc); }
                        cdw[o] = dist_from_acc(acc, xn);
                    }
                }
                __threadfence();          // the staging below re-reads these slots from other threads: no stale L1 lines
                __syncthreads();
            }
        }
        n = sm.offs[a.nblk];
        // (n < k: the scan admitted below an ESTIMATE of the k-th smallest acc -- the embedded scan does, see
        //  psh_capi.hip -- and the estimate fell short: same recovery as an overflow, the exhaustive path)
        if (tid == 0 && (sm.overflow || n < a.k) && a.status) a.status[b] = PSH_STATUS_OVERFLOW_;
    } else {
        n = a.n_fixed;
    }
    mark();                                              // 1: slice prefix done
    if (tid == 0 && a.total) a.total[b] = n;
    // entry j of slice sl: front entries first, then (two-class slices, fallback only) the back ones
    const bool two_class = slices && a.bcount2 != nullptr;
    auto slot = [&](int sl, int j) -> int64_t {
        if (two_class) { const int cf = sm.cnt_front[sl]; if (j >= cf) return (int64_t)sl * a.slice + (a.slice - 1 - (j - cf)); }
        return (int64_t)sl * a.slice + j;
    };
    // candidate e lives at src(e): identity for flat inputs, slice lookup (binary search of
    // the owning block in LDS) otherwise -- no compaction pass over global memory
    auto src = [&](int e) -> int64_t {
        if (!slices) return a.list_stride ? (int64_t)(e / a.list_len) * a.list_stride + (e % a.list_len) : (int64_t)e;
        int lo = 0, hi = a.nblk;              // offs[lo] <= e < offs[hi]
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (sm.offs[mid] <= e) lo = mid; else hi = mid;
        }
        return slot(lo, e - sm.offs[lo]);
    };
    // walk the candidates as (e, src) pairs without a search: a group of threads per slice
    // (flat inputs: e == src).  Four independent loads in flight per thread.
    const int tps = slices ? ((a.nblk >= PSH_SELECT_THREADS) ? 1 : PSH_SELECT_THREADS / a.nblk) : 1;
    auto for_each_cand = [&](auto&& body) {
        if (slices) {
            const int q = tid % tps, sstep = PSH_SELECT_THREADS / tps;
            for (int sl = tid / tps; sl < a.nblk; sl += sstep) {
                const int e0 = sm.offs[sl], cnt = sm.offs[sl + 1] - e0;
                for (int j = q; j < cnt; j += tps) body(e0 + j, slot(sl, j));
            }
        } else {
            for (int e = tid; e < n; e += PSH_SELECT_THREADS) body(e, src(e));
        }
    };
    // distance bits are staged in LDS when they fit: every later pass runs at LDS latency;
    // their min / max fall out of the same pass
    const bool in_lds = n <= a.key_cap;
    unsigned kmin32 = 0xffffffffu, kmax32 = 0u;
    if (in_lds) {
        if (slices) {
            const int q = tid % tps, sstep = PSH_SELECT_THREADS / tps;
            for (int sl = tid / tps; sl < a.nblk; sl += sstep) {
                const int e0 = sm.offs[sl], cnt = sm.offs[sl + 1] - e0;
                for (int j = q; j < cnt; j += 4 * tps) {
                    float v[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) v[u] = (j + u * tps < cnt) ? cd[slot(sl, j + u * tps)] : 0.0f;
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (j + u * tps < cnt) {
                            const unsigned kb = __float_as_uint(v[u]);
                            keys[e0 + j + u * tps] = kb;
                            kmin32 = kb < kmin32 ? kb : kmin32;
                            kmax32 = kb > kmax32 ? kb : kmax32;
                        }
                }
            }
        } else {
#pragma unroll 4
            for (int e = tid; e < n; e += PSH_SELECT_THREADS) keys[e] = __float_as_uint(cd[src(e)]);
        }
        mark();                                          // 2: keys loaded
        if (slices) {      // block min / max of the staged keys
            if (tid == 0) { sm.kmin = ~0ull; sm.kmax = 0ull; }
            __syncthreads();
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const unsigned l2 = __shfl_xor(kmin32, off, 64), h2 = __shfl_xor(kmax32, off, 64);
                kmin32 = l2 < kmin32 ? l2 : kmin32;
                kmax32 = h2 > kmax32 ? h2 : kmax32;
            }
            if ((tid & 63) == 0) {
                atomicMin((unsigned long long*)&sm.kmin, (unsigned long long)kmin32 << 32);
                atomicMax((unsigned long long*)&sm.kmax, (unsigned long long)kmax32 << 32);
            }
        }
        __syncthreads();
    }
    const bool have_mm = in_lds && slices;
    const uint64_t kmin64 = have_mm ? sm.kmin : 0ull, kmax64 = have_mm ? sm.kmax : 0ull;
    auto dkey = [&](int e) -> unsigned { return in_lds ? keys[e] : __float_as_uint(cd[src(e)]); };
    // gathered lists keep distances and indices in separate blocks with different strides
    auto rt_index = [&](int e, int64_t sidx) -> int64_t {
        return (!slices && a.list_stride) ? (int64_t)(e / a.list_len) * a.list_stride_rt + (e % a.list_len) : sidx;
    };
    auto rt_of = [&](int e) -> int2 { return crt[rt_index(e, src(e))]; };

    int2* sel_rt = a.sel_rt + (int64_t)b * a.kpad;
    const bool skip_neg = a.skip_negative_rows != 0;
    auto live = [&](int e) { return !skip_neg || rt_of(e).x >= 0; };

    // flat inputs may carry padding entries (r < 0): only real candidates are ranked
    if (tid == 0) sm.cnt = 0;
    __syncthreads();
    int n_real = n;
    if (skip_neg) {
        int c = 0;
        for (int e = tid; e < n; e += PSH_SELECT_THREADS) c += (rt_of(e).x >= 0) ? 1 : 0;
        if (c) atomicAdd(&sm.cnt, c);
        __syncthreads();
        n_real = sm.cnt;
    }
    const int need = a.k < n_real ? a.k : n_real;

    // ---- which candidates are in
    uint64_t d_prefix = ~0ull, rt_prefix = ~0ull;
    int d_sh = 64, rt_sh = 64;       // 64: no restriction
    bool tie_select = false;
    if (need > 0 && need < n_real) {
        bool exact;
        int rem;
        if (!in_lds && slices && !skip_neg) {
            // more candidates than LDS holds keys for: walk the slices (no per-key search for the owning
            // block, independent loads) instead of indexing candidate e through src(e)
            radix_select64_walk([&](auto&& body) {
                                    for_each_cand([&](int, int64_t sidx) { body((uint64_t)__float_as_uint(cd[sidx]) << 32); });
                                },
                                need, 32, &sm, &d_prefix, &d_sh, &exact, &rem);
        } else {
            radix_select64([&](int e) { return (uint64_t)dkey(e) << 32; }, live, n, need, 32,
                           &sm, &d_prefix, &d_sh, &exact, &rem, have_mm, kmin64, kmax64);
        }
        if (!exact) {
            // the k-th distance VALUE is shared by more candidates than fit: the canonical
            // order keeps the smallest (r, t) among those ties
            tie_select = true;
            const unsigned dk = (unsigned)(d_prefix >> 32);
            bool exact2;
            int rem2;
            radix_select64([&](int e) { const int2 rt = rt_of(e); return ((uint64_t)(unsigned)rt.x << 32) | (uint64_t)(unsigned)rt.y; },
                           [&](int e) { return live(e) && dkey(e) == dk; }, n, rem, 0,
                           &sm, &rt_prefix, &rt_sh, &exact2, &rem2);
        }
    }

    mark();                                              // 4: radix select done
    // ---- collect the selected candidates
    for (int i = tid; i < a.kpad; i += PSH_SELECT_THREADS) items[i] = ~0ull;
    if (tid == 0) sm.nsel = 0;
    __syncthreads();
    if (need > 0) {
        // phase A: slots for the taken candidates, (r,t) still in global memory (a load
        // inside this loop would put one global round trip on every iteration)
        const unsigned dk = (unsigned)(d_prefix >> 32);
        auto park = [&](int slot, unsigned db, int64_t ridx) {
            if (slot < a.kpad) {
                items[slot] = ((uint64_t)db << 32) | (uint64_t)(unsigned)slot;
                sel_rt[slot] = make_int2((int)(ridx & 0xffffffffll), (int)(ridx >> 32));   // parked: where its (r,t) is
            }
        };
        if (in_lds && !skip_neg && !tie_select) {
            // the common case walks the staged keys in lock step, so the waves can claim
            // their slots with ONE LDS atomic per 64 keys (1024 single atomics on one word
            // took 7 us); only the ~k takers look up where their (r,t) lives
            const int n_up = (n + 63) & ~63;
            for (int e = tid; e < n_up; e += PSH_SELECT_THREADS) {
                const unsigned db = e < n ? keys[e] : 0xffffffffu;
                const bool take = e < n && (d_sh >= 64 || ((((uint64_t)db << 32) >> d_sh) <= (d_prefix >> d_sh)));
                const unsigned long long mask = __ballot(take);
                if (!mask) continue;
                int base = 0;
                if ((tid & 63) == 0) base = atomicAdd(&sm.nsel, __popcll(mask));
                base = __builtin_amdgcn_readfirstlane(base);
                // parked as -(e+1): phase B finds where candidate e lives (a search here would
                // serialise 8 dependent LDS reads into every iteration of this loop)
                if (take) park(base + __popcll(mask & ((1ull << (tid & 63)) - 1ull)), db, -(int64_t)e - 1);
            }
        } else {
            for_each_cand([&](int e, int64_t sidx) {
                const unsigned db = in_lds ? keys[e] : __float_as_uint(cd[sidx]);
                bool take;
                if (d_sh >= 64) take = true;
                else if (!tie_select) take = (((uint64_t)db << 32) >> d_sh) <= (d_prefix >> d_sh);
                else take = db <= dk;                      // ties resolved below
                if (!take) return;
                const int64_t ridx = rt_index(e, sidx);
                if (skip_neg || (tie_select && db == dk)) {          // flat inputs / tied values: the index decides
                    const int2 rt = crt[ridx];
                    if (skip_neg && rt.x < 0) return;
                    if (tie_select && db == dk) {
                        const uint64_t rk = ((uint64_t)(unsigned)rt.x << 32) | (uint64_t)(unsigned)rt.y;
                        if (!((rk >> rt_sh) <= (rt_prefix >> rt_sh))) return;
                    }
                }
                park(atomicAdd(&sm.nsel, 1), db, ridx);
            });
        }
    }
    __syncthreads();
    mark();                                              // 5: slots assigned
    {   // phase B: one independent load per selected candidate
        const int ns = sm.nsel < a.kpad ? sm.nsel : a.kpad;
        for (int sl = tid; sl < ns; sl += PSH_SELECT_THREADS) {
            const int2 parked = sel_rt[sl];
            int64_t ridx = ((int64_t)parked.y << 32) | (int64_t)(unsigned)parked.x;
            if (ridx < 0) { const int e = (int)(-ridx - 1); ridx = rt_index(e, src(e)); }
            sel_rt[sl] = crt[ridx];
        }
    }
    __syncthreads();

    mark();                                              // 6: (r,t) fetched
    // ---- bitonic sort of kpad items by (d bits, r, t): strides below 64 stay inside a
    // wave (shuffles, no barrier), only the wider ones go through LDS
    if (a.unsorted_ok) {
        // the caller merges and orders later: the selected items stay where the collection put them
    } else if (a.kpad <= PSH_SELECT_THREADS) {
        // one item per thread.  Pass 0 orders the 64-bit items as plain integers (distance
        // bits, then slot): exact unless two selected candidates share a distance value;
        // only then pass 1 repeats the network with the full (d, r, t) comparison.
        {   // pass 0: every wave sorts its 64 items in registers (21 shuffle steps, no barrier), then each
            // item finds its final position by counting, with one binary search per other run, the items
            // below it: 15 independent 7-probe chains of LDS reads instead of 34 more exchange steps, ten of
            // them through LDS with two block barriers each.  Keys are distinct (the slot is part of the key).
            uint64_t mine = (tid < a.kpad) ? items[tid] : ~0ull;
            if ((unsigned)(mine >> 32) == 0xffffffffu) mine = 0xffffffff00000000ull | (unsigned)tid;   // padding: distinct, last
            // position inside the wave's run: count the lanes holding a smaller key (64 broadcasts and
            // compares, no dependent chain; the 21-step shuffle network cost 5x that)
            int wrank = 0;
#pragma unroll 16
            for (int j = 0; j < 64; ++j) {
                const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)mine, j);
                const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(mine >> 32), j);
                wrank += ((((uint64_t)hi << 32) | lo) < mine) ? 1 : 0;
            }
            __syncthreads();
            if (tid < a.kpad) items[(tid & ~63) + wrank] = mine;      // runs of 64 (or the whole list), ascending
            __syncthreads();
            if (a.kpad > 64) {
                int rank = wrank;
                if (tid < a.kpad) {
                    const int nruns = a.kpad >> 6, w = tid >> 6;
                    // all (up to 16) binary searches advance together: 7 rounds of independent LDS probes
                    int pos[16];
#pragma unroll
                    for (int c = 0; c < 16; ++c) pos[c] = 0;
#pragma unroll
                    for (int step = 32; step > 0; step >>= 1)
#pragma unroll
                        for (int c = 0; c < 16; ++c) {
                            const int v = c < nruns ? c : w;
                            if (items[64 * v + pos[c] + step - 1] < mine) pos[c] += step;
                        }
#pragma unroll
                    for (int c = 0; c < 16; ++c) {
                        const int v = c < nruns ? c : w;
                        if (items[64 * v + pos[c]] < mine) pos[c] += 1;
                        if (v != w) rank += pos[c];
                    }
                }
                __syncthreads();
                if (tid < a.kpad) items[rank] = mine;
                __syncthreads();
            }
            mine = (tid < a.kpad) ? items[tid] : ~0ull;
            if (tid == 0) sm.cnt = 0;
            __syncthreads();
            // any equal distance values next to each other?  Only then the order among them needs (r, t)
            const bool tie = tid + 1 < a.kpad && (unsigned)(mine >> 32) != 0xffffffffu &&
                             (unsigned)(items[tid + 1] >> 32) == (unsigned)(mine >> 32);
            if (tie) sm.cnt = 1;
            __syncthreads();
        }
        if (sm.cnt != 0) {
            // pass 1 (rare): the bitonic network with the full (d, r, t) comparison
            uint64_t mine = (tid < a.kpad) ? items[tid] : ~0ull;
            for (int size = 2; size <= a.kpad; size <<= 1) {
                for (int stride = size >> 1; stride > 0; stride >>= 1) {
                    const bool ascending = ((tid & size) == 0);
                    uint64_t other;
                    if (stride >= 64) {
                        __syncthreads();
                        if (tid < a.kpad) items[tid] = mine;
                        __syncthreads();
                        other = (tid < a.kpad) ? items[tid ^ stride] : ~0ull;
                    } else {
                        other = __shfl_xor(mine, stride, 64);    // (quad DPP moves and a two-buffer exchange were tried: slower)
                    }
                    const bool i_am_low = (tid & stride) == 0;
                    const bool other_less = item_less(other, mine, sel_rt), mine_less = item_less(mine, other, sel_rt);
                    const bool take_other = (i_am_low == ascending) ? other_less : mine_less;
                    if (take_other) mine = other;
                }
            }
            __syncthreads();
            if (tid < a.kpad) items[tid] = mine;
            __syncthreads();
        }
    } else {
        bool need_network = true;
        if (a.sort_buf_ok) {
            // kpad > 1024 (the tutorial's k = 8192, the reference test's k = 10000): a merge sort by RANKING on
            // the 64-bit integer keys -- runs of 64 ordered by in-wave counting, then log2(kpad / 64) levels in
            // which every item binary-searches its sibling run and writes itself to its merged position in the
            // other buffer: ~63 LDS probes per item at kpad = 8192 instead of 91 compare-exchange sweeps with a
            // block barrier each.  The full (d, r, t) comparison throughout: among 8192 selected distances a
            // few equal values are the rule (birthday effect on ~1e7 representable values), and the (r, t)
            // look-up only runs in the lanes that actually meet one.
            uint64_t* bufA = items;
            uint64_t* bufB = reinterpret_cast<uint64_t*>(keys);
            const int per = a.kpad / PSH_SELECT_THREADS;            // 2, 4 or 8
            for (int i = 0; i < per; ++i) {
                const int e = (i * (PSH_SELECT_THREADS / 64) + (tid >> 6)) * 64 + (tid & 63);
                uint64_t mine = bufA[e];
                if ((unsigned)(mine >> 32) == 0xffffffffu) mine = 0xffffffff00000000ull | (unsigned)e;   // padding: distinct, last
                const unsigned myd = (unsigned)(mine >> 32);
                int wrank = 0;
#pragma unroll 16
                for (int j = 0; j < 64; ++j) {
                    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)mine, j);
                    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)myd, j);
                    wrank += (hi < myd) ? 1 : 0;                               // integer compare in the common case
                    if (__any(hi == myd && j != (tid & 63)))                   // an equal distance value: (r, t) decides
                        wrank += (hi == myd && item_less(((uint64_t)hi << 32) | lo, mine, sel_rt)) ? 1 : 0;
                }
                bufB[(e & ~63) + wrank] = mine;
            }
            __syncthreads();
            uint64_t* src = bufB;
            uint64_t* dst = bufA;
            for (int len = 64; len < a.kpad; len <<= 1) {
                const int sh = 31 - __builtin_clz((unsigned)len);
                uint64_t mine[8];
                int lo[8];
                const uint64_t* sib[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int e = tid + (i < per ? i : 0) * PSH_SELECT_THREADS;
                    mine[i] = src[e];
                    sib[i] = src + (size_t)((e >> sh) ^ 1) * len;
                    lo[i] = 0;
                }
                // lower bound on the distance bits alone (branch-free: the 8 searches of a thread interleave),
                // then step over the (almost always empty) range of equal distance values with the full comparison
                for (int step = len >> 1; step > 0; step >>= 1)
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        if (i < per && sort_key_less(sib[i][lo[i] + step - 1], mine[i])) lo[i] += step;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    if (i < per) {
                        const int e = tid + i * PSH_SELECT_THREADS;
                        if (sort_key_less(sib[i][lo[i]], mine[i])) lo[i] += 1;
                        while (lo[i] < len && (unsigned)(sib[i][lo[i]] >> 32) == (unsigned)(mine[i] >> 32)
                               && item_less(sib[i][lo[i]], mine[i], sel_rt)) lo[i] += 1;
                        dst[(size_t)((e >> sh) >> 1) * 2 * len + (e & (len - 1)) + lo[i]] = mine[i];
                    }
                }
                __syncthreads();
                uint64_t* t2 = src; src = dst; dst = t2;
            }
            if (src != items) {
                for (int e = tid; e < a.kpad; e += PSH_SELECT_THREADS) items[e] = src[e];
            }
            need_network = false;
        } else if (a.sort_scratch) {
            // kpad = 16384 (the reference test's k = 10000): the items fill the LDS, so the second buffer of the same
            // merge sort by ranking is GLOBAL scratch (this query's candidate slots, consumed by now): every level
            // searches in LDS, writes the merged order to the scratch and copies it back (128 KB, coalesced, L2) --
            // ~9 us per level against the bitonic network's 105 barrier-separated sweeps.
            uint64_t* G = a.sort_scratch + (int64_t)b * a.cand_stride;
            const int per = a.kpad / PSH_SELECT_THREADS;            // 16
            for (int i = 0; i < per; ++i) {
                const int e = (i * (PSH_SELECT_THREADS / 64) + (tid >> 6)) * 64 + (tid & 63);
                uint64_t mine = items[e];
                if ((unsigned)(mine >> 32) == 0xffffffffu) mine = 0xffffffff00000000ull | (unsigned)e;   // padding: distinct, last
                const unsigned myd = (unsigned)(mine >> 32);
                int wrank = 0;
#pragma unroll 16
                for (int j = 0; j < 64; ++j) {
                    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)mine, j);
                    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)myd, j);
                    wrank += (hi < myd) ? 1 : 0;
                    if (__any(hi == myd && j != (tid & 63)))
                        wrank += (hi == myd && item_less(((uint64_t)hi << 32) | lo, mine, sel_rt)) ? 1 : 0;
                }
                G[(e & ~63) + wrank] = mine;
            }
            __threadfence_block();       // the block's waves share one vL1D: workgroup scope orders the scratch traffic (an agent-scope fence writes back the L2: ~1 ms per level)
            __syncthreads();
            for (int e = tid; e < a.kpad; e += PSH_SELECT_THREADS) items[e] = G[e];
            __syncthreads();
            for (int len = 64; len < a.kpad; len <<= 1) {
                const int sh = 31 - __builtin_clz((unsigned)len);
                for (int i0 = 0; i0 < per; i0 += 8) {
                    uint64_t mine[8];
                    int lo[8];
                    const uint64_t* sib[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const int e = tid + (i0 + i) * PSH_SELECT_THREADS;
                        mine[i] = items[e];
                        sib[i] = items + (size_t)((e >> sh) ^ 1) * len;
                        lo[i] = 0;
                    }
                    for (int step = len >> 1; step > 0; step >>= 1)
#pragma unroll
                        for (int i = 0; i < 8; ++i)
                            if (sort_key_less(sib[i][lo[i] + step - 1], mine[i])) lo[i] += step;
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const int e = tid + (i0 + i) * PSH_SELECT_THREADS;
                        if (sort_key_less(sib[i][lo[i]], mine[i])) lo[i] += 1;
                        while (lo[i] < len && (unsigned)(sib[i][lo[i]] >> 32) == (unsigned)(mine[i] >> 32)
                               && item_less(sib[i][lo[i]], mine[i], sel_rt)) lo[i] += 1;
                        G[(size_t)((e >> sh) >> 1) * 2 * len + (e & (len - 1)) + lo[i]] = mine[i];
                    }
                }
                __threadfence_block();       // the block's waves share one vL1D: workgroup scope orders the scratch traffic (an agent-scope fence writes back the L2: ~1 ms per level)
                __syncthreads();
                for (int e = tid; e < a.kpad; e += PSH_SELECT_THREADS) items[e] = G[e];
                __syncthreads();
            }
            need_network = false;
        }
        if (need_network)
        for (int size = 2; size <= a.kpad; size <<= 1) {
            for (int stride = size >> 1; stride > 0; stride >>= 1) {
                __syncthreads();
                for (int i = tid; i < (a.kpad >> 1); i += PSH_SELECT_THREADS) {
                    const int lo = ((i / stride) * (stride << 1)) + (i % stride);
                    const int hi = lo + stride;
                    const bool ascending = ((lo & size) == 0);
                    const uint64_t x = items[lo], y = items[hi];
                    const bool swap = ascending ? item_less(y, x, sel_rt) : item_less(x, y, sel_rt);
                    if (swap) { items[lo] = y; items[hi] = x; }
                }
            }
        }
    }
    __syncthreads();

    mark();                                              // 7: sorted
    // ---- write out
    const int nsel = sm.nsel < need ? sm.nsel : need;
    for (int i = tid; i < a.k; i += PSH_SELECT_THREADS) {
        float d = __uint_as_float(PSH_INF_BITS);
        int2 rt = make_int2(-1, -1);
        if (i < nsel) {
            const uint64_t it = items[i];
            d = __uint_as_float((unsigned)(it >> 32));
            rt = sel_rt[(unsigned)it];
        }
        a.out_d[(int64_t)b * a.k + i] = d;
        a.out_idx[((int64_t)b * a.k + i) * 2 + 0] = rt.x;
        a.out_idx[((int64_t)b * a.k + i) * 2 + 1] = rt.y;
    }
    if (tid == 0 && a.qstate) a.qstate[b].n_valid = nsel;
    mark();                                              // 6: written
}

// ----------------------------------------------------------------------------------
// merge of per-shard results that arrive SORTED (the cross-GPU merge after the all-gather)
// ----------------------------------------------------------------------------------
// G lists of k_in entries, each ascending by (d, r, t), shard g holding smaller rows than shard g+1
// (padding: d = +inf, r = -1, at the end).  No selection pass, no sort: an entry's position in the
// merged order is its own position plus, per other list, the number of entries that precede it --
// a binary search on the distance bits; equal distances across lists are ordered by the list index,
// which IS the (r, t) order because shards are ascending row blocks.  Only entries that can be among
// the k best take part: with c = ceil(1.25 k / G), everything above P = max_g list_g[c] is out (at
// least G (c + 1) >= k entries are <= P).  One block per query; the distance keys sit in LDS.
__global__ __launch_bounds__(PSH_SELECT_THREADS) void merge_sorted_kernel(MergeSortedArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned mkeys[];     // G x k_in distance bits (non-negative floats: bit order)
    __shared__ int ncut[64];                                              // per list: entries <= P
    __shared__ unsigned pivot;
    const int b = (int)blockIdx.x, tid = (int)threadIdx.x;
    const int G = a.G, kin = a.k_in;
    for (int e = tid; e < G * kin; e += PSH_SELECT_THREADS) {
        const int g = e / kin, j = e - g * kin;
        mkeys[e] = __float_as_uint(a.d[(int64_t)g * a.stride_d + (int64_t)b * kin + j]);
    }
    if (tid == 0) pivot = 0u;
    __syncthreads();
    int c = (5 * a.k + 4 * G - 1) / (4 * G);           // 1.25 k / G: G (c + 1) >= k entries are <= P
    if (c > kin - 1) c = kin - 1;
    if (tid < G) atomicMax(&pivot, mkeys[tid * kin + c]);
    __syncthreads();
    const unsigned P = pivot;
    if (tid < G) {                                      // upper bound of P in list tid
        const unsigned* L = mkeys + tid * kin;
        int lo = 0, hi = kin;                           // L[lo-1] <= P < L[hi]
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (L[mid] <= P) lo = mid + 1; else hi = mid; }
        ncut[tid] = lo;
    }
    __syncthreads();
    // candidates: list g, positions [0, ncut[g]); flattened over (g, j) with a running offset
    int total = 0;
    for (int g = 0; g < G; ++g) total += ncut[g];
    for (int e = tid; e < total; e += PSH_SELECT_THREADS) {
        int g = 0, j = e;
        while (j >= ncut[g]) { j -= ncut[g]; ++g; }
        const unsigned mine = mkeys[g * kin + j];
        int rank = j;
        for (int v = 0; v < G; ++v) {
            if (v == g) continue;
            const unsigned* L = mkeys + v * kin;
            int lo = 0, hi = ncut[v];                   // entries beyond the cut are > P >= mine
            if (v < g) { while (lo < hi) { const int mid = (lo + hi) >> 1; if (L[mid] <= mine) lo = mid + 1; else hi = mid; } }
            else       { while (lo < hi) { const int mid = (lo + hi) >> 1; if (L[mid] < mine) lo = mid + 1; else hi = mid; } }
            rank += lo;
        }
        if (rank < a.k) {
            const int2 rt = a.rt[(int64_t)g * a.stride_rt + (int64_t)b * kin + j];
            a.out_d[(int64_t)b * a.k + rank] = __uint_as_float(mine);
            a.out_idx[((int64_t)b * a.k + rank) * 2 + 0] = rt.x;
            a.out_idx[((int64_t)b * a.k + rank) * 2 + 1] = rt.y;
        }
    }
    // fewer than k entries in all (k larger than the lists together): pad
    for (int i = G * kin + tid; i < a.k; i += PSH_SELECT_THREADS) {
        a.out_d[(int64_t)b * a.k + i] = __uint_as_float(PSH_INF_BITS);
        a.out_idx[((int64_t)b * a.k + i) * 2 + 0] = -1;
        a.out_idx[((int64_t)b * a.k + i) * 2 + 1] = -1;
    }
}

// exhaustive path: the running best goes behind the next chunk's window slots
__global__ void reseed_kernel(ReseedArgs a) {
    const int b = (int)blockIdx.y;
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= a.k) return;
    const bool ok = i < a.qstate[b].n_valid;
    const int64_t o = (int64_t)b * a.cand_stride + a.offset + i;
    a.cand_d[o] = ok ? a.out_d[(int64_t)b * a.k + i] : __uint_as_float(PSH_INF_BITS);
    a.cand_rt[o] = ok ? make_int2(a.out_idx[((int64_t)b * a.k + i) * 2], a.out_idx[((int64_t)b * a.k + i) * 2 + 1])
                      : make_int2(-1, -1);
}

// ----------------------------------------------------------------------------------
// path gather (path_shadowing.py:211-216)
// ----------------------------------------------------------------------------------
__global__ void gather_kernel(GatherArgs a) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // over n * C * len
    const int64_t total = a.n * a.C * a.len;
    if (e >= total) return;
    const int64_t i = e / (a.C * a.len);
    const int64_t c = (e / a.len) % a.C;
    const int64_t j = e % a.len;
    const int64_t r = (int64_t)a.idx[2 * i] - a.r_offset;
    const int64_t t = a.idx[2 * i + 1];
    if (r < 0 || r >= a.R || t < 0 || t + a.len > a.T) return;
    a.out[e] = a.dataset[(r * a.C + c) * a.T + t + j];
}

// ----------------------------------------------------------------------------------
// launchers (host)
// ----------------------------------------------------------------------------------
template <int WT, bool ALIGNED, int MODE>
static hipError_t allow_big_lds(size_t shmem) {
    if (shmem <= 48 * 1024) return hipSuccess;
    return hipFuncSetAttribute((const void*)scan_kernel<WT, ALIGNED, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
}

template <int WT, bool ALIGNED>
static hipError_t launch_scan_mode(const ScanArgs& a, int mode, int grid, size_t shmem, hipStream_t s) {
    hipError_t e = mode == PSH_MODE_BOOT ? allow_big_lds<WT, ALIGNED, PSH_MODE_BOOT>(shmem)
                 : mode == PSH_MODE_FILTER ? allow_big_lds<WT, ALIGNED, PSH_MODE_FILTER>(shmem)
                                           : allow_big_lds<WT, ALIGNED, PSH_MODE_ALL>(shmem);
    if (e != hipSuccess) return e;
    switch (mode) {
        case PSH_MODE_BOOT:
            hipLaunchKernelGGL((scan_kernel<WT, ALIGNED, PSH_MODE_BOOT>), dim3(grid), dim3(PSH_SCAN_THREADS), shmem, s, a);
            break;
        case PSH_MODE_FILTER:
            hipLaunchKernelGGL((scan_kernel<WT, ALIGNED, PSH_MODE_FILTER>), dim3(grid), dim3(PSH_SCAN_THREADS), shmem, s, a);
            break;
        default:
            hipLaunchKernelGGL((scan_kernel<WT, ALIGNED, PSH_MODE_ALL>), dim3(grid), dim3(PSH_SCAN_THREADS), shmem, s, a);
            break;
    }
    return hipGetLastError();
}

size_t scan_shmem_bytes(int tile_floats, int B, int emb_d, int W, int threads) {
    const int nw = threads / 64;
    size_t n = (size_t)tile_floats * nw * sizeof(float)                              // wave-private tiles
               + (size_t)(((B + 3) & ~3) + 4) * sizeof(int)                         // per-query append cursors + work cursor
               + (size_t)nw * PSH_PEND * 16;                                        // wave-private pending admissions
    if (emb_d > 0) n += (size_t)emb_d * ((W + 3) & ~3) * sizeof(float) + (size_t)emb_d * sizeof(int2)    // kernel matrix, tap spans
                      + sizeof(NestHdr) + (size_t)emb_d * 16                                             // suffix-rows fast path: closing order
                      + (size_t)(emb_d + 1) * 32 + 8                                                      //   and support masks
                      + (size_t)nw * 192 * 4;                                                             //   verification scratch
    return n;
}

bool scan_mx_supported(int W, int B) { return W >= 1 && W <= 33 && B == 1; }

size_t scan_mx_shmem_bytes(int tile_floats, int /*B*/) {
    return (size_t)tile_floats * (PSH_SCAN_THREADS / 64) * sizeof(float) + 32
           + (size_t)(PSH_SCAN_THREADS / 64) * PSH_MX_PEND * 16
           + (size_t)(PSH_SCAN_THREADS / 64) * 2 * PSH_MX_NHALF * sizeof(_Float16);
}

// launch a <WT, ALIGNED> kernel family member: W = 20 has its own instantiation, other lengths run WT = 0
template <typename K>
static hipError_t launch_big_lds(K kernel, dim3 grid, int threads, size_t shmem, hipStream_t s, const ScanArgs& a) {
    hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kernel, grid, dim3(threads), shmem, s, a);
    return hipGetLastError();
}

template <bool ALIGNED>
static hipError_t launch_scan_mx(const ScanArgs& a, int grid, hipStream_t s) {
    const size_t shmem = scan_mx_shmem_bytes(a.tile_floats, a.B);
    return a.W == 20 ? launch_big_lds(scan_mx_kernel<20, ALIGNED>, dim3(grid), PSH_SCAN_THREADS, shmem, s, a)
                     : launch_big_lds(scan_mx_kernel<0, ALIGNED>, dim3(grid), PSH_SCAN_THREADS, shmem, s, a);
}

bool scan_mq_supported(int W, int B) { return W >= 1 && W <= 25 && B >= 2; }

size_t scan_mq_shmem_bytes(int tile_floats, int B) {
    constexpr int NW = PSH_MQ_THREADS / 64;
    return (size_t)tile_floats * NW * sizeof(float) + (size_t)(((B + 3) & ~3) + 4) * sizeof(int)
           + (size_t)NW * PSH_PEND * 16 + (size_t)NW * 2 * PSH_MX_NHALF * sizeof(_Float16)
           + (size_t)(PSH_MQ_CHUNK / 4) * 2 * 64 * 8 * sizeof(_Float16) + (size_t)2 * PSH_MQ_CHUNK * sizeof(float)
           + (size_t)PSH_MQ_CHUNK * 25 * sizeof(float) + (size_t)NW * PSH_MQ_QCAP * sizeof(unsigned);
}

int scan_mq_chunks(int B) { return (B + PSH_MQ_CHUNK - 1) / PSH_MQ_CHUNK; }

bool boot_mq_supported(int W) { return W >= 1 && W <= 25; }

size_t boot_mq_shmem_bytes(int tile_floats) {
    constexpr int NW = PSH_MQ_THREADS / 64;
    return (size_t)tile_floats * NW * sizeof(float) + 16 + (size_t)NW * 2 * PSH_MX_NHALF * sizeof(_Float16)
           + (size_t)(PSH_MQ_CHUNK / 4) * 2 * 64 * 8 * sizeof(_Float16) + (size_t)PSH_MQ_CHUNK * sizeof(float);
}

hipError_t launch_boot_mq(const ScanArgs& a, bool aligned, int grid_x, hipStream_t s) {
    const size_t shmem = boot_mq_shmem_bytes(a.tile_floats);
    const dim3 grid(grid_x, scan_mq_chunks(a.B));
    if (a.W == 20)
        return aligned ? launch_big_lds(boot_mq_kernel<20, true>, grid, PSH_MQ_THREADS, shmem, s, a)
                       : launch_big_lds(boot_mq_kernel<20, false>, grid, PSH_MQ_THREADS, shmem, s, a);
    return aligned ? launch_big_lds(boot_mq_kernel<0, true>, grid, PSH_MQ_THREADS, shmem, s, a)
                   : launch_big_lds(boot_mq_kernel<0, false>, grid, PSH_MQ_THREADS, shmem, s, a);
}

hipError_t launch_scan_mq(const ScanArgs& a, bool aligned, int grid_x, hipStream_t s) {
    const size_t shmem = scan_mq_shmem_bytes(a.tile_floats, a.B);
    const dim3 grid(grid_x, scan_mq_chunks(a.B));
    if (a.W == 20)
        return aligned ? launch_big_lds(scan_mq_kernel<20, true>, grid, PSH_MQ_THREADS, shmem, s, a)
                       : launch_big_lds(scan_mq_kernel<20, false>, grid, PSH_MQ_THREADS, shmem, s, a);
    return aligned ? launch_big_lds(scan_mq_kernel<0, true>, grid, PSH_MQ_THREADS, shmem, s, a)
                   : launch_big_lds(scan_mq_kernel<0, false>, grid, PSH_MQ_THREADS, shmem, s, a);
}

#define PSH_EMB_WIDE_THREADS 512
#define PSH_EMB_WIDE_BG 10
#define PSH_EMB_WIDE_NBG 6

template <bool ALIGNED, int MODE>
static hipError_t launch_embed_mode(const ScanArgs& a, int grid, size_t shmem, hipStream_t s) {
    if (a.emb_wide) {
        hipError_t e = hipFuncSetAttribute((const void*)embed_scan_kernel<ALIGNED, MODE, PSH_EMB_WIDE_THREADS, PSH_EMB_WIDE_BG, PSH_EMB_WIDE_NBG>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((embed_scan_kernel<ALIGNED, MODE, PSH_EMB_WIDE_THREADS, PSH_EMB_WIDE_BG, PSH_EMB_WIDE_NBG>), dim3(grid),
                           dim3(PSH_EMB_WIDE_THREADS), shmem, s, a);
        return hipGetLastError();
    }
    if (shmem > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)embed_scan_kernel<ALIGNED, MODE, PSH_SCAN_THREADS, PSH_EMB_BG, PSH_NEST_BG>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((embed_scan_kernel<ALIGNED, MODE, PSH_SCAN_THREADS, PSH_EMB_BG, PSH_NEST_BG>), dim3(grid), dim3(PSH_SCAN_THREADS), shmem, s, a);
    return hipGetLastError();
}

template <bool ALIGNED>
static hipError_t launch_embed(const ScanArgs& a, int mode, int grid, size_t shmem, hipStream_t s) {
    switch (mode) {
        case PSH_MODE_BOOT: return launch_embed_mode<ALIGNED, PSH_MODE_BOOT>(a, grid, shmem, s);
        case PSH_MODE_FILTER: return launch_embed_mode<ALIGNED, PSH_MODE_FILTER>(a, grid, shmem, s);
        default: return launch_embed_mode<ALIGNED, PSH_MODE_ALL>(a, grid, shmem, s);
    }
}

hipError_t launch_scan(const ScanArgs& a, int mode, bool aligned, int grid, hipStream_t s) {
    const size_t shmem = scan_shmem_bytes(a.tile_floats, a.B, a.ker ? a.emb_d : 0, a.W, (a.ker && a.emb_wide) ? PSH_EMB_WIDE_THREADS : PSH_SCAN_THREADS);
    if (a.ker) return aligned ? launch_embed<true>(a, mode, grid, shmem, s) : launch_embed<false>(a, mode, grid, shmem, s);
    if (a.use_mx && mode == PSH_MODE_FILTER && scan_mx_supported(a.W, a.B))
        return aligned ? launch_scan_mx<true>(a, grid, s) : launch_scan_mx<false>(a, grid, s);
    if (a.W == 20) {
        return aligned ? launch_scan_mode<20, true>(a, mode, grid, shmem, s)
                       : launch_scan_mode<20, false>(a, mode, grid, shmem, s);
    }
    return aligned ? launch_scan_mode<0, true>(a, mode, grid, shmem, s)
                   : launch_scan_mode<0, false>(a, mode, grid, shmem, s);
}

hipError_t scan_blocks_per_cu(int W, bool aligned, bool embedded, size_t shmem, int* out) {
    int n = 0;
    hipError_t e;
    if (embedded) {
        e = aligned ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, embed_scan_kernel<true, PSH_MODE_FILTER, PSH_SCAN_THREADS, PSH_EMB_BG, PSH_NEST_BG>, PSH_SCAN_THREADS, shmem)
                    : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, embed_scan_kernel<false, PSH_MODE_FILTER, PSH_SCAN_THREADS, PSH_EMB_BG, PSH_NEST_BG>, PSH_SCAN_THREADS, shmem);
    } else if (W == 20) {
        e = aligned ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, scan_kernel<20, true, PSH_MODE_FILTER>, PSH_SCAN_THREADS, shmem)
                    : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, scan_kernel<20, false, PSH_MODE_FILTER>, PSH_SCAN_THREADS, shmem);
    } else {
        e = aligned ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, scan_kernel<0, true, PSH_MODE_FILTER>, PSH_SCAN_THREADS, shmem)
                    : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, scan_kernel<0, false, PSH_MODE_FILTER>, PSH_SCAN_THREADS, shmem);
    }
    *out = n;
    return e;
}

hipError_t launch_prep(const PrepArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(prep_kernel, dim3(a.B), dim3(64), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_qnorm(const float* q, int B, int W, float* out, hipStream_t s) {
    hipLaunchKernelGGL(qnorm_kernel, dim3((B + 63) / 64), dim3(64), 0, s, q, B, W, out);
    return hipGetLastError();
}
hipError_t launch_threshold(const ThresholdArgs& a0, int B, hipStream_t s) {
    ThresholdArgs a = a0;
    size_t shmem = (size_t)a.n_entries * sizeof(unsigned);
    a.keys_in_lds = shmem <= 128 * 1024;
    if (!a.keys_in_lds) shmem = 0;
    if (shmem > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)threshold_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(threshold_kernel, dim3(B), dim3(PSH_SELECT_THREADS), shmem, s, a);
    return hipGetLastError();
}
hipError_t launch_select(const SelectArgs& a0, int B, hipStream_t s) {
    SelectArgs a = a0;
    // LDS: the k items being sorted + as many staged distance keys as fit next to them
    const size_t lds_budget = 128 * 1024;      // of 160 KB; SelectShared (static) takes ~25 KB
    const size_t items_bytes = (size_t)a.kpad * sizeof(uint64_t);
    int64_t key_cap = items_bytes < lds_budget ? (int64_t)((lds_budget - items_bytes) / sizeof(unsigned)) : 0;
    const int64_t n_max = a.bcount ? (int64_t)a.nblk * a.slice : (int64_t)a.n_fixed;
    if (key_cap > n_max) key_cap = n_max;
    a.key_cap = (int)key_cap;
    // kpad > 1024: the ordering stage wants a second kpad-item buffer behind the items (merge sort by ranking)
    int64_t area = key_cap;
    a.sort_buf_ok = (a.kpad > PSH_SELECT_THREADS && a.kpad <= 8 * PSH_SELECT_THREADS && 2 * items_bytes <= lds_budget) ? 1 : 0;
    // beyond that (kpad = 16384): the second buffer is the query's own candidate slots, if the caller says they are free by then
    if (a.sort_buf_ok || a.kpad <= 8 * PSH_SELECT_THREADS || (int64_t)a.cand_stride < (int64_t)a.kpad) a.sort_scratch = nullptr;
    if (a.sort_buf_ok && area < 2 * (int64_t)a.kpad) area = 2 * (int64_t)a.kpad;
    const size_t shmem = items_bytes + (size_t)area * sizeof(unsigned);
    if (shmem > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)select_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(select_kernel, dim3(B), dim3(PSH_SELECT_THREADS), shmem, s, a);
    return hipGetLastError();
}
hipError_t launch_merge_sorted(const MergeSortedArgs& a, int B, hipStream_t s) {
    const size_t shmem = (size_t)a.G * a.k_in * sizeof(unsigned);
    if (shmem > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)merge_sorted_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(merge_sorted_kernel, dim3(B), dim3(PSH_SELECT_THREADS), shmem, s, a);
    return hipGetLastError();
}
hipError_t launch_reseed(const ReseedArgs& a, int B, hipStream_t s) {
    hipLaunchKernelGGL(reseed_kernel, dim3((a.k + 255) / 256, B), dim3(256), 0, s, a);
    return hipGetLastError();
}
size_t rows_shmem_bytes(int ds, int B) {      // ds: LDS floats per row
    return (size_t)(PSH_ROWS_THREADS / 64) * 64 * (ds + 1) * sizeof(float) + (size_t)((B + 3) & ~3) * sizeof(int)
           + (size_t)(PSH_ROWS_THREADS / 64) * PSH_PEND * 16;
}

// one-window rows: a.n_rows rows from a.row0 at a.row_stride, `grid` blocks (<= PSH_MAX_BLOCKS)
hipError_t launch_rows(ScanArgs a, int mode, int grid, hipStream_t s) {
    const int aligned16 = (((uintptr_t)a.dataset & 15u) == 0 && ((a.T * a.row0) % 4) == 0 && (a.T % 4 == 0 || a.row_stride == 1)) ? 1 : 0;
    int staging = PSH_ROWS_SPLIT;
    int ds = a.W | 1;
    if (aligned16 && a.row_stride == 1 && a.T <= a.W + 12) {
        const int64_t g = a.T & -a.T;                           // largest power of two dividing T
        if (g <= 2) { staging = PSH_ROWS_FLAT; ds = (int)a.T; }
        else { staging = PSH_ROWS_QUADS; ds = (int)((a.W + 3) & ~3) | 1; }
    }
    a.tile_floats = ds;
    const size_t shmem = rows_shmem_bytes(ds, a.B);
    if (mode == PSH_MODE_BOOT) {
        hipError_t e = hipFuncSetAttribute((const void*)rows_kernel<PSH_MODE_BOOT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((rows_kernel<PSH_MODE_BOOT>), dim3(grid), dim3(PSH_ROWS_THREADS), shmem, s, a, aligned16, staging);
    } else if (mode == PSH_MODE_ALL) {
        hipError_t e = hipFuncSetAttribute((const void*)rows_kernel<PSH_MODE_ALL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((rows_kernel<PSH_MODE_ALL>), dim3(grid), dim3(PSH_ROWS_THREADS), shmem, s, a, aligned16, staging);
    } else {
        hipError_t e = hipFuncSetAttribute((const void*)rows_kernel<PSH_MODE_FILTER>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((rows_kernel<PSH_MODE_FILTER>), dim3(grid), dim3(PSH_ROWS_THREADS), shmem, s, a, aligned16, staging);
    }
    return hipGetLastError();
}

hipError_t launch_gather(const GatherArgs& a, hipStream_t s) {
    const int64_t total = a.n * a.C * a.len;
    if (total <= 0) return hipSuccess;
    hipLaunchKernelGGL(gather_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a);
    return hipGetLastError();
}

}  // namespace psh
