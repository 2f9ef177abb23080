// psh_select.hip -- the one-block kernels around the scans: per-query preparation, threshold of the bootstrap sample,
// final top-k selection and ordering, merge of per-shard lists, path gather; with their launchers.  Part of
// libpsh_hip.so; shared device code in psh_device.h, design overview at the top of psh_scan.hip.
#include "psh_device.h"

namespace psh {

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
__global__ __launch_bounds__(PSH_SELECT_THREADS, 8) void threshold_kernel(ThresholdArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned tkeys[];   // n_entries (or nothing)
    __shared__ SelectShared sm;
    const int b = (int)blockIdx.x;
    const int tid = (int)threadIdx.x;
    const float* v = a.minbuf + (int64_t)b * a.min_stride;
    const int n = a.n_entries;
    __shared__ unsigned s_maxbits;                         // largest |value| among the sampled data and this query
    if (tid == 0) { prep_query(a.prep, b); s_maxbits = 0u; sm.prefix_b = ~0ull; }   // ||x||, sum of squares, state reset
    // 8-bit test (end of this kernel): what it reads from memory is requested HERE -- this block's query and the batch's
    // constants (mq_prep_kernel's meta words) go to LDS --, so that the round trips pass behind the selection, not after it
    __shared__ float s_xb[25];
    __shared__ unsigned s_meta[3];                         // largest ||x - s0 x^||^2, largest ||x||^2, largest |x| of the batch (bits)
    __shared__ float s_sc;
    const bool i8prep = a.mq_frag && a.mq_i8;
    if (i8prep) {
        if (tid < 25) s_xb[tid] = tid < a.prep.W ? a.prep.queries[(int64_t)b * a.prep.W + tid] : 0.0f;
        if (tid >= 32 && tid < 35)
            s_meta[tid - 32] = reinterpret_cast<const unsigned*>(reinterpret_cast<const char*>(a.mq_frag) + PSH_MQ_META_OFF((a.prep.B + 3) & ~3))[tid - 30];
        if (tid == 0) s_sc = 0.0f;
    }
    __syncthreads();                                       // (block-scope visibility of qstate[b] for thread 0 below)
    // a level given by the caller (psh_profile.tau_hint): no minima to select from -- tau = tau2 = hint[b]; the f16 scale then
    // comes from the query and the level alone, the way the fused launch derives it (nothing is known about the data)
    const bool hinted = a.tau_hint != nullptr;
    if (a.blockmax || hinted) {
        unsigned mb = 0u;                                  // non-negative floats order as their bit patterns
        if (a.blockmax)
            for (int i = tid; i < a.n_blockmax; i += PSH_SELECT_THREADS) mb = max(mb, __float_as_uint(a.blockmax[i]));
        // batched matrix-core scan: ONE scale for all queries (they share the f16 copy of the data)
        const int64_t xlo = a.mq_frag ? 0 : (int64_t)b * a.prep.W;
        const int64_t xhi = a.mq_frag ? (int64_t)a.prep.B * a.prep.W : xlo + a.prep.W;
        for (int64_t j = xlo + tid; j < xhi; j += PSH_SELECT_THREADS)
            mb = max(mb, __float_as_uint(fabsf(a.prep.queries[j])));
        if (mb) atomicMax(&s_maxbits, mb);
        __syncthreads();
    }
    if (!hinted && n < a.k) return;                        // tau stays +inf (host avoids this)
    const bool in_lds = a.keys_in_lds != 0 && !hinted;
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
    uint64_t prefix = 0;
    int sh = 32, rem;
    bool exact;
    // tau only has to bound the k-th smallest minimum from above: once the digits examined pin it to
    // 2^15 ulps (0.4 %) the bucket's upper edge serves -- usually one pass instead of three
    if (!hinted)
        radix_select64([&](int i) { return (uint64_t)(in_lds ? tkeys[i] : __float_as_uint(v[i])) << 32; },
                       [](int) { return true; }, n, a.k, 32, &sm, &prefix, &sh, &exact, &rem, in_lds, kmin64, kmax64, 32 + 15, a.rank2);
    if (tid == 0) {
        // every sampled value whose bits >> (sh-32) are <= the prefix's is among the k
        // smallest: the largest float with that truncated prefix bounds them all
        const unsigned hi_bits = hinted ? 0u : ((unsigned)(prefix >> 32) | ((sh > 32) ? ((1u << (sh - 32)) - 1u) : 0u));
        if (hi_bits < PSH_INF_BITS) {
            // (a hint that is not a positive finite number leaves tau at +inf: everything is admitted, the slices overflow,
            //  the selection says PSH_STATUS_OVERFLOW -- the documented answer to a useless hint)
            const float tau0 = hinted ? a.tau_hint[b] : __uint_as_float(hi_bits) * PSH_TAU_MARGIN;   // strictly above the k-th value
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
                if ((a.blockmax || hinted) && s_maxbits < PSH_INF_BITS) {
                    // matrix-core filter (scan_mx_kernel): scale = 2^s puts the largest sampled
                    // |value| into [4, 8) -- f16 keeps 11 bits down to 2^-14 and y~^2 stays below
                    // 65504 up to |y~| = 255 -- and mx_thr is the bound derived there, evaluated in
                    // double and rounded up (towards "keep")
                    int e = (int)((s_maxbits >> 23) & 255u) - 126;          // max in [2^(e-1), 2^e)
                    int sexp = 3 - e;
                    if (hinted && !a.mq_frag) {
                        // no sampled data behind the scale: the query's largest |x| 2^sexp < 8 AND tau 4^sexp <= 4096 (the fused
                        // launch's conditions, psh_fused.hip phase B) -- a value the f16 conversion turns into +inf then sits in
                        // windows whose acc is above the level anyway.  (Batches: the scans look at every segment's largest
                        // value themselves and keep everything where it leaves the f16 range.)
                        const int et = (int)((__float_as_uint(tau0) >> 23) & 255u) - 126;
                        const int st = (12 - et) >= 0 ? (12 - et) / 2 : -((et - 12 + 1) / 2);
                        sexp = sexp < st ? sexp : st;
                    }
                    const bool sane = sexp <= 60 && sexp >= -60 && s_maxbits >= 0x00800000u   // normal, squares stay in fp32 range
                                      && (!hinted || __float_as_uint(tau0) >= 0x00800000u);
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
                        if (i8prep) s_sc = sc;
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
    if (a.mq_frag && a.mq_i8) {
        // scan_mq8_kernel: the queries as signed bytes on ONE step s_x for the batch (they share the data's 8-bit copy), the
        // rejection level of this query in the product's integer units, and this query's four byte-shifted copies.
        // With x~ = s_x x^ + ex, y~ = s_y y^ + ey (|ey| <= s_y / 2: y^ = round-to-nearest of y~ / s_y, exactly), c = sum x~ y~:
        //     c <= s_x s_y sum x^ y^ + (s_x s_y / 2) ||x^||_1 + ||ex||_2 sqrt(ny),     2 E sqrt(ny) <= beta ny + E^2 / beta,
        // so a window the exact chain admits -- ny - 2 c < Theta := tau (1 + 2^-16) - nx -- satisfies, in units of 2 s_x s_y,
        //     (1 - beta) ny / (2 s_x s_y) - sum x^ y^  <  (Theta + E^2 / beta) / (2 s_x s_y) + ||x^||_1 / 2.
        // E = the largest ||ex||_2 of the batch and beta = E / the largest ||x~||_2 (clamped) are the same for every query (one
        // block of mq_prep_kernel takes the maxima, through integer atomics: no summation order), so the window side -- C_w, the
        // MFMA's C operand -- is one for all queries.  Everything is evaluated in double and rounded towards "keep".
        __syncthreads();
        const float sc = s_sc;                             // thread 0 above; 0 = filter not armed
        // the batch's step and residues come UNSCALED from mq_prep_kernel (one block's work instead of every block's); the
        // scale is a power of two: s_x = sc s0, E^2 = sc^2 E0^2, max ||x~||^2 = sc^2 max ||x||^2, exactly
        const float xmax = __uint_as_float(s_meta[2]);
        const bool armed = sc > 0.0f && s_meta[2] > 0u && s_meta[2] < PSH_INF_BITS && (xmax * sc) < __uint_as_float(PSH_INF_BITS);
        const float inv_s0 = armed ? 127.0f / xmax : 0.0f;
        const double s_x = armed ? (double)sc / (double)inv_s0 : 0.0;
        const int W = a.prep.W;
        auto quant = [&](float xv) -> int { return mq8_quant(xv, inv_s0); };
        if (tid == 0) {
            QueryState* qs = a.qstate + b;
            const unsigned tb = qs->tau_bits;
            const double E2 = (double)__uint_as_float(s_meta[0]) * (double)sc * (double)sc, NX = (double)__uint_as_float(s_meta[1]) * (double)sc * (double)sc;
            if (armed && tb < PSH_INF_BITS && NX > 0.0 && E2 < (double)__uint_as_float(PSH_INF_BITS)) {
                double beta = sqrt(E2 / NX);
                beta = beta < 1.0 / 4096.0 ? 1.0 / 4096.0 : (beta > 0.25 ? 0.25 : beta);
                double nxs = 0.0, l1 = 0.0;
#pragma unroll
                for (int j = 0; j < 25; ++j) { const double xs = (double)(s_xb[j] * sc); nxs += xs * xs; l1 += fabs((double)quant(s_xb[j])); }
                const double taus = (double)__uint_as_float(tb) * (double)sc * (double)sc;
                const double Theta = taus * (1.0 + 1.0 / 65536.0) - nxs * (1.0 - 1e-12);
                double P = (Theta + E2 * (1.0 + 1e-6) / beta) / (2.0 * s_x);
                P = P > 0.0 ? P * (1.0 + 1.0 / 1048576.0) : P * (1.0 - 1.0 / 1048576.0);    // (the kernel's fma rounds: 2^-24 of the result)
                float Pf = (float)P;
                if ((double)Pf < P) Pf = __uint_as_float(Pf >= 0.0f ? __float_as_uint(Pf) + 1u : __float_as_uint(Pf) - 1u);
                // + 3: the truncating conversion to an integer, the strict comparison, the f16 squares' subnormal tail
                const float Lf = (float)(0.5 * l1 + 3.5) * (1.0f + 1.0f / 65536.0f);
                // the window side from below: the energies come from f16 squares (2^-11 each: 2^-8 covers them with room)
                float k1 = (float)((1.0 - beta) * (1.0 - 1.0 / 256.0) / (2.0 * s_x) * (1.0 - 1.0 / 1048576.0));
                if (Pf == Pf && fabsf(Pf) < __uint_as_float(PSH_INF_BITS) && k1 > 0.0f && k1 < __uint_as_float(PSH_INF_BITS)) {
                    qs->mx8_P = Pf;
                    qs->mx8_L = Lf;
                    qs->mx8_k1 = k1;
                }
            }
        }
        // four byte-shifted copies of xpad = 7 zeros, -x^ (W <= 25 samples), zeros: copy c at dword PSH_MQ8_CDW c holds
        // xpad[i + c] in byte i (PSH_MQ8_QDW = 40 dwords per query, as the f16 table)
        if (tid < 160) {
            const int c = tid / 40, i = tid - 40 * c;
            const int j = i + c - 7;
            const bool in = armed && j >= 0 && j < W;
            reinterpret_cast<signed char*>(a.mq_frag)[(int64_t)b * 160 + tid] = (signed char)(in ? -quant(s_xb[j]) : 0);
        }
    } else if (a.mq_frag) {
        // this query's two zero-padded f16 copies for scan_mq_kernel (PSH_MQ_QDW = 40 dwords: copy c at dword 20 c):
        // xpad[i] = -2 x~[i - 7] inside the query, 0 outside; dword d of copy c = (xpad[2 d + c], xpad[2 d + c + 1])
        __syncthreads();
        const float sc = a.qstate[b].mx_scale;             // thread 0 above; 0 = filter not armed
        if (tid < 80) {
            const int half = tid & 1, dw = tid >> 1, c = dw >= 20 ? 1 : 0, d = dw - 20 * c;
            const int j = 2 * d + c + half - 7;
            const bool in = j >= 0 && j < a.prep.W;
            const float xv = in ? a.prep.queries[(int64_t)b * a.prep.W + j] : 0.0f;
            reinterpret_cast<_Float16*>(a.mq_frag)[(int64_t)b * 80 + tid] = (_Float16)(in ? -2.0f * (xv * sc) : 0.0f);
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

// the distance word of item `idx` of a run (little endian: the high dword of the 64-bit key) -- the searches of the merge
// sort by ranking are bound by LDS throughput (16 waves x ~100 random probes per level), so they read 4 bytes a probe, not 8
__device__ __forceinline__ unsigned run_dist(const uint64_t* run, int idx) {
    return reinterpret_cast<const unsigned*>(run)[2 * idx + 1];
}
__device__ __forceinline__ unsigned run_low(const uint64_t* run, int idx) {
    return reinterpret_cast<const unsigned*>(run)[2 * idx];
}
// first position of run[0, len) whose key is not below `mine` in the order of item_less, given `lo` from the lower bound
// on the distance word: padding (d = 0xffffffff, kpad - k entries, distinct low words in ascending order) takes a second
// lower bound on the low word -- stepping over thousands of "equal" padding keys one by one was 1 ms per level at
// k = 10000 -- anything else steps over the (almost always empty) range of equal distances with the full comparison
__device__ __forceinline__ int finish_rank(const uint64_t* run, int len, int lo, uint64_t mine, const int2* sel_rt) {
    const unsigned myd = (unsigned)(mine >> 32);
    if (lo < len && run_dist(run, lo) < myd) lo += 1;
    if (myd == 0xffffffffu) {
        int n = len - lo;                                  // the padding tail of the run
        const unsigned mylow = (unsigned)mine;
        while (n > 0) {
            const int half = n >> 1;
            if (run_low(run, lo + half) < mylow) { lo += half + 1; n -= half + 1; } else n = half;
        }
        return lo;
    }
    while (lo < len && run_dist(run, lo) == myd && item_less(run[lo], mine, sel_rt)) lo += 1;
    return lo;
}

// ----------------------------------------------------------------------------------
// The selection of a query or two as a RANKING on all CUs (the fused launch's phase D as a kernel of its own): a scan that
// admits below an estimate leaves 2-5 k candidates in the blocks' slices, and the one-block radix select + sort above takes
// 30-45 us over them with 255 CUs idle.  Here PSH_RANK_GRID blocks per query each load ALL candidates (8 per thread, in
// registers), take 1/PSH_RANK_GRID of them as their own and count, for each of their own, the candidates below it in
// (d, r, t): rank < k -> out[rank].  No sort, no single block.  Handles: slices (front lists only), no overflow,
// k <= n <= PSH_RANK_CAP -- anything else leaves handled[b] = 0 and select_kernel, launched behind it, does the work; when
// handled[b] = 1 that launch returns at once.
struct RankShared {
    int offs[PSH_MAX_BLOCKS + 1];
    uint64_t own_key[PSH_RANK_OWN];
    int2 own_rt[PSH_RANK_OWN];
    int rankc[PSH_RANK_OWN];
    int wtot[PSH_SELECT_THREADS / 64];
    int overflow;
};
__global__ __launch_bounds__(PSH_SELECT_THREADS) void rank_select_kernel(SelectArgs a) {
    __shared__ RankShared sm;
    const int b = (int)blockIdx.y, g = (int)blockIdx.x, G = (int)gridDim.x;
    const int tid = (int)threadIdx.x, lane = tid & 63;
    const float* cd = a.cand_d + (int64_t)b * a.cand_stride;
    const int2* crt = a.cand_rt + (int64_t)b * a.cand_stride;
    const int* bc = a.bcount + (int64_t)b * PSH_MAX_BLOCKS;
    const int* bc2 = a.bcount2 ? a.bcount2 + (int64_t)b * PSH_MAX_BLOCKS : nullptr;
    constexpr int OWN = PSH_RANK_OWN;
    int dbg_i = 0;
    auto mark = [&]() { if (a.dbg_times && g == 1 && b == 0 && tid == 0) a.dbg_times[dbg_i] = wall_clock64(); ++dbg_i; };
    mark();
    if (tid == 0) { sm.overflow = 0; sm.offs[0] = 0; }
    if (tid < OWN) sm.rankc[tid] = 0;
    __syncthreads();
    // offs[] = exclusive prefix of the slices' sizes: two slices per thread (PSH_MAX_BLOCKS = 2 x 1024), a DPP scan per wave,
    // the 16 wave totals through LDS
    {
        int c2[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int i = 2 * tid + e;
            c2[e] = 0;
            if (i < a.nblk) {
                const int cf = bc[i], cb = bc2 ? bc2[i] : 0;
                if (cf + cb > a.slice) sm.overflow = 1;
                c2[e] = cf < a.slice ? cf : a.slice;
            }
        }
        int v = c2[0] + c2[1];
        v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);         // inclusive scan over the wave
        v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);
        v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);
        v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);
        v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, true);
        v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, true);
        if (lane == 63) sm.wtot[tid >> 6] = v;
        __syncthreads();
        int before = 0;
        for (int w2 = 0; w2 < (tid >> 6); ++w2) before += sm.wtot[w2];
        const int excl = before + v - (c2[0] + c2[1]);
        if (2 * tid < a.nblk) sm.offs[2 * tid + 1] = excl + c2[0];
        if (2 * tid + 1 < a.nblk) sm.offs[2 * tid + 2] = excl + c2[0] + c2[1];
        __syncthreads();
    }
    const int n = sm.offs[a.nblk];
    mark();
    const bool ok = !sm.overflow && n >= a.k && n <= PSH_RANK_NMAX && n / G + 1 <= OWN - 1;    // the same verdict in every block
    if (g == 0 && tid == 0) a.handled[b] = ok ? 1 : 0;
    if (!ok) return;
    if (g == 0 && tid == 0) {
        if (a.total) a.total[b] = n;
        if (a.qstate) a.qstate[b].n_valid = a.k;
    }
    // candidates as 64-bit keys d << 32 | r << tbits | t (the launcher checked that (r, t) fits 32 bits).  The block's own share
    // goes to LDS first; then ALL candidates pass through the registers PSH_RANK_CAP at a time (8 per thread) and are counted
    // against the own ones (one pass: PSH_RANK_NMAX == PSH_RANK_CAP; more passes were measured and lose to the radix select beyond that).
    constexpr int NE = PSH_RANK_CAP / PSH_SELECT_THREADS;
    const int e_lo = (int)(((int64_t)n * g) / G), e_hi = (int)(((int64_t)n * (g + 1)) / G);
    const int nown = e_hi - e_lo;                                             // <= n / G + 1
    int s0 = 1;
    while (2 * s0 < a.nblk) s0 <<= 1;
    if (tid < nown) {
        const int e = e_lo + tid;
        int l0 = 0;
        for (int step = s0; step >= 1; step >>= 1) { const int m = l0 + step; if (m < a.nblk && sm.offs[m] <= e) l0 = m; }
        const int64_t o = (int64_t)l0 * a.slice + (e - sm.offs[l0]);
        const int2 rt = crt[o];
        sm.own_key[tid] = ((uint64_t)__float_as_uint(cd[o]) << 32) | (uint64_t)(((unsigned)rt.x << a.rank_tbits) | (unsigned)rt.y);
        sm.own_rt[tid] = rt;
    }
    __syncthreads();
    mark();
#pragma unroll 1
    for (int p0 = 0; p0 < n; p0 += PSH_RANK_CAP) {
        const int np = (n - p0) < PSH_RANK_CAP ? (n - p0) : PSH_RANK_CAP;
        const int ns = (np + PSH_SELECT_THREADS - 1) / PSH_SELECT_THREADS;
        // the owning slice of every candidate: the 8 searches of a thread step together (one after the other they were
        // 64 dependent LDS round trips: 4.5 us); then ALL loads in flight together
        int lo[NE], ec[NE];
#pragma unroll
        for (int i = 0; i < NE; ++i) { const int e = p0 + tid + PSH_SELECT_THREADS * i; ec[i] = e < n ? e : n - 1; lo[i] = 0; }
        for (int step = s0; step >= 1; step >>= 1) {
#pragma unroll
            for (int i = 0; i < NE; ++i) {
                const int m = lo[i] + step;
                if (i < ns && m < a.nblk && sm.offs[m] <= ec[i]) lo[i] = m;
            }
        }
        float ld[NE];
        int2 lrt[NE];
#pragma unroll
        for (int i = 0; i < NE; ++i)
            if (i < ns) { const int64_t o = (int64_t)lo[i] * a.slice + (ec[i] - sm.offs[lo[i]]); ld[i] = cd[o]; lrt[i] = crt[o]; }
        uint64_t key[NE];
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            key[i] = ~0ull;
            const int e = p0 + tid + PSH_SELECT_THREADS * i;
            if (i < ns && e < n)
                key[i] = ((uint64_t)__float_as_uint(ld[i]) << 32) | (uint64_t)(((unsigned)lrt[i].x << a.rank_tbits) | (unsigned)lrt[i].y);
        }
        // counting on the vector ALUs, 8 own candidates a turn: a 64-bit compare and an add per pair (a ballot + popcount per
        // pair went through the scalar unit of four waves); two counters share a register for the wave reduction (DPP)
        for (int j0 = 0; j0 < nown; j0 += 8) {
            uint64_t ok8[8];
            int c[8];
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) { ok8[jj] = (j0 + jj < nown) ? sm.own_key[j0 + jj] : 0ull; c[jj] = 0; }
#pragma unroll
            for (int i = 0; i < NE; ++i) {
                if (i < ns) {
#pragma unroll
                    for (int jj = 0; jj < 8; ++jj) c[jj] += key[i] < ok8[jj] ? 1 : 0;
                }
            }
#pragma unroll
            for (int jj = 0; jj < 8; jj += 2) {
                int v = c[jj] | (c[jj + 1] << 16);                // <= 8 per lane and counter: 512 per wave
                v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);     // row_shr 1, 2, 4, 8; row_bcast 15, 31
                v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);
                v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);
                v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);
                v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, true);
                v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, true);
                v = __builtin_amdgcn_readlane(v, 63);
                if (lane == 0) {
                    if (j0 + jj < nown && (v & 0xffff)) atomicAdd(&sm.rankc[j0 + jj], v & 0xffff);
                    if (j0 + jj + 1 < nown && (v >> 16)) atomicAdd(&sm.rankc[j0 + jj + 1], v >> 16);
                }
            }
        }
    }
    __syncthreads();
    mark();
    if (tid < nown) {
        const int rk = sm.rankc[tid];
        if (rk < a.k) {
            const int2 rt = sm.own_rt[tid];
            a.out_d[(int64_t)b * a.k + rk] = __uint_as_float((unsigned)(sm.own_key[tid] >> 32));
            a.out_idx[((int64_t)b * a.k + rk) * 2 + 0] = rt.x;
            a.out_idx[((int64_t)b * a.k + rk) * 2 + 1] = rt.y;
        }
    }
}

__global__ __launch_bounds__(PSH_SELECT_THREADS, 8) void select_kernel(SelectArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint64_t items[];   // kpad entries, then key_cap u32 keys
    __shared__ SelectShared sm;
    unsigned* keys = reinterpret_cast<unsigned*>(items + a.kpad);

    const int b = (int)blockIdx.x;
    const int tid = (int)threadIdx.x;
    if (a.handled && a.handled[b]) return;                // rank_select_kernel has written this query's results
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
    } else if (a.rank_sort) {
        // the ordering is chunk_sort_kernel / chunk_merge_kernel.s (many blocks instead of this one): hand over the items
        uint64_t* G = a.sort_scratch + (int64_t)b * a.cand_stride;
        for (int e = tid; e < a.kpad; e += PSH_SELECT_THREADS) {
            uint64_t it = items[e];
            // padding: distance words of its own above every real one (non-negative floats up to +inf), in slot order
            if ((unsigned)(it >> 32) == 0xffffffffu) it = ((uint64_t)(0x7f800001u + (unsigned)e) << 32) | (unsigned)e;
            G[e] = it;
        }
        if (tid == 0) {
            G[a.kpad] = (uint64_t)(unsigned)(sm.nsel < need ? sm.nsel : need);
            if (a.qstate) a.qstate[b].n_valid = sm.nsel < need ? sm.nsel : need;
        }
        return;
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
                        if (i < per && run_dist(sib[i], lo[i] + step - 1) < (unsigned)(mine[i] >> 32)) lo[i] += step;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    if (i < per) {
                        const int e = tid + i * PSH_SELECT_THREADS;
                        lo[i] = finish_rank(sib[i], len, lo[i], mine[i], sel_rt);
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
                            if (run_dist(sib[i], lo[i] + step - 1) < (unsigned)(mine[i] >> 32)) lo[i] += step;
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const int e = tid + (i0 + i) * PSH_SELECT_THREADS;
                        lo[i] = finish_rank(sib[i], len, lo[i], mine[i], sel_rt);
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
// ordering of a large selection (kpad >= 4096: the tutorial's k = 8192, the reference test's k = 10000) on MANY compute
// units: select_kernel -- one block per query -- leaves its kpad selected items unordered in global scratch (padding
// with distance words of its own, above every real one), and two small launches order them by (d, r, t):
//   chunk_sort_kernel   a block per 1024 items: every wave orders its 64 by counting, every item then finds its place
//                       among the block's 16 runs with 15 binary searches (6 LDS probes each)
//   chunk_merge_kernel  a block per 1024 items again: all distance words in LDS, every item adds to its place in its own
//                       chunk the number of items below it in each other chunk (10 probes each) and writes the result
// ~270 LDS probes per item instead of the in-block merge sort's log2(kpad / 64) levels through one block's LDS and
// global scratch (240 us at kpad = 16384; counting all pairs on all CUs was tried too: 4 VALU operations per pair,
// 90 us).  Equal distance values are met by the full comparison (finish_rank) wherever a search lands on one.
// ----------------------------------------------------------------------------------
#define PSH_CHUNK 1024
// first position of the ascending run whose distance word is not below myd (len a power of two)
__device__ __forceinline__ int lower_bound_dw(const unsigned* run, int len, unsigned myd) {
    int lo = 0;
    for (int step = len >> 1; step > 0; step >>= 1)
        if (run[lo + step - 1] < myd) lo += step;
    if (run[lo] < myd) lo += 1;
    return lo;
}

__global__ __launch_bounds__(PSH_CHUNK) void chunk_sort_kernel(SelectArgs a) {
    __shared__ uint64_t runs[PSH_CHUNK];
    const int b = (int)blockIdx.y, tid = (int)threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (a.handled && a.handled[b]) return;                // rank_select_kernel has written this query's results, in order
    uint64_t* G = a.sort_scratch + (int64_t)b * a.cand_stride + (size_t)blockIdx.x * PSH_CHUNK;
    const int2* sel_rt = a.sel_rt + (int64_t)b * a.kpad;
    const uint64_t mine = G[tid];
    const unsigned myd = (unsigned)(mine >> 32);
    // position inside the wave's run: the lanes holding a smaller item
    int wrank = 0;
#pragma unroll 16
    for (int j = 0; j < 64; ++j) {
        const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)mine, j);
        const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)myd, j);
        wrank += (hi < myd) ? 1 : 0;
        if (__any(hi == myd && j != lane))                                   // an equal distance value: (r, t) decides
            wrank += (hi == myd && j != lane && item_less(((uint64_t)hi << 32) | lo, mine, sel_rt)) ? 1 : 0;
    }
    runs[(tid & ~63) + wrank] = mine;
    __syncthreads();
    // place among the 16 runs: all searches advance together (6 rounds of independent LDS probes)
    int pos[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) pos[c] = 0;
#pragma unroll
    for (int step = 32; step > 0; step >>= 1)
#pragma unroll
        for (int c = 0; c < 16; ++c)
            if (run_dist(runs + 64 * c, pos[c] + step - 1) < myd) pos[c] += step;
    int rank = wrank;
#pragma unroll
    for (int c = 0; c < 16; ++c)
        if (c != w) rank += finish_rank(runs + 64 * c, 64, pos[c], mine, sel_rt);
    G[rank] = mine;
}

__global__ __launch_bounds__(PSH_CHUNK) void chunk_merge_kernel(SelectArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned dw[];      // the distance words of the query's kpad items, chunk by chunk
    const int b = (int)blockIdx.y, tid = (int)threadIdx.x, me = (int)blockIdx.x;
    if (a.handled && a.handled[b]) return;
    const uint64_t* G = a.sort_scratch + (int64_t)b * a.cand_stride;
    const int2* sel_rt = a.sel_rt + (int64_t)b * a.kpad;
    {
        const u32x4* src = reinterpret_cast<const u32x4*>(G);          // two items per 16 bytes
        for (int i = tid; i < a.kpad / 2; i += PSH_CHUNK) {
            const u32x4 v = src[i];
            dw[2 * i] = v[1];
            dw[2 * i + 1] = v[3];
        }
    }
    __syncthreads();
    const uint64_t mine = G[(size_t)me * PSH_CHUNK + tid];
    const unsigned myd = (unsigned)(mine >> 32), mylo = (unsigned)mine;
    const int nch = a.kpad / PSH_CHUNK;
    int rank = tid;
    for (int c0 = 0; c0 < nch; c0 += 8) {                    // eight searches advance together
        int lo[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) lo[i] = 0;
        for (int step = PSH_CHUNK >> 1; step > 0; step >>= 1)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int c = c0 + i < nch ? c0 + i : me;
                if (dw[c * PSH_CHUNK + lo[i] + step - 1] < myd) lo[i] += step;
            }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c = c0 + i;
            if (c >= nch || c == me) continue;
            int l = lo[i];
            if (dw[c * PSH_CHUNK + l] < myd) l += 1;
            // equal distance values (rare): the full comparison decides
            while (l < PSH_CHUNK && dw[c * PSH_CHUNK + l] == myd && item_less(G[(size_t)c * PSH_CHUNK + l], mine, sel_rt)) l += 1;
            rank += l;
        }
    }
    const int nsel = (int)(unsigned)G[a.kpad];
    if (rank < a.k) {
        float d = __uint_as_float(PSH_INF_BITS);
        int2 rt = make_int2(-1, -1);
        if (rank < nsel && myd <= 0x7f800000u) { d = __uint_as_float(myd); rt = sel_rt[mylo]; }
        a.out_d[(int64_t)b * a.k + rank] = d;
        a.out_idx[((int64_t)b * a.k + rank) * 2 + 0] = rt.x;
        a.out_idx[((int64_t)b * a.k + rank) * 2 + 1] = rt.y;
    }
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
    // many queries (a block each): 8192 staged keys -- twice what a query of a large batch brings -- leave room for TWO
    // blocks per compute unit (8 + 32 + 25 KB each): 512 queries are one round of blocks instead of two.  A query with
    // more candidates than that selects from global memory (in_lds = false), as it does beyond the LDS anyway.
    if (B > 256 && a.kpad <= 1024 && key_cap > 8192) key_cap = 8192;
    a.key_cap = (int)key_cap;
    // kpad > 1024: the ordering stage wants a second kpad-item buffer behind the items (merge sort by ranking)
    int64_t area = key_cap;
    // large selections are ordered by chunk_sort_kernel / chunk_merge_kernel when the caller.s scratch can take the items
    a.rank_sort = (!a.unsorted_ok && a.kpad >= 4096 && a.kpad <= 32768 && a.sort_scratch != nullptr && (int64_t)a.cand_stride > (int64_t)a.kpad) ? 1 : 0;
    a.sort_buf_ok = (!a.rank_sort && a.kpad > PSH_SELECT_THREADS && a.kpad <= 8 * PSH_SELECT_THREADS && 2 * items_bytes <= lds_budget) ? 1 : 0;
    // beyond that (kpad = 16384): the second buffer is the query's own candidate slots, if the caller says they are free by then
    if (!a.rank_sort && (a.sort_buf_ok || a.kpad <= 8 * PSH_SELECT_THREADS || (int64_t)a.cand_stride < (int64_t)a.kpad)) a.sort_scratch = nullptr;
    if (a.sort_buf_ok && area < 2 * (int64_t)a.kpad) area = 2 * (int64_t)a.kpad;
    const size_t shmem = items_bytes + (size_t)area * sizeof(unsigned);
    if (shmem > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)select_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        if (e != hipSuccess) return e;
    }
    // up to 8 queries with few candidates expected: the ranking on all CUs first (rank_select_kernel), 256 blocks for one or two
    // queries, 256 / B for more (a block's own share stays within PSH_RANK_OWN down to 16 blocks)
    if (a.handled) {
        // (k <= PSH_RANK_CAP / 2: beyond that the candidates outnumber one pass -- measured with up to five passes: k = 8192 146 us
        //  per call against 153, k = 16384 235 against 186, six queries with k = 8192 0.42 ms against 0.30: the counting is O(n^2 / blocks))
        if (a.bcount && B <= PSH_RANK_MAX_B && a.k <= PSH_RANK_CAP / 2 && !a.skip_negative_rows && a.rank_tbits >= 0) {
            int G = PSH_RANK_GRID;
            if (B > 2) { G = PSH_RANK_GRID / B; if (G < 16) G = 16; }
            hipLaunchKernelGGL(rank_select_kernel, dim3(G, B), dim3(PSH_SELECT_THREADS), 0, s, a);
        } else
            a.handled = nullptr;
    }
    hipLaunchKernelGGL(select_kernel, dim3(B), dim3(PSH_SELECT_THREADS), shmem, s, a);
    if (a.rank_sort) {
        const size_t ms_shmem = (size_t)a.kpad * sizeof(unsigned);
        if (ms_shmem > 48 * 1024) {
            hipError_t e = hipFuncSetAttribute((const void*)chunk_merge_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ms_shmem);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(chunk_sort_kernel, dim3(a.kpad / PSH_CHUNK, B), dim3(PSH_CHUNK), 0, s, a);
        hipLaunchKernelGGL(chunk_merge_kernel, dim3(a.kpad / PSH_CHUNK, B), dim3(PSH_CHUNK), ms_shmem, s, a);
    }
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
hipError_t launch_gather(const GatherArgs& a, hipStream_t s) {
    const int64_t total = a.n * a.C * a.len;
    if (total <= 0) return hipSuccess;
    hipLaunchKernelGGL(gather_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a);
    return hipGetLastError();
}


}  // namespace psh
