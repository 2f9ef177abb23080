// psh_stream.hip -- the single-query step as THREE launches sized to share the chip with another stream's step (gfx950).
// Identity + RelativeMSE (reference path_shadowing.py:149-173, path_distance.py:62-65), the same arithmetic and the same
// results as scan_fused_kernel / scan_mx_kernel: f16 rejection test on the matrix cores, exact fp32 chain for the
// survivors, ranking by (d, r, t).  Part of libpsh_hip.so; selected by PSH_FLAG_OVERLAP.
//
// Why.  The fused launch (psh_fused.hip) needs every one of its blocks resident, owns every CU for ~100 us, and HBM
// idles for ~25 us of them (sample, two grid barriers, threshold, ranking, launch ramp).  Those idle stretches cannot be
// shortened much further inside one launch -- but they need not be idle: a caller that has INDEPENDENT queries (a server,
// a sharded run that overlaps its exchange) issues them on two or three streams, and then the latency-bound parts of one
// step can run while another step streams the ensemble.  That only works if the launches can be co-resident:
//   P  stream_sample_kernel   one-wave blocks, <= 64 VGPRs, one 4.5 KB tile of LDS: the bootstrap sample (one minimum per
//                             sampled unit) and, in the block that arrives LAST (one device-scope ticket), the admission
//                             level tau2, the f16 scale and the rejection threshold (scan_fused_kernel's phase B);
//   S  stream_scan_kernel     16-wave blocks (112 VGPRs, ~146 KB of LDS): scan_fused_kernel's phase C and nothing else -- no
//                             grid barrier, no polling, no residency requirement.  A block appends its handful of
//                             candidates to ONE compact list behind one device-scope atomicAdd at its end;
//   R  stream_rank_kernel     one-wave blocks, <= 64 VGPRs: every block ranks its share of the ~2k candidates against all of
//                             them by counting (rank_select_kernel's loop) and writes out[rank].
// Four S waves per SIMD leave 64 VGPRs, a wave slot and 14 KB of LDS per CU: exactly what a P or R wave needs, so P(i+1)
// and R(i-1) run BESIDE S(i) of another stream, and the blocks of S(i+1) start on a CU the moment S(i)'s block leaves it
// (no barrier inside S: an XCD that finishes early is refilled early).  Kernel boundaries are the only synchronisation;
// the status protocol is the fused launch's (anything unusual -> PSH_STATUS_RETRY -> the caller's separate launches).
//
// Two or three queries (PSH_STREAM_MAX_Q) ride the same three launches: a sampled unit and a scanned unit are loaded and
// converted ONCE, the window energies are one banded product, and every query adds four MFMAs with its own shifted-query
// fragments and a threshold test, so a batch of 2 or 3 costs little more than one query, where the batched scan
// (scan_mq_kernel: 8 waves per CU, sized for hundreds of queries) streams at half rate.  One f16 scale serves all queries of
// the step: the largest that every query's proof allows (each query bounds it from above by its own max|x| and tau2).
#include "psh_device.h"

namespace psh {

typedef unsigned long long u64;
typedef __attribute__((address_space(1))) u64 gu64;
typedef __attribute__((address_space(1))) unsigned gu32;
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
#define PSH_AUX_SC1 16

__device__ __forceinline__ void st_sc1(unsigned* p, unsigned v) {
    __hip_atomic_store((gu32*)(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t st_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}

// ------------------------------------------------------------------------------------------------------------------
// P: the sample and the admission levels
// ------------------------------------------------------------------------------------------------------------------
#define PSH_STREAM_HIST 1024

// scan_fused_kernel's phase B in two steps (same formulas, same roundings towards "keep"): the largest scale exponent a
// query allows -- scale = 2^sexp with max|x| 2^sexp < 8 and tau0 4^sexp <= 4096 -- then, under the step's common scale,
// the query's rejection threshold.
__device__ inline bool stream_sexp_of(unsigned qmaxbits, float tau0, int* sexp_out);
__device__ inline bool stream_sexp(const_f32p xq, int W, float tau0, int* sexp_out) {
    unsigned qmaxbits = 0u;
#pragma unroll 1
    for (int j = 0; j < W; ++j) qmaxbits = max(qmaxbits, __float_as_uint(fabsf(xq[j])));
    return stream_sexp_of(qmaxbits, tau0, sexp_out);
}
// (qmaxbits: the bit pattern of the query's largest |x|)
__device__ inline bool stream_sexp_of(unsigned qmaxbits, float tau0, int* sexp_out) {
    if (!(tau0 > 0.0f && tau0 < __uint_as_float(PSH_INF_BITS) && qmaxbits < PSH_INF_BITS)) return false;
    // (exponents of the bit patterns: value in [2^(e-1), 2^e))
    const int et = (int)((__float_as_uint(tau0) >> 23) & 255u) - 126;
    int sexp = (12 - et) >= 0 ? (12 - et) / 2 : -((et - 12 + 1) / 2);
    if (qmaxbits >= 0x00800000u) {
        const int eq = (int)((qmaxbits >> 23) & 255u) - 126;
        sexp = sexp < 3 - eq ? sexp : 3 - eq;
    }
    if (!(sexp <= 60 && sexp >= -60 && __float_as_uint(tau0) >= 0x00800000u)) return false;
    *sexp_out = sexp;
    return true;
}
__device__ inline bool stream_threshold_of(double nxs, int W, float tau0, float sc, float* thr_out);
__device__ inline bool stream_threshold(const_f32p xq, int W, float tau0, float sc, float* thr_out) {
    double nxs = 0.0;
#pragma unroll 1
    for (int j = 0; j < W; ++j) { const double vv = (double)xq[j] * (double)sc; nxs += vv * vv; }
    return stream_threshold_of(nxs, W, tau0, sc, thr_out);
}
// (nxs: the sum of (x_j sc)^2 in double)
__device__ inline bool stream_threshold_of(double nxs, int W, float tau0, float sc, float* thr_out) {
    // (b: the absolute part of the bound -- f16 subnormals, one unit of 2^-24 per product -- grows with the taps: 2^-18 covers the
    //  2 W + 1 = 41 .. 67 terms of W <= 33; a long window takes (2 W + 2) / 64 of it)
    // a: the relative part.  W <= 33: 2^-9 covers the f16 roundings of x~, y~ AND (y~^2)^ plus the fp32 accumulation of both banded
    // products.  A long window's energies are fp32 prefix sums whose error the tile's C operand takes off separately (round 6:
    // ce <= E by construction), so only the correlation is left:  |c^ - c| <= (2^-10 (1 + 2^-12) + 2 x 288 x 2^-24)(nx~ + E) / 2 x 2,
    // and with E <= 2 (nx~ + acc~):  t^ <= acc~ (1 + 2 a') - nx~ (1 - 3 a'),  a' = 1.012e-3 -> 1 / 900 with a tenth to spare.
    const double am = W > 33 ? 1.0 / 900.0 : 1.0 / 512.0, bm = (1.0 / 262144.0) * (W > 31 ? (double)(2 * W + 2) / 64.0 : 1.0);
    const double taus = (double)tau0 * (double)sc * (double)sc;
    const double T = taus * (1.0 + 1.0 / 131072.0) * (1.0 + 2.0 * am) - nxs * (1.0 - 3.0 * am) * (1.0 - 1e-12) + bm;
    float Tf = (float)T;
    if ((double)Tf < T) Tf = __uint_as_float(Tf >= 0.0f ? __float_as_uint(Tf) + 1u : __float_as_uint(Tf) - 1u);
    if (!(Tf == Tf && fabsf(Tf) < __uint_as_float(PSH_INF_BITS))) return false;
    *thr_out = Tf;
    return true;
}
// K-steps of 16 the banded product of a window of W samples takes: the band of column n (the query shifted down by n, n < 32)
// ends at tap W + 30 -- four steps up to W = 33 (the kernels with the band in registers), ceil((W + 31) / 16) beyond
__host__ __device__ inline int stream_ksteps(int W) { return W <= 33 ? 4 : (W + 31 + 15) / 16; }

// What follows the sample in the block that arrives LAST (one device-scope ticket): per query the admission level tau2 from the
// rank-th smallest minimum (or the caller's hint), then ONE f16 scale for the step, every query's rejection threshold and the
// scan's B fragments of the shifted query (scan_fused_kernel's phase B).  Shared by stream_sample_kernel (exact minima) and
// stream_sample_long_kernel (matrix-core upper bounds of the minima, long windows): `tile` is the wave's scratch (>= 1024 words).
// `xs` (nullable; a long window's launch passes nq x W floats of LDS that are dead by now): the last block stages the queries
// there and runs the per-query work -- max|x|, nx~, the scan's fragment table -- a lane a sample instead of a serial loop over the
// window by lane 0 with every sample fetched from memory: 10 us of a 25 us launch at W = 126, twice that at 252.
template <int NWP>
__device__ __forceinline__ void stream_sample_finish(const ScanArgs& a, const FusedArgs& f, float* tile, const int W, const int nq,
                                                     const unsigned nbu, const bool hinted, const int lane, const int wave,
                                                     float* xs = nullptr) {
    __shared__ int sh_last, sh_sexp[4], sh_armed[4];
    __shared__ float sh_tau0[4];
    FusedHdr* hdr = f.hdr;
    StreamCtl* ctl = &hdr->stream;
    // arrive: the minima have left this CU (drain), then ONE device-scope ticket per block; the last arriver goes on
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (NWP > 1) __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned tk = __hip_atomic_fetch_add((gu32*)&ctl->ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sh_last = tk == gridDim.x - 1u ? 1 : 0;
    }
    if (NWP > 1) __syncthreads(); else wave_lds_fence();
    if (!sh_last) return;
    if (xs) {
        for (int e = (int)threadIdx.x; e < nq * W; e += 64 * NWP) xs[e] = a.queries[e];
        if (NWP > 1) __syncthreads(); else wave_lds_fence();
    }

    // ---- the last block, per query: rank-th smallest minimum -> tau0 (its bucket's upper edge) and the scale it allows
    unsigned* hist = reinterpret_cast<unsigned*>(tile);
    const int n4 = ((int)nbu + 3) >> 2;                      // 16-byte groups of a query's minima
#pragma unroll 1
    for (int q = (NWP == 1 ? 0 : wave); q < nq; q += NWP) {
        bool armed = true;
        const __amdgpu_buffer_rsrc_t rmin = st_rsrc(hdr->minima + (size_t)q * f.units_stride, (unsigned)(4 * n4) * 4u);
        wave_lds_fence();
        for (int i = lane; i < PSH_STREAM_HIST; i += 64) hist[i] = 0u;
        unsigned kmin = 0xffffffffu, kmax = 0u;
        int nfin = 0;
        for (int g0 = 0; g0 < n4; g0 += 64 * 4) {
            u32x4v mv[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int g = g0 + lane + 64 * c;
                mv[c] = __builtin_amdgcn_raw_buffer_load_b128(rmin, (g < n4 ? g : 0) * 16, 0, PSH_AUX_SC1);
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int g = g0 + lane + 64 * c;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (g < n4 && 4 * g + e < (int)nbu && mv[c][e] < PSH_INF_BITS) {
                        kmin = mv[c][e] < kmin ? mv[c][e] : kmin; kmax = mv[c][e] > kmax ? mv[c][e] : kmax; ++nfin;
                    }
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const unsigned l2 = __shfl_xor(kmin, off, 64), h2 = __shfl_xor(kmax, off, 64);
            kmin = l2 < kmin ? l2 : kmin;
            kmax = h2 > kmax ? h2 : kmax;
            nfin += __shfl_xor(nfin, off, 64);
        }
        unsigned edge = 0u;
        if (hinted) {
            // (a hint that is not a positive finite number fails stream_sexp below: not armed -> PSH_STATUS_RETRY)
        } else if (nfin >= f.rank) {
            const unsigned range = kmax - kmin;
            const int hb = range ? 32 - __builtin_clz(range) : 0;
            const int shift = hb > 10 ? hb - 10 : 0;                      // (range >> shift) < 1024
            wave_lds_fence();
            for (int g0 = 0; g0 < n4; g0 += 64 * 4) {
                u32x4v mv[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int g = g0 + lane + 64 * c;
                    mv[c] = __builtin_amdgcn_raw_buffer_load_b128(rmin, (g < n4 ? g : 0) * 16, 0, PSH_AUX_SC1);
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int g = g0 + lane + 64 * c;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (g < n4 && 4 * g + e < (int)nbu && mv[c][e] < PSH_INF_BITS) atomicAdd(&hist[(mv[c][e] - kmin) >> shift], 1u);
                }
            }
            wave_lds_fence();
            constexpr int PER = PSH_STREAM_HIST / 64;
            unsigned h[PER];
            unsigned sl = 0;
#pragma unroll
            for (int c = 0; c < PER; ++c) { h[c] = hist[PER * lane + c]; sl += h[c]; }
            unsigned inc = sl;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const unsigned t2 = __shfl_up(inc, off, 64);
                if (lane >= off) inc += t2;
            }
            unsigned cum = inc - sl;
            const unsigned rk = (unsigned)f.rank;
            unsigned my_edge = 0u;
            const bool mine = cum < rk && inc >= rk;                     // exactly one lane
            if (mine) {
                int bucket = PER * lane;
#pragma unroll
                for (int c = 0; c < PER; ++c) {
                    if (cum < rk && cum + h[c] >= rk) bucket = PER * lane + c;
                    cum += h[c];
                }
                // every minimum in buckets <= `bucket` is at or below the bucket's upper edge, and there are >= rank of them
                u64 e2 = (u64)kmin + (((u64)bucket + 1ull) << shift) - 1ull;
                if (e2 > (u64)kmax) e2 = kmax;
                my_edge = (unsigned)e2;
            }
            const u64 who = __ballot(mine);
            edge = (unsigned)__builtin_amdgcn_readlane((int)my_edge, who ? (int)__builtin_ctzll(who) : 0);
        } else {
            armed = false;
        }
        const float tau0 = hinted ? f.tau_hint[q] : __uint_as_float(edge) * PSH_TAU_MARGIN;
        int sx = 0;
        if (armed && xs) {
            // (stream_sexp with the query's largest |x| taken a lane a sample: a maximum does not depend on the order)
            unsigned mb = 0u;
            for (int j = lane; j < W; j += 64) mb = max(mb, __float_as_uint(fabsf(xs[(size_t)q * W + j])));
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) mb = max(mb, (unsigned)__shfl_xor((int)mb, off, 64));
            if (!stream_sexp_of(mb, tau0, &sx)) armed = false;
        } else if (armed && !stream_sexp((const_f32p)a.queries + (size_t)q * W, W, tau0, &sx)) armed = false;   // (uniform: scalar inputs)
        if (lane == 0) { sh_tau0[q] = tau0; sh_sexp[q] = sx; sh_armed[q] = armed ? 1 : 0; }
    }
    if (NWP > 1) __syncthreads(); else wave_lds_fence();
    // ---- ONE scale for the step (every query's conditions bound it from above), then every query's threshold and fragments
    bool armed = true;
    int sexp_common = 1000;
    for (int q = 0; q < nq; ++q) { armed = armed && sh_armed[q] != 0; sexp_common = sh_sexp[q] < sexp_common ? sh_sexp[q] : sexp_common; }
    const float scale = armed ? __uint_as_float((unsigned)(127 + sexp_common) << 23) : 0.0f;
#pragma unroll 1
    for (int q = (NWP == 1 ? 0 : wave); q < nq; q += NWP) {
        const const_f32p xq = (const_f32p)a.queries + (size_t)q * W;
        const float tau0q_ = sh_tau0[q];
        float thr = 0.0f;
        if (armed && xs) {
            // (stream_threshold with nx~ summed a lane a sample, in double: whatever the order, its rounding is 10^-16 of the sum
            //  where the formula keeps 10^-12 in hand)
            double part = 0.0;
            for (int j = lane; j < W; j += 64) { const double vv = (double)xs[(size_t)q * W + j] * (double)scale; part += vv * vv; }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
            if (!stream_threshold_of(part, W, tau0q_, scale, &thr)) sh_armed[q] = 0;
        } else if (armed && !stream_threshold(xq, W, tau0q_, scale, &thr)) sh_armed[q] = 0;
        if (lane == 0) {
            // (||x||: the reference's order -- sumsq8 --, from the staged copy when there is one)
            const float s2 = xs ? sumsq8([&](int j) { return xs[(size_t)q * W + j]; }, W) : sumsq8([&](int j) { return xq[j]; }, W);
            ctl->tau2_bits[q] = __float_as_uint(tau0q_);
            ctl->thr2_bits[q] = __float_as_uint(thr);
            ctl->xn_bits[q] = __float_as_uint(f.qnorm_in ? f.qnorm_in[q] : __builtin_sqrtf(s2));
            ctl->ncand[q] = 0u;
        }
        // the scan's B fragments of the shifted query (column n of K-step s holds -2 x~[16 s + 8 hk + i - n]): prepared ONCE
        // here, a 4 KB table every scan block fetches with four 16-byte loads per lane instead of 32 scattered ones
        const int n = lane & 31, hk = lane >> 5;
        const int nks = stream_ksteps(W);                    // (more than four: one query with a long window, stream_scan_long_kernel)
#pragma unroll 1
        for (int s = 0; s < nks; ++s) {                      // (once per launch: compact code, not speed)
            f16x8 b;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int j = 16 * s + 8 * hk + i - n;
                const bool in = j >= 0 && j < W;
                const float xv = xs ? xs[(size_t)q * W + (in ? j : 0)] : a.queries[(size_t)q * W + (in ? j : 0)];
                b[i] = (_Float16)(in ? -2.0f * (xv * scale) : 0.0f);
            }
            *reinterpret_cast<f16x8*>(hdr->bxtab + ((size_t)q * nks * 64 + (size_t)(s * 64 + lane)) * 8) = b;    // (nks = 4 up to W = 33)
        }
    }
    if (NWP > 1) __syncthreads(); else wave_lds_fence();
    if (threadIdx.x == 0) {
        for (int q = 0; q < nq; ++q) armed = armed && sh_armed[q] != 0;   // (a threshold that came out non-finite)
        ctl->scale_bits = __float_as_uint(scale);
        ctl->armed = armed ? 1u : 0u;
        ctl->ovf = 0u;
        ctl->ticket = 0u;                                                 // the next launch on this workspace counts from zero
    }
}

// NWP waves per block: 1 when the launch has to fit beside another step's scan blocks (one query, PSH_FLAG_OVERLAP); 4 for the
// two- and three-query step, whose last block then works on its queries side by side (a wave per query: the tail of the
// launch -- selection, threshold, fragment table -- is 12 us per query when one wave does them in turn)
template <int WT, bool ALIGNED, int NWP>
__global__ __launch_bounds__(64 * NWP) __attribute__((amdgpu_waves_per_eu(8, 8)))
void stream_sample_kernel(ScanArgs a, FusedArgs f) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // This wave shares its SIMD with four scan waves of another step, all OLDER than it: the arbiter serves the oldest
    // first, and at the default priority a sample / ranking launch took 70 us beside a scan (10-35 us alone) -- long enough
    // to gate the next scan of its own stream.  Its work is a few per cent of a scan's: it goes first.
    __builtin_amdgcn_s_setprio(3);
    const int lane = lane_id();
    const int wave = NWP == 1 ? 0 : __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    float* tile = smem + (size_t)wave * a.tile_floats;       // a tile per wave; the last block's histograms live there too
    FusedHdr* hdr = f.hdr;
    StreamCtl* ctl = &hdr->stream;
    const int W = WT > 0 ? WT : a.W;
    const int nfloat = PSH_SEG + W - 1;
    const int nq = f.nq;
    // levels given by the caller (psh_profile.tau_hint; the launch is then ONE block): nothing is sampled, the last-block
    // part below takes tau0 = hint[q]
    const bool hinted = f.tau_hint != nullptr;
    const unsigned nbu = hinted ? 0u : (unsigned)f.boot_units;

    auto boot_load = [&](Stage& sx, unsigned uu) {
        const unsigned ri = fast_div(uu, a.magic_nseg, (unsigned)a.nseg);
        const unsigned sg = uu - ri * (unsigned)a.nseg;
        stage_load<ALIGNED>(sx, a.dataset + (f.boot_row0 + (int64_t)ri * f.boot_row_stride) * a.T, a.T, (int)sg * PSH_SEG, nfloat, lane);
    };
    unsigned u = blockIdx.x * NWP + (unsigned)wave;
    for (int p = lane; p < a.tile_floats; p += 64) tile[p] = 0.0f;           // no slot is ever read uninitialised
    // a header psh_workspace_init never saw: no ticket can be trusted -- block 0 says so, the scan and the ranking return
    const bool armed_hdr = hdr->magic == PSH_FUSED_MAGIC;
    if (!armed_hdr) {
        if (blockIdx.x == 0 && threadIdx.x == 0) { ctl->armed = 0u; ctl->ovf = 0u; for (int q = 0; q < 4; ++q) ctl->ncand[q] = 0u; }
        return;
    }
    while (u < nbu) {
        const unsigned ri = fast_div(u, a.magic_nseg, (unsigned)a.nseg);
        const unsigned sg = u - ri * (unsigned)a.nseg;
        const int seg_start = (int)sg * PSH_SEG;
        {   // (no register prefetch of the next unit: 20 VGPRs this wave does not have; the other sample waves of the chip
            //  cover the latency)
            Stage st;
            boot_load(st, u);
            stage_store<false>(st, tile, nfloat, lane);
        }
        wave_lds_fence();
        const unsigned un = u + gridDim.x * NWP;
        const int t_lane = seg_start + PSH_L * lane;
        int nvalid = a.Tp - t_lane;
        nvalid = nvalid < 0 ? 0 : (nvalid > PSH_L ? PSH_L : nvalid);
        // the exact chains of the lane's 16 windows, taps from the scalar cache (few registers: this wave lives in the 64
        // VGPRs four scan waves leave on a SIMD; the sample is ~3 % of a scan's arithmetic); the unit serves every query
#pragma unroll 1
        for (int q = 0; q < nq; ++q) {
            float acc[PSH_L];
            accumulate16<WT, false>(tile, lane, (const_f32p)a.queries + (size_t)q * W, W, acc);
            float m = __uint_as_float(PSH_INF_BITS);
#pragma unroll
            for (int i = 0; i < PSH_L; ++i) m = (i < nvalid) ? fminf(m, acc[i]) : m;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) m = fminf(m, __shfl_xor(m, off, 64));
            if (!(m == m)) m = __uint_as_float(PSH_INF_BITS);              // NaN data: the segment carries no information
            if (lane == 0) st_sc1(&hdr->minima[(size_t)q * f.units_stride + u], __float_as_uint(m));   // write-through
        }
        wave_lds_fence();
        u = un;
    }
    stream_sample_finish<NWP>(a, f, tile, W, nq, nbu, hinted, lane, wave);
}

// ------------------------------------------------------------------------------------------------------------------
// S: the scan
// ------------------------------------------------------------------------------------------------------------------
#define PSH_STREAM_FIXED_BYTES 256    // control words; the block's candidate list follows
// A block's list of admitted windows is full (clustered matches: a smooth ensemble -- price levels, not returns -- puts a
// window's neighbours in t next to it in distance too): the entry goes straight to the query's compact list in memory, one
// device-scope atomic per entry (r05; until then such a step gave up -- PSH_STATUS_RETRY -- and the caller ran the separate launches)
__device__ __forceinline__ void spill_candidate(FusedHdr* hdr, void* cand_list, int cand_cap, int q, float xn, float acc, int r_global, int t) {
    const unsigned slot = __hip_atomic_fetch_add((gu32*)&hdr->stream.ncand[q], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (slot < (unsigned)cand_cap)
        reinterpret_cast<u32x4*>(cand_list)[(size_t)q * cand_cap + slot] = u32x4{__float_as_uint(dist_from_acc(acc, xn)), (unsigned)r_global, (unsigned)t, (unsigned)q};
}
#define PSH_STREAM_FL(NQ) ((NQ) == 1 ? PSH_FUSED_FRONT : 2 * PSH_FUSED_FRONT)
enum { S_FRONT = 0, S_NEXT = 1 };

template <int WT, bool ALIGNED, int NQ>
__device__ __forceinline__ void stream_scan_body(const ScanArgs& a, const FusedArgs& f) {
    static_assert(WT >= 0 && WT <= 33, "the shifted-query band must fit K = 64 (WT = 0: run-time W <= 33)");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NW = PSH_SCAN_THREADS / 64;
    constexpr int NFL = PSH_STREAM_FL(NQ);
    const int lane = lane_id();
    const int tid = (int)threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    int* ctl = reinterpret_cast<int*>(smem);                                 // 64 control words
    u32x4* fl = reinterpret_cast<u32x4*>(ctl + 64);                           // NFL entries {acc bits, r, t, query}
    float* tiles = reinterpret_cast<float*>(fl + NFL);
    float* tile = tiles + (size_t)wave * a.tile_floats;
    _Float16* ah0 = reinterpret_cast<_Float16*>(tiles + (size_t)NW * a.tile_floats);
    _Float16* a1 = ah0 + (size_t)wave * 2 * PSH_MX_NHALF;                     // y^
    _Float16* a2 = a1 + PSH_MX_NHALF;                                         // (y~^2)^
    _Float16* bxl = ah0 + (size_t)NW * 2 * PSH_MX_NHALF;                      // NQ > 1: the queries' fragment tables, 4 KB each
    FusedHdr* hdr = f.hdr;
    const StreamCtl* sc = &hdr->stream;
    auto stamp = [&](int i) { if (a.dbg_times && tid == 0) a.dbg_times[(size_t)blockIdx.x * 8 + i] = (unsigned long long)wall_clock64(); };
    stamp(0);

    const int W = WT > 0 ? WT : a.W;
    const int nfloat = PSH_SEG + W - 1;
    const unsigned n_rs = (unsigned)a.n_rows * (unsigned)a.nseg;
    const unsigned u_lo = (unsigned)(((u64)n_rs * blockIdx.x) / gridDim.x);
    const unsigned u_hi = (unsigned)(((u64)n_rs * (blockIdx.x + 1u)) / gridDim.x);

    auto decode = [&](unsigned uu, unsigned& ri, unsigned& sg) {
        ri = fast_div(uu, a.magic_nseg, (unsigned)a.nseg);
        sg = uu - ri * (unsigned)a.nseg;
    };
    auto load_unit = [&](Stage& sx, unsigned uu) {
        unsigned ri, sg;
        decode(uu, ri, sg);
        stage_load<ALIGNED>(sx, a.dataset + (a.row0 + (int64_t)ri * a.row_stride) * a.T, a.T, (int)sg * PSH_SEG, nfloat, lane);
    };
    // the first unit of every wave is requested before anything else (static: unit u_lo + wave; the queue starts behind them)
    Stage st;
    unsigned u = u_lo + (unsigned)wave;
    if (u < u_hi) load_unit(st, u);
    // what the sample kernel left (an earlier launch on this stream: plain loads): the shifted-query fragments -- one query's in
    // registers; two or three queries' block-shared in LDS, read per use (a second set in registers spills 15 of them inside
    // the loop at the 128-register cap of four waves per SIMD: 142 us for two queries against 107 from LDS; the LDS pipe is
    // this kernel's co-limit -- 27 KB of traffic per unit and wave are 80 % of what it delivers in a unit's time -- so every
    // query read from LDS costs ~12 %) ...
    f16x8 bx[4], bo[4];
    if constexpr (NQ == 1) {
#pragma unroll
        for (int s = 0; s < 4; ++s) bx[s] = *reinterpret_cast<const f16x8*>(hdr->bxtab + (size_t)(s * 64 + lane) * 8);
    } else {
        for (int i = tid; i < NQ * 4 * 64; i += PSH_SCAN_THREADS)
            *reinterpret_cast<f16x8*>(bxl + (size_t)i * 8) = *reinterpret_cast<const f16x8*>(hdr->bxtab + (size_t)i * 8);
    }
    // ... and the admission levels (scalar loads; first needed inside the loop, so the set-up below runs under their latency)
    const unsigned armed_w = sc->armed;
    const float scale = __uint_as_float(sc->scale_bits);
    float tau2[NQ], thr2[NQ], xn[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        tau2[q] = __uint_as_float(sc->tau2_bits[q]);
        thr2[q] = __uint_as_float(sc->thr2_bits[q]);
        xn[q] = __uint_as_float(sc->xn_bits[q]);
    }
    if (tid == 0) { ctl[S_FRONT] = 0; ctl[S_NEXT] = NW; }
    {   // every slot of the f16 arrays a segment does not write must be finite (0 * NaN poisons a row)
        unsigned* z = reinterpret_cast<unsigned*>(a1);
        for (int i = lane; i < PSH_MX_NHALF; i += 64) z[i] = 0u;              // 2 arrays x NHALF halves = NHALF dwords
    }
    {   // the band of ones (window energies): arithmetic only
        const int n = lane & 31, hk = lane >> 5;
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int j = 16 * s + 8 * hk + i - n;
                bo[s][i] = (_Float16)((j >= 0 && j < W) ? 1.0f : 0.0f);
            }
    }
    __syncthreads();
    stamp(1);
    if (armed_w == 0u) return;                                                // uniform: the ranking reports PSH_STATUS_RETRY
    auto grab = [&]() -> unsigned {
        int v = 0;
        if (lane == 0) v = atomicAdd(&ctl[S_NEXT], 1);
        return u_lo + (unsigned)__builtin_amdgcn_readfirstlane(v);
    };
    // the survivors of one query's accumulator tile: exact chain from the fp32 tile, admitted below the query's level
    auto admit = [&](const f32x16& acc, int q, float thr, float tau, int seg_start, int r_global) {
        const int m = lane & 31, hk = lane >> 5;
        const const_f32p x = (const_f32p)a.queries + (size_t)q * W;
        unsigned hm = 0u;
#pragma unroll
        for (int r = 0; r < 16; ++r) hm |= !(acc[r] > thr) ? (1u << r) : 0u;
#pragma unroll 1
        for (int r = 0; r < 16; ++r) {
            const int p = 32 * ((r & 3) + 8 * (r >> 2) + 4 * hk) + m;         // C layout: row -> window
            bool hit = (((hm >> r) & 1u) != 0u) && (seg_start + p < a.Tp);
            if (!__ballot(hit)) continue;
            float v = 0.0f;
            if (hit) { if constexpr (WT > 0) v = exact_one<(WT > 0 ? WT : 20)>(tile, p, x); else v = exact_one_rt(tile, p, x, W); }
            hit = hit && (v < tau);
            const unsigned long long mask = __ballot(hit);
            if (!mask) continue;
            int base = 0;
            if (lane == 0) base = atomicAdd(&ctl[S_FRONT], __popcll(mask));
            base = __builtin_amdgcn_readfirstlane(base);
            if (hit) {
                const int slot = base + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
                if (slot < NFL) fl[slot] = u32x4{__float_as_uint(v), (unsigned)r_global, (unsigned)(seg_start + p), (unsigned)q};
                else spill_candidate(hdr, f.cand_list, f.cand_cap, q, __uint_as_float(sc->xn_bits[q]), v, r_global, seg_start + p);
            }
        }
    };
    while (u < u_hi) {
        unsigned ri, sg;
        decode(u, ri, sg);
        const int64_t row = a.row0 + (int64_t)ri * a.row_stride;
        const int seg_start = (int)sg * PSH_SEG;
        const int r_global = (int)(row + a.r_offset);

        stage_store(st, tile, nfloat, lane);
        if (a.dbg_times && tid == 0 && u == u_lo) a.dbg_times[(size_t)blockIdx.x * 8 + 4] = (unsigned long long)wall_clock64();   // first data in the tile
        {
            const int nq4 = (nfloat + 3) >> 2;
#pragma unroll
            for (int q = 0; q < PSH_NSTAGE; ++q) {
                const int m = lane + 64 * q;
                if (q < PSH_NSTAGE - 1 || m < nq4) {
                    const f32x4 v = st.v[q] * scale;
                    const f32x4 v2 = v * v;
                    *reinterpret_cast<f16x4*>(a1 + mx_half(4 * m)) = __builtin_convertvector(v, f16x4);
                    *reinterpret_cast<f16x4*>(a2 + mx_half(4 * m)) = __builtin_convertvector(v2, f16x4);
                }
            }
        }
        wave_lds_fence();
        const unsigned un = grab();
        if (un < u_hi) load_unit(st, un);

        const int m = lane & 31, hk = lane >> 5;
        f16x8 fa[4];
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
#pragma unroll
        for (int s = 0; s < 4; ++s) fa[s] = *reinterpret_cast<const f16x8*>(a2 + mx_half(32 * m + 16 * s + 8 * hk));
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[s], bo[s], acc, 0, 0, 0);
#pragma unroll
        for (int s = 0; s < 4; ++s) fa[s] = *reinterpret_cast<const f16x8*>(a1 + mx_half(32 * m + 16 * s + 8 * hk));
        if constexpr (NQ == 1) {
#pragma unroll
            for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[s], bx[s], acc, 0, 0, 0);
            bool keep = false;                             // NaN-safe: !(t^ > thr)
#pragma unroll
            for (int r = 0; r < 16; ++r) keep = keep || !(acc[r] > thr2[0]);
            if (__any(keep)) admit(acc, 0, thr2[0], tau2[0], seg_start, r_global);
        } else {
            // the window energies are in `acc`; every query adds its own banded product on top of them (the energies are the
            // C operand of its first MFMA: no copy)
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const _Float16* bp = bxl + ((size_t)q * 4 * 64 + (size_t)lane) * 8;
                f32x16 aq = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[0], *reinterpret_cast<const f16x8*>(bp), acc, 0, 0, 0);
#pragma unroll
                for (int s = 1; s < 4; ++s) aq = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[s], *reinterpret_cast<const f16x8*>(bp + (size_t)s * 64 * 8), aq, 0, 0, 0);
                bool keep = false;
#pragma unroll
                for (int r = 0; r < 16; ++r) keep = keep || !(aq[r] > thr2[q]);
                if (__any(keep)) admit(aq, q, thr2[q], tau2[q], seg_start, r_global);
            }
        }
        wave_lds_fence();  // all lanes done with the tile before it is overwritten
        u = un;
    }
    stamp(2);
    __syncthreads();
    // the block's candidates (distances in place of acc) go to ONE compact list per query behind ONE device-scope atomicAdd per
    // block and query (0.3 us of a block's 75; a list per block made the ranking search for every candidate's slot: 60 us
    // instead of 9)
    if (wave == 0) {
        const int nfront = ctl[S_FRONT];
        const int mown = nfront < NFL ? nfront : NFL;
#pragma unroll
        for (int c0 = 0; c0 < NFL; c0 += 64) {
            if (c0 >= mown) break;
            const bool have = c0 + lane < mown;
            u32x4 e = have ? fl[c0 + lane] : u32x4{0u, 0u, 0u, 0xffffffffu};
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const bool mine = have && (int)e[3] == q;
                const unsigned long long mask = __ballot(mine);
                if (!mask) continue;
                unsigned base = 0u;
                if (lane == 0) base = __hip_atomic_fetch_add((gu32*)&hdr->stream.ncand[q], (unsigned)__popcll(mask), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
                if (mine) {
                    const unsigned slot = base + __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
                    if (slot < (unsigned)f.cand_cap) {
                        u32x4 o = e;
                        o[0] = __float_as_uint(dist_from_acc(__uint_as_float(e[0]), xn[q]));
                        reinterpret_cast<u32x4*>(f.cand_list)[(size_t)q * f.cand_cap + slot] = o;
                    }
                }
            }
        }
    }
    stamp(3);
}

// ------------------------------------------------------------------------------------------------------------------
// S for ONE query with a LONG window (34 <= W <= 256; the reference takes any Identity(dimension): path_embedding.py:135-139,
// the tutorial's context is 126 samples).  The same rejection test -- t^ = sum y~^2 - 2 sum x~ y~ on f16 copies, same bound,
// same exact recheck -- with the band split over ceil((W + 31) / 16) K-steps that accumulate into the one 32 x 32 tile of a row
// group: 2 MFMAs per K-step (energies: the band of ones; correlation: the shifted query) instead of 8 per segment.  What changes
// around it: the shifted-query fragments and the boundary fragments of the band of ones are block-shared tables in LDS (18
// steps do not fit registers; the INTERIOR steps of the ones band are one constant register set), and there is no fp32 tile --
// the f16 arrays grow with W and the tables take its place -- so a segment that holds survivors fetches its fp32 values again (one
// coalesced round trip: the segment was streamed a microsecond ago, L2 / MALL) into the f16 arrays' LDS for the exact chains.  Candidates, status protocol and the launches around it (sample + levels,
// ranking) are stream_scan_kernel's.
// The long-window scan's f16 arrays (round 5, second layout): ROWS of 32 samples, a row of y^ followed by the same row of
// (y~^2)^ and 8 halves of padding -- 72 halves = 144 bytes = 36 dwords a row: the 16 lanes of a ds_read_b128 group (rows m
// with all 16 residues mod 16, 4 dwords each) fall on 36 m mod 64 = 16 different multiples of 4.  The fragment of row m,
// K-step s lies at row m + (s >> 1), halves 16 (s & 1) + 8 hk .. + 7 of it: per PAIR of K-steps one pointer moves by one row
// and every other offset is an immediate -- the slot-rotation layout of the short kernels (mx_half) cost the K-loop 7 vector
// instructions per step for the address alone, and on this part vector instructions ADD to the matrix cores' time (A.5).
// inc += (inc of the lane CTRL names, 0 where there is none): one step of a wave scan, a v_add_f32 with a DPP operand
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_addf(float inc) {
    return inc + __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(inc), CTRL, ROW_MASK, 0xf, true));
}
#define PSH_LONG_ROW 40
#define PSH_LONG_SFLOATS 1280         // fp32 prefix sums of a segment's squares: entries 0 .. SEG + W - 1 <= 1279; the survivors' scratch afterwards
#define PSH_LONG_QCAP 64              // deferred survivors a wave keeps before it verifies them (8-byte entries)
#define PSH_LONG_GAMMA 7.62939453125e-06f   // 2^-17: what the energies' prefix sums may be off by, per unit of the prefix S[p + W] (see below)
__host__ __device__ inline int stream_long_rows(int W) { return 31 + (stream_ksteps(W) + 1) / 2; }                 // rows the band of row 31 reaches
__host__ __device__ inline int stream_long_nhalf(int W) { return stream_long_rows(W) * PSH_LONG_ROW; }             // halves per wave (both arrays)
__device__ __forceinline__ int long_half(int idx) { return (idx >> 5) * PSH_LONG_ROW + (idx & 31); }                // logical sample -> its y^ half ((y~^2)^: + 32)

// The band's K-steps of one query: the fragment of K-step s lies at pa0 + (s >> 1) rows + 16 (s & 1) halves, its table at
// pb + s ts -- every offset an immediate off two registers for all 18 steps a window of 256 takes; one exit.
__device__ __forceinline__ f32x16 long_chain(const _Float16* pa0, const _Float16* pb, const int ts, const f32x16& c0, const int nks) {
    auto ld = [](const _Float16* p) { return *reinterpret_cast<const f16x8*>(p); };
    f32x16 c = c0;
    f16x8 a0 = ld(pa0), b0 = ld(pb), a1, b1;
    // step S multiplies set CUR and requests step S + 1 into set NXT (spelled out: an unrolled loop with a break became a real
    // loop that picked its register set with v_cndmask).  (The compiler sinks a pair of reads down to the step that uses it, so a
    // step is still read - wait - multiply; reads as inline assembly with hand-placed wait counts kept them ahead but cost the
    // allocator 200 spills over the chain's eighteen exits -- four waves per SIMD hide the round trip well enough.)
#define PSH_LONG_STEP(S, CA, CB, NA, NB)                                                        \
    if ((S) >= nks) goto done;                                                                  \
    NA = ld(pa0 + (((S) + 1) >> 1) * PSH_LONG_ROW + 16 * (((S) + 1) & 1));                      \
    NB = ld(pb + ((S) + 1) * ts);                                                               \
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(CA, CB, c, 0, 0, 0);
    PSH_LONG_STEP(0, a0, b0, a1, b1) PSH_LONG_STEP(1, a1, b1, a0, b0) PSH_LONG_STEP(2, a0, b0, a1, b1) PSH_LONG_STEP(3, a1, b1, a0, b0)
    PSH_LONG_STEP(4, a0, b0, a1, b1) PSH_LONG_STEP(5, a1, b1, a0, b0) PSH_LONG_STEP(6, a0, b0, a1, b1) PSH_LONG_STEP(7, a1, b1, a0, b0)
    PSH_LONG_STEP(8, a0, b0, a1, b1) PSH_LONG_STEP(9, a1, b1, a0, b0) PSH_LONG_STEP(10, a0, b0, a1, b1) PSH_LONG_STEP(11, a1, b1, a0, b0)
    PSH_LONG_STEP(12, a0, b0, a1, b1) PSH_LONG_STEP(13, a1, b1, a0, b0) PSH_LONG_STEP(14, a0, b0, a1, b1) PSH_LONG_STEP(15, a1, b1, a0, b0)
    PSH_LONG_STEP(16, a0, b0, a1, b1) PSH_LONG_STEP(17, a1, b1, a0, b0)
#undef PSH_LONG_STEP
    static_assert(PSH_STREAM_LONG_KS == 18, "the chain is spelled out for 18 steps");
done:
    return c;
}

// NQ: one, two or three queries ride one pass (round 5): the window energies' MFMA of a K-step is shared, a query adds one
// MFMA with its own fragment -- 1 + NQ per step where NQ one-query steps issue 2 NQ -- and the segment is staged and converted
// once (three queries with W = 126: one pass where the loop of one-query steps made three).
template <bool ALIGNED, int NQ>
__device__ __forceinline__ void stream_scan_long_body(const ScanArgs& a, const FusedArgs& f) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NW = PSH_SCAN_THREADS / 64;
    constexpr int NFL = PSH_STREAM_FL(NQ);
    const int lane = lane_id();
    const int tid = (int)threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int W = a.W;
    const int nks = stream_ksteps(W), nhalf = stream_long_nhalf(W);
    int* ctl = reinterpret_cast<int*>(smem);                                 // 64 control words
    u32x4* fl = reinterpret_cast<u32x4*>(ctl + 64);                           // NFL entries {acc bits, r, t, query}
    _Float16* bxl = reinterpret_cast<_Float16*>(fl + NFL);                    // [K-step][query][lane][8]: -2 x~_q shifted by the lane's column
    constexpr int TS = NQ * 64 * 8;                                           // halves of a K-step's tables
    float* sp0 = reinterpret_cast<float*>(bxl + (size_t)nks * TS);            // per wave: PSH_LONG_SFLOATS prefix sums, then the queue, then the rows
    float* sp = sp0 + (size_t)wave * (PSH_LONG_SFLOATS + 2 * PSH_LONG_QCAP + nhalf / 2);
    u64* sq = reinterpret_cast<u64*>(sp + PSH_LONG_SFLOATS);                  // deferred survivors: row | t << 32 | query << 62
    _Float16* a1 = reinterpret_cast<_Float16*>(sq + PSH_LONG_QCAP);           // rows of {y^ [32], pad [8]}
    FusedHdr* hdr = f.hdr;
    const StreamCtl* sc = &hdr->stream;

    const int nfloat = PSH_SEG + W - 1;
    const unsigned n_rs = (unsigned)a.n_rows * (unsigned)a.nseg;
    const unsigned u_lo = (unsigned)(((u64)n_rs * blockIdx.x) / gridDim.x);
    const unsigned u_hi = (unsigned)(((u64)n_rs * (blockIdx.x + 1u)) / gridDim.x);
    auto decode = [&](unsigned uu, unsigned& ri, unsigned& sg) {
        ri = fast_div(uu, a.magic_nseg, (unsigned)a.nseg);
        sg = uu - ri * (unsigned)a.nseg;
    };
    // A segment is staged by BUFFER loads (round 6): the row is the resource (base = its first float, records = its bytes), a lane's
    // five 16-byte pieces sit at one constant VGPR offset + immediates, the segment's start is the scalar offset -- no address
    // arithmetic on the vector ALUs, and what lies beyond the row reads as zero (it only feeds inadmissible windows).  Rows need
    // 4-byte alignment only: one code path for aligned and unaligned ensembles.
    const int lane16 = lane * 16;
    auto stage_row = [&](Stage& sx, int64_t row, int seg_start) {
        const int64_t bytes = a.T * 4;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dataset + row * a.T), 0,
                                                                             (int)(bytes > 0x7ffffffc ? 0x7ffffffc : bytes), 0x00020000);
        const int nq4 = (nfloat + 3) >> 2;
#pragma unroll
        for (int q = 0; q < PSH_NSTAGE; ++q)
            if (q < PSH_NSTAGE - 1 || lane + 64 * q < nq4) {
                const u32x4v w = __builtin_amdgcn_raw_buffer_load_b128(rs, lane16 + 1024 * q, seg_start * 4, 2 /* nt */);
                sx.v[q] = f32x4{__uint_as_float(w[0]), __uint_as_float(w[1]), __uint_as_float(w[2]), __uint_as_float(w[3])};
            }
    };
    auto load_unit = [&](Stage& sx, unsigned uu) {
        unsigned ri, sg;
        decode(uu, ri, sg);
        stage_row(sx, a.row0 + (int64_t)ri * a.row_stride, (int)sg * PSH_SEG);
    };
    Stage st;
    unsigned u = u_lo + (unsigned)wave;
    if (u < u_hi) load_unit(st, u);
    // the tables: the sample kernel's fragments of the shifted query (plain loads: an earlier launch on this stream)
    for (int i = tid; i < nks * 64; i += PSH_SCAN_THREADS) {
        const int s = i >> 6, l = i & 63;
#pragma unroll
        for (int q = 0; q < NQ; ++q)
            *reinterpret_cast<f16x8*>(bxl + (size_t)s * TS + ((size_t)q * 64 + l) * 8) = *reinterpret_cast<const f16x8*>(hdr->bxtab + ((size_t)q * nks * 64 + i) * 8);
    }
    const unsigned armed_w = sc->armed;
    const float scale = __uint_as_float(sc->scale_bits);
    float tau2[NQ], thr2[NQ], xn[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        tau2[q] = __uint_as_float(sc->tau2_bits[q]); thr2[q] = __uint_as_float(sc->thr2_bits[q]); xn[q] = __uint_as_float(sc->xn_bits[q]);
    }
    if (tid == 0) { ctl[S_FRONT] = 0; ctl[S_NEXT] = NW; }
    {   // every slot of the rows a segment does not write must be finite (0 * NaN poisons a row)
        unsigned* z = reinterpret_cast<unsigned*>(a1);
        for (int i = lane; i < nhalf / 2; i += 64) z[i] = 0u;
    }
    __syncthreads();
    if (armed_w == 0u) return;                                                // uniform: the ranking reports PSH_STATUS_RETRY
    auto grab = [&]() -> unsigned {
        int v = 0;
        if (lane == 0) v = atomicAdd(&ctl[S_NEXT], 1);
        return u_lo + (unsigned)__builtin_amdgcn_readfirstlane(v);
    };
    // ---- deferred survivors (round 6).  A window that survives the test goes to the wave's queue -- row, t, query: 8 bytes --
    // and the queue is verified when it is full (64 entries) and when the wave has no unit left: about ten survivors in a wave's
    // whole life on ordinary data, so ONE batch at its end instead of a stall in one segment out of four (re-fetching the segment
    // and running the chains in place was 20 of the scan's 125 us at W = 126).
    int qn = 0;                                                               // entries in the queue (uniform)
    typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));     // a window starts at any float
    auto verify_queue = [&]() {
        // a lane per queued window: its W samples straight from memory, 16 bytes a load (64 scattered requests an instruction,
        // all of them L2 / MALL hits: the segment was streamed microseconds ago), the chain in the reference's order
        const bool have = lane < qn;
        const u64 mine = have ? sq[lane] : 0ull;
        const int q = (int)(mine >> 62);
        const unsigned t = (unsigned)(mine >> 32) & 0x3fffffffu;
        const float* y = a.dataset + (int64_t)(unsigned)mine * a.T + t;
        const const_f32p x = (const_f32p)a.queries + (size_t)(NQ == 1 ? 0 : q) * W;
        float v = 0.0f;
        if (have) {
            int j = 0;
#pragma unroll 2
            for (; j + 4 <= W; j += 4) {
                const f32x4u yy = *reinterpret_cast<const f32x4u*>(y + j);
#pragma unroll
                for (int c = 0; c < 4; ++c) { const float D = __fsub_rn(x[j + c], yy[c]); v = __builtin_fmaf(D, D, v); }
            }
            for (; j < W; ++j) { const float D = __fsub_rn(x[j], y[j]); v = __builtin_fmaf(D, D, v); }
        }
        float tq = tau2[0], xq = xn[0];
#pragma unroll
        for (int qq = 1; qq < NQ; ++qq) { tq = q == qq ? tau2[qq] : tq; xq = q == qq ? xn[qq] : xq; }
        const bool hit = have && (v < tq);
        const unsigned long long mask = __ballot(hit);
        if (mask) {
            int base = 0;
            if (lane == 0) base = atomicAdd(&ctl[S_FRONT], __popcll(mask));
            base = __builtin_amdgcn_readfirstlane(base);
            if (hit) {
                const int slot = base + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
                const int r_global = (int)((int64_t)(unsigned)mine + a.r_offset);
                if (slot < NFL) fl[slot] = u32x4{__float_as_uint(v), (unsigned)r_global, t, (unsigned)q};
                else spill_candidate(hdr, f.cand_list, f.cand_cap, q, xq, v, r_global, (int)t);
            }
        }
        wave_lds_fence();                                                     // (the queue is read: it may be written again)
        qn = 0;
    };
    const int m = lane & 31, hk = lane >> 5;
    const _Float16* pa0 = a1 + m * PSH_LONG_ROW + 8 * hk;
    const _Float16* pb0 = bxl + lane * 8;
    const float* ps_lo = sp + m + 128 * hk;                                   // S[p] of the lane's 16 windows: p = m + 128 hk + 32 (r & 3) + 256 (r >> 2)
    const float* ps_hi = ps_lo + W;
    while (u < u_hi) {
        unsigned ri, sg;
        decode(u, ri, sg);
        const int64_t row = a.row0 + (int64_t)ri * a.row_stride;
        const int seg_start = (int)sg * PSH_SEG;
        {
            // the segment's f16 copy y^ (rows of 32 samples), and -- round 6 -- the WINDOW ENERGIES from fp32 prefix sums of the
            // squares instead of a second banded product: S[i] = sum of y~_j^2, j < i, 3 adds inside a lane, six DPP adds across
            // the wave, the carry from one 256-sample group to the next (embed_px_kernel's scan, psh_embed_px.hip).  All terms are
            // >= 0, every S[i] is a sum of them in SOME order with at most 17 roundings on a term's way:
            //     |S^[i] - S[i]| <= 17 u S[i] (1 + ...)   (u = 2^-24)    =>    |(S^[p + W] - S^[p]) - E(p)| <= 35 u S[p + W]
            // which the test below takes off the energy as PSH_LONG_GAMMA S^[p + W] = 128 u S[p + W] (3.6x the bound: room for the
            // C operand's own rounding and its share of the product's accumulation).  The correlation keeps the f16 bound of
            // stream_threshold unchanged -- its error model had the energies' f16 roundings in it, which are gone.  The K-loop
            // is ONE MFMA and TWO fragment reads per step instead of two and four (the energies' MFMAs were what made a long
            // window's K-loop ADD to the streaming time: W = 126 105 -> 87 us, W = 252 155 -> 113 us for the scan without them).
            // A NaN / inf sample makes every LATER prefix of the segment NaN / inf: those windows survive the test (NaN-safe
            // compare) and the exact chains sort them out -- slow on such a segment, never wrong.
            const int nq4 = (nfloat + 3) >> 2;
            // (the five groups' scans step by step side by side: a DPP operand written by the instruction before costs two wait
            //  states, five independent chains fill them)
            float d0[PSH_NSTAGE], d1[PSH_NSTAGE], d2[PSH_NSTAGE], d3[PSH_NSTAGE], inc[PSH_NSTAGE];
#pragma unroll
            for (int q = 0; q < PSH_NSTAGE; ++q) {
                const int mm = lane + 64 * q;
                const bool on = q < PSH_NSTAGE - 1 || mm < nq4;
                const f32x4 v = on ? st.v[q] * scale : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                const f32x4 z = v * v;
                if (on) *reinterpret_cast<f16x4*>(a1 + (mm >> 3) * PSH_LONG_ROW + 4 * (mm & 7)) = __builtin_convertvector(v, f16x4);
                d0[q] = z[0]; d1[q] = d0[q] + z[1]; d2[q] = d1[q] + z[2]; d3[q] = d2[q] + z[3];
                inc[q] = d3[q];
            }
#ifdef PSH_TUNING
            if (!(a.dbg & 64))                                                // ablation: no scan, no sums stored (results invalid)
#endif
            {
#pragma unroll
            for (int q = 0; q < PSH_NSTAGE; ++q) inc[q] = dpp_addf<0x111, 0xf>(inc[q]);
#pragma unroll
            for (int q = 0; q < PSH_NSTAGE; ++q) inc[q] = dpp_addf<0x112, 0xf>(inc[q]);
#pragma unroll
            for (int q = 0; q < PSH_NSTAGE; ++q) inc[q] = dpp_addf<0x114, 0xf>(inc[q]);
#pragma unroll
            for (int q = 0; q < PSH_NSTAGE; ++q) inc[q] = dpp_addf<0x118, 0xf>(inc[q]);
#pragma unroll
            for (int q = 0; q < PSH_NSTAGE; ++q) inc[q] = dpp_addf<0x142, 0xa>(inc[q]);
#pragma unroll
            for (int q = 0; q < PSH_NSTAGE; ++q) inc[q] = dpp_addf<0x143, 0xc>(inc[q]);
            float carry = 0.0f;
#pragma unroll
            for (int q = 0; q < PSH_NSTAGE; ++q) {
                const int mm = lane + 64 * q;
                const float x0 = carry + (inc[q] - d3[q]);                    // exclusive: the lane's own total taken off again
                carry += __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(inc[q]), 63));
                if (q < PSH_NSTAGE - 1 || mm <= nq4) *reinterpret_cast<f32x4*>(sp + 4 * mm) = f32x4{x0, x0 + d0[q], x0 + d1[q], x0 + d2[q]};
            }
            }
        }
        wave_lds_fence();
        const unsigned un = grab();
        if (un < u_hi) load_unit(st, un);
#ifdef PSH_TUNING
        const int nks_run = (a.dbg & 8) ? 1 : nks;                            // ablation: one K-step (results invalid)
#else
        const int nks_run = nks;
#endif
        // the tile's C operand: what the prefix sums say about the 16 windows of this lane, E^(p) - gamma S^[p + W]
        f32x16 ce;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int off = 32 * (r & 3) + 256 * (r >> 2);
            ce[r] = __builtin_fmaf(ps_hi[off], 1.0f - PSH_LONG_GAMMA, -ps_lo[off]);
        }
        if (a.Tp - seg_start < PSH_SEG) {
            // A row's last segment: the windows past the last admissible one never pass the test (+inf in their slots of the C
            // operand).  Left alone they often do: what lies beyond the row reads as zero, a window of zeros is at acc = ||x||^2,
            // and for a long window that is about where the admission level sits -- the survivors' walk below then ran in most
            // last segments for windows that do not exist.
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (seg_start + 32 * ((r & 3) + 8 * (r >> 2) + 4 * hk) + m >= a.Tp) ce[r] = __uint_as_float(PSH_INF_BITS);
        }
#ifdef PSH_TUNING
        if (a.dbg & 32) {                                                     // ablation: no energies read back (results invalid)
#pragma unroll
            for (int r = 0; r < 16; ++r) ce[r] = 0.0f;
        }
#endif
        // The band's K-steps as ONE unrolled chain: the fragment of K-step s lies at pa0 + (s >> 1) rows + 16 (s & 1) halves, its
        // table at pb0 + s TS -- every offset an immediate off TWO registers for all 18 steps a window of 256 takes, a uniform
        // branch per step.  The test that follows is 16 compares into scalar masks OR-ed on the scalar unit; only a segment that
        // holds a survivor builds per-lane masks.
        auto ld = [](const _Float16* p) { return *reinterpret_cast<const f16x8*>(p); };
        unsigned hmq[NQ];                                                     // per query: bit r = accumulator r's window survives the test
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const f32x16 c = long_chain(pa0, pb0 + q * 64 * 8, TS, ce, nks_run);
            unsigned long long any = 0ull;                                    // NaN-safe: !(t^ > thr)
#pragma unroll
            for (int r = 0; r < 16; ++r) any |= __ballot(!(c[r] > thr2[q]));
            unsigned hm = 0u;
            if (any) {
#pragma unroll
                for (int r = 0; r < 16; ++r) hm |= !(c[r] > thr2[q]) ? (1u << r) : 0u;
            }
            hmq[q] = hm;
        }
        unsigned hany = 0u;
#pragma unroll
        for (int q = 0; q < NQ; ++q) hany |= hmq[q];
#ifdef PSH_TUNING
        if (a.dbg & 4) hany = 0u;                                             // ablation: no survivor handling (results invalid)
#endif
        if (__any(hany != 0u)) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const unsigned hm = hmq[q];
                if (!__any(hm != 0u)) continue;
#pragma unroll 1
                for (int r = 0; r < 16; ++r) {
                    const int p = 32 * ((r & 3) + 8 * (r >> 2) + 4 * hk) + m; // C layout: row -> window
                    const bool hit = (((hm >> r) & 1u) != 0u) && (seg_start + p < a.Tp);
                    const unsigned long long mask = __ballot(hit);
                    if (!mask) continue;
                    const int n = (int)__popcll(mask);
                    if (qn + n > PSH_LONG_QCAP) verify_queue();               // (uniform; the prefix sums are dead by now: its scratch)
                    if (hit) {
                        const int slot = qn + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
                        sq[slot] = (u64)(unsigned)row | ((u64)(unsigned)(seg_start + p) << 32) | ((u64)(unsigned)q << 62);
                    }
                    qn += n;
                }
            }
        }
        wave_lds_fence();  // all lanes done with the arrays before they are overwritten
        u = un;
    }
    if (qn > 0) verify_queue();
    __syncthreads();
    if (wave == 0) {                                                          // (stream_scan_body's publication)
        const int nfront = ctl[S_FRONT];
        const int mown = nfront < NFL ? nfront : NFL;
#pragma unroll
        for (int c0 = 0; c0 < NFL; c0 += 64) {
            if (c0 >= mown) break;
            const bool have = c0 + lane < mown;
            u32x4 e = have ? fl[c0 + lane] : u32x4{0u, 0u, 0u, 0xffffffffu};
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const bool mine = have && (int)e[3] == q;
                const unsigned long long mask = __ballot(mine);
                if (!mask) continue;
                unsigned base = 0u;
                if (lane == 0) base = __hip_atomic_fetch_add((gu32*)&hdr->stream.ncand[q], (unsigned)__popcll(mask), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
                if (mine) {
                    const unsigned slot = base + __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
                    if (slot < (unsigned)f.cand_cap) {
                        u32x4 o = e;
                        o[0] = __float_as_uint(dist_from_acc(__uint_as_float(e[0]), xn[q]));
                        reinterpret_cast<u32x4*>(f.cand_list)[(size_t)q * f.cand_cap + slot] = o;
                    }
                }
            }
        }
    }
}
template <bool ALIGNED, int NQ>
__global__ __launch_bounds__(PSH_SCAN_THREADS) void stream_scan_long_kernel(ScanArgs a, FusedArgs f) {
    stream_scan_long_body<ALIGNED, NQ>(a, f);
}

// one query: 112 registers (56 arch + 56 acc of the unified file), so that a sample or ranking wave of another stream's step fits
// beside four of these on a SIMD; two or three queries: the whole file (their steps are rarely run beside others)
template <int WT, bool ALIGNED>
__global__ __launch_bounds__(PSH_SCAN_THREADS) __attribute__((amdgpu_num_vgpr(56))) void stream_scan_kernel(ScanArgs a, FusedArgs f) {
    stream_scan_body<WT, ALIGNED, 1>(a, f);
}
template <int WT, bool ALIGNED, int NQ>
__global__ __launch_bounds__(PSH_SCAN_THREADS) void stream_scan_q_kernel(ScanArgs a, FusedArgs f) {
    stream_scan_body<WT, ALIGNED, NQ>(a, f);
}


// ------------------------------------------------------------------------------------------------------------------
// P for long windows (round 6): the sample on the matrix cores
// ------------------------------------------------------------------------------------------------------------------
// stream_sample_kernel's exact chains cost a sampled unit 1024 x W x 2 vector instructions -- 8 us of a wave at W = 126, 51 us a
// launch beside the scans, and at W = 252 (where no sample block fits in the LDS a long-window scan block leaves) 25 us in front of
// every scan.  Here a sampled unit takes the long-window scan's own road: f16 rows, window energies from fp32 prefix sums, the
// banded product with the shifted query on the matrix cores -- and gives an UPPER bound of its smallest acc per query, exactly
// scan_lq_kernel's BOOT construction (psh_lq.hip: the C operand is E^ + gamma S, the bound (t^ + nx~)(1 + 3 a + 6 gamma) + b, the
// query's fragments eight shifted copies of -2 x~ under a scale that needs the query alone: max|x| 2^s < 8).  The rank-th smallest
// upper bound is a level with at least `rank` windows below it, as the rank-th smallest exact minimum is; it sits half a per cent
// higher and admits ~30 % more candidates at W = 126.  A unit that holds a near-match (its estimate below nx~ / 6: smooth
// ensembles, where the bound would be the correlation's error term alone) is sampled with the exact chains instead.  The tail is
// stream_sample_finish.
// NWP: 1 (one query: 13-15 KB of LDS and <= 128 VGPRs, beside a scan block up to W ~ 200) or 4 (two / three queries).
template <int NKS, int NWP>
__device__ __forceinline__ void stream_sample_long_body(const ScanArgs& a, const FusedArgs& f) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __builtin_amdgcn_s_setprio(3);                                            // (see stream_sample_kernel)
    const int lane = lane_id();
    const int tid = (int)threadIdx.x;
    const int wave = NWP == 1 ? 0 : __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int W = a.W, nq = f.nq;
    constexpr int CP = lq_copy_chunks(NKS), QS = lq_query_bytes(NKS) / 2;     // chunks a copy, HALVES a query
    constexpr int NROWS = lq_rows(NKS);
    int* ctlw = reinterpret_cast<int*>(smem);                                 // [0]: the step's sample exponent
    float* nxq = smem + 4;                                                    // nx~ (1 + 1e-6) per query
    _Float16* tab = reinterpret_cast<_Float16*>(smem + 16);                   // [query][copy c < 8][chunk < CP][8 halves]
    float* sp = reinterpret_cast<float*>(tab + (size_t)nq * QS) + (size_t)wave * (PSH_LONG_SFLOATS + NROWS * PSH_LONG_ROW / 2);
    _Float16* a1 = reinterpret_cast<_Float16*>(sp + PSH_LONG_SFLOATS);        // rows of {y^ [32], pad [8]}
    FusedHdr* hdr = f.hdr;
    StreamCtl* ctl = &hdr->stream;
    const unsigned nbu = (unsigned)f.boot_units;
    const int nfloat = PSH_SEG + W - 1;
    const bool armed_hdr = hdr->magic == PSH_FUSED_MAGIC;                     // (stream_sample_kernel: a header psh_workspace_init never saw)
    if (!armed_hdr) {
        if (blockIdx.x == 0 && threadIdx.x == 0) { ctl->armed = 0u; ctl->ovf = 0u; for (int q = 0; q < 4; ++q) ctl->ncand[q] = 0u; }
        return;
    }
    const int lane16 = lane * 16;
    auto load_unit = [&](Stage& sx, unsigned uu) {
        const unsigned ri = fast_div(uu, a.magic_nseg, (unsigned)a.nseg);
        const unsigned sg = uu - ri * (unsigned)a.nseg;
        const int64_t row = f.boot_row0 + (int64_t)ri * f.boot_row_stride;
        const int64_t bytes = a.T * 4;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dataset + row * a.T), 0,
                                                                             (int)(bytes > 0x7ffffffc ? 0x7ffffffc : bytes), 0x00020000);
        const int nq4 = (nfloat + 3) >> 2;
#pragma unroll
        for (int q = 0; q < PSH_NSTAGE; ++q)
            if (q < PSH_NSTAGE - 1 || lane + 64 * q < nq4) {
                const u32x4v w = __builtin_amdgcn_raw_buffer_load_b128(rs, lane16 + 1024 * q, (int)sg * PSH_SEG * 4, 0);
                sx.v[q] = f32x4{__uint_as_float(w[0]), __uint_as_float(w[1]), __uint_as_float(w[2]), __uint_as_float(w[3])};
            }
    };
    Stage st;
    unsigned u = blockIdx.x * NWP + (unsigned)wave;
    if (u < nbu) load_unit(st, u);
    // ---- the sample's scale (every query: max|x| 2^sexp < 8), nx~, the tables.  The queries are staged in LDS first (the first
    //      wave's prefix-sum scratch: <= 3 x 256 floats): the tables read every sample a dozen times, and from memory a block's
    //      set-up was five dependent round trips -- half the life of a block that samples two units.
    float* xs = reinterpret_cast<float*>(tab + (size_t)nq * QS);             // [query][W]
    for (int e = tid; e < nq * W; e += 64 * NWP) xs[e] = a.queries[e];
    if (tid == 0) ctlw[0] = 60;
    if (NWP > 1) __syncthreads(); else wave_lds_fence();
    for (int q = wave; q < nq; q += NWP) {
        const float* xq = xs + (size_t)q * W;
        unsigned mb = 0u;
        for (int j = lane; j < W; j += 64) mb = max(mb, __float_as_uint(fabsf(xq[j])));
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) mb = max(mb, (unsigned)__shfl_xor((int)mb, off, 64));
        // (a query with a NaN / inf sample does not bound the scale: its tables hold NaN, its upper bounds come out +inf and the
        //  finish finds fewer finite minima than its rank -> not armed -> PSH_STATUS_RETRY)
        if (lane == 0 && mb >= 0x00800000u && mb < PSH_INF_BITS) atomicMin(&ctlw[0], 3 - ((int)((mb >> 23) & 255u) - 126));
    }
    if (NWP > 1) __syncthreads(); else wave_lds_fence();
    const int sexp = ctlw[0] < -60 ? -60 : ctlw[0];
    const float scale = __uint_as_float((unsigned)(127 + sexp) << 23);
    for (int q = wave; q < nq; q += NWP) {
        const float* xq = xs + (size_t)q * W;
        double part = 0.0;
        for (int j = lane; j < W; j += 64) { const double vv = (double)xq[j] * (double)scale; part += vv * vv; }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
        if (lane == 0) nxq[q] = (float)(part * (1.0 + 1e-6));
    }
    for (int e = tid; e < nq * 8 * CP; e += 64 * NWP) {                       // copy c, chunk v, half i: -2 x~[8 (v - 3) + i - c]
        const int v = e % CP, c = (e / CP) & 7, q = e / (8 * CP);
        const float* xq = xs + (size_t)q * W;
        f16x8 b;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int j = 8 * (v - 3) + i - c;
            const bool in = j >= 0 && j < W;
            const float xv = xq[in ? j : 0];
            b[i] = (_Float16)(in ? -2.0f * (xv * scale) : 0.0f);
        }
        *reinterpret_cast<f16x8*>(tab + (size_t)q * QS + ((size_t)c * CP + v) * 8) = b;
    }
    if (NWP > 1) __syncthreads(); else wave_lds_fence();                     // (the staged queries are dead: the first wave's scratch)
    {   // every slot of the rows a segment does not write must be finite (0 * NaN poisons a row)
        unsigned* z = reinterpret_cast<unsigned*>(a1);
        for (int i = lane; i < NROWS * PSH_LONG_ROW / 2; i += 64) z[i] = 0u;
    }
    if (NWP > 1) __syncthreads(); else wave_lds_fence();

    const int m = lane & 31, hk = lane >> 5;
    const _Float16* pa0 = a1 + m * PSH_LONG_ROW + 8 * hk;
    const _Float16* pb0 = tab + ((size_t)(m & 7) * CP + (hk - (m >> 3) + 3)) * 8;   // copy n & 7, chunk hk - (n >> 3) + 3 (+ 2 s a K-step)
    const float* ps_lo = sp + m + 128 * hk;
    const float* ps_hi = ps_lo + W;
    while (u < nbu) {
        const unsigned ri = fast_div(u, a.magic_nseg, (unsigned)a.nseg);
        const int seg_start = (int)(u - ri * (unsigned)a.nseg) * PSH_SEG;
        {   // f16 rows and the fp32 prefix sums of the squares (stream_scan_long_body: the construction and its bound are there)
            const int nq4 = (nfloat + 3) >> 2;
            float d0[PSH_NSTAGE], d1[PSH_NSTAGE], d2[PSH_NSTAGE], d3[PSH_NSTAGE], inc[PSH_NSTAGE];
#pragma unroll
            for (int q = 0; q < PSH_NSTAGE; ++q) {
                const int mm = lane + 64 * q;
                const bool on = q < PSH_NSTAGE - 1 || mm < nq4;
                const f32x4 v = on ? st.v[q] * scale : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                const f32x4 z = v * v;
                if (on) *reinterpret_cast<f16x4*>(a1 + (mm >> 3) * PSH_LONG_ROW + 4 * (mm & 7)) = __builtin_convertvector(v, f16x4);
                d0[q] = z[0]; d1[q] = d0[q] + z[1]; d2[q] = d1[q] + z[2]; d3[q] = d2[q] + z[3];
                inc[q] = d3[q];
            }
#pragma unroll
            for (int q = 0; q < PSH_NSTAGE; ++q) inc[q] = dpp_addf<0x111, 0xf>(inc[q]);
#pragma unroll
            for (int q = 0; q < PSH_NSTAGE; ++q) inc[q] = dpp_addf<0x112, 0xf>(inc[q]);
#pragma unroll
            for (int q = 0; q < PSH_NSTAGE; ++q) inc[q] = dpp_addf<0x114, 0xf>(inc[q]);
#pragma unroll
            for (int q = 0; q < PSH_NSTAGE; ++q) inc[q] = dpp_addf<0x118, 0xf>(inc[q]);
#pragma unroll
            for (int q = 0; q < PSH_NSTAGE; ++q) inc[q] = dpp_addf<0x142, 0xa>(inc[q]);
#pragma unroll
            for (int q = 0; q < PSH_NSTAGE; ++q) inc[q] = dpp_addf<0x143, 0xc>(inc[q]);
            float carry = 0.0f;
#pragma unroll
            for (int q = 0; q < PSH_NSTAGE; ++q) {
                const int mm = lane + 64 * q;
                const float x0 = carry + (inc[q] - d3[q]);
                carry += __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(inc[q]), 63));
                if (q < PSH_NSTAGE - 1 || mm <= nq4) *reinterpret_cast<f32x4*>(sp + 4 * mm) = f32x4{x0, x0 + d0[q], x0 + d1[q], x0 + d2[q]};
            }
        }
        wave_lds_fence();
        const unsigned un = u + gridDim.x * NWP;
        if (un < nbu) load_unit(st, un);
        // the tile's C operand: an UPPER bound of the 16 windows' energies, E^(p) + gamma S^[p + W]
        f32x16 ce;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int off = 32 * (r & 3) + 256 * (r >> 2);
            ce[r] = __builtin_fmaf(ps_hi[off], 1.0f + PSH_LONG_GAMMA, -ps_lo[off]);
        }
        const int nvalid = a.Tp - seg_start;
        auto ld = [](const _Float16* p) { return *reinterpret_cast<const f16x8*>(p); };
        bool loose = false;                                                   // (uniform)
#pragma unroll 1
        for (int q = 0; q < nq; ++q) {
            f32x16 c = ce;
            const _Float16* pb = pb0 + (size_t)q * QS;
#pragma unroll
            for (int s2 = 0; s2 < NKS; ++s2)
                c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ld(pa0 + (s2 >> 1) * PSH_LONG_ROW + 16 * (s2 & 1)), ld(pb + 16 * s2), c, 0, 0, 0);
            float mn = __uint_as_float(PSH_INF_BITS);
            if (nvalid >= PSH_SEG) {
#pragma unroll
                for (int r = 0; r < 16; ++r) mn = fminf(mn, c[r]);
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int p = 32 * ((r & 3) + 8 * (r >> 2) + 4 * hk) + m;
                    mn = p < nvalid ? fminf(mn, c[r]) : mn;
                }
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) mn = fminf(mn, __shfl_xor(mn, off, 64));
            // An upper bound of the smallest acc of the unit (scan_lq_kernel's BOOT has the derivation):
            //     acc~ (1 - 2 a) <= t^ + nx~ (1 + 3 a) + b,   back to the data's scale.
            // A unit whose estimate t^ + nx~ is below nx~ / 6 -- a near-match: smooth ensembles -- would get a bound made of the
            // 3 a nx~ term alone: such a unit is sampled with the exact chains below instead.
            const float nx = nxq[q];
            const float est = fmaxf(mn + nx, 0.0f);
            if (est < nx * (1.0f / 6.0f)) loose = true;                       // (NaN: not loose -- the bound comes out NaN -> +inf)
            float ub = (est + nx * (3.0f / 900.0f) + (float)(2 * W + 2) / 64.0f / 262144.0f)
                       * (1.0f + 2.0f / 900.0f + 1.0e-5f + 6.0f * PSH_LONG_GAMMA) * (1.0f + 1e-6f);
            const float inv = __uint_as_float((unsigned)(127 - sexp) << 23);
            ub = ub * inv * inv;
            if (!(ub == ub)) ub = __uint_as_float(PSH_INF_BITS);              // NaN data: the unit carries no information
            if (lane == 0) st_sc1(&hdr->minima[(size_t)q * f.units_stride + u], __float_as_uint(ub));   // write-through
        }
        wave_lds_fence();                                                     // all lanes done with the arrays before they are overwritten
        if (__builtin_amdgcn_readfirstlane(loose ? 1 : 0)) {
            // the unit again, as stream_sample_kernel samples it: its fp32 samples (an L2 hit) into the scratch -- prefix sums and
            // rows are dead --, the exact chains of the lane's 16 windows, the exact minimum over the unit
            {
                Stage s2;
                stage_load<false>(s2, a.dataset + (f.boot_row0 + (int64_t)ri * f.boot_row_stride) * a.T, a.T, seg_start, nfloat, lane);
                stage_store<false>(s2, sp, nfloat, lane);
            }
            wave_lds_fence();
            const int t_lane = seg_start + PSH_L * lane;
            int nv = a.Tp - t_lane;
            nv = nv < 0 ? 0 : (nv > PSH_L ? PSH_L : nv);
#pragma unroll 1
            for (int q = 0; q < nq; ++q) {
                float acc[PSH_L];
                accumulate16<0, false>(sp, lane, (const_f32p)a.queries + (size_t)q * W, W, acc);
                float mq = __uint_as_float(PSH_INF_BITS);
#pragma unroll
                for (int i = 0; i < PSH_L; ++i) mq = (i < nv) ? fminf(mq, acc[i]) : mq;
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) mq = fminf(mq, __shfl_xor(mq, off, 64));
                if (!(mq == mq)) mq = __uint_as_float(PSH_INF_BITS);
                if (lane == 0) st_sc1(&hdr->minima[(size_t)q * f.units_stride + u], __float_as_uint(mq));
            }
            wave_lds_fence();
            {   // (the rows held fp32 samples: every slot a segment does not write must be finite again)
                unsigned* z = reinterpret_cast<unsigned*>(a1);
                for (int i = lane; i < NROWS * PSH_LONG_ROW / 2; i += 64) z[i] = 0u;
            }
            wave_lds_fence();
        }
        u = un;
    }
    // (the rows of the block's first wave -- 620 floats and more, >= 3 W -- take the staged queries of the launch's tail; with four
    //  waves the others may still be sampling: the barrier in front of the finish's ticket is their last use of anything here)
    stream_sample_finish<NWP>(a, f, sp, W, nq, nbu, false, lane, wave,
                              reinterpret_cast<float*>(reinterpret_cast<_Float16*>(reinterpret_cast<float*>(tab + (size_t)nq * QS) + PSH_LONG_SFLOATS)));
}
// one query: 128 registers, so that the wave fits beside the four scan waves (96 registers each) of another step on its SIMD
template <int NKS>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 8))) void stream_sample_long_kernel1(ScanArgs a, FusedArgs f) {
    stream_sample_long_body<NKS, 1>(a, f);
}
// (W > 161: no sample block fits in the LDS a scan block leaves, whatever its registers -- and the cap would spill the longer chains)
template <int NKS>
__global__ __launch_bounds__(64) void stream_sample_long_kernel1u(ScanArgs a, FusedArgs f) {
    stream_sample_long_body<NKS, 1>(a, f);
}
template <int NKS>
__global__ __launch_bounds__(256) void stream_sample_long_kernel4(ScanArgs a, FusedArgs f) {
    stream_sample_long_body<NKS, 4>(a, f);
}

// ------------------------------------------------------------------------------------------------------------------
// R: ranking by counting (grid.y = query)
// ------------------------------------------------------------------------------------------------------------------
template <bool PACKED>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(8, 8)))
void stream_rank_kernel(ScanArgs a, FusedArgs f) {
    __builtin_amdgcn_s_setprio(3);                           // (see stream_sample_kernel)
    const int lane = lane_id();
    const int q = (int)blockIdx.y;
    FusedHdr* hdr = f.hdr;
    const StreamCtl* sc = &hdr->stream;
    const int ncand = (int)sc->ncand[q];
    const bool good = sc->armed != 0u && sc->ovf == 0u && ncand >= a.k && ncand <= f.cand_cap;
    if (blockIdx.x == 0 && lane == 0) {
        f.status[q] = good ? PSH_STATUS_OK_ : PSH_STATUS_RETRY_;
        if (f.total) f.total[q] = ncand;
        if (a.qstate) {                                     // diagnostics / the separate launches' state, kept coherent
            QueryState qs;
            qs.xn = __uint_as_float(sc->xn_bits[q]); qs.tau_bits = sc->tau2_bits[q]; qs.n_valid = good ? a.k : 0; qs.nx = 0.0f;
            qs.thr_base = __uint_as_float(PSH_INF_BITS); qs.mx_scale = __uint_as_float(sc->scale_bits);
            qs.mx_thr = __uint_as_float(sc->thr2_bits[q]); qs.tau2_bits = sc->tau2_bits[q]; qs.mx_thr2 = qs.mx_thr;
            qs.mx8_P = qs.mx8_L = qs.mx8_k1 = 0.0f;
            a.qstate[q] = qs;
        }
    }
    if (!good) {                                             // (uniform over the launch: no block writes a rank then)
        if (blockIdx.x == 0) poison_results(f.out_d + (size_t)q * f.k_out, f.out_idx + (size_t)q * f.k_out * 2, a.k, lane, 64);
        return;
    }
    const u32x4v* cand = reinterpret_cast<const u32x4v*>(f.cand_list) + (size_t)q * f.cand_cap;
    float* out_d = f.out_d + (size_t)q * f.k_out;
    int32_t* out_idx = f.out_idx + (size_t)q * f.k_out * 2;
    const int per = (ncand + (int)gridDim.x - 1) / (int)gridDim.x;
    const int lo = (int)blockIdx.x * per;
    const int hi = lo + per < ncand ? lo + per : ncand;
    constexpr int NB = 4;                                    // candidate loads in flight per lane
    for (int j0 = lo; j0 < hi; j0 += 8) {
        // the (up to) 8 own candidates of this turn: lane jj holds candidate j0 + jj, its key goes to scalar registers
        const u32x4v mine = cand[j0 + (lane & 7) < hi ? j0 + (lane & 7) : lo];
        const u64 mkey = PACKED ? (((u64)mine[0] << 32) | (u64)((mine[1] << f.tbits) | mine[2]))
                                : (((u64)mine[1] << 32) | (u64)mine[2]);
        u64 okey[8];
        unsigned od[8];
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            okey[jj] = ((u64)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(mkey >> 32), jj) << 32)
                       | (u64)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)mkey, jj);
            od[jj] = (unsigned)__builtin_amdgcn_readlane((int)mine[0], jj);
        }
        int c[8];
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) c[jj] = 0;
        for (int i0 = 0; i0 < ncand; i0 += 64 * NB) {
            u32x4v e[NB];
#pragma unroll
            for (int q = 0; q < NB; ++q) {
                const int ci = i0 + lane + 64 * q;
                e[q] = cand[ci < ncand ? ci : ncand - 1];
            }
#pragma unroll
            for (int q = 0; q < NB; ++q) {
                const bool live = i0 + lane + 64 * q < ncand;
                if (PACKED) {
                    const u64 key = live ? (((u64)e[q][0] << 32) | (u64)((e[q][1] << f.tbits) | e[q][2])) : ~0ull;
#pragma unroll
                    for (int jj = 0; jj < 8; ++jj) c[jj] += key < okey[jj] ? 1 : 0;
                } else {
                    const unsigned kd = live ? e[q][0] : 0xffffffffu;
                    const u64 krt = live ? (((u64)e[q][1] << 32) | (u64)e[q][2]) : ~0ull;
#pragma unroll
                    for (int jj = 0; jj < 8; ++jj) c[jj] += (kd < od[jj] || (kd == od[jj] && krt < okey[jj])) ? 1 : 0;
                }
            }
        }
        int rank = 0;                                        // lane jj keeps the rank of own candidate jj
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            int v = c[jj];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
            rank = lane == jj ? v : rank;
        }
        if (lane < 8 && j0 + lane < hi && rank < a.k) {
            out_d[rank] = __uint_as_float(mine[0]);
            out_idx[2 * rank + 0] = (int)mine[1];
            out_idx[2 * rank + 1] = (int)mine[2];
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
size_t stream_scan_shmem_bytes_q(int tile_floats, int nq) {
    return (size_t)PSH_STREAM_FIXED_BYTES + (size_t)(nq == 1 ? PSH_FUSED_FRONT : 2 * PSH_FUSED_FRONT) * 16
           + (size_t)tile_floats * (PSH_SCAN_THREADS / 64) * sizeof(float)
           + (size_t)(PSH_SCAN_THREADS / 64) * 2 * PSH_MX_NHALF * sizeof(_Float16)
           + (nq > 1 ? (size_t)nq * 4 * 64 * 8 * sizeof(_Float16) : 0);
}
size_t stream_scan_shmem_bytes(int tile_floats) { return stream_scan_shmem_bytes_q(tile_floats, 1); }
bool stream_long_supported(int W) { return W >= 34 && W <= 256; }
size_t stream_scan_long_shmem_bytes(int W, int nq) {
    // control words, the block's list, the queries' fragment tables; per wave: the prefix sums, the survivors' queue, the f16 rows
    return (size_t)PSH_STREAM_FIXED_BYTES + (size_t)(nq == 1 ? PSH_FUSED_FRONT : 2 * PSH_FUSED_FRONT) * 16
           + (size_t)nq * stream_ksteps(W) * 64 * 8 * sizeof(_Float16)
           + (size_t)(PSH_SCAN_THREADS / 64) * ((size_t)(PSH_LONG_SFLOATS + 2 * PSH_LONG_QCAP) * sizeof(float) + (size_t)stream_long_nhalf(W) * sizeof(_Float16));
}
size_t stream_sample_shmem_bytes(int tile_floats) {
    const size_t t = (size_t)tile_floats * sizeof(float), h = (size_t)PSH_STREAM_HIST * sizeof(unsigned);
    return t > h ? t : h;
}

template <typename K>
static hipError_t launch_k(K kernel, dim3 grid, int threads, size_t shmem, hipStream_t s, const ScanArgs& a, const FusedArgs& f) {
    if (shmem > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kernel, grid, dim3(threads), shmem, s, a, f);
    return hipGetLastError();
}

template <int NWP>
static hipError_t launch_stream_sample_w(const ScanArgs& b, const FusedArgs& f, bool aligned, int grid, size_t shmem, hipStream_t s) {
    if (b.W == 20)
        return aligned ? launch_k(stream_sample_kernel<20, true, NWP>, dim3(grid), 64 * NWP, shmem, s, b, f)
                       : launch_k(stream_sample_kernel<20, false, NWP>, dim3(grid), 64 * NWP, shmem, s, b, f);
    return aligned ? launch_k(stream_sample_kernel<0, true, NWP>, dim3(grid), 64 * NWP, shmem, s, b, f)
                   : launch_k(stream_sample_kernel<0, false, NWP>, dim3(grid), 64 * NWP, shmem, s, b, f);
}
// grid: one-wave blocks for one query (they have to fit beside another step's scan), a quarter as many four-wave blocks for two
// or three
hipError_t launch_stream_sample(const ScanArgs& a, const FusedArgs& f, bool aligned, int grid, int sample_tile_floats, hipStream_t s) {
    ScanArgs b = a;
    b.tile_floats = sample_tile_floats;
    if (f.nq == 1) return launch_stream_sample_w<1>(b, f, aligned, grid, stream_sample_shmem_bytes(sample_tile_floats), s);
    return launch_stream_sample_w<4>(b, f, aligned, (grid + 3) / 4, 4 * stream_sample_shmem_bytes(sample_tile_floats), s);
}

size_t stream_sample_long_shmem_bytes(int W, int nq) {
    const int nks = lq_bucket(W), nwp = nq == 1 ? 1 : 4;
    return 64 + (size_t)nq * lq_query_bytes(nks) + (size_t)nwp * ((size_t)PSH_LONG_SFLOATS * 4 + (size_t)lq_rows(nks) * PSH_LONG_ROW * 2);
}
static hipError_t launch_stream_sample_long_1(const ScanArgs& a, const FusedArgs& f, int grid, hipStream_t s) {
    const int nks = lq_bucket(a.W);
    const size_t shmem = stream_sample_long_shmem_bytes(a.W, f.nq);
    if (nks == 6) return launch_k(stream_sample_long_kernel1<6>, dim3(grid), 64, shmem, s, a, f);
    if (nks == 10) return launch_k(stream_sample_long_kernel1<10>, dim3(grid), 64, shmem, s, a, f);
    if (nks == 14) return launch_k(stream_sample_long_kernel1u<14>, dim3(grid), 64, shmem, s, a, f);
    return launch_k(stream_sample_long_kernel1u<18>, dim3(grid), 64, shmem, s, a, f);
}
static hipError_t launch_stream_sample_long_4(const ScanArgs& a, const FusedArgs& f, int grid, hipStream_t s) {
    const int nks = lq_bucket(a.W);
    const size_t shmem = stream_sample_long_shmem_bytes(a.W, f.nq);
    if (nks == 6) return launch_k(stream_sample_long_kernel4<6>, dim3(grid), 256, shmem, s, a, f);
    if (nks == 10) return launch_k(stream_sample_long_kernel4<10>, dim3(grid), 256, shmem, s, a, f);
    if (nks == 14) return launch_k(stream_sample_long_kernel4<14>, dim3(grid), 256, shmem, s, a, f);
    return launch_k(stream_sample_long_kernel4<18>, dim3(grid), 256, shmem, s, a, f);
}
// the long-window step's sample (34 <= W <= 256, no hint): one-wave blocks for one query, a quarter as many four-wave blocks for
// two or three (launch_stream_sample's rule)
hipError_t launch_stream_sample_long(const ScanArgs& a, const FusedArgs& f, int grid, hipStream_t s) {
    if (f.nq == 1) return launch_stream_sample_long_1(a, f, grid, s);
    return launch_stream_sample_long_4(a, f, (grid + 3) / 4, s);
}

template <int NQ>
static hipError_t launch_stream_scan_q(const ScanArgs& a, const FusedArgs& f, bool aligned, int grid, hipStream_t s) {
    const size_t shmem = stream_scan_shmem_bytes_q(a.tile_floats, NQ);
    if (a.W == 20)
        return aligned ? launch_k(stream_scan_q_kernel<20, true, NQ>, dim3(grid), PSH_SCAN_THREADS, shmem, s, a, f)
                       : launch_k(stream_scan_q_kernel<20, false, NQ>, dim3(grid), PSH_SCAN_THREADS, shmem, s, a, f);
    return aligned ? launch_k(stream_scan_q_kernel<0, true, NQ>, dim3(grid), PSH_SCAN_THREADS, shmem, s, a, f)
                   : launch_k(stream_scan_q_kernel<0, false, NQ>, dim3(grid), PSH_SCAN_THREADS, shmem, s, a, f);
}
hipError_t launch_stream_scan(const ScanArgs& a, const FusedArgs& f, bool aligned, int grid, hipStream_t s) {
    if (f.nq == 2) return launch_stream_scan_q<2>(a, f, aligned, grid, s);
    if (f.nq == 3) return launch_stream_scan_q<3>(a, f, aligned, grid, s);
    const size_t shmem = stream_scan_shmem_bytes_q(a.tile_floats, 1);
    if (a.W == 20)
        return aligned ? launch_k(stream_scan_kernel<20, true>, dim3(grid), PSH_SCAN_THREADS, shmem, s, a, f)
                       : launch_k(stream_scan_kernel<20, false>, dim3(grid), PSH_SCAN_THREADS, shmem, s, a, f);
    return aligned ? launch_k(stream_scan_kernel<0, true>, dim3(grid), PSH_SCAN_THREADS, shmem, s, a, f)
                   : launch_k(stream_scan_kernel<0, false>, dim3(grid), PSH_SCAN_THREADS, shmem, s, a, f);
}

template <int NQ>
static hipError_t launch_stream_scan_long_q(const ScanArgs& a, const FusedArgs& f, bool aligned, int grid, hipStream_t s) {
    const size_t shmem = stream_scan_long_shmem_bytes(a.W, NQ);
    return aligned ? launch_k(stream_scan_long_kernel<true, NQ>, dim3(grid), PSH_SCAN_THREADS, shmem, s, a, f)
                   : launch_k(stream_scan_long_kernel<false, NQ>, dim3(grid), PSH_SCAN_THREADS, shmem, s, a, f);
}
hipError_t launch_stream_scan_long(const ScanArgs& a, const FusedArgs& f, bool aligned, int grid, hipStream_t s) {
    return f.nq == 1 ? launch_stream_scan_long_q<1>(a, f, aligned, grid, s)
         : f.nq == 2 ? launch_stream_scan_long_q<2>(a, f, aligned, grid, s) : launch_stream_scan_long_q<3>(a, f, aligned, grid, s);
}

hipError_t launch_stream_rank(const ScanArgs& a, const FusedArgs& f, int grid, hipStream_t s) {
    return f.tbits >= 0 ? launch_k(stream_rank_kernel<true>, dim3(grid, f.nq), 64, 0, s, a, f)
                        : launch_k(stream_rank_kernel<false>, dim3(grid, f.nq), 64, 0, s, a, f);
}

}  // namespace psh
