// psh_lq.hip -- BATCHED queries with a LONG window (gfx950): B >= 4 queries with 26 <= W <= 256 (two / three when their tables do
// not ride one pass of the three launches, psh_stream.hip: W > 97 / 145), Identity + RelativeMSE (reference
// path_embedding.py:135-139 takes any Identity(dimension); predict() loops over many query dates, path_shadowing.py:286-301).
// Part of libpsh_hip.so.  Until round 6 such a call was a loop of two- or three-query passes of the single-query long-window
// scan (psh_stream.hip): W = 126, 64 queries = 32 passes over the ensemble, 6.4 ms.  Here ONE pass per chunk of queries:
//
//   * a block of 8 waves (two per SIMD: 256 registers a lane) owns a share of the (row, segment) units and a CHUNK of the
//     queries -- as many as put their fragment tables (eight shifted copies of -2 x~ per query, lq_copy_chunks below: 4.6 KB a
//     query at W = 126) in LDS beside the waves' buffers: 20 at W = 126, 13 at W = 252, 32 at W = 64; grid.y = the chunks;
//   * a wave stages a segment (buffer loads), converts it to f16 rows and takes the WINDOW ENERGIES from fp32 prefix sums of the
//     squares (stream_scan_long_kernel's construction: the tile's C operand is E^ - gamma S, shared by every query);
//   * the chunk's queries in groups of FOUR, K-step by K-step: a step's A fragment multiplies four queries' B fragments into four
//     independent tiles, ONE aligned 16-byte LDS read per MFMA (128 B/clk a CU, half the LDS rate).  The matrix cores' share:
//     B x (W + 31 rounded up to K-steps of 16) x 2 flop a window;
//   * FILTER: windows whose t^ = E^ - 2 c^ does not exceed the query's rejection threshold (stream_threshold's bound) go to the
//     wave's queue and are verified with the reference's exact chain straight from memory (a lane a window), the admitted ones
//     (acc < tau) to the query's slice of this block -- the layout select_kernel reads (psh_select.hip);
//     BOOT: the minimum of an UPPER bound of acc per (unit, query) -> minbuf, for threshold_kernel (the sample that gives every
//     query its admission level; upper bounds, so also a provable level when the plan wants one).
// The launches around it are the separate launches' (psh_capi.hip): bootstrap -> threshold_kernel -> scan -> select_kernel.
// Results are the exact top-k: what is ranked is always the sequential fp32 chain (include/psh.h).
#include <type_traits>
#include "psh_device.h"

namespace psh {

typedef unsigned long long u64;
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));

#define PSH_LQ_THREADS 512
#define PSH_LQ_ROW 40                 // halves a row of 32 samples takes (8 of padding: ds_read_b128 of 16 rows on 16 bank quads)
#define PSH_LQ_SFLOATS 1280           // prefix sums of a segment's squares: entries 0 .. SEG + W - 1 <= 1279
#define PSH_LQ_QCAP 64                // deferred survivors a wave keeps
#define PSH_LQ_GAMMA 7.62939453125e-06f   // 2^-17 (psh_stream.hip, PSH_LONG_GAMMA: the same construction, the same bound)
#define PSH_LQ_FIXED 2048             // control words, per-query constants and counters
#define PSH_LQ_MAXQ 32                // queries a chunk holds at most

// (lq_ksteps, lq_bucket, lq_rows, lq_copy_chunks, lq_query_bytes: psh_kernels.h -- the long-window sample of the three launches,
//  psh_stream.hip, lays its tables out the same way)
__host__ __device__ inline size_t lq_wave_bytes(int nks) {
    return (size_t)PSH_LQ_SFLOATS * 4 + (size_t)PSH_LQ_QCAP * 8 + (size_t)lq_rows(nks) * PSH_LQ_ROW * 2;
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float lq_dpp_add(float inc) {
    return inc + __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(inc), CTRL, ROW_MASK, 0xf, true));
}

// MODE: PSH_MODE_BOOT / PSH_MODE_FILTER.  NKS: K-steps compiled in.
template <int MODE, int NKS>
__global__ __launch_bounds__(PSH_LQ_THREADS) void scan_lq_kernel(ScanArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NW = PSH_LQ_THREADS / 64;
    const int lane = lane_id();
    const int tid = (int)threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int W = a.W;
    const int q0 = (int)blockIdx.y * a.q_per_group;
    const int nq = a.B - q0 < a.q_per_group ? a.B - q0 : a.q_per_group;
    // fixed area: control words [0, 16), per-query {thr or nx~, tau, -, -} [16, 16 + 4 MAXQ), counters
    int* ctl = reinterpret_cast<int*>(smem);
    float* qc = smem + 16;                                                    // 4 floats a query
    int* lcount = reinterpret_cast<int*>(smem + 16 + 4 * PSH_LQ_MAXQ);        // this block's cursor in the slice of every query of its chunk
    _Float16* tab = reinterpret_cast<_Float16*>(smem + PSH_LQ_FIXED / 4);     // [query][copy c < 8][chunk < CP][8 halves]
    constexpr int CP = lq_copy_chunks(NKS), QS = lq_query_bytes(NKS) / 2;     // chunks a copy, HALVES a query
    char* wbase = reinterpret_cast<char*>(tab + (size_t)a.q_per_group * QS) + (size_t)wave * lq_wave_bytes(NKS);
    float* sp = reinterpret_cast<float*>(wbase);                              // prefix sums
    unsigned* sq_row = reinterpret_cast<unsigned*>(sp + PSH_LQ_SFLOATS);      // deferred survivors: row, t | query << 27
    unsigned* sq_tq = sq_row + PSH_LQ_QCAP;
    _Float16* a1 = reinterpret_cast<_Float16*>(sq_tq + PSH_LQ_QCAP);          // rows of {y^ [32], pad [8]}

    const int nfloat = PSH_SEG + W - 1;
    const unsigned n_rs = (unsigned)a.n_rows * (unsigned)a.nseg;
    const unsigned u_lo = (unsigned)(((u64)n_rs * blockIdx.x) / gridDim.x);
    const unsigned u_hi = (unsigned)(((u64)n_rs * (blockIdx.x + 1u)) / gridDim.x);
    auto decode = [&](unsigned uu, unsigned& ri, unsigned& sg) {
        ri = fast_div(uu, a.magic_nseg, (unsigned)a.nseg);
        sg = uu - ri * (unsigned)a.nseg;
    };
    const int lane16 = lane * 16;
    // (two units in flight per wave -- a second Stage, the loop body twice -- moved the kernel's floor at W = 64 from 1.33 to 1.02 ms
    //  and nothing at W >= 126, where the floor is the per-segment work, not the loaded latency: not kept)
    Stage st;
    auto stage_row = [&](Stage& st, int64_t row, int seg_start) {
        const int64_t bytes = a.T * 4;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dataset + row * a.T), 0,
                                                                             (int)(bytes > 0x7ffffffc ? 0x7ffffffc : bytes), 0x00020000);
        const int nq4 = (nfloat + 3) >> 2;
#pragma unroll
        for (int q = 0; q < PSH_NSTAGE; ++q)
            if (q < PSH_NSTAGE - 1 || lane + 64 * q < nq4) {
                const u32x4v w = __builtin_amdgcn_raw_buffer_load_b128(rs, lane16 + 1024 * q, seg_start * 4, 0);
                st.v[q] = f32x4{__uint_as_float(w[0]), __uint_as_float(w[1]), __uint_as_float(w[2]), __uint_as_float(w[3])};
            }
    };
    auto load_unit = [&](Stage& sx, unsigned uu) {
        unsigned ri, sg;
        decode(uu, ri, sg);
        stage_row(sx, a.row0 + (int64_t)ri * a.row_stride, (int)sg * PSH_SEG);
    };
    unsigned u = u_lo + (unsigned)wave;
    if (u < u_hi) load_unit(st, u);

    // ---- the chunk's scale and per-query constants.  One f16 scale 2^sexp for the chunk: max|x_q| 2^sexp < 8 for every query,
    //      and (FILTER) tau_q 4^sexp <= 4096 -- the fused launch's rule (psh_fused.hip, derive_levels).  A query whose level or
    //      samples are not finite positive numbers is not armed: its threshold is +inf, every window is verified exactly.
    //      The chunk's queries are staged in LDS first (the waves' buffers, not in use yet): the sums and the tables below read
    //      every sample dozens of times, and from memory a block's set-up was 126 dependent round trips per query -- 35 us a
    //      block, once per chunk and launch.
    float* xs = reinterpret_cast<float*>(tab + (size_t)a.q_per_group * QS);   // [query][W]
    for (int e = tid; e < nq * W; e += PSH_LQ_THREADS) xs[e] = a.queries[(size_t)q0 * W + e];
    if (tid == 0) { ctl[0] = 60; ctl[1] = NW; }                              // ctl[0]: the chunk's exponent (minimum); ctl[1]: the next unit
    if (tid < PSH_LQ_MAXQ) lcount[tid] = 0;
    __syncthreads();
    for (int ql = wave; ql < nq; ql += NW) {
        const float* xq = xs + (size_t)ql * W;
        unsigned mb = 0u;
        for (int j = lane; j < W; j += 64) mb = max(mb, __float_as_uint(fabsf(xq[j])));
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) mb = max(mb, (unsigned)__shfl_xor((int)mb, off, 64));
        if (lane == 0) {
            int sexp = 60;
            bool ok = mb < PSH_INF_BITS;
            if (mb >= 0x00800000u) { const int eq = (int)((mb >> 23) & 255u) - 126; sexp = 3 - eq; }
            float tau0 = 0.0f;
            if (MODE == PSH_MODE_FILTER) {
                tau0 = __uint_as_float(a.qstate[q0 + ql].tau_bits);
                ok = ok && tau0 > 0.0f && tau0 < __uint_as_float(PSH_INF_BITS) && __float_as_uint(tau0) >= 0x00800000u;
                if (ok) {
                    const int et = (int)((__float_as_uint(tau0) >> 23) & 255u) - 126;
                    const int st2 = (12 - et) >= 0 ? (12 - et) / 2 : -((et - 12 + 1) / 2);
                    sexp = sexp < st2 ? sexp : st2;
                }
            }
            ok = ok && sexp >= -60;
            qc[4 * ql + 1] = tau0;
            qc[4 * ql + 2] = ok ? 1.0f : 0.0f;
            if (ok) atomicMin(&ctl[0], sexp);
        }
    }
    __syncthreads();
    const int sexp = ctl[0] > 60 ? 60 : ctl[0];
    const float scale = __uint_as_float((unsigned)(127 + sexp) << 23);
    if (tid < nq) {
        // the query's constant under the chunk's scale: FILTER the rejection threshold (stream_threshold, psh_stream.hip: a = 2^-9
        // relative, b absolute growing with the taps), BOOT nx~ = sum x~^2
        const float* xq = xs + (size_t)tid * W;
        double nxs = 0.0;
        for (int j = 0; j < W; ++j) { const double vv = (double)xq[j] * (double)scale; nxs += vv * vv; }
        float out = __uint_as_float(PSH_INF_BITS);
        if (MODE == PSH_MODE_FILTER) {
            if (qc[4 * tid + 2] != 0.0f) {
                const double am = 1.0 / 900.0, bm = (1.0 / 262144.0) * ((double)(2 * W + 2) / 64.0);   // (stream_threshold, W > 33: only the correlation's f16 error is left)
                const double taus = (double)qc[4 * tid + 1] * (double)scale * (double)scale;
                const double T = taus * (1.0 + 1.0 / 131072.0) * (1.0 + 2.0 * am) - nxs * (1.0 - 3.0 * am) * (1.0 - 1e-12) + bm;
                float Tf = (float)T;
                if ((double)Tf < T) Tf = __uint_as_float(Tf >= 0.0f ? __float_as_uint(Tf) + 1u : __float_as_uint(Tf) - 1u);
                if (Tf == Tf && fabsf(Tf) < __uint_as_float(PSH_INF_BITS)) out = Tf;
            }
        } else {
            out = (float)(nxs * (1.0 + 1e-6));
        }
        qc[4 * tid + 0] = out;
    }
    // the tables: copy c, chunk v, half i holds -2 x~[8 (v - 3) + i - c]; zero outside the window
    for (int e = tid; e < nq * 8 * CP; e += PSH_LQ_THREADS) {
        const int v = e % CP, c = (e / CP) & 7, ql = e / (8 * CP);
        const float* xq = xs + (size_t)ql * W;
        f16x8 b;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int j = 8 * (v - 3) + i - c;
            const bool in = j >= 0 && j < W;
            const float xv = xq[in ? j : 0];
            b[i] = (_Float16)(in ? -2.0f * (xv * scale) : 0.0f);
        }
        *reinterpret_cast<f16x8*>(tab + (size_t)ql * QS + ((size_t)c * CP + v) * 8) = b;
    }
    __syncthreads();                                                          // (the staged queries are dead: the waves' buffers)
    {   // every slot of the rows a segment does not write must be finite (0 * NaN poisons a row)
        unsigned* z = reinterpret_cast<unsigned*>(a1);
        for (int i = lane; i < lq_rows(NKS) * PSH_LQ_ROW / 2; i += 64) z[i] = 0u;
    }
    wave_lds_fence();
    auto grab = [&]() -> unsigned {
        int v = 0;
        if (lane == 0) v = atomicAdd(&ctl[1], 1);
        return u_lo + (unsigned)__builtin_amdgcn_readfirstlane(v);
    };

    // ---- deferred survivors (stream_scan_long_kernel's scheme): verified a lane a window, straight from memory
    int qn = 0;
    typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
    auto verify_queue = [&]() {
#ifdef PSH_TUNING
        if ((a.dbg & 16) && lane == 0) atomicAdd(&a.bcount[PSH_MAX_BLOCKS - 1], qn);   // probe: survivors verified (tools/lq_ablate.py)
#endif
#ifdef PSH_TUNING
        if (a.dbg & 8) { qn = 0; return; }                                   // ablation: queued survivors dropped unverified (results invalid)
#endif
        const bool have = lane < qn;
        const unsigned row = have ? sq_row[lane] : 0u, tq = have ? sq_tq[lane] : 0u;
        const int ql = (int)(tq >> 27);
        const unsigned t = tq & 0x07ffffffu;
        const int b = q0 + ql;
        const float* y = a.dataset + (int64_t)row * a.T + t;
        const float* x = a.queries + (size_t)b * W;
        float v = 0.0f;
        if (have) {
            int j = 0;
#pragma unroll 2
            for (; j + 4 <= W; j += 4) {
                const f32x4u yy = *reinterpret_cast<const f32x4u*>(y + j);
                const f32x4u xx = *reinterpret_cast<const f32x4u*>(x + j);
#pragma unroll
                for (int c = 0; c < 4; ++c) { const float D = __fsub_rn(xx[c], yy[c]); v = __builtin_fmaf(D, D, v); }
            }
            for (; j < W; ++j) { const float D = __fsub_rn(x[j], y[j]); v = __builtin_fmaf(D, D, v); }
        }
        const bool hit = have && (v < qc[4 * ql + 1]);
        if (hit) {
            const int pos = atomicAdd(&lcount[ql], 1);
            if (pos < a.slice) {
                const int64_t o = (int64_t)b * a.cap + (int64_t)blockIdx.x * a.slice + pos;
                a.cand_d[o] = dist_from_acc(v, a.qstate[b].xn);
                a.cand_rt[o] = make_int2((int)((int64_t)row + a.r_offset), (int)t);
            }
        }
        wave_lds_fence();
        qn = 0;
    };

    const int m = lane & 31, hk = lane >> 5;
    const _Float16* pa0 = a1 + m * PSH_LQ_ROW + 8 * hk;
    const _Float16* pb0 = tab + ((size_t)(m & 7) * CP + (hk - (m >> 3) + 3)) * 8;       // copy n & 7, chunk hk - (n >> 3) + 3 (+ 2 s a K-step)
    const float* ps_lo = sp + m + 128 * hk;
    const float* ps_hi = ps_lo + W;
    while (u < u_hi) {
        unsigned ri, sg;
        decode(u, ri, sg);
        const int64_t row = a.row0 + (int64_t)ri * a.row_stride;
        const int seg_start = (int)sg * PSH_SEG;
        bool clean;                                                           // the segment holds no NaN / inf and nothing the f16 rows overflow on
        {
            // f16 rows and the fp32 prefix sums of the squares (psh_stream.hip, stream_scan_long_kernel: the bound is there)
            const int nq4 = (nfloat + 3) >> 2;
            float d0[PSH_NSTAGE], d1[PSH_NSTAGE], d2[PSH_NSTAGE], d3[PSH_NSTAGE], inc[PSH_NSTAGE];
#pragma unroll
            for (int q = 0; q < PSH_NSTAGE; ++q) {
                const int mm = lane + 64 * q;
                const bool on = q < PSH_NSTAGE - 1 || mm < nq4;
                const f32x4 v = on ? st.v[q] * scale : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                const f32x4 z = v * v;
                if (on) *reinterpret_cast<f16x4*>(a1 + (mm >> 3) * PSH_LQ_ROW + 4 * (mm & 7)) = __builtin_convertvector(v, f16x4);
                d0[q] = z[0]; d1[q] = d0[q] + z[1]; d2[q] = d1[q] + z[2]; d3[q] = d2[q] + z[3];
                inc[q] = d3[q];
            }
#pragma unroll
            for (int q = 0; q < PSH_NSTAGE; ++q) inc[q] = lq_dpp_add<0x111, 0xf>(inc[q]);
#pragma unroll
            for (int q = 0; q < PSH_NSTAGE; ++q) inc[q] = lq_dpp_add<0x112, 0xf>(inc[q]);
#pragma unroll
            for (int q = 0; q < PSH_NSTAGE; ++q) inc[q] = lq_dpp_add<0x114, 0xf>(inc[q]);
#pragma unroll
            for (int q = 0; q < PSH_NSTAGE; ++q) inc[q] = lq_dpp_add<0x118, 0xf>(inc[q]);
#pragma unroll
            for (int q = 0; q < PSH_NSTAGE; ++q) inc[q] = lq_dpp_add<0x142, 0xa>(inc[q]);
#pragma unroll
            for (int q = 0; q < PSH_NSTAGE; ++q) inc[q] = lq_dpp_add<0x143, 0xc>(inc[q]);
            float carry = 0.0f;
#pragma unroll
            for (int q = 0; q < PSH_NSTAGE; ++q) {
                const int mm = lane + 64 * q;
                const float x0 = carry + (inc[q] - d3[q]);
                carry += __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(inc[q]), 63));
                if (q < PSH_NSTAGE - 1 || mm <= nq4) *reinterpret_cast<f32x4*>(sp + 4 * mm) = f32x4{x0, x0 + d0[q], x0 + d1[q], x0 + d2[q]};
            }
            // the segment's total of squares says whether any tile value can be NaN: a NaN / inf sample makes it NaN / inf, a sample
            // beyond the f16 range (|y~| > 65504: the rows hold inf, 0 x inf = NaN in its row's tiles) makes it > 4.29e9
            // (a wave-uniform value, but a vector compare's mask: without the readfirstlane the compiler treats every branch on it
            //  as divergent and carries the test's sixteen scalar masks through vector registers)
            clean = __builtin_amdgcn_readfirstlane(carry < 4.0e9f ? 1 : 0) != 0;
        }
        wave_lds_fence();
        const unsigned un = grab();
        if (un < u_hi) load_unit(st, un);
        // the tile's C operand from the prefix sums: FILTER a LOWER bound of the energies (E^ - gamma S), BOOT an UPPER one
        f32x16 ce;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int off = 32 * (r & 3) + 256 * (r >> 2);
            ce[r] = __builtin_fmaf(ps_hi[off], MODE == PSH_MODE_FILTER ? 1.0f - PSH_LQ_GAMMA : 1.0f + PSH_LQ_GAMMA, -ps_lo[off]);
        }
        const int nvalid = a.Tp - seg_start;                                  // windows of this segment that exist (>= 1024: all)
        if (nvalid < PSH_SEG) {
            // A row's last segment: the windows past the last admissible one never pass the test (+inf in their slots of the C
            // operand).  Left alone they pass it in most tiles: what lies beyond the row reads as zero, a window of zeros is at
            // acc = ||x||^2, and for a long window that IS about where the admission level sits (the k-th smallest of 10^8 sums of
            // W squared differences) -- a quarter of all tiles went through the survivors' code for windows that do not exist.
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (32 * ((r & 3) + 8 * (r >> 2) + 4 * hk) + m >= nvalid) ce[r] = __uint_as_float(PSH_INF_BITS);
        }
        // BOOT: the minimum of an upper bound of acc per (unit, query)
        auto boot_finish = [&](const f32x16& c, const int ql) __attribute__((always_inline)) {
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
            if (lane == 0) {
                // An upper bound of the smallest acc of the unit, back in the data's scale.  With 2 |c^ - c| <= a (nx~ + E) and
                // E <= 2 (nx~ + acc~) (stream_threshold's model, read the other way round):
                //     acc~ (1 - 2 a) <= t^ + nx~ (1 + 3 a) + b.
                // A unit whose estimate t^ + nx~ is below nx~ / 6 -- a near-match: smooth ensembles -- would get a bound made of the
                // 3 a nx~ term alone, several times its value; the ESTIMATE (1 + 3 a)(t^ + nx~) stands in there.  Nothing rests on
                // either being a proof: the level is checked by what it admits (select_kernel: at least k, or the fallback).
                const float nx = qc[4 * ql + 0];
                const float est = fmaxf(mn + nx, 0.0f);
                const float bb = (float)(2 * W + 2) / 64.0f / 262144.0f;
                float ub = est < nx * (1.0f / 6.0f)
                               ? (est * (1.0f + 3.0f / 900.0f + 6.0f * PSH_LQ_GAMMA) + bb) * (1.0f + 1e-6f)
                               : (est + nx * (3.0f / 900.0f) + bb) * (1.0f + 2.0f / 900.0f + 1.0e-5f + 6.0f * PSH_LQ_GAMMA) * (1.0f + 1e-6f);
                const float inv = __uint_as_float((unsigned)(127 - sexp) << 23);
                ub = ub * inv * inv;
                if (!(ub == ub)) ub = __uint_as_float(PSH_INF_BITS);      // NaN data: the unit carries no information
                a.minbuf[(int64_t)(q0 + ql) * a.min_stride + (int64_t)u] = ub;
            }
        };
        // FILTER, the test of a tile: does it hold a window with t^ <= thr (NaN-safe: !(t^ > thr))?  On a clean segment no NaN can
        // sit in the tile and its smallest value decides: 8 v_min3 and one compare; the tile that holds none -- 19 in 20 -- is
        // done.  What follows for the others is ONE piece of code per kernel, behind the group (tile_survivors below): inlined into
        // every tile's test the survivors' code made the kernel 70 KB of instructions and every visit of it -- a tile in twenty --
        // a string of instruction-cache misses: 1.3 of the scan's 3.9 ms at W = 126, 64 queries.
        auto tile_hit = [&](const f32x16& c, const int ql) __attribute__((always_inline)) -> bool {
#ifdef PSH_TUNING
            if (a.dbg & 2) { if (c[0] == 12345.678f) a.minbuf[0] = c[1]; return false; }   // ablation: no test (results invalid)
#endif
            if (!clean) return true;
            const float thr = qc[4 * ql + 0];
            float mn = fminf(fminf(c[0], c[1]), c[2]);
#pragma unroll
            for (int r = 3; r + 1 < 16; r += 2) mn = fminf(fminf(mn, c[r]), c[r + 1]);
            mn = fminf(mn, c[15]);
            return __ballot(!(mn > thr)) != 0ull;
        };
        // the windows of a tile that pass the test, a bit per accumulator slot (slot r of lane (m, hk) is window
        // 32 ((r & 3) + 8 (r >> 2) + 4 hk) + m)
        auto tile_bits = [&](const f32x16& c, const int ql) __attribute__((always_inline)) -> unsigned {
            const float thr = qc[4 * ql + 0];
            unsigned hm = 0u;
#pragma unroll
            for (int r = 0; r < 16; ++r) hm |= !(c[r] > thr) ? (1u << r) : 0u;
            return hm;
        };
        // ... go to the wave's queue: a loop over the LANES that hold one (one or two in a tile that holds any)
        auto queue_bits = [&](unsigned hm, const int ql) __attribute__((always_inline)) {
            if (nvalid < PSH_SEG) {                                           // a row's last segment: windows that do not exist
#pragma unroll 1
                for (int r = 0; r < 16; ++r)
                    if (32 * ((r & 3) + 8 * (r >> 2) + 4 * hk) + m >= nvalid) hm &= ~(1u << r);
            }
#ifdef PSH_TUNING
            if (a.dbg & 4) hm = 0u;                                           // ablation: no survivor handling (results invalid)
#endif
            unsigned long long lanes = __ballot(hm != 0u);
#pragma unroll 1
            while (lanes) {
                const int l = (int)__builtin_ctzll(lanes);
                lanes &= lanes - 1ull;
                unsigned h = (unsigned)__builtin_amdgcn_readlane((int)hm, l);
                const int n = (int)__popc(h);
                if (qn + n > PSH_LQ_QCAP) verify_queue();
                if (lane == l) {
                    int slot = qn;
#pragma unroll 1
                    for (; h; h &= h - 1u, ++slot) {
                        const int r = __builtin_ctz(h);
                        sq_row[slot] = (unsigned)row;
                        sq_tq[slot] = (unsigned)(seg_start + 32 * ((r & 3) + 8 * (r >> 2) + 4 * hk) + m) | ((unsigned)ql << 27);
                    }
                }
                qn += n;
            }
        };
        // The chunk's queries in GROUPS of up to four, K-step by K-step (round 6, second form): a step's A fragment is read once and
        // multiplies the fragments of the group's queries into FOUR independent tiles -- no MFMA waits for the one before it, and
        // when the last step has been issued for the fourth query the first query's tile is nearly ready for its test (eight tiles
        // left no registers to request fragments ahead: every other MFMA waited for an LDS read issued just before it).
        // (Query by query -- the segment's A fragments in registers, one dependent chain of MFMAs per query -- every chain ended
        // in a drain and a test with nothing to overlap them: 1460 cycles a query and wave where the chain itself is 320.)
        f32x16 acc[4];
        // returns a bit per tile of the group that holds a survivor (FILTER)
        auto run_group = [&](auto ng_tag, const int g0) __attribute__((always_inline)) -> unsigned {
            constexpr int NG = decltype(ng_tag)::value;
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = ce;
            const _Float16* pbg = pb0 + (size_t)g0 * QS;
            auto ld = [](const _Float16* p) { return *reinterpret_cast<const f16x8*>(p); };
            // two sets of fragments take turns: the next step's A fragment and NG B fragments are requested before this step's MFMAs
            f16x8 fa[2], fb[2][NG];
            fa[0] = ld(pa0);
#pragma unroll
            for (int j = 0; j < NG; ++j) fb[0][j] = ld(pbg + (size_t)j * QS);
            // (the first step's reads are a group of the pipeline too: without it every group below takes the reads of the step
            //  before its own and the LAST step's five are left to the default scheduler, which sinks each in front of its MFMA)
            __builtin_amdgcn_sched_group_barrier(0x100, NG + 1, 0);
#ifdef PSH_TUNING
            if (!(a.dbg & 1))                                                 // ablation: no MFMAs (results invalid)
#endif
#pragma unroll
            for (int s = 0; s < NKS; ++s) {
                if (s + 1 < NKS) {
                    fa[(s + 1) & 1] = ld(pa0 + ((s + 1) >> 1) * PSH_LQ_ROW + 16 * ((s + 1) & 1));
#pragma unroll
                    for (int j = 0; j < NG; ++j) fb[(s + 1) & 1][j] = ld(pbg + (size_t)j * QS + 16 * (s + 1));
                }
#pragma unroll
                for (int j = 0; j < NG; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[s & 1], fb[s & 1][j], acc[j], 0, 0, 0);
                // the order the scheduler has to keep: the next step's NG + 1 reads, THEN this step's NG MFMAs (left alone it issues
                // the reads two at a time right in front of the MFMA that needs the first of them)
                if (s + 1 < NKS) __builtin_amdgcn_sched_group_barrier(0x100, NG + 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, NG, 0);
            }
            unsigned hitm = 0u;
            if (MODE == PSH_MODE_BOOT) {
#pragma unroll
                for (int j = 0; j < NG; ++j) boot_finish(acc[j], g0 + j);
            } else {
#pragma unroll
                for (int j = 0; j < NG; ++j) hitm |= tile_hit(acc[j], g0 + j) ? (1u << j) : 0u;
            }
            return (unsigned)__builtin_amdgcn_readfirstlane((int)hitm);       // (uniform: said so, the compiler keeps it in a vector register otherwise)
        };
        for (int g0 = 0; g0 < nq; g0 += 4) {
            unsigned hitm;
            switch (nq - g0 < 4 ? nq - g0 : 4) {
                case 1: hitm = run_group(std::integral_constant<int, 1>{}, g0); break;
                case 2: hitm = run_group(std::integral_constant<int, 2>{}, g0); break;
                case 3: hitm = run_group(std::integral_constant<int, 3>{}, g0); break;
                default: hitm = run_group(std::integral_constant<int, 4>{}, g0); break;
            }
            if (MODE == PSH_MODE_FILTER && hitm) {
#ifdef PSH_TUNING
                if (a.dbg & 32) hitm = 0u;                                    // ablation: the tiles' min trees alone (results invalid)
#endif
                unsigned hmj[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) hmj[j] = ((hitm >> j) & 1u) ? tile_bits(acc[j], g0 + j) : 0u;
#pragma unroll 1
                for (int j = 0; j < 4; ++j) {
                    if (!((hitm >> j) & 1u)) continue;
                    queue_bits(j == 0 ? hmj[0] : (j == 1 ? hmj[1] : (j == 2 ? hmj[2] : hmj[3])), g0 + j);
                }
            }
        }
        wave_lds_fence();  // all lanes done with the arrays before they are overwritten
        u = un;
    }
    if (MODE == PSH_MODE_FILTER) {
        if (qn > 0) verify_queue();
        __syncthreads();
        for (int q = q0 + tid; q < q0 + nq; q += PSH_LQ_THREADS) a.bcount[(int64_t)q * PSH_MAX_BLOCKS + blockIdx.x] = lcount[q - q0];
    }
}

// (W >= 26: nothing in the kernel needs a window longer than the batched 8-bit / f16 scans' bands reach -- W <= 25 is theirs --;
//  batches with 26 <= W <= 33 were a loop of three-query steps until late in round 6)
bool scan_lq_supported(int W, int B, int64_t T) { return W >= 26 && W <= 256 && B >= 2 && B <= PSH_MAX_B_PER_LAUNCH && T < (1ll << 27); }

// queries a chunk takes: what puts its tables in LDS beside the eight waves' buffers
int scan_lq_chunk(int W, int B) {
    const int nks = lq_bucket(W);
    const size_t fixed = (size_t)PSH_LQ_FIXED;
    const size_t room = (size_t)PSH_LDS_BYTES - fixed - (size_t)(PSH_LQ_THREADS / 64) * lq_wave_bytes(nks);
    int qc = (int)(room / (size_t)lq_query_bytes(nks));
    if (qc > PSH_LQ_MAXQ) qc = PSH_LQ_MAXQ;
    if (qc < 1) qc = 1;
    const int chunks = (B + qc - 1) / qc;
    return (B + chunks - 1) / chunks;                       // even chunks
}
size_t scan_lq_shmem_bytes(int W, int B, int q_per_group) {
    const int nks = lq_bucket(W);
    return (size_t)PSH_LQ_FIXED + (size_t)q_per_group * lq_query_bytes(nks) + (size_t)(PSH_LQ_THREADS / 64) * lq_wave_bytes(nks);
}

template <int MODE>
static hipError_t launch_lq_m(const ScanArgs& a, int grid_x, hipStream_t s) {
    const int nks = lq_bucket(a.W);
    const size_t shmem = scan_lq_shmem_bytes(a.W, a.B, a.q_per_group);
    const dim3 grid((unsigned)grid_x, (unsigned)a.n_qgroups);
#define PSH_LQ_LAUNCH(KERNEL)                                                                                                        \
    {                                                                                                                                \
        hipError_t e = hipFuncSetAttribute((const void*)KERNEL, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);             \
        if (e != hipSuccess) return e;                                                                                               \
        hipLaunchKernelGGL(KERNEL, grid, dim3(PSH_LQ_THREADS), shmem, s, a);                                                         \
        return hipGetLastError();                                                                                                    \
    }
    if (nks == 6) PSH_LQ_LAUNCH((scan_lq_kernel<MODE, 6>))
    if (nks == 10) PSH_LQ_LAUNCH((scan_lq_kernel<MODE, 10>))
    if (nks == 14) PSH_LQ_LAUNCH((scan_lq_kernel<MODE, 14>))
    PSH_LQ_LAUNCH((scan_lq_kernel<MODE, 18>))
#undef PSH_LQ_LAUNCH
}

hipError_t launch_scan_lq(const ScanArgs& a, int mode, int grid_x, hipStream_t s) {
    return mode == PSH_MODE_BOOT ? launch_lq_m<PSH_MODE_BOOT>(a, grid_x, s) : launch_lq_m<PSH_MODE_FILTER>(a, grid_x, s);
}

}  // namespace psh
