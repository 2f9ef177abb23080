// psh_scan.hip -- gfx950 (MI355X, CDNA4) kernels of the k-nearest-path scan: the Identity scans (scan_kernel,
// scan_mx_kernel, scan_mq_kernel, scan_mq8_kernel, mq_prep_kernel, boot_mq_kernel) and their launchers.  The embedded / one-window-row scans are in
// psh_embed.hip, thresholding / selection / merge / gather in psh_select.hip, shared device code in psh_device.h.
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
//     scan_mq_kernel / boot_mq_kernel, batches), the same band as an 8-BIT product with a
//     quantisation bound per segment (scan_mq8_kernel: batches of 32 queries and more), or an
//     fp32 correlation + prefix sums on the VALU (scan_kernel, every other window length;
//     PSH_FLAG_FILTER_VALU).
//   * an admission threshold tau (a provable upper bound of the k-th smallest acc: the k-th
//     smallest over ANY subset of the windows bounds the global one) keeps all but ~1e4 of
//     the 1e8 windows out of the candidate lists; it comes from a bootstrap pass over 1/16
//     of the rows.  Survivors are appended to per-block slices with LDS cursors -- no global
//     atomics anywhere (device-scope atomics on one line cost ~25 ns each on this 8-XCD
//     part; 5e4 of them were 5x the whole scan).  A one-block radix select then picks the
//     k best and orders them by (d, r, t).
//   * the embedded scan (embed_scan_kernel) runs the same pipeline behind a linear embedding.
#include "psh_device.h"

namespace psh {

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
        if (MODE == PSH_MODE_FILTER && npend > 0) {   // last iteration's admissions, ahead of the prefetch
            pend_flush(pend, npend, lcount, a, lane);
            npend = 0;
        }
        const unsigned un = grab();
        {   // prefetch the next unit of this wave while this one is computed.  (Spreading
            // these five loads over the arithmetic through a hook was tried: +6 % time --
            // the extra live ranges cost a spill at the 128-VGPR cap.)
            if (un < u_hi) {
                unsigned rsn, rin, sgn, qgn;
                decode(un, rsn, rin, sgn, qgn);
                const int64_t rown = a.row0 + (int64_t)rin * a.row_stride;
                stage_load<ALIGNED>(st, a.dataset + rown * a.T, a.T, (int)sgn * PSH_SEG, nfloat, lane);
            }
        }

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
                approx16<WX>(tile, lane, xv, acc, NY);     // acc[] holds t_i = ny_i - 2 c_i here
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
        u = un;
    }
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
// (f16x8 / f16x4 / f32x16, PSH_MX_SLOTS / NHALF / PEND and mx_half(): psh_device.h -- shared with psh_fused.hip)

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
#define PSH_MQ_THREADS 512
#define PSH_MQ_QCAP 192           // survivors of the cheap test queued per wave before a dense exact pass
// scan_mq_kernel keeps TWO zero-padded f16 copies of a query -- xpad = 7 zeros, the W <= 25 scaled taps times -2, zeros; copy 0
// from half 0, copy 1 from half 1, 20 dwords each -- instead of the 8 shifted copies a [group][K-step][lane] fragment table
// holds: the fragment of lane (query, shift, hk), K-step s is the 8 halves xpad[o + 16 s ..], o = 7 - shift + 8 hk, i.e. dwords
// o / 2 + 8 s .. + 3 of copy (o & 1): two dword-aligned ds_read2_b32.  (One copy and a 2-byte-aligned ds_read_b128 is what
// the compiler would emit and gfx950 serves -- at an eighth of the aligned rate: tools/ubench_lds_unaligned.hip.)  A
// ds_read2_b32 is two dword reads of 32 lanes each on 32 banks: the queries of a group sit PSH_MQ_QDW = 40 dwords apart, a
// query's two copies 20, so that the 4 x 8 lanes of a read -- 4 consecutive dwords of either copy of 4 queries -- fall on 32
// different banks ({0..3, 20..23} + 8 q mod 32).
// 160 bytes per query instead of 512 put PSH_MQS_CHUNK queries beside the waves' tiles: the ensemble is staged, converted
// and its window energies taken once per segment and 256 queries, not once per 112 (that per-segment work was a quarter of
// the kernel: profiles/r04_mq_ablations.txt).
#define PSH_MQ_QDW 40
#define PSH_MQ_CDW 20
#define PSH_MQS_CHUNK 512         // scan_mq_kernel: queries per pass over the ensemble
#define PSH_MQB_CHUNK 512         // boot_mq_kernel (it keeps an fp32 tile of its own for segments beyond f16 range: 156 KB of LDS in all)
// halves of LDS per wave of scan_mq_kernel: the two f16 arrays, or the fp32 tile that takes their place (whichever is larger)
__host__ __device__ inline int mq_wave_halves(int tile_floats) {
    const int t = 2 * tile_floats, h = 2 * PSH_MX_NHALF;
    return ((t > h ? t : h) + 7) & ~7;
}

template <int WT, bool ALIGNED>
__global__ __launch_bounds__(PSH_MQ_THREADS) void scan_mq_kernel(ScanArgs a) {
    static_assert(WT >= 0 && WT <= 25, "query + 7 shifts must fit K = 32 (WT = 0: run-time W <= 25)");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NW = PSH_MQ_THREADS / 64;
    const int lane = lane_id();
    const int wave_in_block = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    int* lcount = reinterpret_cast<int*>(smem);
    int* next_unit = lcount + ((a.B + 3) & ~3);
    u32x4* pend0 = reinterpret_cast<u32x4*>(next_unit + 4);
    u32x4* pend = pend0 + (size_t)wave_in_block * PSH_PEND;
    // A wave's f16 copies of its segment (y^, (y~^2)^) only feed the A fragments and the window energies, which stay in
    // registers for the whole query loop: once those are read, the SAME LDS holds the segment's fp32 values for the exact
    // rechecks (the tile is stored after the fragment reads, not before).  What the tile used to take -- 37 KB of the CU's
    // 160 -- now holds the padded copies of a whole batch of 512 queries.
    const int wave_halves = mq_wave_halves(a.tile_floats);
    _Float16* hbase = reinterpret_cast<_Float16*>(pend0 + (size_t)NW * PSH_PEND);
    _Float16* a1 = hbase + (size_t)wave_in_block * wave_halves;          // y^
    _Float16* a2 = a1 + PSH_MX_NHALF;                                      // (y~^2)^
    float* tile = reinterpret_cast<float*>(a1);                            // (after the segment's fragments are in registers)
    unsigned* fragL = reinterpret_cast<unsigned*>(hbase + (size_t)NW * wave_halves);   // [query of the chunk] x PSH_MQ_QDW dwords
    float* thrL = reinterpret_cast<float*>(fragL + (size_t)PSH_MQS_CHUNK * PSH_MQ_QDW);
    float* tauL = thrL + PSH_MQS_CHUNK;
    unsigned* sq = reinterpret_cast<unsigned*>(tauL + PSH_MQS_CHUNK) + (size_t)wave_in_block * PSH_MQ_QCAP;   // survivor queue
    const int W = WT > 0 ? WT : a.W;
    int npend = 0;
    int nsq = 0;

    const int q0 = (int)blockIdx.y * PSH_MQS_CHUNK;                        // this block's queries: [q0, q0 + nq)
    const int nq = (a.B - q0) < PSH_MQS_CHUNK ? (a.B - q0) : PSH_MQS_CHUNK;
    const int ngroups = (nq + 3) >> 2;
    if (threadIdx.x == 0) *next_unit = 0;
    for (int q = (int)threadIdx.x; q < a.B; q += PSH_MQ_THREADS) lcount[q] = 0;
    {
        unsigned* z = reinterpret_cast<unsigned*>(a1);
        for (int i = lane; i < PSH_MX_NHALF; i += 64) z[i] = 0u;
        // the padded copies of the chunk's queries (PSH_MQ_QDW dwords = ten 16-byte pieces each); queries past the end of
        // the batch in the last group: zeros
        const f32x4* src = reinterpret_cast<const f32x4*>(a.mq_frag) + (size_t)q0 * (PSH_MQ_QDW / 4);
        f32x4* dst = reinterpret_cast<f32x4*>(fragL);
        for (int i = (int)threadIdx.x; i < 4 * ngroups * (PSH_MQ_QDW / 4); i += PSH_MQ_THREADS)
            dst[i] = i < nq * (PSH_MQ_QDW / 4) ? src[i] : f32x4{0.f, 0.f, 0.f, 0.f};
        for (int i = (int)threadIdx.x; i < 4 * ngroups; i += PSH_MQ_THREADS) {
            thrL[i] = i < nq ? a.qstate[q0 + i].mx_thr : -__uint_as_float(PSH_INF_BITS);   // -inf: reject everything
            tauL[i] = i < nq ? __uint_as_float(a.qstate[q0 + i].tau_bits) : 0.0f;
        }
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

#ifdef PSH_TUNING
    const int dbg = __builtin_amdgcn_readfirstlane(a.dbg);
    if ((dbg & 1) && wave_in_block >= 4) __builtin_amdgcn_s_setprio(1);       // the second wave of every SIMD ahead of the first
#endif
    Stage st;
    unsigned u = grab();
    if (u < u_hi) load_unit(st, u);
    while (u < u_hi) {
        const unsigned ri = fast_div(u, a.magic_nseg, (unsigned)a.nseg);
        const unsigned sg = u - ri * (unsigned)a.nseg;
        const int64_t row = a.row0 + (int64_t)ri * a.row_stride;
        const int seg_start = (int)sg * PSH_SEG;
        const int r_global = (int)(row + a.r_offset);

        float lmax = 0.0f, nanq = 0.0f;
        {
            const int nqd = (nfloat + 3) >> 2;
#pragma unroll
            for (int q = 0; q < PSH_NSTAGE; ++q) {
                const int m = lane + 64 * q;
                if (q < PSH_NSTAGE - 1 || m < nqd) {
                    const f32x4 v = st.v[q] * scale;
                    const f32x4 v2 = v * v;
                    lmax = fmaxf(fmaxf(lmax, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
                    nanq += (v2[0] + v2[1]) + (v2[2] + v2[3]);             // (v_max drops a NaN; a sum of squares keeps it)
                    *reinterpret_cast<f16x4*>(a1 + mx_half(4 * m)) = __builtin_convertvector(v, f16x4);
                    *reinterpret_cast<f16x4*>(a2 + mx_half(4 * m)) = __builtin_convertvector(v2, f16x4);
                } else if (4 * m < PSH_MX_NHALF) {
                    // the arrays' tail past the segment: zeros again (the previous segment's fp32 tile lay here, and the
                    // banded product multiplies these slots by its zero taps: 0 x NaN would poison a row)
                    *reinterpret_cast<f16x4*>(a1 + mx_half(4 * m)) = f16x4{0, 0, 0, 0};
                    *reinterpret_cast<f16x4*>(a2 + mx_half(4 * m)) = f16x4{0, 0, 0, 0};
                }
            }
        }
        wave_lds_fence();
        if (npend > 0) { pend_flush(pend, npend, lcount, a, lane); npend = 0; }
        // a value beyond f16 range, a NaN (the banded product spreads it over the 32 outputs of its A row -- 0 x NaN -- and the
        // minimum over a tile drops NaNs: clean windows beside it would be rejected unseen), or no armed filter: nothing may
        // be rejected in this segment
        const bool keep_all = __any(!(lmax <= 128.0f) || !(nanq == nanq)) || !(scale > 0.0f);

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
        // the fragments are in registers: the segment's fp32 values take the arrays' place (for the exact rechecks), and only
        // then is the next unit requested into the staging registers
        wave_lds_fence();
        stage_store(st, tile, nfloat, lane);
        const unsigned un = grab();
        if (un < u_hi) load_unit(st, un);

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
                    const float* xq = a.queries + (int64_t)(q0 + ql2) * W;       // (rare path: the exact query comes from memory)
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

        // (running pointers: the group's two fragments sit 1 KB apart behind one address register, the threshold behind
        //  another -- six VALU instructions of index arithmetic per group were a seventh of the loop's fast path)
        // (the LDS byte address of the lane's first fragment dword: the low half of the flat address of a __shared__ object)
        unsigned frag_addr = (unsigned)(size_t)(fragL + qsub * PSH_MQ_QDW + ((7 - shift) & 1) * PSH_MQ_CDW + ((7 - shift + 8 * hk) >> 1));
        const float* thrp = thrL + qsub;
#pragma unroll 1
        for (int G = 0; G < ngroups; ++G, frag_addr += 4 * PSH_MQ_QDW * 4, thrp += 4) {
            // four ds_read2_b32 off one address register.  By hand: left to itself the compiler folds the adjacent dword-aligned
            // pieces into ONE 4-byte-aligned ds_read_b128, which gfx950 serves at an eighth of the aligned rate
            // (tools/ubench_lds_unaligned.hip); the wait is part of the statement -- the compiler does not count these reads
            u32x2 f00, f01, f10, f11;
            asm volatile("ds_read2_b32 %0, %4 offset1:1\n\t"
                         "ds_read2_b32 %1, %4 offset0:2 offset1:3\n\t"
                         "ds_read2_b32 %2, %4 offset0:8 offset1:9\n\t"
                         "ds_read2_b32 %3, %4 offset0:10 offset1:11\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : "=&v"(f00), "=&v"(f01), "=&v"(f10), "=&v"(f11) : "v"(frag_addr) : "memory");
            const u32x4 w0 = u32x4{f00[0], f00[1], f01[0], f01[1]}, w1 = u32x4{f10[0], f10[1], f11[0], f11[1]};
            f16x8 b0, b1;
            __builtin_memcpy(&b0, &w0, 16);
            __builtin_memcpy(&b1, &w1, 16);
            const float thr = keep_all ? __uint_as_float(PSH_INF_BITS) : *thrp;
            // all 8 MFMAs of the group first (4 independent accumulator tiles), then the tests:
            // a test-and-branch per tile serialises MFMA latency, min tree and branch 4 times
            f32x16 acc[4];
#ifdef PSH_TUNING
            if (dbg & 2) __builtin_amdgcn_s_setprio(2);
#endif
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fy[g][0], b0, ny[g], 0, 0, 0);
#ifdef PSH_TUNING
            if (!(dbg & 32))                                               // ablation: ONE K-step (what an 8-bit product would issue; results invalid)
#endif
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fy[g][1], b1, acc[g], 0, 0, 0);
#ifdef PSH_TUNING
            if (dbg & 2) __builtin_amdgcn_s_setprio(0);
            if (dbg & 8) {                                                 // ablation: no epilogue (results invalid)
                asm volatile("" :: "v"(acc[0][0]), "v"(acc[1][0]), "v"(acc[2][0]), "v"(acc[3][0]));
                continue;
            }
#endif
            float mn[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                mn[g] = tile_min16(acc[g]);
            }
            // values are finite here unless keep_all (then thr = +inf keeps NaN too)
            if (!__any(!(min3f(min3f(mn[0], mn[1], mn[2]), mn[3], mn[3]) > thr))) continue;
#ifdef PSH_TUNING
            if (dbg & 4) continue;                                         // ablation: no survivor handling (results invalid)
#endif
            // survivors are only QUEUED here (window, query): a lane-by-lane exact chain would run ~140
            // instructions for the one or two lanes that hold a survivor; the queue is drained 64 at a time
            const int ql = 4 * G + qsub;                                   // this lane's query within the chunk
            const bool lane_ok = ql < nq;
            const int nsq0 = nsq;
            bool full = false;                                             // wave-uniform
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (!__any(!(mn[g] > thr))) continue;
                // the survivors of this tile as a per-lane bit mask (pure VALU), then one queue round per
                // survivor of the busiest lane (usually one): a ballot per accumulator register would put 16
                // VALU -> SALU round trips on every tile that holds a survivor
                // (a compare and an add-with-carry per accumulator, hm = 2 hm + [!(acc > thr)], from the last register down:
                //  the compiler's select + or3 form is 3 instructions per accumulator.  The accumulators were read by
                //  tile_min16 above and the branch depends on that: no MFMA is in flight on them here.)
                unsigned hm = 0u;
#pragma unroll
                for (int r = 15; r >= 0; --r)
                    asm volatile("v_cmp_ngt_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(hm) : "v"(acc[g][r]), "v"(thr) : "vcc");
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
// the same scan with the rejection test as an 8-BIT product (round 4)
// ----------------------------------------------------------------------------------
// scan_mq_kernel issues two K = 16 steps of f16 per tile for 20 useful taps of 32, and the matrix cores are what bounds it
// (75 % busy at the clock the part sustains under that load).  v_mfma_i32_32x32x32_i8 takes the whole band, K = 32, in ONE
// instruction at the same issue rate: half the MFMAs, half the fragment bytes (4 operand registers per tile and per query
// group instead of 8), and an integer accumulator whose only error is the quantisation itself -- which is bounded
// rigorously (threshold_kernel, "scan_mq8_kernel"): the data of a segment on its own step s_y = max|y~| / 127, the queries
// on one step for the batch, the window energies (f16 squares, as before) turned into the MFMA's integer C operand, and a
// window is rejected iff  C_w - sum x^ y^  >  P_q / s_y + L_q.  The margin that buys -- ~9 % of the level at the benchmark's
// sizes, 2.2x the survivors of the f16 test -- costs less than the MFMAs it saves.
// The queries' copies: 4 per query, shifted by 0..3 BYTES, 10 dwords each (PSH_MQ8_CDW), queries PSH_MQ8_QDW = 40 dwords
// apart: lane (query, shift, hk) reads dwords (o >> 2) + 4 hk .. + 3 of copy o & 3, o = 7 - shift, as two ds_read2_b32; the
// 32 lanes of a pass fall on banks {0, 1, 10, 11, 20, 21, 30, 31} + 8 q mod 32 -- all different.
#define PSH_MQ8_QDW 40
#define PSH_MQ8_CDW 10
typedef int i32x4v __attribute__((ext_vector_type(4)));
typedef int i32x16v __attribute__((ext_vector_type(16)));
__device__ __forceinline__ int imin3(int a, int b, int c) { const int m = a < b ? a : b; return m < c ? m : c; }   // v_min3_i32
__device__ __forceinline__ int tile_min16_i(const i32x16v& t) {
    const int m0 = imin3(t[0], t[1], t[2]), m1 = imin3(t[3], t[4], t[5]), m2 = imin3(t[6], t[7], t[8]);
    const int m3 = imin3(t[9], t[10], t[11]), m4 = imin3(t[12], t[13], t[14]);
    return imin3(imin3(m0, m1, m2), imin3(m3, m4, t[15]), 0x7fffffff);
}

template <int WT, bool ALIGNED>
__global__ __launch_bounds__(PSH_MQ_THREADS) void scan_mq8_kernel(ScanArgs a) {
    static_assert(WT >= 0 && WT <= 25, "query + 7 shifts must fit K = 32 (WT = 0: run-time W <= 25)");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NW = PSH_MQ_THREADS / 64;
    const int lane = lane_id();
    const int wave_in_block = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    int* lcount = reinterpret_cast<int*>(smem);
    int* next_unit = lcount + ((a.B + 3) & ~3);                           // [0] the work cursor, [1] the batch's k1 (float bits)
    u32x4* pend0 = reinterpret_cast<u32x4*>(next_unit + 4);
    u32x4* pend = pend0 + (size_t)wave_in_block * PSH_PEND;
    // a wave's LDS: the segment's signed bytes y^ (PSH_MX_NHALF of them) and the f16 squares behind them, for as long as the
    // A fragments and the energies are being read; the segment's fp32 values for the exact rechecks afterwards (as in
    // scan_mq_kernel: the tile is stored after the fragment reads)
    const int wave_halves = mq_wave_halves(a.tile_floats);
    _Float16* hbase = reinterpret_cast<_Float16*>(pend0 + (size_t)NW * PSH_PEND);
    _Float16* a1 = hbase + (size_t)wave_in_block * wave_halves;
    signed char* a8 = reinterpret_cast<signed char*>(a1);                  // y^
    _Float16* a2 = a1 + PSH_MX_NHALF;                                      // (y~^2)^
    float* tile = reinterpret_cast<float*>(a1);
    unsigned* fragL = reinterpret_cast<unsigned*>(hbase + (size_t)NW * wave_halves);   // [query of the chunk] x PSH_MQ8_QDW dwords
    f32x2* plL = reinterpret_cast<f32x2*>(fragL + (size_t)PSH_MQS_CHUNK * PSH_MQ8_QDW);     // {P, L} per query
    float* tauL = reinterpret_cast<float*>(plL + PSH_MQS_CHUNK);
    unsigned* sq = reinterpret_cast<unsigned*>(tauL + PSH_MQS_CHUNK) + (size_t)wave_in_block * PSH_MQ_QCAP;   // survivor queue
    int* thrI = reinterpret_cast<int*>(reinterpret_cast<unsigned*>(tauL + PSH_MQS_CHUNK) + (size_t)NW * PSH_MQ_QCAP)
                + (size_t)wave_in_block * PSH_MQS_CHUNK;                  // the wave's segment: every query's level in the product's units
    const int W = WT > 0 ? WT : a.W;
    int npend = 0;
    int nsq = 0;

    const int q0 = (int)blockIdx.y * PSH_MQS_CHUNK;                        // this block's queries: [q0, q0 + nq)
    const int nq = (a.B - q0) < PSH_MQS_CHUNK ? (a.B - q0) : PSH_MQS_CHUNK;
    const int ngroups = (nq + 3) >> 2;
    if (threadIdx.x == 0) { next_unit[0] = 0; next_unit[1] = 0; }
    for (int q = (int)threadIdx.x; q < a.B; q += PSH_MQ_THREADS) lcount[q] = 0;
    __syncthreads();
    {
        unsigned* z = reinterpret_cast<unsigned*>(a1);
        for (int i = lane; i < PSH_MX_NHALF; i += 64) z[i] = 0u;
        const f32x4* src = reinterpret_cast<const f32x4*>(a.mq_frag) + (size_t)q0 * (PSH_MQ8_QDW / 4);
        f32x4* dst = reinterpret_cast<f32x4*>(fragL);
        for (int i = (int)threadIdx.x; i < 4 * ngroups * (PSH_MQ8_QDW / 4); i += PSH_MQ_THREADS)
            dst[i] = i < nq * (PSH_MQ8_QDW / 4) ? src[i] : f32x4{0.f, 0.f, 0.f, 0.f};
        for (int i = (int)threadIdx.x; i < 4 * ngroups; i += PSH_MQ_THREADS) {
            // a query whose test is not armed (no level yet, a non-finite sample): P = +inf keeps everything; so do the
            // places past the end of the batch in the last group -- their lanes never queue anything (lane_ok)
            const float k1q = i < nq ? a.qstate[q0 + i].mx8_k1 : 0.0f;
            f32x2 plq = f32x2{__uint_as_float(PSH_INF_BITS), 0.0f};
            if (k1q > 0.0f) plq = f32x2{a.qstate[q0 + i].mx8_P, a.qstate[q0 + i].mx8_L};
            plL[i] = plq;
            tauL[i] = i < nq ? __uint_as_float(a.qstate[q0 + i].tau_bits) : 0.0f;
            if (k1q > 0.0f) atomicMax(reinterpret_cast<unsigned*>(next_unit + 1), __float_as_uint(k1q));   // (the armed queries all hold the same value)
        }
    }
    __syncthreads();

    const int nfloat = PSH_SEG + W - 1;
    const unsigned n_rs = (unsigned)a.n_rows * (unsigned)a.nseg;
    const unsigned u_lo = (unsigned)(((unsigned long long)n_rs * blockIdx.x) / gridDim.x);
    const unsigned u_hi = (unsigned)(((unsigned long long)n_rs * (blockIdx.x + 1)) / gridDim.x);
    typedef const __attribute__((address_space(4))) QueryState* const_qsp;
    const float scale = ((const_qsp)a.qstate)[0].mx_scale;                // one f16 / 8-bit scale domain for the whole batch
    const float k1 = __uint_as_float((unsigned)__builtin_amdgcn_readfirstlane(next_unit[1]));
    const int n = lane & 31, hk = lane >> 5, qsub = n >> 3, shift = n & 7;

    f16x8 bo[2];                                                           // the band of ones (window energies)
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

#ifdef PSH_TUNING
    const int dbg = __builtin_amdgcn_readfirstlane(a.dbg);
    unsigned long long tacc[3] = {0ull, 0ull, 0ull}, tlast = __builtin_amdgcn_s_memtime();   // per segment: set-up / group loop / drain + hand-over
    auto tstamp = [&](int i) { const unsigned long long t = __builtin_amdgcn_s_memtime(); tacc[i] += t - tlast; tlast = t; };
#else
    auto tstamp = [&](int) {};
#endif
    Stage st;
    unsigned u = grab();
    if (u < u_hi) load_unit(st, u);
    while (u < u_hi) {
        const unsigned ri = fast_div(u, a.magic_nseg, (unsigned)a.nseg);
        const unsigned sg = u - ri * (unsigned)a.nseg;
        const int64_t row = a.row0 + (int64_t)ri * a.row_stride;
        const int seg_start = (int)sg * PSH_SEG;
        const int r_global = (int)(row + a.r_offset);

        // pass 1 over the staged segment: the f16 squares, the largest |y~|, a sum that keeps a NaN
        float lmax = 0.0f, nanq = 0.0f;
        const int nqd = (nfloat + 3) >> 2;
#pragma unroll
        for (int q = 0; q < PSH_NSTAGE; ++q) {
            const int m = lane + 64 * q;
            if (q < PSH_NSTAGE - 1 || m < nqd) {
                const f32x4 v = st.v[q] * scale;
                const f32x4 v2 = v * v;
                lmax = fmaxf(fmaxf(lmax, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
                nanq += (v2[0] + v2[1]) + (v2[2] + v2[3]);
                *reinterpret_cast<f16x4*>(a2 + mx_half(4 * m)) = __builtin_convertvector(v2, f16x4);
            } else if (4 * m < PSH_MX_NHALF) {
                *reinterpret_cast<f16x4*>(a2 + mx_half(4 * m)) = f16x4{0, 0, 0, 0};   // (the previous segment's fp32 tile lay here: 0 x NaN)
            }
        }
        const float lm = wave_max_nonneg(lmax);                            // (per-lane maxima are never NaN: fmaxf drops it)
        // nothing may be rejected in this segment: a value beyond the f16 range of the squares, a NaN, no armed filter
        const bool keep_all = !(lm <= 128.0f) || __any(!(nanq == nanq)) || !(scale > 0.0f) || !(k1 > 0.0f);
        // the segment's step: s_y = max(|y~|, 2^-6) / 127 (a floor keeps the f16 squares' subnormal tail below one unit)
        const float inv_sy = 127.0f / fmaxf(lm, 0.015625f);
        // pass 2: y^ = round-to-nearest-even of y~ / s_y through the 1.5 * 2^23 trick (the fma rounds the EXACT product to an
        // integer: |y^ - y~ / s_y| <= 1/2, what the bound assumes), four to a dword
#pragma unroll
        for (int q = 0; q < PSH_NSTAGE; ++q) {
            const int m = lane + 64 * q;
            if (q < PSH_NSTAGE - 1 || m < nqd) {
                const f32x4 v = st.v[q] * scale;
                const unsigned r0 = __float_as_uint(__builtin_fmaf(v[0], inv_sy, 12582912.0f)), r1 = __float_as_uint(__builtin_fmaf(v[1], inv_sy, 12582912.0f));
                const unsigned r2 = __float_as_uint(__builtin_fmaf(v[2], inv_sy, 12582912.0f)), r3 = __float_as_uint(__builtin_fmaf(v[3], inv_sy, 12582912.0f));
                const unsigned w01 = __builtin_amdgcn_perm(r1, r0, 0x0c0c0400u), w23 = __builtin_amdgcn_perm(r3, r2, 0x04000c0cu);
                *reinterpret_cast<unsigned*>(a8 + 4 * m) = w01 | w23;
            } else if (4 * m < PSH_MX_NHALF) {
                *reinterpret_cast<unsigned*>(a8 + 4 * m) = 0u;
            }
        }
        // every query's level for THIS segment's step, in the product's units, rounded towards "keep" by the constants'
        // margins; the clamp keeps the conversion inside 32 bits (beyond it: reject nothing / everything, as it should).
        // Once per segment and query here instead of once per group and lane in the loop.
        for (int i = lane; i < 4 * ngroups; i += 64) {
            const f32x2 pl = plL[i];
            thrI[i] = keep_all ? 0x7fffffff
                               : (int)__builtin_amdgcn_fmed3f(__builtin_fmaf(pl[0], inv_sy, pl[1]), -2147483520.0f, 2147483520.0f);
        }
        wave_lds_fence();
        if (npend > 0) { pend_flush(pend, npend, lcount, a, lane); npend = 0; }

        // window energies of the 4 row groups -> the integer C operand, and the y^ fragments, once per segment
        i32x16v cw[4];
        i32x4v fy[4];
        {
            // C_w = floor(ny kC) <= 2^30 (ny <= 32 * 128^2 where anything may be rejected: the product stays in 32 bits)
            const float kC = fminf(inv_sy * k1, 2048.0f);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f16x8 e0 = *reinterpret_cast<const f16x8*>(a2 + mx_half(256 * g + 8 * n + 8 * hk));
                const f16x8 e1 = *reinterpret_cast<const f16x8*>(a2 + mx_half(256 * g + 8 * n + 16 + 8 * hk));
                const u32x2 y0 = *reinterpret_cast<const u32x2*>(a8 + 256 * g + 8 * n + 16 * hk);
                const u32x2 y1 = *reinterpret_cast<const u32x2*>(a8 + 256 * g + 8 * n + 16 * hk + 8);
                fy[g] = i32x4v{(int)y0[0], (int)y0[1], (int)y1[0], (int)y1[1]};
                f32x16 ny;
#pragma unroll
                for (int i = 0; i < 16; ++i) ny[i] = 0.0f;
                ny = __builtin_amdgcn_mfma_f32_32x32x16_f16(e0, bo[0], ny, 0, 0, 0);
                ny = __builtin_amdgcn_mfma_f32_32x32x16_f16(e1, bo[1], ny, 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 16; ++i) cw[g][i] = (int)fminf(ny[i] * kC, 1073741824.0f);   // (ny >= 0; the clamp only matters under keep_all: inf, NaN)
            }
        }
        wave_lds_fence();
        stage_store(st, tile, nfloat, lane);
        const unsigned un = grab();
        if (un < u_hi) load_unit(st, un);

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
                    const float* xq = a.queries + (int64_t)(q0 + ql2) * W;       // (rare path: the exact query comes from memory)
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

        // The group loop.  The next group's fragment and level are requested right behind this group's MFMAs and waited for
        // at the top of the next turn: an LDS round trip per group would otherwise sit between a wave's tests and its next
        // MFMAs.  (The loop software-pipelined in halves of a group -- the MFMAs of tiles 2, 3 issued before the tests of
        // tiles 0, 1 and so on, same 64 accumulators -- was built and measured: 3.0 ms per step against 2.85.)
        tstamp(0);
        const int o = 7 - shift;
        unsigned frag_addr = (unsigned)(size_t)(fragL + qsub * PSH_MQ8_QDW + (o & 3) * PSH_MQ8_CDW + (o >> 2) + 4 * hk);
        const int* thrp = thrI + qsub;
        // (fragment reads by hand, as in scan_mq_kernel: the compiler would fold the adjacent dwords into one 4-byte-aligned
        //  ds_read_b128; the wait is a statement of its own -- tied to the registers, so that nothing reads them before it)
        u32x2 f0, f1;
        asm volatile("ds_read2_b32 %0, %2 offset1:1\n\tds_read2_b32 %1, %2 offset0:2 offset1:3"
                     : "=&v"(f0), "=&v"(f1) : "v"(frag_addr) : "memory");
        int thr_next = *thrp;
#pragma unroll 1
        for (int G = 0; G < ngroups; ++G) {
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f0), "+v"(f1) :: "memory");
            const i32x4v bq = i32x4v{(int)f0[0], (int)f0[1], (int)f1[0], (int)f1[1]};
            const int thr = thr_next;
            i32x16v acc[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fy[g], bq, cw[g], 0, 0, 0);
            frag_addr += 4 * PSH_MQ8_QDW * 4;
            thrp += 4;
            if (G + 1 < ngroups) {
                asm volatile("ds_read2_b32 %0, %2 offset1:1\n\tds_read2_b32 %1, %2 offset0:2 offset1:3"
                             : "=&v"(f0), "=&v"(f1) : "v"(frag_addr) : "memory");
                thr_next = *thrp;
            }
#ifdef PSH_TUNING
            if (dbg & 8) {                                                 // ablation: no epilogue (results invalid)
                asm volatile("" :: "v"(acc[0][0]), "v"(acc[1][0]), "v"(acc[2][0]), "v"(acc[3][0]));
                continue;
            }
#endif
            int mn[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) mn[g] = tile_min16_i(acc[g]);
            if (!__any(!(imin3(imin3(mn[0], mn[1], mn[2]), mn[3], mn[3]) > thr))) continue;
#ifdef PSH_TUNING
            if (dbg & 4) continue;                                         // ablation: no survivor handling (results invalid)
#endif
            // survivors are only QUEUED here (window, query); the queue is drained 64 at a time (scan_mq_kernel)
            const int ql = 4 * G + qsub;                                   // this lane's query within the chunk
            const bool lane_ok = ql < nq;
            const int nsq0 = nsq;
            bool full = false;                                             // wave-uniform
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (!__any(!(mn[g] > thr))) continue;
                unsigned hm = 0u;                                          // hm = 2 hm + [acc <= thr], from the last register down
#pragma unroll
                for (int r = 15; r >= 0; --r)
                    asm volatile("v_cmp_le_i32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(hm) : "v"(acc[g][r]), "v"(thr) : "vcc");
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
                // more survivors in one group than the queue holds: forget the group's entries and run its queries exactly
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
            if (nsq >= 64) drain();
        }
        tstamp(1);
        if (nsq > 0) drain();
        wave_lds_fence();  // all lanes done with the tile before it is overwritten
        u = un;
        tstamp(2);
    }
#ifdef PSH_TUNING
    if (a.dbg_times && lane == 0)
        for (int i = 0; i < 3; ++i) a.dbg_times[((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * NW + wave_in_block) * 3 + i] = tacc[i];
#endif
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
// What every block of the bootstrap needs of the batch -- the scale, every query's two padded f16 copies (the scan's layout,
// PSH_MQ_QDW dwords a query) and nx~ -- written ONCE by one block (round 3: every block of every query chunk rebuilt its
// chunk's 57 KB fragment table from the queries: most of the bootstrap's 0.25 ms at 512 queries).  Layout behind `mq`
// (the workspace's 512 x B4 bytes, B4 = B rounded up to 4; the threshold kernel later writes the SCAN's copies, at its own
// scale, at offset 0): boot copies at 192 B4 bytes, nx~ at 384 B4, {scale, 1 / scale^2} at 392 B4.
// (PSH_MQ_BOOT_OFF / PSH_MQ_NX_OFF / PSH_MQ_META_OFF: psh_kernels.h -- the threshold kernel reads the meta words too)
#define PSH_MQ_PREP_Q 16          // queries per block of mq_prep_kernel (every block finds the batch's scale for itself)
__global__ __launch_bounds__(1024) void mq_prep_kernel(const float* __restrict__ queries, int B, int W, void* mq) {
    __shared__ unsigned s_max;
    const int tid = (int)threadIdx.x;
    if (tid == 0) s_max = 0u;
    __syncthreads();
    unsigned mb = 0u;                                       // the largest |x| of the whole batch into [4, 8)
    for (int64_t j = tid; j < (int64_t)B * W; j += 1024) mb = max(mb, __float_as_uint(fabsf(queries[j])));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mb = max(mb, (unsigned)__shfl_xor((int)mb, off, 64));
    if ((tid & 63) == 0 && mb) atomicMax(&s_max, mb);
    __syncthreads();
    const unsigned qmaxbits = s_max;
    const int sexp = 3 - ((int)((qmaxbits >> 23) & 255u) - 126);
    const bool sane = sexp <= 60 && sexp >= -60 && qmaxbits >= 0x00800000u && qmaxbits < PSH_INF_BITS;
    const float scale = sane ? __uint_as_float((unsigned)(127 + sexp) << 23) : 0.0f;     // 0: exact chain everywhere
    const float unscale2 = sane ? __uint_as_float((unsigned)(127 - 2 * sexp) << 23) : 0.0f;
    const int B4 = (B + 3) & ~3;
    char* base = reinterpret_cast<char*>(mq);
    _Float16* tab = reinterpret_cast<_Float16*>(base + PSH_MQ_BOOT_OFF(B4));
    float* nx = reinterpret_cast<float*>(base + PSH_MQ_NX_OFF(B4));
    const int b_lo = (int)blockIdx.x * PSH_MQ_PREP_Q, b_hi = (b_lo + PSH_MQ_PREP_Q) < B4 ? (b_lo + PSH_MQ_PREP_Q) : B4;
    for (int64_t i = (int64_t)b_lo * 2 * PSH_MQ_QDW + tid; i < (int64_t)b_hi * 2 * PSH_MQ_QDW; i += 1024) {
        const int b = (int)(i / (2 * PSH_MQ_QDW)), e = (int)(i - (int64_t)b * 2 * PSH_MQ_QDW);
        const int half = e & 1, dw = e >> 1, c = dw >= PSH_MQ_CDW ? 1 : 0, d = dw - PSH_MQ_CDW * c;
        const int j = 2 * d + c + half - 7;
        const bool in = b < B && j >= 0 && j < W;
        const float xv = in ? queries[(int64_t)b * W + j] : 0.0f;
        tab[i] = (_Float16)(in ? -2.0f * (xv * scale) : 0.0f);
    }
    for (int b = b_lo + tid; b < b_hi; b += 1024) {
        float sq = 0.0f;
        if (b < B)
            for (int j = 0; j < W; ++j) { const float v = queries[(int64_t)b * W + j] * scale; sq = __builtin_fmaf(v, v, sq); }
        nx[b] = sq;
    }
    if (tid == 0 && blockIdx.x == 0) {
        float* meta = reinterpret_cast<float*>(base + PSH_MQ_META_OFF(B4));
        meta[0] = scale;
        meta[1] = unscale2;
    }
    if (blockIdx.x == gridDim.x - 1) {
        // the 8-bit test's batch constants (threshold_kernel, "scan_mq8_kernel"), once per batch instead of once per query's
        // block: with the step s0 = max|x| / 127 and x^ = round(x / s0), the largest ||x - s0 x^||^2 and the largest ||x||^2 of
        // the batch (UNSCALED: the power-of-two scale the threshold kernel settles on later multiplies both exactly)
        __shared__ unsigned s_e2, s_nx;
        if (tid == 0) { s_e2 = 0u; s_nx = 0u; }
        __syncthreads();
        const float xmax = __uint_as_float(qmaxbits);
        const bool ok = qmaxbits > 0u && qmaxbits < PSH_INF_BITS;
        const float inv_s0 = ok ? 127.0f / xmax : 0.0f;
        const double s0 = ok ? 1.0 / (double)inv_s0 : 0.0;
        for (int q = tid; q < B; q += 1024) {
            float xv[25];
#pragma unroll
            for (int j = 0; j < 25; ++j) xv[j] = j < W ? queries[(int64_t)q * W + j] : 0.0f;
            double e2 = 0.0, nxq = 0.0;
#pragma unroll
            for (int j = 0; j < 25; ++j) {
                const double r = (double)xv[j] - s0 * (double)mq8_quant(xv[j], inv_s0);
                e2 += r * r;
                nxq += (double)xv[j] * (double)xv[j];
            }
            float e2f = (float)e2, nxf = (float)nxq;
            if ((double)e2f < e2) e2f = __uint_as_float(__float_as_uint(e2f) + 1u);          // rounded up (non-negative)
            if ((double)nxf < nxq) nxf = __uint_as_float(__float_as_uint(nxf) + 1u);
            if (ok && e2f == e2f) atomicMax(&s_e2, __float_as_uint(e2f));
            if (ok && nxf == nxf) atomicMax(&s_nx, __float_as_uint(nxf));
        }
        __syncthreads();
        if (tid == 0) {
            unsigned* meta = reinterpret_cast<unsigned*>(base + PSH_MQ_META_OFF(B4));
            meta[2] = ok ? s_e2 : 0u;
            meta[3] = ok ? s_nx : 0u;
            meta[4] = ok ? qmaxbits : 0u;
        }
    }
}

hipError_t launch_mq_prep(const float* queries, int B, int W, void* mq, hipStream_t s) {
    const int B4 = (B + 3) & ~3;
    hipLaunchKernelGGL(mq_prep_kernel, dim3((unsigned)((B4 + PSH_MQ_PREP_Q - 1) / PSH_MQ_PREP_Q)), dim3(1024), 0, s, queries, B, W, mq);
    return hipGetLastError();
}

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
    unsigned* fragL = reinterpret_cast<unsigned*>(hbase + (size_t)NW * 2 * PSH_MX_NHALF);   // [query of the chunk] x PSH_MQ_QDW dwords
    float* nxL = reinterpret_cast<float*>(fragL + (size_t)PSH_MQB_CHUNK * PSH_MQ_QDW);       // nx~ per query of the chunk

    const int W = WT > 0 ? WT : a.W;
    const int q0 = (int)blockIdx.y * PSH_MQB_CHUNK;
    const int nq = (a.B - q0) < PSH_MQB_CHUNK ? (a.B - q0) : PSH_MQB_CHUNK;
    const int ngroups = (nq + 3) >> 2;
    if (threadIdx.x == 0) { next_unit[0] = 0; next_unit[1] = 0; next_unit[2] = 0; }
    {
        unsigned* z = reinterpret_cast<unsigned*>(a1);
        for (int i = lane; i < PSH_MX_NHALF; i += 64) z[i] = 0u;
    }
    // the batch's scale, this chunk's padded query copies and nx~: written once by mq_prep_kernel (B4 is a multiple of 4:
    // the last group's queries past the end of the batch are zeros there)
    const int B4 = (a.B + 3) & ~3;
    const char* mqb = reinterpret_cast<const char*>(a.mq_frag);
    const float scale = reinterpret_cast<const float*>(mqb + PSH_MQ_META_OFF(B4))[0];
    const float unscale2 = reinterpret_cast<const float*>(mqb + PSH_MQ_META_OFF(B4))[1];
    {
        const f32x4* src = reinterpret_cast<const f32x4*>(mqb + PSH_MQ_BOOT_OFF(B4)) + (size_t)q0 * (PSH_MQ_QDW / 4);
        f32x4* dst = reinterpret_cast<f32x4*>(fragL);
        for (int i = (int)threadIdx.x; i < 4 * ngroups * (PSH_MQ_QDW / 4); i += PSH_MQ_THREADS) dst[i] = src[i];
        const float* nxs = reinterpret_cast<const float*>(mqb + PSH_MQ_NX_OFF(B4)) + q0;
        for (int i = (int)threadIdx.x; i < 4 * ngroups; i += PSH_MQ_THREADS) nxL[i] = nxs[i];
    }
    __syncthreads();

    const int nfloat = PSH_SEG + W - 1;
    const unsigned n_rs = (unsigned)a.n_rows * (unsigned)a.nseg;
    const unsigned u_lo = (unsigned)(((unsigned long long)n_rs * blockIdx.x) / gridDim.x);
    const unsigned u_hi = (unsigned)(((unsigned long long)n_rs * (blockIdx.x + 1)) / gridDim.x);
    const int n = lane & 31, hk = lane >> 5, qsub = n >> 3, shift = n & 7;
    const const_f32p xk = (const_f32p)a.queries;
    // acc~ <= (nx~ (1 + 3a) + t^ + b) / (1 - 2a), a = 2^-9, b = 2^-18; constants rounded up, fp32 slack included
    // When the caller admits below an ESTIMATE anyway (a.boot_estimate: the r-th smallest minimum of a thin sample, a shortfall
    // reported by the selection -- every large batch does), a minimum need not be an upper bound either: the value itself,
    // nx~ + t^, is the better estimate of the segment's smallest acc -- the bound's 5a nx~ are ~5 % of an acc near the level,
    // and the level's 10th power counts the candidates (4.4 k per query with bounds, 2.7 k with values, for k = 1024).
    const bool est = a.boot_estimate != 0;
    const float C1 = est ? 1.0f : 1.0f + 3.0f / 512.0f + 1.0f / 65536.0f;
    const float C2 = est ? 1.0f : (1.0f / (1.0f - 2.0f / 512.0f)) * (1.0f + 1.0f / 32768.0f);
    const float BB = est ? 0.0f : 1.0f / 262144.0f;

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
            if (ragged) {
                // windows past the row's last admissible one: their ENERGY becomes +inf, once per unit -- it is the C operand of
                // every group's MFMAs, so their accumulators are +inf in every group and no minimum sees them (masking the 64
                // accumulators group after group made the last segment of every row four times as long as the others: the
                // bootstrap's 179 us per launch at 512 queries were its ragged units)
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int p = 256 * g + 8 * ((r & 3) + 8 * (r >> 2) + 4 * hk) + shift;
                        ny[g][r] = (seg_start + p < a.Tp) ? ny[g][r] : __uint_as_float(PSH_INF_BITS);
                    }
            }
            unsigned frag_addr = (unsigned)(size_t)(fragL + qsub * PSH_MQ_QDW + ((7 - shift) & 1) * PSH_MQ_CDW + ((7 - shift + 8 * hk) >> 1));
#pragma unroll 1
            for (int G = 0; G < ngroups; ++G, frag_addr += 4 * PSH_MQ_QDW * 4) {
                // (the scan's fragment reads: four ds_read2_b32 off one address register, see scan_mq_kernel)
                u32x2 f00, f01, f10, f11;
                asm volatile("ds_read2_b32 %0, %4 offset1:1\n\t"
                             "ds_read2_b32 %1, %4 offset0:2 offset1:3\n\t"
                             "ds_read2_b32 %2, %4 offset0:8 offset1:9\n\t"
                             "ds_read2_b32 %3, %4 offset0:10 offset1:11\n\t"
                             "s_waitcnt lgkmcnt(0)"
                             : "=&v"(f00), "=&v"(f01), "=&v"(f10), "=&v"(f11) : "v"(frag_addr) : "memory");
                const u32x4 w0 = u32x4{f00[0], f00[1], f01[0], f01[1]}, w1 = u32x4{f10[0], f10[1], f11[0], f11[1]};
                f16x8 b0, b1;
                __builtin_memcpy(&b0, &w0, 16);
                __builtin_memcpy(&b1, &w1, 16);
                const int ql = 4 * G + qsub;
                float mn = __uint_as_float(PSH_INF_BITS);
                f32x16 acc[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fy[g][0], b0, ny[g], 0, 0, 0);
#pragma unroll
                for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fy[g][1], b1, acc[g], 0, 0, 0);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float m2 = fminf(fminf(acc[g][0], acc[g][1]), acc[g][2]);
#pragma unroll
                    for (int i = 3; i + 1 < 16; i += 2) m2 = fminf(fminf(m2, acc[g][i]), acc[g][i + 1]);
                    mn = fminf(mn, fminf(m2, acc[g][15]));
                }
                // lanes of one query: 8 shifts x 2 halves.  On the DPP path of the vector ALUs (quad_perm [1,0,3,2], [2,3,0,1],
                // row_half_mirror: lane i <-> 7 - i of its 8) and one v_permlane32_swap for the halves: four dependent
                // ds_bpermute round trips per group were most of this kernel's time (179 us per launch at 512 queries)
                mn = fminf(mn, __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(mn), 0xB1, 0xf, 0xf, false)));
                mn = fminf(mn, __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(mn), 0x4E, 0xf, 0xf, false)));
                mn = fminf(mn, __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(mn), 0x141, 0xf, 0xf, false)));
                {
                    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mn), __float_as_uint(mn), false, false);
                    mn = fminf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));      // lane l: min of lanes l % 32 and l % 32 + 32
                }
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
    return (size_t)(((B + 3) & ~3) + 4) * sizeof(int)
           + (size_t)NW * PSH_PEND * 16 + (size_t)NW * mq_wave_halves(tile_floats) * sizeof(_Float16)
           + (size_t)PSH_MQS_CHUNK * PSH_MQ_QDW * sizeof(unsigned) + (size_t)3 * PSH_MQS_CHUNK * sizeof(float)   // (8-bit test: {P, L} per query)
           + (size_t)NW * PSH_MQ_QCAP * sizeof(unsigned) + (size_t)NW * PSH_MQS_CHUNK * sizeof(int);   // (... and a wave's levels for its segment)
}

int scan_mq_chunks(int B) { return (B + PSH_MQS_CHUNK - 1) / PSH_MQS_CHUNK; }
int boot_mq_chunks(int B) { return (B + PSH_MQB_CHUNK - 1) / PSH_MQB_CHUNK; }

bool boot_mq_supported(int W) { return W >= 1 && W <= 25; }

size_t boot_mq_shmem_bytes(int tile_floats) {
    constexpr int NW = PSH_MQ_THREADS / 64;
    return (size_t)tile_floats * NW * sizeof(float) + 16 + (size_t)NW * 2 * PSH_MX_NHALF * sizeof(_Float16)
           + (size_t)PSH_MQB_CHUNK * PSH_MQ_QDW * sizeof(unsigned) + (size_t)PSH_MQB_CHUNK * sizeof(float);
}

hipError_t launch_boot_mq(const ScanArgs& a, bool aligned, int grid_x, hipStream_t s) {
    const size_t shmem = boot_mq_shmem_bytes(a.tile_floats);
    const dim3 grid(grid_x, boot_mq_chunks(a.B));
    if (a.W == 20)
        return aligned ? launch_big_lds(boot_mq_kernel<20, true>, grid, PSH_MQ_THREADS, shmem, s, a)
                       : launch_big_lds(boot_mq_kernel<20, false>, grid, PSH_MQ_THREADS, shmem, s, a);
    return aligned ? launch_big_lds(boot_mq_kernel<0, true>, grid, PSH_MQ_THREADS, shmem, s, a)
                   : launch_big_lds(boot_mq_kernel<0, false>, grid, PSH_MQ_THREADS, shmem, s, a);
}

hipError_t launch_scan_mq(const ScanArgs& a, bool aligned, int grid_x, hipStream_t s) {
    const size_t shmem = scan_mq_shmem_bytes(a.tile_floats, a.B);
    const dim3 grid(grid_x, scan_mq_chunks(a.B));
    if (a.mq_i8) {
        if (a.W == 20)
            return aligned ? launch_big_lds(scan_mq8_kernel<20, true>, grid, PSH_MQ_THREADS, shmem, s, a)
                           : launch_big_lds(scan_mq8_kernel<20, false>, grid, PSH_MQ_THREADS, shmem, s, a);
        return aligned ? launch_big_lds(scan_mq8_kernel<0, true>, grid, PSH_MQ_THREADS, shmem, s, a)
                       : launch_big_lds(scan_mq8_kernel<0, false>, grid, PSH_MQ_THREADS, shmem, s, a);
    }
    if (a.W == 20)
        return aligned ? launch_big_lds(scan_mq_kernel<20, true>, grid, PSH_MQ_THREADS, shmem, s, a)
                       : launch_big_lds(scan_mq_kernel<20, false>, grid, PSH_MQ_THREADS, shmem, s, a);
    return aligned ? launch_big_lds(scan_mq_kernel<0, true>, grid, PSH_MQ_THREADS, shmem, s, a)
                   : launch_big_lds(scan_mq_kernel<0, false>, grid, PSH_MQ_THREADS, shmem, s, a);
}

hipError_t launch_scan(const ScanArgs& a, int mode, bool aligned, int grid, hipStream_t s) {
    if (a.ker) return launch_embed_scan(a, mode, aligned, grid, s);           // psh_embed.hip
    const size_t shmem = scan_shmem_bytes(a.tile_floats, a.B, 0, a.W, PSH_SCAN_THREADS);
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
    if (embedded) return embed_blocks_per_cu(aligned, shmem, out);              // psh_embed.hip
    int n = 0;
    hipError_t e;
    if (W == 20) {
        e = aligned ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, scan_kernel<20, true, PSH_MODE_FILTER>, PSH_SCAN_THREADS, shmem)
                    : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, scan_kernel<20, false, PSH_MODE_FILTER>, PSH_SCAN_THREADS, shmem);
    } else {
        e = aligned ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, scan_kernel<0, true, PSH_MODE_FILTER>, PSH_SCAN_THREADS, shmem)
                    : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, scan_kernel<0, false, PSH_MODE_FILTER>, PSH_SCAN_THREADS, shmem);
    }
    *out = n;
    return e;
}

}  // namespace psh
