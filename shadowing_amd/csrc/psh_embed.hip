// psh_embed.hip -- the scans behind a linear embedding (embed_scan_kernel: dense fma chains, and the suffix-rows
// rejection test for Foveal-like kernels) and over one-window rows (rows_kernel: PathDistance.forward_topk), with their
// launchers.  Part of libpsh_hip.so; shared device code in psh_device.h, design overview at the top of psh_scan.hip.
#include "psh_device.h"

namespace psh {

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
    // supports that form one interval: embed_px_kernel (psh_embed_px.hip), launched beside this one, does the stage
    if (MODE != PSH_MODE_ALL && a.plan != nullptr && __builtin_amdgcn_readfirstlane(a.plan->contig) != 0) return;
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
        // PSH_FLAG_EMBED_DENSE: EVERY tap is multiplied, the zeros in front of and behind a row's span included -- the same bits
        // on finite data (fma(0, y, e) = e), and 0 * NaN = NaN exactly where the reference's zero-padded conv has it (what the
        // callers' exhaustive pass over the dirty rows of an ensemble relies on: psh.h, psh_rows_nonfinite)
        if (a.emb_dense) { lo = 0; hi = K; }
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
#define PSH_ROWS_NB 10               // 16-byte loads a lane keeps in flight while a chunk of 64 rows is staged

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
            // PSH_ROWS_NB 16-byte loads of a lane are in flight TOGETHER before the first of them is stored (a plain
            // "load, store, next" loop waits for every load in turn: one KB in flight per wave, 4.8 TB/s; rows of 34
            // floats are 9 loads per lane -- one batch)
            const int64_t n4 = nfl >> 2;                     // whole float4s of the chunk
            const f32x4* src4 = reinterpret_cast<const f32x4*>(src);
            if (staging == PSH_ROWS_FLAT) {
                for (int64_t b4 = 0; b4 < n4; b4 += 64 * PSH_ROWS_NB) {
                    f32x4 v[PSH_ROWS_NB];
#pragma unroll
                    for (int u2 = 0; u2 < PSH_ROWS_NB; ++u2) {     // unconditional (clamped): a predicated load is a branch with its own wait
                        const int64_t e4 = b4 + lane + 64 * u2;
                        v[u2] = __builtin_nontemporal_load(src4 + (e4 < n4 ? e4 : n4 - 1));
                    }
#pragma unroll
                    for (int u2 = 0; u2 < PSH_ROWS_NB; ++u2) {
                        const int64_t e4 = b4 + lane + 64 * u2;
                        if (e4 < n4) *reinterpret_cast<f32x4*>(tile + 4 * e4) = v[u2];
                    }
                }
                if (lane < (int)(nfl & 3)) tile[4 * n4 + lane] = src[4 * n4 + lane];      // the last < 4 floats of a ragged chunk
            } else if (staging == PSH_ROWS_QUADS) {
                const unsigned T4 = (unsigned)(T >> 2);
                const unsigned magic4 = (unsigned)((1ull << 32) / (unsigned long long)T4);
                for (int64_t b4 = 0; b4 < n4; b4 += 64 * PSH_ROWS_NB) {
                    f32x4 v[PSH_ROWS_NB];
#pragma unroll
                    for (int u2 = 0; u2 < PSH_ROWS_NB; ++u2) {     // unconditional (clamped): a predicated load is a branch with its own wait
                        const int64_t e4 = b4 + lane + 64 * u2;
                        v[u2] = __builtin_nontemporal_load(src4 + (e4 < n4 ? e4 : n4 - 1));
                    }
#pragma unroll
                    for (int u2 = 0; u2 < PSH_ROWS_NB; ++u2) {
                        const int64_t e4 = b4 + lane + 64 * u2;
                        if (e4 < n4) {
                            const unsigned r = fast_div((unsigned)e4, magic4, T4);
                            const unsigned c2 = 4u * ((unsigned)e4 - r * T4);
                            if (c2 < (unsigned)W) {
                                float* dstp = tile + r * ds + c2;
                                dstp[0] = v[u2][0]; dstp[1] = v[u2][1]; dstp[2] = v[u2][2]; dstp[3] = v[u2][3];   // (columns >= W of the last quad: unused slots of the row)
                            }
                        }
                    }
                }
            } else {
            const unsigned magic = (unsigned)((1ull << 32) / (unsigned long long)T);
            for (int64_t e4 = lane; 4 * e4 < nfl; e4 += 64) {
                float v[4];
                if (aligned16 && 4 * e4 + 3 < nfl) {
                    const f32x4 q4 = __builtin_nontemporal_load(src4 + e4);
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
                if (a.boot_wave_min) {
                    // an ESTIMATE's rank is a few dozen among thousands of sampled rows: the minimum of a chunk of 64 rows
                    // says as much as its 64 values (two of the best rows in one chunk: the estimate comes out a shade
                    // higher), and the threshold kernel selects among 64x fewer entries
                    float m = (valid && acc == acc) ? acc : __uint_as_float(PSH_INF_BITS);
#pragma unroll
                    for (int off = 32; off > 0; off >>= 1) m = fminf(m, __shfl_xor(m, off, 64));
                    if (lane == 0) a.minbuf[(int64_t)b * a.min_stride + c] = m;
                } else if (valid) a.minbuf[(int64_t)b * a.min_stride + i] = acc;
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

hipError_t launch_embed_scan(const ScanArgs& a, int mode, bool aligned, int grid, hipStream_t s) {
    if (a.emb_mx && (mode == PSH_MODE_FILTER || (mode == PSH_MODE_BOOT && a.boot_per_wave == 2)))
        return launch_embed_mx(a, mode, aligned, grid, s);                   // psh_embed_mx.hip: dense kernel, matrix cores
    if (a.plan != nullptr && mode != PSH_MODE_ALL) {                         // psh_embed_px.hip: one of the two returns at once
        const hipError_t e = launch_embed_px(a, mode, aligned, grid, s);
        if (e != hipSuccess) return e;
    }
    const size_t shmem = scan_shmem_bytes(a.tile_floats, a.B, a.emb_d, a.W, a.emb_wide ? PSH_EMB_WIDE_THREADS : PSH_SCAN_THREADS);
    return aligned ? launch_embed<true>(a, mode, grid, shmem, s) : launch_embed<false>(a, mode, grid, shmem, s);
}

hipError_t embed_blocks_per_cu(bool aligned, size_t shmem, int* out) {
    int n = 0;
    const hipError_t e =
        aligned ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, embed_scan_kernel<true, PSH_MODE_FILTER, PSH_SCAN_THREADS, PSH_EMB_BG, PSH_NEST_BG>, PSH_SCAN_THREADS, shmem)
                : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, embed_scan_kernel<false, PSH_MODE_FILTER, PSH_SCAN_THREADS, PSH_EMB_BG, PSH_NEST_BG>, PSH_SCAN_THREADS, shmem);
    *out = n;
    return e;
}

// ----------------------------------------------------------------------------------
// embedding of the FIRST window of every row: out[r][i] = sum_j ker[i][j] * y[r][j], fma chain over increasing j
// (all K taps, zeros included: the oracle's embedded_acc_one_window).  One-window rows behind a linear embedding
// (T == K + h) are embedded once with this and then scanned as N pre-embedded points by rows_kernel, whose numerator is
// the contiguous 8-lane reduce the reference uses for that layout (path_embedding.py:129-132, path_distance.py:65).
// ----------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void embed_rows_kernel(const float* __restrict__ dataset, int64_t R, int64_t T,
                                                         const float* __restrict__ ker, int d, int K, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float kerS[];          // d x K
    for (int e = (int)threadIdx.x; e < d * K; e += 256) kerS[e] = ker[e];
    __syncthreads();
    const int64_t n = R * (int64_t)d;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
        const int64_t r = e / d;
        const int i = (int)(e - r * d);
        const float* y = dataset + r * T;
        const float* kr = kerS + (size_t)i * K;
        float acc = 0.0f;
        for (int j = 0; j < K; ++j) acc = __builtin_fmaf(kr[j], y[j], acc);
        out[e] = acc;
    }
}

hipError_t launch_embed_rows(const float* dataset, int64_t R, int64_t T, const float* ker, int d, int K, float* out, hipStream_t s) {
    const size_t shmem = (size_t)d * K * sizeof(float);
    if (shmem > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)embed_rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        if (e != hipSuccess) return e;
    }
    int64_t grid = (R * d + 255) / 256;
    if (grid > 8192) grid = 8192;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(embed_rows_kernel, dim3((unsigned)grid), dim3(256), shmem, s, dataset, R, T, ker, d, K, out);
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


}  // namespace psh
