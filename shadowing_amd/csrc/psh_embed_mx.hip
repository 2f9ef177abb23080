// psh_embed_mx.hip -- the embedded scan for DENSE linear embeddings (wavelet banks, user kernels) with its rejection test on
// the matrix cores: psh_scan_topk_embedded + PSH_FLAG_EMBED_MX, BOOT and FILTER stages (the exhaustive stage stays with
// embed_scan_kernel).  Reference: path_embedding.py:117-132 (conv1d with a (d,1,K) kernel) feeding path_distance.py:62-65.
//
// Per window  H_i = sum_j ker[i][j] y[t+j]  (i < d)  is a Toeplitz product: (1024 windows x K taps) . (K x d).  On the vector
// ALUs it is d*K fma per window and 4 bytes (2772 for the 11 x 252 bank of BASELINE configs[4]): 10.8 ms per 16 queries.
// Here it runs on v_mfma_f32_16x16x32_f16, laid out so that NO operand needs an unaligned read and every lane ends up
// with ALL d values of its windows:
//   A  row m = the samples from window 16 m on (64 rows per segment, taken 32 at a time: two M tiles per half segment):
//      fragment = 8 consecutive halves at 16 m + 32 ks + 8 kq -- an aligned 16-byte LDS read of the segment's f16 copy;
//   B  one N tile PER KERNEL ROW i, its 16 columns the 16 shifts s:  B[k][s] = ker_i[k - s]  -> fragment = 8 consecutive
//      taps at 32 ks + 8 kq - s: the full scan (one product) reads it as ONE aligned 16 bytes from the copy of the row shifted by
//      (-s) mod 8 (8 copies, placed on the LDS banks so that a read's four lane groups are conflict-free: emx_dims); the
//      bootstrap (split products) as two aligned 8-byte reads of the copy shifted by (-s) mod 4 (4 copies, hi and lo);
//   C  tile (M tile, i), column s (the lane), row (the register): H_i of window 16 m + s.  K' = K + 15 taps in 32-tap steps.
// A half segment (512 windows) keeps 2 x 12 tiles x 4 = 96 accumulator registers per lane -- all of them VGPRs the epilogue
// can read (a whole segment's 192 spill: the vector ALUs cannot take AGPR operands).
// f16 alone is too coarse (its 2^-11 per tap let ~2 % of the windows through, r01): data and kernel are SPLIT,
// v = hi + lo (22 bits), and three products are accumulated, hi.hi + lo.hi + hi.lo (lo.lo ~ 2^-22 is dropped): per output
//   |C_i - S H_i| <= eps * sum_j |k~_ij| |y~_j|,   eps = 3 * 2^-22 (splits, dropped term) + 3 K' * 2^-24 (fp32 accumulation
//   of 3 K' exact products, any order) -- doubled below for whatever the unit does inside -- plus 2^-6 absolute (subnormal
//   lo parts), S = the two power-of-two scales that put max|y| of the segment and max|ker| into [256, 512).
// In the embedding space that is a ball of radius  R = ymax * eps * sqrt(sum_i ||ker_i||_1^2): a window survives unless
//   acc^ > (sqrt(tau)(1 + 2^-15) + R)^2 (1 + 2^-14),  acc^ = sum_i (hx_i - H^_i)^2 = ||hx||^2 + sum_i H^_i^2 - 2 sum_i hx_i H^_i
// (13 VALU operations per window and query, the query's 12 coordinates in SGPRs); survivors -- true candidates plus a
// fraction of a percent -- get the exact dense chains in the oracle's order (embedded_acc), and only those values are ranked:
// results are bit-identical to embed_scan_kernel's.  The bootstrap uses the bound the other way round (upper bounds).
// 1296 16x16x32 MFMAs per segment (8.6 us per SIMD) against ~15 000 VALU instructions for the dense chains.
//
// r03, the full scan (FILTER, one product: 432 MFMAs a segment + 128 of the per-query pass).  Measured per half segment and wave
// with tools/emx_phases.py: the matrix cores and the vector ALUs of a SIMD do NOT overlap between its two waves here (a raised
// priority or a lock that keeps the two out of the product loop together only moves time from one phase to the other), so the
// kernel's time is the SUM of its MFMA cycles, its vector-ALU issue cycles and its stalls -- and each was cut:
//   * product loop: fragments software-pipelined by hand, ONE 16-byte read between two MFMAs (a block of reads behind 8 MFMAs
//     costs ~50 cycles per read: tools/ubench_emx_loop.hip), every offset an immediate, two steps per turn: no vector-ALU
//     instruction in the loop, bank conflicts 3.7 -> 0.3 cycles per LDS cycle;
//   * energies and the A fragments of the per-query pass from the UNSCALED accumulators with packed operations on the register
//     pairs as they lie (450 -> 234 instructions), one f16 scale for the batch's queries so that the energies enter the per-query
//     MFMAs as their C operand (59 -> 44 per four queries);
//   * the next unit is touched into the L2 at a unit's top and read at its end (kept in registers across a unit it was spilled
//     right behind its loads: a wait for HBM per unit); survivors' exact chains read 16 taps per LDS round trip.
// configs[4]'s shard (R = 32768, 16 queries): 1.61 -> 1.15 ms.
#include <cstdlib>
#include <type_traits>

#include "psh_device.h"

namespace psh {

#define PSH_EMX_THREADS 512                  // 8 waves, two per SIMD: one wave's epilogue / conversion runs beside the other's MFMAs
#define PSH_EMX_TILE 0                       // fp32 copy of the segment in LDS for the exact verification (0: the survivors' windows are re-read from global memory)
#define PSH_EMX_MAX_D 12
#define PSH_EMX_PADL 16                      // zero taps in front of a kernel row (shifts reach 15 taps back)
#define PSH_EMX_CS8 312                     // halves per copy of the 8-copy layout: 39 slots of 16 bytes, 7 (mod 16)
#define PSH_EMX_QCAP 128                     // survivors (window | query << 12) queued per wave before exact verification
#define PSH_EMX_QM_MAX_B 256                 // the per-query pass runs on the matrix cores for 3 .. 256 queries (their tables sit in LDS)
#define PSH_EMX_QM_MIN_B 3
#define PSH_EMX_EPS_REL (2.0f * (3.0f / 4194304.0f))          // 2 x 3 * 2^-22; the accumulation part is added per K (below)

struct EmxDims {
    int KS;        // 32-tap steps: ceil((K + 15) / 32)
    int CS;        // split products (NP = 3): halves per shifted copy of a kernel row (zero padded on both sides), 4 copies, hi and lo
    int CS8;       // one product (NP = 1): halves per copy, 8 copies (every fragment one aligned 16-byte read), hi only
    unsigned pos8; // nibble c: where copy c of a row sits among the row's 8 (see below)
    int nhalf;     // halves of a segment's f16 copy: 1024 + 32 KS
};
// The 8-copy layout and the LDS banks.  A ds_read_b128 is served in four groups of 16 lanes ({0-3, 12-15, 20-27}, {4-11, 16-19,
// 28-31} and the same + 32), one cycle per group when its 16 lanes touch 16 different 16-byte slots of the 256-byte bank row.
// Lane (s = lane & 15, kq = lane >> 4) reads copy c = -s & 7 at slot  base_c + 2 - (s + c) / 8 + kq: a group touches 12 different
// addresses in 8 copies, and they fall on 12 different slots iff the copies' bases do (mod 16) what an exhaustive search allows --
// with one stride S between neighbouring copies: S = 3, 7, 9 or 13 (mod 16) slots and the copies in the orders below.  A copy is
// 4 KS + 2 slots long, so S = that + 1, + 3 or + 5 (39 slots = 312 halves for K = 252: the size the 4 + 4 split copies have).
// (r02: 4 copies read with ds_read2_b64 -- half the bytes per LDS cycle of a b128, and 5 two-way conflicts in every 16 lanes:
//  SQ_LDS_BANK_CONFLICT 3.7 cycles per LDS instruction cycle, the LDS the product loop's bound at 2.2x its MFMA time.)
__host__ __device__ inline EmxDims emx_dims(int K) {
    EmxDims d;
    d.KS = (K + 15 + 31) / 32;
    int cs = 32 * d.KS + PSH_EMX_PADL + 8;   // the last fragment ends at 32 (KS - 1) + 24 + PADL + 7
    if ((cs & 31) == 0) cs += 8;
    d.CS = cs;
    // (orders for the four strides, position of copy c = 0 .. 7 lowest nibble first: 3: 0x64175302, 7: 0x52176304, 9: 0x54176302,
    //  13: 0x74265310.)  ONE stride serves every K <= 256: 39 slots (KS <= 9: 4 KS + 2 <= 38), so that the 12 rows' and 8 copies'
    //  offsets are instruction immediates in the product loop instead of an address add per read.
    d.CS8 = PSH_EMX_CS8;
    d.pos8 = 0x52176304u;
    d.nhalf = 1024 + 32 * d.KS;
    return d;
}

__host__ __device__ inline size_t emx_shmem_bytes(int K, int d, int B, int tile_floats) {
    const EmxDims m = emx_dims(K);
    size_t n = (size_t)PSH_EMX_MAX_D * 4 * m.CS * sizeof(_Float16) * 2;              // B operand: 12 rows x 4 shifted copies, hi and lo
    const size_t n8 = (size_t)PSH_EMX_MAX_D * 8 * m.CS8 * sizeof(_Float16);           // ... or 12 rows x 8 copies, hi
    n = n > n8 ? n : n8;
    n += (size_t)d * ((K + 3) & ~3) * sizeof(float);                                  // the fp32 kernel (exact verification), rows padded to 16 bytes
    n += (size_t)(PSH_EMX_THREADS / 64) * ((size_t)2 * m.nhalf * sizeof(_Float16) + (size_t)(PSH_EMX_TILE ? tile_floats : 0) * sizeof(float)
                                           + (size_t)PSH_PEND * 16 + (size_t)PSH_EMX_QCAP * 8 + 64 * 4);
    n += (size_t)(((B + 3) & ~3) + 8) * sizeof(int) + 64;
    // the per-query pass on the matrix cores (B <= PSH_EMX_QM_MAX_B): scaled f16 query coordinates, per-query constants,
    // a wave's transposed window energies
    if (B <= PSH_EMX_QM_MAX_B) n += (size_t)((B + 3) & ~3) * (32 + 16) + (size_t)(PSH_EMX_THREADS / 64) * 8 * 64 * sizeof(float);
    return n;
}

// logical half index -> padded LDS index (8 halves of padding per 256)
__device__ __forceinline__ int emx_pad(int p) { return p; }
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef _Float16 f16x4v __attribute__((ext_vector_type(4)));

// NG: groups of 4 kernel rows computed (ceil(d / 4)); a compile-time constant so that the product loop has NO branch around
// its MFMAs -- with one, the accumulators are shuffled between AGPRs and VGPRs at every merge (96 + 96 moves per K step).
// NP: products accumulated per tile.  3 = the split above (the bootstrap: its minima feed the admission level).  1 = hi . hi
// alone (the full scan): eps = 2^-10 + 2^-22 + 2 K' 2^-24 -- ten times the split's radius, a few times more survivors, each one
// exact dense chain -- for a third of the MFMAs and half the fragment reads; matrix cores and vector ALUs do not overlap on
// this part (their busy times ADD UP to the kernel's in every counter run), so an MFMA saved is time saved.
// QM: the per-query pass of the full scan -- acc^ - ||hx||^2 = ||H||^2 - 2 sum_i hx_i H_i for every window and query -- on the
// matrix cores too (3 .. 256 queries): a lane's accumulator registers ARE an A fragment (row = its column s, K chunk = its
// row quarter kq: 8 coordinates of ONE of its windows), and a B tile whose column (kq', q) carries query q's coordinates
// in K chunk kq' only (zeros elsewhere) picks them apart again:  D[s][(kq', q)] = sum_i H_i(window(kq', s)) hx_q[i] -- 2 MFMAs per
// accumulator slot and 4 queries where the vector ALUs spent 110 packed instructions per query and half segment.  One f16
// product again: |c^ - c| <= eps2 sqrt(nh nx) <= eps2 (nh + nx) / 2, so the test compares  nh (1 - eps2) - 2 c^  with
// thr - nx (1 - eps2): nothing per window and query is added.  The window energies reach the D layout through LDS (8 floats a lane).
template <bool ALIGNED, int MODE, int NG, int NP, bool QM>
__global__ __launch_bounds__(PSH_EMX_THREADS) void embed_mx_kernel(ScanArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NW = PSH_EMX_THREADS / 64;
    const int lane = lane_id();
    const int tid = (int)threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int K = a.W, d = a.emb_d;
    const EmxDims dm = emx_dims(K);
    // ---- LDS carve
    _Float16* bh = reinterpret_cast<_Float16*>(smem);                                  // [12][4][CS] hi  (NP = 1: [12][8][CS8])
    _Float16* bl = bh + (size_t)PSH_EMX_MAX_D * 4 * dm.CS;                              // [12][4][CS] lo
    const size_t bop_halves = (size_t)PSH_EMX_MAX_D * (8 * dm.CS > 8 * dm.CS8 ? 8 * dm.CS : 8 * dm.CS8);
    const int Kst = (K + 3) & ~3;                                                       // row stride of the fp32 kernel: rows 16-byte aligned, zero tail
    float* kerF = reinterpret_cast<float*>(bh + bop_halves);                            // d x Kst fp32
    char* pw = reinterpret_cast<char*>(kerF) + (size_t)d * Kst * 4;
    const int tile_fl = PSH_EMX_TILE ? a.tile_floats : 0;
    const size_t per_wave = (size_t)2 * dm.nhalf * 2 + (size_t)tile_fl * 4 + (size_t)PSH_PEND * 16 + (size_t)PSH_EMX_QCAP * 8 + 64 * 4;
    char* mine = pw + (size_t)wave * per_wave;
    _Float16* yh = reinterpret_cast<_Float16*>(mine);
    _Float16* yl = yh + dm.nhalf;
    float* tile = reinterpret_cast<float*>(yl + dm.nhalf);
    u32x4* pend = reinterpret_cast<u32x4*>(tile + tile_fl);
    unsigned* sq = reinterpret_cast<unsigned*>(pend + PSH_PEND);
    unsigned* squ = sq + PSH_EMX_QCAP;                                                  // ... and the (row, segment) index of the unit that queued it
    float* Dl = reinterpret_cast<float*>(squ + PSH_EMX_QCAP);                           // 4 survivors x 16 row differences
    int* lcount = reinterpret_cast<int*>(pw + (size_t)NW * per_wave);
    int* ctl = lcount + ((a.B + 3) & ~3);                                               // [0] work cursor, [1] max|ker| bits, [2] cerr^2, [3] max ||ker_i||_1 (float bits)
    const int Bp = (a.B + 3) & ~3;
    _Float16* qh = reinterpret_cast<_Float16*>(reinterpret_cast<char*>(ctl) + 8 * sizeof(int) + 64);   // QM: [Bp][16] scaled query coordinates
    f32x4* qc = reinterpret_cast<f32x4*>(qh + (size_t)Bp * 16);                          // QM: [Bp] {sqrt(tau)(1+2^-15), nx(1-eps2), -2/s_q, -}
    float* nhL = reinterpret_cast<float*>(qc + Bp) + (size_t)wave * 8 * 64;             // QM: the wave's window energies, [slot][kq][s]
    int npend = 0, nsq = 0;
#ifdef PSH_TUNING
    unsigned long long tacc[4] = {0ull, 0ull, 0ull, 0ull}, tlast = __builtin_amdgcn_s_memtime();     // set-up+convert / product / energies + per-query pass / verification at the unit's end
    unsigned tcount[2] = {0u, 0u};                                                                   // survivors verified at unit ends, passes of 4
    auto tstamp = [&](int i) { const unsigned long long t = __builtin_amdgcn_s_memtime(); tacc[i] += t - tlast; tlast = t; };
#else
    auto tstamp = [&](int) {};
#endif

    // ---- per-block set-up: kernel scale, the shifted hi/lo copies of every row, the radius constant
    if (tid == 0) { ctl[0] = 0; ctl[1] = 0; ctl[2] = 0; ctl[3] = 0; ctl[4] = 0; ctl[5] = 0; ctl[6] = 0; ctl[7] = 0; }
    if (MODE == PSH_MODE_FILTER)
        for (int q = tid; q < a.B; q += PSH_EMX_THREADS) lcount[q] = 0;
    for (int e = tid; e < d * Kst; e += PSH_EMX_THREADS) { const int i = e / Kst, j = e - i * Kst; kerF[e] = j < K ? a.ker[i * K + j] : 0.0f; }
    {
        unsigned* z = reinterpret_cast<unsigned*>(yh);
        for (int i = lane; i < dm.nhalf; i += 64) z[i] = 0u;                            // 2 arrays x nhalf halves = nhalf dwords
    }
    __syncthreads();
    {
        unsigned mb = 0u;
        for (int e = tid; e < d * Kst; e += PSH_EMX_THREADS) mb = max(mb, __float_as_uint(fabsf(kerF[e])));
        if (mb) atomicMax(reinterpret_cast<unsigned*>(&ctl[1]), mb);
        if (tid < d) {                                                                  // ||ker_i||_1^2 summed over the rows
            float l1 = 0.0f;
            for (int j = 0; j < K; ++j) l1 += fabsf(kerF[tid * Kst + j]);
            atomicAdd(reinterpret_cast<float*>(&ctl[2]), l1 * l1 * 1.0001f);
            atomicMax(reinterpret_cast<unsigned*>(&ctl[3]), __float_as_uint(l1 * 1.0001f));
        }
    }
    __syncthreads();
    // ---- leading zero taps (round 5).  A bank of localised filters anchored at the window's end (wavelets at dyadic scales:
    // configs[4]'s 28 / 55 / 110 / 220 / 252-tap rows) leaves whole 32-tap steps of a row's band empty: row i first meets a
    // non-zero tap at step f_i / 32 (f_i = its first non-zero tap; the band of step ks covers taps 32 ks - 15 .. 32 ks + 31).
    // The accumulator slots take the rows SORTED by that step, latest first, so that the three 4-row groups of the product
    // loop start at steps gs[0] >= gs[1] >= gs[2] and the loop runs its MFMAs only where a group has anything to multiply
    // (configs[4]: 20 of 27 group-steps).  perm[slot] = the kernel row in accumulator slot `slot` (>= d: an all-zero slot).
    int* permL = ctl + 8;                                                               // 12 ints (the 64 bytes behind the control words)
    {
        int* stL = reinterpret_cast<int*>(bh);                                          // scratch: the B copies are built below
        constexpr int NS = 4 * NG;                                                      // accumulator slots this instantiation computes (>= d)
        if (tid < PSH_EMX_MAX_D) {
            int f = K;
            if (tid < d) for (int j = K - 1; j >= 0; --j) f = kerF[tid * Kst + j] != 0.0f ? j : f;
            stL[tid] = (tid < d && f < K) ? f / 32 : dm.KS;                             // (an all-zero row, a slot past d: never active)
            permL[tid] = PSH_EMX_MAX_D;                                                 // (slots past NS: no row)
        }
        __syncthreads();
        if (tid < NS) {                                                                 // rank among the NS slots by (start, latest first; row index)
            const int mine = stL[tid];
            int rank = 0;
            for (int j = 0; j < NS; ++j) rank += (stL[j] > mine || (stL[j] == mine && j < tid)) ? 1 : 0;
            permL[rank] = tid;
        }
        __syncthreads();
        if (tid < 3) {
            // (the last slot of the last group -- the row with the longest support: a bank's low-pass -- runs ALONE from its
            //  own first step, ctl[20], to the first step of the group's three other rows)
            int m = dm.KS;
            if (tid < NG)
                for (int r4 = 0; r4 < (NG == 3 && tid == 2 ? 3 : 4); ++r4) { const int v = stL[permL[4 * tid + r4]]; m = v < m ? v : m; }
            ctl[5 + tid] = m;
            if (tid == 2) ctl[20] = NG == 3 ? stL[permL[11]] : dm.KS;
        }
        __syncthreads();
    }
    const unsigned kmb = (unsigned)ctl[1];
    int ek = kmb >= 0x00800000u ? 9 - ((int)((kmb >> 23) & 255u) - 126) : 0;            // max|ker| 2^ek in [256, 512)
    if constexpr (QM) {
        // The per-query pass takes the accumulators AS THEY ARE for its f16 A fragments (round 5: scaling them first was 48
        // packed multiplies per half segment): |acc| <= (ymax 2^ey < 512) (max_i ||ker_i||_1 2^ek) must stay inside f16, and as
        // far up as the scale of rounds 3-4 put it ([8192, 16384) exactly; now [8192, 32768) by the two factors' binades):
        // max_i ||ker_i||_1 2^ek in [32, 64).  The copies' taps are then at most 64 -- a tap below 2^-14 / 2^ek is a subnormal
        // f16, absolute error 2^-25 instead of 2^-11 relative: sqrt(12) K 2^-25 / 32 < 2^-20 of max ||ker||_1 on the radius (eps).
        const unsigned lb = (unsigned)ctl[3];
        ek = lb >= 0x00800000u ? 6 - ((int)((lb >> 23) & 255u) - 126) : 0;
    }
    // (wave-uniform constants through readfirstlane: scalar registers -- as vector registers three of them were spilled once
    //  the product loop's phases took theirs)
    const int ekc = __builtin_amdgcn_readfirstlane(ek > 100 ? 100 : (ek < -100 ? -100 : ek));
    const float sk = __uint_as_float((unsigned)(127 + ekc) << 23);
    if (NP == 3) {
        for (int e = tid; e < PSH_EMX_MAX_D * 4 * dm.CS; e += PSH_EMX_THREADS) {
            // copy c of row i:  copy[x] = L_i[x + c],  L_i[x] = ker_i[x - PADL] (zero outside [0, K))
            const int i = e / (4 * dm.CS), rem = e - i * 4 * dm.CS, c = rem / dm.CS, x2 = rem - c * dm.CS;
            const int j = x2 + c - PSH_EMX_PADL;
            const int ir = permL[i];                                                    // the kernel row in accumulator slot i
            float v = (ir < d && j >= 0 && j < K) ? kerF[ir * Kst + j] * sk : 0.0f;
            const _Float16 hi = (_Float16)v;
            const _Float16 lo = (_Float16)(v - (float)hi);
            bh[e] = hi;
            bl[e] = lo;
        }
    } else {
        for (int e = tid; e < PSH_EMX_MAX_D * 8 * dm.CS8; e += PSH_EMX_THREADS) {
            // the copy at position ps of row i is copy c with pos8[c] == ps
            const int i = e / (8 * dm.CS8), rem = e - i * 8 * dm.CS8, ps = rem / dm.CS8, x2 = rem - ps * dm.CS8;
            int c = 0;
#pragma unroll
            for (int cc = 0; cc < 8; ++cc) c = ((dm.pos8 >> (4 * cc)) & 15u) == (unsigned)ps ? cc : c;
            const int j = x2 + c - PSH_EMX_PADL;
            const int ir = permL[i];
            const float v = (ir < d && j >= 0 && j < K) ? kerF[ir * Kst + j] * sk : 0.0f;
            bh[e] = (_Float16)v;
        }
    }
    __syncthreads();
    // eps: splits and the dropped lo.lo term (3 * 2^-22) + fp32 accumulation of 3 K' products (3 * 32 KS * 2^-24), doubled
    const float eps = NP == 3 ? PSH_EMX_EPS_REL + 2.0f * (float)(3 * 32 * dm.KS) / 16777216.0f
                              : 1.001f * (1.0f / 1024.0f + 1.0f / 4194304.0f) + 2.0f * (float)(32 * dm.KS) / 16777216.0f + (QM ? 1.0f / 524288.0f : 0.0f);
    const float cerr = __uint_as_float((unsigned)__builtin_amdgcn_readfirstlane((int)__float_as_uint(eps * __builtin_sqrtf(__uint_as_float((unsigned)ctl[2])) * 1.001f)));
    const float kl1max = __uint_as_float((unsigned)__builtin_amdgcn_readfirstlane(ctl[3]));
    constexpr float eps2 = 1.001f * (1.0f / 1024.0f + 1.0f / 4194304.0f) + 64.0f / 16777216.0f;
    float sqc = 1.0f, thrG = 0.0f;
    int eqc = 0;
    if (QM) {
        // the queries' coordinates as f16 at ONE power-of-two scale (the largest |hx_i| of the batch in [256, 512)): with one scale
        // the window energies can enter the per-query MFMAs as their C operand (below).  A coordinate more than 2^-22 below the
        // batch's largest is a subnormal f16 -- absolute error 2^-25 / s instead of 2^-11 relative:  2 sum_i |H_i| 2^-25 / s <=
        // 2^-24 sqrt(12 nh) / s <= 2^-12 nh + 12 * 2^-38 / s^2  -- a 2^-12 on the energies' factor and thrG on the thresholds.
        unsigned mq = 0u;
        for (int e = tid; e < a.B * d; e += PSH_EMX_THREADS) mq = max(mq, __float_as_uint(fabsf(a.hx[e])));
        if (mq) atomicMax(reinterpret_cast<unsigned*>(&ctl[4]), mq);
        __syncthreads();
        const unsigned mb2 = (unsigned)ctl[4];
        int eq = mb2 >= 0x00800000u ? 9 - ((int)((mb2 >> 23) & 255u) - 126) : 0;
        eq = eq > 100 ? 100 : (eq < -100 ? -100 : eq);
        eqc = eq;
        sqc = __uint_as_float((unsigned)(127 + eq) << 23);
        {
            const float mxf = __uint_as_float(mb2);
            thrG = (mxf * (1.0f / 33554432.0f)) * (mxf * (1.0f / 33554432.0f));        // 2^-50 max|hx|^2 >= 12 * 2^-38 / s^2  (1 / s <= max|hx| / 256)
        }
        for (int q = tid; q < Bp; q += PSH_EMX_THREADS) {
            float hq[16], nxq = 0.0f;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int ir = i < PSH_EMX_MAX_D ? permL[i] : d;                        // slot i holds kernel row ir
                hq[i] = (q < a.B && ir < d) ? a.hx[(int64_t)q * d + ir] : 0.0f;
                nxq = __builtin_fmaf(hq[i], hq[i], nxq);
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) qh[q * 16 + i] = (_Float16)(hq[i] * sqc);
            const float tau = q < a.B ? __uint_as_float(a.qstate[q].tau2_bits) : 0.0f;
            qc[q] = f32x4{__builtin_sqrtf(tau) * (1.0f + 1.0f / 32768.0f), nxq * (1.0f - eps2) * (1.0f - 1.0f / 1048576.0f), 0.0f, 0.0f};
        }
        __syncthreads();
    }

    const int nfloat = PSH_SEG + K - 1;
    const unsigned n_rs = (unsigned)a.n_rows * (unsigned)a.nseg;
    const unsigned n_units = n_rs * (unsigned)a.n_qgroups;
    const unsigned u_lo = (unsigned)(((unsigned long long)n_units * blockIdx.x) / gridDim.x);
    const unsigned u_hi = (unsigned)(((unsigned long long)n_units * (blockIdx.x + 1)) / gridDim.x);
    const const_f32p hxk = (const_f32p)a.hx;
    typedef const __attribute__((address_space(4))) QueryState* const_qsp;
    const const_qsp qstate_k = (const_qsp)a.qstate;
    const int scol = lane & 15, kq = lane >> 4;                                         // column = shift s (and A row), K quarter
    int perm_s[PSH_EMX_MAX_D];                                                          // the slots' kernel rows, wave-uniform
#pragma unroll
    for (int i = 0; i < PSH_EMX_MAX_D; ++i) perm_s[i] = __builtin_amdgcn_readfirstlane(permL[i]);
    // first steps of the product loop's groups (slots 0-3 / 4-7 / 8-11), latest first; NG < 3: the plain loop runs every step
    const int gs0 = __builtin_amdgcn_readfirstlane(ctl[5]), gs1 = __builtin_amdgcn_readfirstlane(ctl[6]), gs2 = __builtin_amdgcn_readfirstlane(ctl[7]);
    const int gs3 = __builtin_amdgcn_readfirstlane(ctl[20]);                            // slot 11 alone: steps [gs3, gs2)

    // exact verification of the queued survivors: lane (e, i) runs row i of survivor e (4 per pass), the oracle's order:
    // hy_i = fma chain over all K taps, D_i = hx_i - hy_i, acc = fma chain over i.
    // A queue entry carries the (row, segment) index of the unit that found it: at a unit's end only WHOLE passes of four run,
    // the last one to three survivors wait for the next units' (a pass costs the same 6 - 8 k cycles for one survivor as for
    // four, and a unit finds 0.5 on average -- 0.41 passes per unit became 0.13); the last unit of the wave flushes.
    auto coords = [&](unsigned rsu, int& seg_start_e, int64_t& row_e) {
        const unsigned ri2 = fast_div(rsu, a.magic_nseg, (unsigned)a.nseg);
        seg_start_e = (int)(rsu - ri2 * (unsigned)a.nseg) * PSH_SEG;
        row_e = a.row0 + (int64_t)ri2 * a.row_stride;
    };
    auto verify_impl = [&](bool whole_passes_only, auto fast_c) {
        constexpr bool FAST = decltype(fast_c)::value;
        wave_lds_fence();
        const int el = lane >> 4, il = lane & 15;
        const int n_run = whole_passes_only ? (nsq & ~3) : nsq;
#pragma unroll 1
        for (int e0 = 0; e0 < n_run; e0 += 4) {
            const bool lv = e0 + el < n_run;
            const unsigned ent = lv ? sq[e0 + el] : 0u;
            const int pwin = (int)(ent & 4095u), b = (int)(ent >> 12);
            int seg_start_e;
            int64_t row_e;
            coords(lv ? squ[e0 + el] : 0u, seg_start_e, row_e);
            float hy = 0.0f;
            if constexpr (!FAST) {
                // (mid-unit, a full queue -- rare: the accumulators are live, so the chain reads memory tap by tap and needs no registers)
                if (lv && il < d) {
                    const float* kr = kerF + il * Kst;
                    const float* yw = a.dataset + row_e * a.T + seg_start_e + pwin;
                    for (int j = 0; j < K; ++j) hy = __builtin_fmaf(kr[j], yw[j], hy);
                    Dl[el * 16 + il] = __fsub_rn(a.hx[(int64_t)b * d + il], hy);
                }
            } else {
                // (unit end: the segment's f16 copies are done with) the 4 survivors' windows are first copied into that LDS,
                // all loads in flight together, and the chains run at LDS latency: a load per step of the chain was a round
                // trip to memory per tap (38 us a pass of 4 survivors, a quarter of the 16-query scan)
                // (survivors 0, 1 at the start of the hi copy, 2, 3 at the start of the lo copy: 2 Kst floats = 4 Kst halves each,
                //  inside the part every segment rewrites -- the zero tails of the copies must stay zero: 0 * garbage)
                float* ysA = reinterpret_cast<float*>(yh);
                float* ysB = reinterpret_cast<float*>(yl);
                {
                    float v[16];
#pragma unroll
                    for (int m4 = 0; m4 < 4; ++m4) {                            // survivor m4, taps lane + 64 (0 .. 3)
                        const int e2 = e0 + m4 < n_run ? e0 + m4 : e0;
                        int ss2;
                        int64_t row2;
                        coords(squ[e2], ss2, row2);
                        const float* yw2 = a.dataset + row2 * a.T + ss2 + (int)(sq[e2] & 4095u);
#pragma unroll
                        for (int m1 = 0; m1 < 4; ++m1) {
                            int j = lane + 64 * m1;
                            j = j < K ? j : K - 1;
                            v[4 * m4 + m1] = yw2[j];
                        }
                    }
                    wave_lds_fence();                                           // the pass before this one has read ys
#pragma unroll
                    for (int m = 0; m < 16; ++m) {
                        const int j = lane + 64 * (m & 3);
                        if (j < Kst) (((m >> 2) & 2) ? ysB : ysA)[Kst * ((m >> 2) & 1) + j] = v[m];
                    }
                    wave_lds_fence();
                }
                const float* kr = kerF + (il < d ? il : 0) * Kst;
                const float* yw = ((el & 2) ? ysB : ysA) + Kst * (el & 1);
                const float hxv = (lv && il < d) ? a.hx[(int64_t)b * d + il] : 0.0f;
                {
                    // the chain in the oracle's order, its operands 16 taps at a time as 16-byte LDS reads, the next 16 in flight
                    // (a read per operand and tap, four taps at a time, sat at LDS latency every four fma: 16 k cycles a pass)
                    const f32x4* kr4 = reinterpret_cast<const f32x4*>(kr);
                    const f32x4* yw4 = reinterpret_cast<const f32x4*>(yw);
                    const int nv = K >> 2, gmax = (Kst >> 2) - 1;
                    f32x4 kb[4], yb[4];
#pragma unroll
                    for (int u2 = 0; u2 < 4; ++u2) { const int g4 = u2 < gmax ? u2 : gmax; kb[u2] = kr4[g4]; yb[u2] = yw4[g4]; }
#pragma unroll 1
                    for (int g0 = 0; g0 < nv; g0 += 4) {
                        f32x4 kn[4], yn[4];
#pragma unroll
                        for (int u2 = 0; u2 < 4; ++u2) { const int g4 = g0 + 4 + u2 < gmax ? g0 + 4 + u2 : gmax; kn[u2] = kr4[g4]; yn[u2] = yw4[g4]; }
#pragma unroll
                        for (int u2 = 0; u2 < 4; ++u2)
                            if (g0 + u2 < nv) {
#pragma unroll
                                for (int e4 = 0; e4 < 4; ++e4) hy = __builtin_fmaf(kb[u2][e4], yb[u2][e4], hy);
                            }
#pragma unroll
                        for (int u2 = 0; u2 < 4; ++u2) { kb[u2] = kn[u2]; yb[u2] = yn[u2]; }
                    }
                    for (int j = 4 * nv; j < K; ++j) hy = __builtin_fmaf(kr[j], yw[j], hy);
                }
                if (lv && il < d) Dl[el * 16 + il] = __fsub_rn(hxv, hy);
            }
            wave_lds_fence();
            float ea = 0.0f;
            bool hit = false;
            if (lv && il == 0) {
                for (int i = 0; i < d; ++i) { const float D = Dl[el * 16 + i]; ea = __builtin_fmaf(D, D, ea); }
                hit = ea < __uint_as_float(a.qstate[b].tau2_bits);
            }
            const unsigned long long mask = __ballot(hit);
            wave_lds_fence();                                // Dl is rewritten by the next pass
            if (!mask) continue;
            const int nh = __popcll(mask);
            if (npend + nh > PSH_PEND) {
                pend_flush(pend, npend, lcount, a, lane);
                npend = 0;
                wave_lds_fence();
            }
            if (hit) {
                const int slot = npend + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
                pend[slot] = u32x4{__float_as_uint(ea), (unsigned)(int)(row_e + a.r_offset), (unsigned)(seg_start_e + pwin), (unsigned)b};
            }
            npend += nh;
        }
        // the survivors that wait: to the front of the queue
        const int left = nsq - n_run;
        unsigned keep0 = 0u, keep1 = 0u;
        int ln = lane;
        asm volatile("" : "+v"(ln));                        // (an address taken here, not one kept in a register across the unit)
        if (ln < left) { keep0 = sq[n_run + ln]; keep1 = squ[n_run + ln]; }
        wave_lds_fence();
        if (ln < left) { sq[ln] = keep0; squ[ln] = keep1; }
        nsq = left;
        wave_lds_fence();
    };
    auto verify = [&]() { verify_impl(false, std::false_type{}); };
    auto verify_end = [&](bool whole_passes_only) { verify_impl(whole_passes_only, std::true_type{}); };

    auto grab = [&]() -> unsigned {
        int v0 = 0;
        if (lane == 0) v0 = atomicAdd(&ctl[0], 1);
        return u_lo + (unsigned)__builtin_amdgcn_readfirstlane(v0);
    };
    auto load_unit = [&](Stage& sx, unsigned uu) {
        const unsigned qg2 = fast_div(uu, a.magic_nrs, n_rs);
        const unsigned rs2 = uu - qg2 * n_rs;
        const unsigned ri2 = fast_div(rs2, a.magic_nseg, (unsigned)a.nseg);
        const unsigned sg2 = rs2 - ri2 * (unsigned)a.nseg;
        sx.v[PSH_NSTAGE - 1] = f32x4{0.f, 0.f, 0.f, 0.f};  // (the last, partial stage: lanes past the segment would otherwise KEEP the old one -- live across the unit)
        stage_load<ALIGNED>(sx, a.dataset + (a.row0 + (int64_t)ri2 * a.row_stride) * a.T, a.T, (int)sg2 * PSH_SEG, nfloat, lane);
    };
    // One segment in flight from HBM besides the one being worked on.  The full scan (LATE) has no 20 registers to keep it in
    // across a unit -- the compiler spilled part of it right behind the loads, i.e. WAITED for HBM twice per unit (r03: 7.6 k of a
    // unit's 35 k cycles) -- so there the next unit is only TOUCHED at the top of this one (one dword per 128-byte line: the
    // segment comes to the L2) and read into registers at its end, when the accumulators are dead.
    constexpr bool LATE = (MODE == PSH_MODE_FILTER && NP == 1);
    auto touch_unit = [&](unsigned uu) -> float {
        const unsigned qg2 = fast_div(uu, a.magic_nrs, n_rs);
        const unsigned rs2 = uu - qg2 * n_rs;
        const unsigned ri2 = fast_div(rs2, a.magic_nseg, (unsigned)a.nseg);
        const unsigned sg2 = rs2 - ri2 * (unsigned)a.nseg;
        const float* rowp = a.dataset + (a.row0 + (int64_t)ri2 * a.row_stride) * a.T;
        int p = (int)sg2 * PSH_SEG + 32 * lane;
        const int pl = (int)sg2 * PSH_SEG + nfloat - 1;
        p = p < pl ? p : pl;
        p = p < (int)a.T - 1 ? p : (int)a.T - 1;
        return rowp[p];
    };
    Stage st;
    unsigned u = grab();
    if (u < u_hi) load_unit(st, u);
    while (u < u_hi) {
        const unsigned qgi = fast_div(u, a.magic_nrs, n_rs);
        const unsigned rs = u - qgi * n_rs;
        const unsigned ri = fast_div(rs, a.magic_nseg, (unsigned)a.nseg);
        const unsigned sg = rs - ri * (unsigned)a.nseg;
        const int64_t row = a.row0 + (int64_t)ri * a.row_stride;
        const int seg_start = (int)sg * PSH_SEG;
        const int r_global = (int)(row + a.r_offset);

        // ---- the segment: fp32 tile (verification), scale, hi/lo f16 copies
        float ymax = 0.0f;
        int ey;
        unsigned un;
        {
            const int nq = (nfloat + 3) >> 2;
#pragma unroll
            for (int q = 0; q < PSH_NSTAGE; ++q)
                if (q < PSH_NSTAGE - 1 || lane + 64 * q < nq)
                    ymax = fmaxf(ymax, fmaxf(fmaxf(fabsf(st.v[q][0]), fabsf(st.v[q][1])), fmaxf(fabsf(st.v[q][2]), fabsf(st.v[q][3]))));
            ymax = wave_max_nonneg(ymax);
            const unsigned yb = __float_as_uint(ymax);
            ey = yb >= 0x00800000u ? 9 - ((int)((yb >> 23) & 255u) - 126) : 0;          // ymax 2^ey in [256, 512)
            ey = ey > 100 ? 100 : (ey < -100 ? -100 : ey);
            const float sy = __uint_as_float((unsigned)(127 + ey) << 23);
            if (MODE == PSH_MODE_FILTER && PSH_EMX_TILE) stage_store(st, tile, nfloat, lane);
#pragma unroll
            for (int q = 0; q < PSH_NSTAGE; ++q) {
                int m = lane + 64 * q;
                if (q == PSH_NSTAGE - 1) asm volatile("" : "+v"(m));   // (the partial stage's address: computed here, not kept -- spilled -- across the unit)
                if (q < PSH_NSTAGE - 1 || m < nq) {
                    const f32x4 v = st.v[q] * sy;
                    const f16x4v hi = __builtin_convertvector(v, f16x4v);
                    const f32x4 res = v - __builtin_convertvector(hi, f32x4);
                    *reinterpret_cast<f16x4v*>(yh + emx_pad(4 * m)) = hi;
                    if (NP == 3) *reinterpret_cast<f16x4v*>(yl + emx_pad(4 * m)) = __builtin_convertvector(res, f16x4v);
                }
            }
            if (MODE == PSH_MODE_FILTER && npend > 0) {   // stores ahead of the prefetch: vmcnt retires in order
                pend_flush(pend, npend, lcount, a, lane);
                npend = 0;
            }
            un = grab();
            if (!LATE && un < u_hi) load_unit(st, un);
        }
        float touched = 0.0f;
        if (LATE && un < u_hi) touched = touch_unit(un);
        wave_lds_fence();

        const float inv = __uint_as_float((unsigned)(127 - ey - ekc) << 23);
        const float Rad = ymax * cerr + 1.0e-30f;
        const int q_begin = (int)qgi * a.q_per_group;
        const int q_end = (q_begin + a.q_per_group) < a.B ? (q_begin + a.q_per_group) : a.B;
#pragma unroll 1
        for (int hf = 0; hf < 2; ++hf) {
            // ---- the banded product of a half segment: 2 M tiles x 12 kernel rows x KS steps x 3 products
            f32x4 C[2][4 * NG];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int i = 0; i < 4 * NG; ++i) C[mt][i] = f32x4{0.f, 0.f, 0.f, 0.f};
            tstamp(0);
            if constexpr (NP == 1) {
                // ---- one product, 8 copies: every fragment ONE aligned 16-byte read, software-pipelined by hand (left to itself
                // the compiler reads one fragment into ONE register quad, waits, issues its two MFMAs, reads the next: the LDS
                // latency of every fragment in the open, r02).  NG == 3: a ring of three 4-row groups, a group's reads issued two
                // groups (16 MFMAs = 256 cycles) before its MFMAs; NG < 3: all of a step's reads, then its MFMAs.
                const int cpy = (-scol) & 7;
                const int ps = (int)((dm.pos8 >> (4 * cpy)) & 15u);
                const _Float16* bp = bh + (size_t)ps * PSH_EMX_CS8 + (PSH_EMX_PADL - scol - cpy) + 8 * kq;
                constexpr int rstride = 8 * PSH_EMX_CS8;
                const _Float16* ap = yh + 16 * (32 * hf + scol) + 8 * kq;               // A row of M tile 0: m = 32 hf + (lane & 15); M tile 1: + 256
                auto ldB = [&](f16x8 (&f)[4], int g, int ks) {
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) f[r4] = *reinterpret_cast<const f16x8*>(bp + (size_t)(4 * g + r4) * rstride + 32 * ks);
                };
                auto mm = [&](const f16x8 (&f)[4], int g, const f16x8& a0, const f16x8& a1) {
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) {
                        C[0][4 * g + r4] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, f[r4], C[0][4 * g + r4], 0, 0, 0);
                        C[1][4 * g + r4] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, f[r4], C[1][4 * g + r4], 0, 0, 0);
                    }
                };
                if constexpr (NG == 3) {
                    f16x8 F0[4], F1[4], F2[4];
                    f16x8 a0, a1, b0, b1;
                    // One read BETWEEN two MFMAs, into the buffer whose MFMAs were issued a group earlier: a block of 4 - 6 reads
                    // behind 8 MFMAs holds the wave's next MFMA back by ~50 cycles per read (tools/ubench_emx_loop.hip: 1141 cycles
                    // per step for one wave in the loop -- and the partner wave is in its epilogue more often than not -- against
                    // 545 interleaved; the MFMAs alone: 405).  Two steps per turn (the A fragments alternate between two register
                    // sets, no moves), every offset an immediate: 2 vector-ALU instructions per 48 MFMAs besides them.
#define PSH_EMX_STEP(A0, A1, FM, gm, FL, gl, OFS, EXTRA0, EXTRA1)                                                                      \
                    _Pragma("unroll") for (int r4 = 0; r4 < 4; ++r4) {                                                                 \
                        C[0][4 * (gm) + r4] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A0, FM[r4], C[0][4 * (gm) + r4], 0, 0, 0);       \
                        __builtin_amdgcn_sched_barrier(0);                                                                             \
                        FL[r4] = *reinterpret_cast<const f16x8*>(bpk + (size_t)(4 * (gl) + r4) * rstride + (OFS));                     \
                        if (r4 == 0) { EXTRA0; }                                                                                       \
                        if (r4 == 2) { EXTRA1; }                                                                                       \
                        __builtin_amdgcn_sched_barrier(0);                                                                             \
                        C[1][4 * (gm) + r4] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A1, FM[r4], C[1][4 * (gm) + r4], 0, 0, 0);       \
                        __builtin_amdgcn_sched_barrier(0);                                                                             \
                    }
                    // Phases by the groups' first steps (rows sorted above): [gs2, gs1) slots 8-11 alone, [gs1, gs0) slots 4-11,
                    // [gs0, KS) all twelve -- the ring below.  The first two keep the ring's rule (one read between two MFMAs, a
                    // group's fragments requested a group of MFMAs ahead) with two buffers; a step's A fragments move between
                    // two register sets (8 moves a step: these phases are short).
                    int ks0 = gs3 < gs2 ? gs3 : gs2;
                    ks0 = ks0 < dm.KS ? ks0 : dm.KS;
                    const _Float16* bpk = bp + 32 * ks0;
                    const _Float16* apk = ap + 32 * ks0;
                    a0 = *reinterpret_cast<const f16x8*>(apk); a1 = *reinterpret_cast<const f16x8*>(apk + 256);
                    // (In all three phases the register sets ROTATE BY NAME -- two or three steps per turn, the set the next phase
                    //  expects restored once at the exit: as "next = current" moves inside the loops they were 24 v_mov per step,
                    //  96 cycles beside a step's 32 - 128 cycles of MFMAs.)
                    if (ks0 < gs2) {
                        // slot 11 alone (configs[4]: the 252-tap low-pass beside 220- and 110-tap rows -- three steps of 2 MFMAs
                        // instead of 8): its fragments two steps ahead (two MFMAs do not cover an LDS read); reads past step KS --
                        // inside the block's LDS -- feed nothing
                        const int lim0 = gs2 < dm.KS ? gs2 : dm.KS;
                        f16x8 fa = *reinterpret_cast<const f16x8*>(bpk + (size_t)11 * rstride);
                        f16x8 fb = *reinterpret_cast<const f16x8*>(bpk + (size_t)11 * rstride + 32), fc;
                        f16x8 c0, c1;
                        b0 = *reinterpret_cast<const f16x8*>(apk + 32); b1 = *reinterpret_cast<const f16x8*>(apk + 256 + 32);
#define PSH_EMX_ONE(FU, U0, U1, FL, L0, L1)                                                                                            \
                        FL = *reinterpret_cast<const f16x8*>(bpk + (size_t)11 * rstride + 64);                                         \
                        L0 = *reinterpret_cast<const f16x8*>(apk + 64); L1 = *reinterpret_cast<const f16x8*>(apk + 256 + 64);          \
                        __builtin_amdgcn_sched_barrier(0);                                                                             \
                        C[0][11] = __builtin_amdgcn_mfma_f32_16x16x32_f16(U0, FU, C[0][11], 0, 0, 0);                                  \
                        C[1][11] = __builtin_amdgcn_mfma_f32_16x16x32_f16(U1, FU, C[1][11], 0, 0, 0);                                  \
                        __builtin_amdgcn_sched_barrier(0);                                                                             \
                        bpk += 32; apk += 32; ++ks0;
                        for (;;) {
                            PSH_EMX_ONE(fa, a0, a1, fc, c0, c1)
                            if (ks0 >= lim0) { a0 = b0; a1 = b1; break; }
                            PSH_EMX_ONE(fb, b0, b1, fa, a0, a1)
                            if (ks0 >= lim0) { a0 = c0; a1 = c1; break; }
                            PSH_EMX_ONE(fc, c0, c1, fb, b0, b1)
                            if (ks0 >= lim0) break;
                        }
#undef PSH_EMX_ONE
                    }
                    if (ks0 < gs1 && ks0 < dm.KS) {
                        const int lim1 = gs1 < dm.KS ? gs1 : dm.KS;
#pragma unroll
                        for (int r4 = 0; r4 < 4; ++r4) F2[r4] = *reinterpret_cast<const f16x8*>(bpk + (size_t)(8 + r4) * rstride);
                        for (;;) {
                            PSH_EMX_STEP(a0, a1, F2, 2, F0, 2, 32, b0 = *reinterpret_cast<const f16x8*>(apk + 32),
                                         b1 = *reinterpret_cast<const f16x8*>(apk + 256 + 32))
                            bpk += 32; apk += 32; ++ks0;
                            if (ks0 >= lim1) { a0 = b0; a1 = b1; break; }
                            PSH_EMX_STEP(b0, b1, F0, 2, F2, 2, 32, a0 = *reinterpret_cast<const f16x8*>(apk + 32),
                                         a1 = *reinterpret_cast<const f16x8*>(apk + 256 + 32))
                            bpk += 32; apk += 32; ++ks0;
                            if (ks0 >= lim1) break;
                        }
                        // (the next phase fetches group 2's fragments of its first step itself)
                    }
                    if (ks0 < gs0 && ks0 < dm.KS) {
                        const int lim2 = gs0 < dm.KS ? gs0 : dm.KS;
#pragma unroll
                        for (int r4 = 0; r4 < 4; ++r4) F1[r4] = *reinterpret_cast<const f16x8*>(bpk + (size_t)(4 + r4) * rstride);
                        for (;;) {
                            PSH_EMX_STEP(a0, a1, F1, 1, F2, 2, 0, (void)0, (void)0)
                            PSH_EMX_STEP(a0, a1, F2, 2, F1, 1, 32, b0 = *reinterpret_cast<const f16x8*>(apk + 32),
                                         b1 = *reinterpret_cast<const f16x8*>(apk + 256 + 32))
                            bpk += 32; apk += 32; ++ks0;
                            if (ks0 >= lim2) { a0 = b0; a1 = b1; break; }
                            PSH_EMX_STEP(b0, b1, F1, 1, F2, 2, 0, (void)0, (void)0)
                            PSH_EMX_STEP(b0, b1, F2, 2, F1, 1, 32, a0 = *reinterpret_cast<const f16x8*>(apk + 32),
                                         a1 = *reinterpret_cast<const f16x8*>(apk + 256 + 32))
                            bpk += 32; apk += 32; ++ks0;
                            if (ks0 >= lim2) break;
                        }
                    } else if (ks0 < dm.KS) {
#pragma unroll
                        for (int r4 = 0; r4 < 4; ++r4) F1[r4] = *reinterpret_cast<const f16x8*>(bpk + (size_t)(4 + r4) * rstride);
                    }
                    if (ks0 < dm.KS) {
#pragma unroll
                        for (int r4 = 0; r4 < 4; ++r4) F0[r4] = *reinterpret_cast<const f16x8*>(bpk + (size_t)r4 * rstride);
                    }
#pragma unroll 1
                    for (int ks = ks0; ks < dm.KS; ks += 2) {
                        // (the reads of step KS -- past the copies and the segment's tail, inside the block's LDS -- feed nothing)
                        PSH_EMX_STEP(a0, a1, F0, 0, F2, 2, 0, (void)0, (void)0)
                        PSH_EMX_STEP(a0, a1, F1, 1, F0, 0, 32, b0 = *reinterpret_cast<const f16x8*>(apk + 32),
                                     b1 = *reinterpret_cast<const f16x8*>(apk + 256 + 32))
                        PSH_EMX_STEP(a0, a1, F2, 2, F1, 1, 32, (void)0, (void)0)
                        if (ks + 1 < dm.KS) {
                            PSH_EMX_STEP(b0, b1, F0, 0, F2, 2, 32, (void)0, (void)0)
                            PSH_EMX_STEP(b0, b1, F1, 1, F0, 0, 64, a0 = *reinterpret_cast<const f16x8*>(apk + 64),
                                         a1 = *reinterpret_cast<const f16x8*>(apk + 256 + 64))
                            PSH_EMX_STEP(b0, b1, F2, 2, F1, 1, 64, (void)0, (void)0)
                        }
                        bpk += 64;
                        apk += 64;
                    }
#undef PSH_EMX_STEP
                } else {
#pragma unroll 1
                    for (int ks = 0; ks < dm.KS; ++ks) {
                        f16x8 F[NG][4];
                        const f16x8 a0 = *reinterpret_cast<const f16x8*>(ap + 32 * ks), a1 = *reinterpret_cast<const f16x8*>(ap + 256 + 32 * ks);
#pragma unroll
                        for (int g = 0; g < NG; ++g) ldB(F[g], g, ks);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int g = 0; g < NG; ++g) mm(F[g], g, a0, a1);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            } else {
                const int cpy = (-scol) & 3;                                            // the copy whose fragment is 8-byte aligned for this shift
                const _Float16* bhc = bh + (size_t)cpy * dm.CS + (PSH_EMX_PADL - scol - cpy) + 8 * kq;
                const _Float16* blc = bl + (size_t)cpy * dm.CS + (PSH_EMX_PADL - scol - cpy) + 8 * kq;
                const int arow = 16 * (32 * hf + scol) + 8 * kq;                        // A row of M tile 0: m = 32 hf + (lane & 15)
                // B fragments of a group of 4 rows: hi and lo, two 8-byte reads each
                auto load_b = [&](f16x4v (&fb)[4][4], int g, int ks) {
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) {
                        const _Float16* ph = bhc + (size_t)(4 * g + r4) * 4 * dm.CS + 32 * ks;
                        const _Float16* pl = blc + (size_t)(4 * g + r4) * 4 * dm.CS + 32 * ks;
                        fb[r4][0] = *reinterpret_cast<const f16x4v*>(ph);
                        fb[r4][1] = *reinterpret_cast<const f16x4v*>(ph + 4);
                        if (NP == 3) {
                            fb[r4][2] = *reinterpret_cast<const f16x4v*>(pl);
                            fb[r4][3] = *reinterpret_cast<const f16x4v*>(pl + 4);
                        }
                    }
                };
                // the 24 MFMAs of a group: the three products of a tile are 8 instructions apart (no back-to-back dependence)
                auto mma_group = [&](const f16x4v (&fb)[4][4], int g, const f16x8& ah0, const f16x8& ah1, const f16x8& al0, const f16x8& al1) {
                    f16x8 bhf[4], blf[4];
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) {
                        // (a concatenation of register pairs, not eight element moves: built element by element the fragments cost
                        //  ~5700 VALU instructions per segment, 1.5 ms of a 3.5 ms scan)
                        bhf[r4] = __builtin_shufflevector(fb[r4][0], fb[r4][1], 0, 1, 2, 3, 4, 5, 6, 7);
                        if (NP == 3) blf[r4] = __builtin_shufflevector(fb[r4][2], fb[r4][3], 0, 1, 2, 3, 4, 5, 6, 7);
                    }
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) {
                        C[0][4 * g + r4] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah0, bhf[r4], C[0][4 * g + r4], 0, 0, 0);
                        C[1][4 * g + r4] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah1, bhf[r4], C[1][4 * g + r4], 0, 0, 0);
                    }
                    if (NP == 3) {
#pragma unroll
                        for (int r4 = 0; r4 < 4; ++r4) {
                            C[0][4 * g + r4] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al0, bhf[r4], C[0][4 * g + r4], 0, 0, 0);
                            C[1][4 * g + r4] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al1, bhf[r4], C[1][4 * g + r4], 0, 0, 0);
                        }
#pragma unroll
                        for (int r4 = 0; r4 < 4; ++r4) {
                            C[0][4 * g + r4] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah0, blf[r4], C[0][4 * g + r4], 0, 0, 0);
                            C[1][4 * g + r4] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah1, blf[r4], C[1][4 * g + r4], 0, 0, 0);
                        }
                    }
                };
#pragma unroll 1
                for (int ks = 0; ks < dm.KS; ++ks) {
                    const int p0 = emx_pad(arow + 32 * ks), p1 = emx_pad(arow + 256 + 32 * ks);
                    const f16x8 ah0 = *reinterpret_cast<const f16x8*>(yh + p0), ah1 = *reinterpret_cast<const f16x8*>(yh + p1);
                    f16x8 al0 = ah0, al1 = ah1;
                    if (NP == 3) { al0 = *reinterpret_cast<const f16x8*>(yl + p0); al1 = *reinterpret_cast<const f16x8*>(yl + p1); }
#pragma unroll
                    for (int g = 0; g < NG; ++g) {                                      // (rows >= d of the last group: zero copies)
                        f16x4v fb[4][4];
                        load_b(fb, g, ks);
                        mma_group(fb, g, ah0, ah1, al0, al1);
                    }
                }
            }
            tstamp(1);
            // ---- back to the data's units; ||H||^2 per window.  Slot (mt, r) of the lane: window 16 (32 hf + 16 mt + 4 kq + r) + scol
            float nh[8];
#pragma unroll
            for (int sl = 0; sl < 8; ++sl) nh[sl] = 0.0f;
            if constexpr (QM) {
                // the accumulators stay in the product's units (inv is a power of two: the energies are scaled instead, exactly) and
                // are squared as the register pairs lie -- (r0, r1), (r2, r3) of a tile -- with packed fma: 48 instructions where
                // scaling + squaring element by element cost 96 v_pk_mul + 48 v_pk_fma + 184 moves that gathered the pairs
                f32x2v np[2][2];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) { np[mt][0] = f32x2v{0.f, 0.f}; np[mt][1] = f32x2v{0.f, 0.f}; }
#pragma unroll
                for (int i = 0; i < 4 * NG; ++i) {                                     // (rows >= d: zero copies of the kernel -> exact zeros)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) {
                        const f32x2v lo2 = __builtin_shufflevector(C[mt][i], C[mt][i], 0, 1), hi2 = __builtin_shufflevector(C[mt][i], C[mt][i], 2, 3);
                        np[mt][0] = __builtin_elementwise_fma(lo2, lo2, np[mt][0]);
                        np[mt][1] = __builtin_elementwise_fma(hi2, hi2, np[mt][1]);
                    }
                }
#pragma unroll
                for (int sl = 0; sl < 8; ++sl) nh[sl] = (np[sl >> 2][(sl & 3) >> 1][sl & 1] * inv) * inv;
            } else {
#pragma unroll
                for (int i = 0; i < 4 * NG; ++i) {                                     // (rows >= d: zero copies of the kernel -> exact zeros)
#pragma unroll
                    for (int sl = 0; sl < 8; ++sl) { C[sl >> 2][i][sl & 3] *= inv; nh[sl] = __builtin_fmaf(C[sl >> 2][i][sl & 3], C[sl >> 2][i][sl & 3], nh[sl]); }
                }
            }
            float nh_raw[8];
#pragma unroll
            for (int sl = 0; sl < 8; ++sl) nh_raw[sl] = nh[sl];
#pragma unroll
            for (int sl = 0; sl < 8; ++sl) {                                            // an inadmissible window never wins, never survives
                const int p = 16 * (32 * hf + 16 * (sl >> 2) + 4 * kq + (sl & 3)) + scol;
                nh[sl] = (seg_start + p < a.Tp) ? nh[sl] : __uint_as_float(PSH_INF_BITS);
            }
            if constexpr (QM) {
                // ---- the per-query pass on the matrix cores
                // the scale of the accumulators: |H_i| <= ymax max_i ||ker_i||_1 -> [8192, 32768) in the product's units (f16 keeps 11 bits down to 6e-5)
                // (sc = 1 / inv: the accumulators are the A fragments' values already)
                int ec = ey + ekc;
                const bool clamped = ec > 100 || ec < -100;                             // (fp32's ends: such a half segment keeps every window)
                ec = ec > 100 ? 100 : (ec < -100 ? -100 : ec);
                const float sc = __uint_as_float((unsigned)(127 + ec) << 23);
                wave_lds_fence();                                                       // the half before this one has read nhL
                // a non-finite accumulator (non-finite data) spoils the D values of the three other windows of its A row (0 * NaN):
                // such a half segment takes the careful path for every group (a NaN fails '>' and is kept)
                bool bad0 = false;
#pragma unroll
                for (int sl = 0; sl < 8; ++sl) bad0 = bad0 || !(nh_raw[sl] < __uint_as_float(PSH_INF_BITS));
                // one scale for the 4 x 4 queries of every group:  v = nt + kk D,  kk = -2 / (s sc) < 0  ->  D' = nt / kk + D  with nt / kk
                // the C operand of the slot's first MFMA;  v > thr  <=>  D' < thr / kk, the smallest v the largest D'.  (The MFMA adds
                // nt / kk in fp32 like any partial sum: the (1 - 2^-20) on the energies covers 16 such roundings.)
                const float inv_kk = (-0.5f * sqc) * sc;
                const bool badC = __any(bad0) || clamped || eqc + ec > 120 || eqc + ec < -120;
                f16x8 A2[8][(4 * NG > 8) ? 2 : 1];
                {
                    const float nf = ((1.0f - eps2 - 1.0f / 4096.0f) * (1.0f - 1.0f / 1048576.0f)) * inv_kk;
#pragma unroll
                    for (int sl = 0; sl < 8; ++sl) {
                        nhL[sl * 64 + kq * 16 + scol] = nh[sl] * nf;
#pragma unroll
                        for (int h2 = 0; h2 < ((4 * NG > 8) ? 2 : 1); ++h2)
#pragma unroll
                            for (int j = 0; j < 8; ++j)
                                A2[sl][h2][j] = (8 * h2 + j < 4 * NG) ? (_Float16)C[sl >> 2][(8 * h2 + j < 4 * NG) ? 8 * h2 + j : 0][sl & 3] : (_Float16)0.0f;
                    }
                }
                wave_lds_fence();
                const int ncol = lane & 15, gq = lane >> 4;                             // D layout: column (kq', q) = ncol, rows 4 gq + rr = s
                const int kqp = ncol >> 2;
#pragma unroll 1
                for (int Q4 = q_begin & ~3; Q4 < q_end; Q4 += 4) {
                    const int qn = Q4 + (ncol & 3);
                    const bool qv = qn >= q_begin && qn < q_end;
                    f16x8 B2[2];
                    const f16x8 z8 = {(_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f};
                    const bool mine2 = qv && gq == kqp;                                 // K chunk gq of column (kq', q): query q's coordinates iff gq == kq'
                    B2[0] = mine2 ? *reinterpret_cast<const f16x8*>(qh + (size_t)qn * 16) : z8;
                    B2[1] = mine2 ? *reinterpret_cast<const f16x8*>(qh + (size_t)qn * 16 + 8) : z8;
                    const f32x4 qcv = qc[qv ? qn : q_begin];
                    // D' of the 8 slots: energies in, the slots' first MFMAs, then their second ones (independent accumulators back to back)
                    f32x4 D[8];
#pragma unroll
                    for (int sl = 0; sl < 8; ++sl) D[sl] = *reinterpret_cast<const f32x4*>(nhL + sl * 64 + kqp * 16 + 4 * gq);
                    const float st2 = qcv[0] + Rad;
                    const float thr = qv ? (st2 * st2 * (1.0f + 1.0f / 16384.0f) - qcv[1]) + thrG : -__uint_as_float(PSH_INF_BITS);
                    const float thrp = clamped ? -__uint_as_float(PSH_INF_BITS) : thr * inv_kk;   // (a query that is not there: +inf -- nothing is above it)
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int sl = 0; sl < 8; ++sl) D[sl] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A2[sl][0], B2[0], D[sl], 0, 0, 0);
                    if (4 * NG > 8) {
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int sl = 0; sl < 8; ++sl) D[sl] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A2[sl][(4 * NG > 8) ? 1 : 0], B2[1], D[sl], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    float vm = -__uint_as_float(PSH_INF_BITS);
#pragma unroll
                    for (int sl = 0; sl < 8; ++sl) vm = max3f(vm, max3f(D[sl][0], D[sl][1], D[sl][2]), D[sl][3]);
                    if (!badC && !__any(qv && !(vm < thrp))) continue;                  // the common case: nothing of these 4 queries here
                    unsigned hm = 0u;                                                   // bit 4 sl + rr: the window survives (a NaN fails '<' and is kept)
#pragma unroll
                    for (int sl = 0; sl < 8; ++sl)
#pragma unroll
                        for (int rr = 0; rr < 4; ++rr) hm |= !(D[sl][rr] < thrp) ? (1u << (4 * sl + rr)) : 0u;
                    if (!qv) hm = 0u;
                    while (__any(hm != 0u)) {
                        const bool has0 = hm != 0u;
                        const int bit = has0 ? (int)__builtin_ctz(hm) : 0;
                        hm &= hm - 1u;
                        const int sl = bit >> 2, rr = bit & 3;
                        const int p = 16 * (32 * hf + 16 * (sl >> 2) + 4 * kqp + (sl & 3)) + 4 * gq + rr;
                        const bool has = has0 && (seg_start + p < a.Tp);                // (an infinite threshold -- non-finite data -- keeps inadmissible windows too)
                        const unsigned long long sm = __ballot(has);
                        if (!sm) continue;
                        const int ne = __popcll(sm);
                        if (nsq + ne > PSH_EMX_QCAP) verify();
                        if (has) {
                            const int slot = nsq + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(sm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)sm, 0u));
                            sq[slot] = (unsigned)p | ((unsigned)qn << 12);
                            squ[slot] = (unsigned)__builtin_amdgcn_readfirstlane((int)rs);
                        }
                        nsq += ne;
                    }
                }
            } else {
            // the query's coordinates and level come through the scalar cache, one query AHEAD of their use (one wave per SIMD:
            // nothing else would hide the load)
            float hnx[4 * NG];
            unsigned taunx = 0u;
            auto fetch_query = [&](int bq) {
#pragma unroll
                for (int i = 0; i < 4 * NG; ++i) hnx[i] = hxk[(int64_t)bq * d + (perm_s[i] < d ? perm_s[i] : 0)];
                if (MODE == PSH_MODE_FILTER) taunx = qstate_k[bq].tau2_bits;
            };
            if (q_begin < q_end) fetch_query(q_begin);
#pragma unroll 1
            for (int b = q_begin; b < q_end; ++b) {
                float hxs[4 * NG];
                float nx = 0.0f;
#pragma unroll
                for (int i = 0; i < 4 * NG; ++i) {
                    hxs[i] = perm_s[i] < d ? hnx[i] : 0.0f;
                    asm volatile("" : "+v"(hxs[i]));                                    // a VGPR copy: an fma with an SGPR operand issues at ~0.6 of the rate (ubench_dot2)
                    nx = __builtin_fmaf(hxs[i], hxs[i], nx);
                }
                const unsigned tau_bits = taunx;
                if (b + 1 < q_end) fetch_query(b + 1);
                // acc^ - nx = nh - 2 sum_i hx_i H_i  (nh of an inadmissible window is +inf: it never wins a minimum, never survives);
                // 13 operations per window and query -- the minimum alone is kept, the rare wave that holds a survivor goes
                // through its slots again
                auto vslot = [&](int sl) -> float {
                    float c = 0.0f;
#pragma unroll
                    for (int i = 0; i < 4 * NG; ++i) c = __builtin_fmaf(hxs[i], C[sl >> 2][i][sl & 3], c);
                    return __builtin_fmaf(-2.0f, c, nh[sl]);
                };
                float mn = vslot(0);
#pragma unroll
                for (int sl = 1; sl < 8; ++sl) mn = fminf(mn, vslot(sl));
                if (MODE == PSH_MODE_BOOT) {
#pragma unroll
                    for (int off = 32; off > 0; off >>= 1) mn = fminf(mn, __shfl_xor(mn, off, 64));
                    // one minimum per HALF segment (the launch plan's boot_per_wave == 2): an upper bound of the exact acc of
                    // the half's best window
                    float m2 = mn + nx;
                    m2 = m2 > 0.0f ? m2 : 0.0f;                                         // (cancellation can leave a tiny negative)
                    const float su = __builtin_sqrtf(m2) * (1.0f + 1.0f / 32768.0f) + Rad;
                    const float ub = su * su * (1.0f + 1.0f / 16384.0f);
                    if (lane == 0) a.minbuf[(int64_t)b * a.min_stride + 2 * (int64_t)rs + hf] = ub;
                } else {
                    const float tau = __uint_as_float(tau_bits);
                    const float st2 = __builtin_sqrtf(tau) * (1.0f + 1.0f / 32768.0f) + Rad;
                    const float thr = st2 * st2 * (1.0f + 1.0f / 16384.0f) - nx * (1.0f - 1.0f / 4194304.0f);   // compared with acc^ - nx (rounded towards "keep")
                    if (!__any(!(mn > thr))) continue;                                  // the common case: nothing of this query here
                    unsigned hm = 0u;
#pragma unroll
                    for (int sl = 0; sl < 8; ++sl) hm |= !(vslot(sl) > thr) ? (1u << sl) : 0u;
                    while (__any(hm != 0u)) {
                        const bool has0 = hm != 0u;
                        const int sl = has0 ? (int)__builtin_ctz(hm) : 0;
                        hm &= hm - 1u;
                        const int p = 16 * (32 * hf + 16 * (sl >> 2) + 4 * kq + (sl & 3)) + scol;
                        const bool has = has0 && (seg_start + p < a.Tp);                // (an infinite threshold -- non-finite data -- keeps inadmissible windows too)
                        const unsigned long long sm = __ballot(has);
                        if (!sm) continue;
                        const int ne = __popcll(sm);
                        if (nsq + ne > PSH_EMX_QCAP) verify();
                        if (has) {
                            const int slot = nsq + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(sm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)sm, 0u));
                            sq[slot] = (unsigned)p | ((unsigned)b << 12);
                            squ[slot] = (unsigned)__builtin_amdgcn_readfirstlane((int)rs);
                        }
                        nsq += ne;
                    }
                }
            }
            }
            tstamp(2);
        }
#ifdef PSH_TUNING
        tcount[0] += (unsigned)(un < u_hi ? (nsq & ~3) : nsq); tcount[1] += (unsigned)(un < u_hi ? (nsq >> 2) : ((nsq + 3) >> 2));
#endif
        // (the accumulators are dead here: the register-hungry fast chain; whole passes of four while more units follow)
        if (MODE == PSH_MODE_FILTER && (un < u_hi ? nsq >= 4 : nsq > 0)) verify_end(un < u_hi);
        if (LATE) {
            asm volatile("" ::"v"(touched));                                                               // (the touch has landed: one register across the unit)
            load_unit(st, un < u_hi ? un : u);             // (unconditionally: a segment kept "as it was" would be live across the whole unit)
        }
        tstamp(3);
        wave_lds_fence();
        u = un;
    }
#ifdef PSH_TUNING
    if (MODE == PSH_MODE_FILTER && a.dbg_times && lane == 0)
    {
        for (int i = 0; i < 4; ++i) a.dbg_times[((size_t)blockIdx.x * NW + wave) * 6 + i] = tacc[i];
        for (int i = 0; i < 2; ++i) a.dbg_times[((size_t)blockIdx.x * NW + wave) * 6 + 4 + i] = tcount[i];
    }
#endif
    if (MODE == PSH_MODE_FILTER) {
        if (npend > 0) pend_flush(pend, npend, lcount, a, lane);
        __syncthreads();
        for (int q = tid; q < a.B; q += PSH_EMX_THREADS) a.bcount[(int64_t)q * PSH_MAX_BLOCKS + blockIdx.x] = lcount[q];
    }
}

bool embed_mx_supported(int d, int K, int B, int tile_floats) {
    return d >= 1 && d <= PSH_EMX_MAX_D && K >= 1 && K <= 256 && emx_shmem_bytes(K, d, B, tile_floats) <= PSH_LDS_BYTES;
}

template <bool ALIGNED, int MODE, int NG, int NP, bool QM>
static hipError_t launch_emx_ng(const ScanArgs& a, int grid, hipStream_t s) {
    const size_t shmem = emx_shmem_bytes(a.W, a.emb_d, a.B, a.tile_floats);
    hipError_t e = hipFuncSetAttribute((const void*)embed_mx_kernel<ALIGNED, MODE, NG, NP, QM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((embed_mx_kernel<ALIGNED, MODE, NG, NP, QM>), dim3(grid), dim3(PSH_EMX_THREADS), shmem, s, a);
    return hipGetLastError();
}

template <bool ALIGNED, int MODE, int NP, bool QM>
static hipError_t launch_emx(const ScanArgs& a, int grid, hipStream_t s) {
    const int ng = (a.emb_d + 3) >> 2;
    return ng == 1 ? launch_emx_ng<ALIGNED, MODE, 1, NP, QM>(a, grid, s) : ng == 2 ? launch_emx_ng<ALIGNED, MODE, 2, NP, QM>(a, grid, s)
                                                                                  : launch_emx_ng<ALIGNED, MODE, 3, NP, QM>(a, grid, s);
}

// the bootstrap with the split products; the full scan with one product and -- 3 .. 256 queries -- the per-query pass on the
// matrix cores too, or (a.emb_mx == 3: PSH_FLAG_EMBED_MX_SPLIT) the split and the vector ALUs as in the bootstrap
hipError_t launch_embed_mx(const ScanArgs& a0, int mode, bool aligned, int grid, hipStream_t s) {
    ScanArgs a = a0;
#ifdef PSH_TUNING
    if (const char* e = getenv("PSH_DBG_TIMES_PTR")) a.dbg_times = (unsigned long long*)strtoull(e, nullptr, 0);   // tools/emx_phases.py
#endif
    if (mode == PSH_MODE_BOOT) return aligned ? launch_emx<true, PSH_MODE_BOOT, 3, false>(a, grid, s) : launch_emx<false, PSH_MODE_BOOT, 3, false>(a, grid, s);
    if (a.emb_mx == 3) return aligned ? launch_emx<true, PSH_MODE_FILTER, 3, false>(a, grid, s) : launch_emx<false, PSH_MODE_FILTER, 3, false>(a, grid, s);
    if (a.B >= PSH_EMX_QM_MIN_B && a.B <= PSH_EMX_QM_MAX_B)
        return aligned ? launch_emx<true, PSH_MODE_FILTER, 1, true>(a, grid, s) : launch_emx<false, PSH_MODE_FILTER, 1, true>(a, grid, s);
    return aligned ? launch_emx<true, PSH_MODE_FILTER, 1, false>(a, grid, s) : launch_emx<false, PSH_MODE_FILTER, 1, false>(a, grid, s);
}

}  // namespace psh
