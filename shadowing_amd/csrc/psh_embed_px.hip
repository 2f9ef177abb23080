// psh_embed_px.hip -- the embedded scan for suffix-rows kernels whose supports form ONE interval (Foveal itself:
// reference path_embedding.py:142-172), BOOT and FILTER stages: the running sums of the rows as differences of PREFIX
// SUMS of the segment.  Part of libpsh_hip.so; the general suffix-rows pass (tap walk, kernels with a gap) and the dense
// chains are embed_scan_kernel's (psh_embed.hip), which this kernel replaces when the structure is there.
//
// Row i of such a kernel is one constant c_i on the taps [a_i, ktop) and zero elsewhere, so for window t
//     h_i(t) = c_i S_i(t),   S_i(t) = E[t + ktop] - E[t + a_i],   E[m] = y_0 + .. + y_{m-1}   (of the segment's samples)
// and the cheap embedding costs ONE LDS read and three packed-able operations per row, window and query
//     S = Pk - E[t + a_i];  e = hx_i - c_i S;  acc^ += e^2
// instead of one packed add per TAP and window (115 taps against 34 rows -- 27 distinct -- for Foveal(1.15, 0.9, 126)).
//   * E is a scan over the staged registers in fp32 (4 samples per lane serially, the lane totals through six DPP steps, the
//     five 256-sample blocks chained through a scalar double): an entry is off by at most 16 u A, A = the segment's sum
//     of |y| (prefix_store), so  |S^_i - S_i| <= 32 u A.  (Round 2 summed in double and rounded once -- 2 u max|E| --
//     for 260 of the kernel's 1220 vector instructions per segment; the radius below is 0.5 % of sqrt(tau) either way.)
//   * IDENTICAL rows (same a_i, same c_i: Foveal's short scales repeat) are merged, up to 4 to a group:
//     sum_j (hx_j - c S)^2 = m (mean hx - c S)^2 + V  -- one row with c' = sqrt(m) c and h' = sqrt(m) mean hx; V >= 0 is
//     dropped by the rejection test (conservative) and added back by the bootstrap's upper bounds.
//   * Window w of a lane is  lane + 64 w : every LDS read is 64 consecutive floats (no bank conflict) and a register PAIR
//     holds two windows 64 apart (ds_read2st64_b32), ready for v_pk_add_f32 / v_pk_fma_f32.
// Bound-then-verify like the tap walk (psh_embed.hip): with the exact chain's own n_i^2 u |c_i| ymax, the rounding of c'
// and h' (<= 2 u n_i |c_i| ymax, 6 u ||hx||) the cheap embedding lies within
//     Rad = ymax * u sqrt(sum_i (c_i (n_i + 2)^2)^2) + A * 32 u ||c||_2 + 6 u ||hx||_2            (each with a 5 % margin)
// of the exact one; a window survives unless  acc^ > (sqrt(tau)(1 + 2^-15) + Rad)^2 (1 + 2^-14);  survivors get the exact
// dense chain in the oracle's order (oracle/psh_oracle.c: embedded_acc) -- their rows spread evenly over the lanes by the plan,
// their samples re-read from global memory (the tile holds E) --, and only exact values are ever ranked: results are bit-identical to the tap walk's and the dense chains'.
// Non-finite data: an infinite A / ymax makes the threshold infinite, a NaN in E fails every '>' -- either way the
// windows are verified exactly.
//
// The structure is recognised ON THE DEVICE (embed_plan_kernel, one small launch per call: the library cannot look at the
// matrix without a synchronisation): the plan it leaves in the workspace says which of the two kernels launched for a
// stage does the work -- the other one returns at once.
#include <type_traits>

#include "psh_device.h"

namespace psh {

typedef float f32x2 __attribute__((ext_vector_type(2)));

// ---- the plan: one block looks at the d x K matrix -----------------------------------------------------------------------
// (everything in LDS, rows across the threads: the launch is in front of every sampled call, ~10 us)
#define PSH_PLAN_THREADS 256
__global__ __launch_bounds__(PSH_PLAN_THREADS) void embed_plan_kernel(const float* __restrict__ ker, int d, int K, EmbedPlan* plan) {
    extern __shared__ __attribute__((aligned(16))) float s_ker[];        // d x K
    __shared__ int s_ok, s_ngroups;
    __shared__ unsigned long long s_U[4];
    __shared__ int4 s_row[PSH_EMB_MAX_D];                    // {first tap, row, c bits, taps}
    __shared__ int4 s_prog[PSH_EMB_MAX_D];                   // the same, longest support last
    __shared__ int s_rep[PSH_EMB_MAX_D];                     // rank among the identical rows before it
    __shared__ int4 s_grp[PSH_EMB_MAX_D];                    // merged rows in the order they were found
    __shared__ float s_gw[PSH_EMB_MAX_D], s_gws[PSH_EMB_MAX_D];   // their weights; the weights in ranked order
    __shared__ float s_e2[PSH_EMB_MAX_D], s_c2[PSH_EMB_MAX_D];
    const int tid = (int)threadIdx.x;
    if (d > PSH_EMB_MAX_D || K > 256) { if (tid == 0) { plan->contig = 0; plan->ngroups = 0; } return; }
    for (int e = tid; e < d * K; e += PSH_PLAN_THREADS) s_ker[e] = ker[e];
    if (tid == 0) { s_ok = 1; s_ngroups = 0; s_U[0] = s_U[1] = s_U[2] = s_U[3] = 0ull; }
    __syncthreads();
    // support mask, size, constant of every row: a wave per row, the taps across its lanes (64 per ballot)
    __shared__ unsigned long long s_m[PSH_EMB_MAX_D][4];
    {
        const int lane = tid & 63, wv = tid >> 6;
        for (int r = wv; r < d; r += PSH_PLAN_THREADS / 64) {
            const float* row = s_ker + (size_t)r * K;
            unsigned long long mm[4];
            float v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                v[q] = (64 * q + lane < K) ? row[64 * q + lane] : 0.0f;
                mm[q] = __ballot(v[q] != 0.0f);              // (NaN included: it then fails the comparison with c below)
            }
            int nn = 0, low = 1 << 20, top = -1;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (mm[q]) {
                    if (low == (1 << 20)) low = 64 * q + (int)__builtin_ctzll(mm[q]);
                    top = 64 * q + 63 - (int)__builtin_clzll(mm[q]);
                    nn += (int)__popcll(mm[q]);
                }
            unsigned cb = 0u;                                // the constant: the value of the last non-zero tap
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (top >= 64 * q && top < 64 * q + 64) cb = (unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(v[q]), top & 63);
            bool bad = false;
#pragma unroll
            for (int q = 0; q < 4; ++q) bad = bad || (v[q] != 0.0f && __float_as_uint(v[q]) != cb) || (v[q] != v[q]);
            const bool okc = !__any(bad) && (fabsf(__uint_as_float(cb)) <= 3.0e38f);
            if (lane == 0) {
#pragma unroll
                for (int q = 0; q < 4; ++q) { s_m[r][q] = mm[q]; if (mm[q]) atomicOr(&s_U[q], mm[q]); }
                s_row[r] = make_int4(low, r, (int)cb, nn);
                if (!okc) atomicAnd(&s_ok, 0);
            }
        }
    }
    __syncthreads();
    unsigned long long m[4] = {0ull, 0ull, 0ull, 0ull};
    int n = 0, lowest = 1 << 20;
    float c = 0.0f;
    if (tid < d) {
        const int4 o = s_row[tid];
        lowest = o.x; n = o.w; c = __uint_as_float((unsigned)o.z);
#pragma unroll
        for (int q = 0; q < 4; ++q) m[q] = s_m[tid][q];
    }
    // the union of the supports: one interval [lowU, topU)?
    int lowU = -1, topU = 0, nU = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const unsigned long long uw = s_U[q];
        if (uw) {
            if (lowU < 0) lowU = 64 * q + (int)__builtin_ctzll(uw);
            topU = 64 * q + 64 - (int)__builtin_clzll(uw);
            nU += (int)__popcll(uw);
        }
    }
    if (tid < d) {                                           // the row must be all of U from its first tap up
        bool oks = true;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int lo_bit = lowest - 64 * q;
            const unsigned long long keep = lo_bit <= 0 ? ~0ull : (lo_bit >= 64 ? 0ull : (~0ull << lo_bit));
            oks = oks && (m[q] == (s_U[q] & keep));
        }
        if (n == 0) oks = true;
        if (!oks) atomicAnd(&s_ok, 0);
        int rk = 0;                                          // short supports first (the schedule below takes them longest first)
        for (int i2 = 0; i2 < d; ++i2) { const int l2 = s_row[i2].x; rk += (l2 > lowest || (l2 == lowest && i2 < tid)) ? 1 : 0; }
        s_prog[rk] = make_int4(lowest, tid, (int)__float_as_uint(c), n);
    }
    __syncthreads();
    // merged rows: identical rows (same constant, same first tap) in groups of up to 4, a group where its first row stands
    int4 me = make_int4(0, 0, 0, 0);
    int off = 0, rep = 0;
    if (tid < d) {
        me = s_prog[tid];
        off = 4 * (me.w > 0 ? me.x : topU);                  // byte offset of E[a_i] (E[ktop] for an empty row: S = 0)
        for (int j = 0; j < tid; ++j) { const int4 o = s_prog[j]; rep += (o.z == me.z && 4 * (o.w > 0 ? o.x : topU) == off) ? 1 : 0; }
        s_rep[tid] = rep;
        const float n2 = (float)(me.w + 2);
        const float t = fabsf(__uint_as_float((unsigned)me.z)) * n2 * n2;
        s_e2[tid] = t * t;
        s_c2[tid] = __uint_as_float((unsigned)me.z) * __uint_as_float((unsigned)me.z);
    }
    __syncthreads();
    if (tid < d && (rep & 3) == 0) {                         // a group starts here: its index = the starts before it
        int g = 0;
        for (int j = 0; j < tid; ++j) g += (s_rep[j] & 3) == 0 ? 1 : 0;
        int members = me.y, cnt = 1;
        for (int j = tid + 1; j < d; ++j) {
            const int4 o = s_prog[j];
            if (o.z == me.z && 4 * (o.w > 0 ? o.x : topU) == off && s_rep[j] > rep && s_rep[j] < rep + 4) { members |= o.y << (8 * (s_rep[j] - rep)); ++cnt; }
        }
        const float rm = cnt == 1 ? 1.0f : (cnt == 2 ? 1.41421356f : (cnt == 3 ? 1.7320508f : 2.0f));
        const float cm = __fmul_rn(__uint_as_float((unsigned)me.z), rm);
        s_grp[g] = make_int4((int)__float_as_uint(cm), off, members, cnt);
        // what the group adds to the squared distance of an unrelated window, up to the data's variance: c'^2 x taps
        s_gw[g] = (me.w > 0 && fabsf(cm) < 1.0e18f) ? cm * cm * (float)me.w : 0.0f;
        atomicAdd(&s_ngroups, 1);
    }
    __syncthreads();
    // The scan walks the merged rows HEAVIEST FIRST and stops early for windows whose partial sum of squares is already
    // above the threshold (embed_px_kernel): rank by weight, ties by position; r1 = the rows of the first phase -- the
    // smallest even count that carries 85 % of the total weight (kernels with fewer than 8 merged rows: no early exit).
    if (tid < s_ngroups) {
        const float w = s_gw[tid];
        int rk = 0;
        for (int j = 0; j < s_ngroups; ++j) { const float wj = s_gw[j]; rk += (wj > w || (wj == w && j < tid)) ? 1 : 0; }
        plan->gtab[rk] = s_grp[tid];
        s_gws[rk] = w;
    }
    __syncthreads();
    if (tid == 0) {
        const int G = s_ngroups;
        float tot = 0.0f;
        for (int j = 0; j < G; ++j) tot += s_gws[j];
        int r1 = G;
        if (G >= 8 && tot > 0.0f) {
            float cum = 0.0f;
            for (int j = 0; j < G; ++j) {
                cum += s_gws[j];
                if (cum >= 0.85f * tot && ((j + 1) & 1) == 0) { r1 = j + 1; break; }
            }
            if (r1 > G - 4) r1 = G;                          // nothing worth a second phase
        }
        plan->r1 = r1;
    }
    // the verification's schedule: rows longest first, each onto the lane slot with the fewest taps so far (wave 0 -- one
    // DPP minimum per row --, the rest across the threads)
    __shared__ int s_bin[PSH_EMB_MAX_D], s_t4[PSH_EMB_MAX_D], s_tot, s_long, s_nl, s_lmax;
    if (tid == 0) { s_tot = 0; s_long = 1; }
    __syncthreads();
    int my_t4 = 0;
    if (tid < d) {                                           // the dense chain's span of a row, in whole groups of 4 taps (an empty row: one)
        const int n = me.w > 0 ? topU - (me.x & ~3) : 0;
        my_t4 = n > 0 ? (n + 3) >> 2 : 1;
        s_t4[tid] = my_t4;
        atomicAdd(&s_tot, my_t4);
        atomicMax(&s_long, my_t4);
    }
    __syncthreads();
    if (tid < 64) {
        int NL = (s_tot + s_long - 1) / s_long;
        NL = NL < 1 ? 1 : (NL > 64 ? 64 : NL);
        int load = 0;
        for (int i = d - 1; i >= 0; --i) {                   // (s_prog: short supports first)
            int key = tid < NL ? ((load << 6) | tid) : 0x7fffffff;        // the least loaded slot: a minimum over the wave on the DPP path
            { int k2 = __builtin_amdgcn_update_dpp(key, key, 0x111, 0xf, 0xf, false); key = k2 < key ? k2 : key; }
            { int k2 = __builtin_amdgcn_update_dpp(key, key, 0x112, 0xf, 0xf, false); key = k2 < key ? k2 : key; }
            { int k2 = __builtin_amdgcn_update_dpp(key, key, 0x114, 0xf, 0xf, false); key = k2 < key ? k2 : key; }
            { int k2 = __builtin_amdgcn_update_dpp(key, key, 0x118, 0xf, 0xf, false); key = k2 < key ? k2 : key; }
            { int k2 = __builtin_amdgcn_update_dpp(key, key, 0x142, 0xa, 0xf, false); key = k2 < key ? k2 : key; }
            { int k2 = __builtin_amdgcn_update_dpp(key, key, 0x143, 0xc, 0xf, false); key = k2 < key ? k2 : key; }
            key = __builtin_amdgcn_readlane(key, 63);
            if ((key & 63) == tid) { load += s_t4[i]; s_bin[i] = tid; }
        }
        int lmax = load;
#pragma unroll
        for (int o2 = 32; o2 > 0; o2 >>= 1) { const int v2 = __shfl_xor(lmax, o2, 64); lmax = v2 > lmax ? v2 : lmax; }
        if (tid == 0) { s_nl = NL; s_lmax = lmax; plan->vnl = NL; plan->vmax = 4 * lmax; }
    }
    __syncthreads();
    if (tid < 66) {                                          // where a slot's rows start: the rows of the slots before it
        int w = 0;
        for (int i = 0; i < d; ++i) w += s_bin[i] < tid ? 1 : 0;
        plan->vstart[tid] = w;
    }
    if (tid < d) {                                           // a slot's rows longest first
        const int b = s_bin[tid];
        int w = 0;
        for (int i = 0; i < d; ++i) w += (s_bin[i] < b || (s_bin[i] == b && i > tid)) ? 1 : 0;
        const int lo = me.w > 0 ? (me.x & ~3) : 0, ath = me.w > 0 ? me.x : 0;
        plan->vrow[w] = make_int2(lo | (my_t4 << 8) | (ath << 16) | (me.y << 24), me.z);
    }
    __syncthreads();
    if (tid == 0) {
        float e2p = 0.0f, c2s = 0.0f;
        for (int i = 0; i < d; ++i) { e2p += s_e2[i]; c2s += s_c2[i]; }
        e2p *= 1.0001f; c2s *= 1.0001f;                      // (the order of this sum does not matter to the margin)
        const int G = s_ngroups;
        // (at most 64 merged rows: a lane of the scan per row; larger kernels keep the tap walk)
        const bool ok = s_ok != 0 && nU > 0 && topU - lowU == nU && e2p < 3.0e38f && c2s < 3.0e38f && G <= 64;
        plan->gtab[G] = make_int4(0, 0, 0, 0);
        plan->ktop = topU;
        plan->ngroups = G;
        plan->d = d;
        plan->cerr_y = 1.05f * 5.9604645e-8f * __builtin_sqrtf(e2p);
        plan->cerr_p = 1.05f * 2.0f * 5.9604645e-8f * __builtin_sqrtf(c2s);
        plan->contig = ok ? 1 : 0;
    }
}

hipError_t launch_embed_plan(const float* ker, int d, int K, EmbedPlan* plan, hipStream_t s) {
    const size_t shmem = (size_t)d * K * sizeof(float);
    if (shmem > 48 * 1024) {
        const hipError_t e = hipFuncSetAttribute((const void*)embed_plan_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(embed_plan_kernel, dim3(1), dim3(PSH_PLAN_THREADS), shmem, s, ker, d, K, plan);
    return hipGetLastError();
}

// inc += (inc of the lane CTRL names, 0 where there is none): one step of the wave scan, a v_add_f32 with a DPP operand
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float inc) {
    return inc + __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(inc), CTRL, ROW_MASK, 0xf, true));
}

// Exclusive prefix sums of the staged segment (element 4 (lane + 64 q) + r of Stage) -> dst[0 .. 4 nq]; returns the sum of
// |y| over the staged samples (what bounds the rounding of the sums, below).
// fp32 all the way, except the carry from one 256-sample block to the next (a double: two scalar-rate instructions a block):
// 3 adds inside a lane, six DPP adds across the wave, one subtraction and two adds to place an entry -- every E^[m] is a
// sum of the y_j, j < m, in SOME order with at most PSH_PX_SCAN_OPS roundings on a term's way, so
//     |E^[m] - E[m]| <= PSH_PX_SCAN_OPS u sum_j |y_j|        (u = 2^-24; Higham's bound for any summation order)
// -- 400x the error of sums taken in double and rounded once, and still ~0.5 % of the radius sqrt(tau) the rejection
// test works with at the benchmark sizes; the double-precision scan it replaces was 260 of the kernel's 1220 vector
// instructions per segment, this one is 70.
#define PSH_PX_SCAN_OPS 16
__device__ __forceinline__ float prefix_store(const Stage& st, float* dst, int nfloat, int lane) {
    const int nq = (nfloat + 3) >> 2;
    double carry = 0.0;
    float ab = 0.0f;
#pragma unroll
    for (int q = 0; q < PSH_NSTAGE; ++q) {
        const int m = lane + 64 * q;
        const bool on = q < PSH_NSTAGE - 1 || m < nq;      // (group nq, all zeros, carries E[nfloat] when nfloat % 4 == 0)
        const float v0 = on ? st.v[q][0] : 0.0f, v1 = on ? st.v[q][1] : 0.0f, v2 = on ? st.v[q][2] : 0.0f, v3 = on ? st.v[q][3] : 0.0f;
        ab += (fabsf(v0) + fabsf(v1)) + (fabsf(v2) + fabsf(v3));
        const float d0 = v0, d1 = d0 + v1, d2 = d1 + v2, d3 = d2 + v3;
        // inclusive scan of the lane totals: within the rows of 16 lanes (row_shr 1, 2, 4, 8), then across them
        // (row_bcast15 into rows 1 and 3, row_bcast31 into rows 2 and 3)
        float inc = d3;
        inc = dpp_add<0x111, 0xf>(inc);
        inc = dpp_add<0x112, 0xf>(inc);
        inc = dpp_add<0x114, 0xf>(inc);
        inc = dpp_add<0x118, 0xf>(inc);
        inc = dpp_add<0x142, 0xa>(inc);
        inc = dpp_add<0x143, 0xc>(inc);
        const float x = (float)carry + (inc - d3);          // exclusive: the lane's own total taken off again
        const f32x4 E = f32x4{x, x + d0, x + d1, x + d2};
        carry += (double)__uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(inc), 63));
        if (on || m == nq) *reinterpret_cast<f32x4*>(dst + 4 * m) = E;
    }
    ab = wave_sum_dpp(ab);                                   // (DPP + SGPRs: six dependent ds_bpermute were ~100 cycles each)
    return ab * 1.0001f;                                     // (its own fp32 rounding: ~30 u)
}

#define PSH_PX_DLCAP 320          // row differences per wave: (survivors per verification pass) x d floats
#define PSH_PX_ALIVE 64           // live windows per wave, segment and query that the sparse second phase of the row loop takes
__host__ __device__ inline size_t px_shmem_bytes(int tile_floats, int B, int d, int threads) {
    const int nw = threads / 64;
    const int nbg = threads == 512 ? 6 : 2;                                         // (PSH_PX_WIDE_NBG / PSH_PX_NBG)
    return (size_t)tile_floats * nw * sizeof(float)                                  // wave-private tiles (E)
           + (size_t)(((B + 3) & ~3) + 4) * sizeof(int)                             // per-query append cursors + work cursor
           + (size_t)nw * PSH_PEND * 16                                             // wave-private pending admissions
           + (((size_t)d * 8 + 15) & ~(size_t)15) + 272                             // verification schedule: rows, slot starts
           + (size_t)(d + 1) * 16 + (size_t)(d + 4) * 16                            // merged rows, their {c', c', offset}
           + (size_t)nw * (128 + PSH_PX_DLCAP) * 4                                  // verification scratch: survivor list (+ units), row differences
           + (size_t)nw * (2 * PSH_PX_ALIVE + nbg * 64) * 4;                        // second phase: live windows, coordinates by row
}

// THREADS / NBG: 1024 threads (4 waves per SIMD, 128 VGPRs) with 2 queries per pass over the rows, or -- batches of 7 and
// more -- 512 threads (256 VGPRs) with 6.
template <bool ALIGNED, int MODE, int THREADS, int NBG>
__global__ __launch_bounds__(THREADS) void embed_px_kernel(ScanArgs a) {
    const EmbedPlan* __restrict__ plan = a.plan;
    if (__builtin_amdgcn_readfirstlane(plan->contig) == 0) return;      // not this kernel's structure: embed_scan_kernel works
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = lane_id();
    const int wave_in_block = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    constexpr int NW = THREADS / 64;
    float* tile = smem + (size_t)wave_in_block * a.tile_floats;
    int* lcount = reinterpret_cast<int*>(smem + (size_t)NW * a.tile_floats);
    int* next_unit = lcount + ((a.B + 3) & ~3);
    u32x4* pend0 = reinterpret_cast<u32x4*>(lcount + ((a.B + 3) & ~3) + 4);
    u32x4* pend = pend0 + (size_t)wave_in_block * PSH_PEND;
    const int K = a.W, d = a.emb_d;
    int2* vrow = reinterpret_cast<int2*>(pend0 + (size_t)NW * PSH_PEND);     // the verification's rows, slot by slot
    int* vstart = reinterpret_cast<int*>(reinterpret_cast<char*>(vrow) + (((size_t)d * 8 + 15) & ~(size_t)15));   // 66 slot starts
    int4* gtab = reinterpret_cast<int4*>(vstart + 68);                       // ngroups (+1) merged rows
    int4* rtab = gtab + d + 1;                                               // ngroups (+3) x {c' bits twice, byte offset of E[a_i], -}
    int* sl = reinterpret_cast<int*>(rtab + d + 4) + (size_t)wave_in_block * (128 + PSH_PX_DLCAP);  // wave-private: 64 survivors,
    int* slu = sl + 64;                                                              //   the (row, segment) index of each,
    float* Dl = reinterpret_cast<float*>(slu + 64);                                  //   PSH_PX_DLCAP row differences
    // wave-private, second phase of the row loop: live windows (index, partial sum), the query group's coordinates by merged row
    int* alp = reinterpret_cast<int*>(rtab + d + 4) + (size_t)NW * (128 + PSH_PX_DLCAP) + (size_t)wave_in_block * (2 * PSH_PX_ALIVE + NBG * 64);
    float* ala = reinterpret_cast<float*>(alp + PSH_PX_ALIVE);
    float* hq = ala + PSH_PX_ALIVE;
    int npend = 0;

    const int ktop = __builtin_amdgcn_readfirstlane(plan->ktop);
    const int ngroups = __builtin_amdgcn_readfirstlane(plan->ngroups);
    int r1_plan = __builtin_amdgcn_readfirstlane(a.emb_r1 > 0 ? a.emb_r1 : plan->r1);
    if (r1_plan < 2 || (r1_plan & 1) || r1_plan > ngroups) r1_plan = ngroups;          // (phases are whole pairs of rows)
    const float cerr_y = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(plan->cerr_y)));
    const float cerr_p = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(plan->cerr_p)));
    if (threadIdx.x == 0) *next_unit = 0;
    if (MODE == PSH_MODE_FILTER)
        for (int q = (int)threadIdx.x; q < a.B; q += THREADS) lcount[q] = 0;
    for (int i = (int)threadIdx.x; i < d; i += THREADS) vrow[i] = plan->vrow[i];
    for (int i = (int)threadIdx.x; i < 66; i += THREADS) vstart[i] = plan->vstart[i];
    const int vnl = __builtin_amdgcn_readfirstlane(plan->vnl);
    const int vmax = __builtin_amdgcn_readfirstlane(plan->vmax);
    for (int i = (int)threadIdx.x; i <= ngroups + 2; i += THREADS) {
        const int4 ge = i <= ngroups ? plan->gtab[i] : make_int4(0, 0, 0, 0);
        if (i <= ngroups) gtab[i] = ge;
        rtab[i] = make_int4(ge.x, ge.x, ge.y, 0);
    }
    for (int p = lane; p < a.tile_floats; p += 64) tile[p] = 0.0f;
    __syncthreads();

    const int nfloat = PSH_SEG + K - 1;
    const unsigned n_rs = (unsigned)a.n_rows * (unsigned)a.nseg;
    const unsigned n_units = n_rs * (unsigned)a.n_qgroups;
    const unsigned u_lo = (unsigned)(((unsigned long long)n_units * blockIdx.x) / gridDim.x);
    const unsigned u_hi = (unsigned)(((unsigned long long)n_units * (blockIdx.x + 1)) / gridDim.x);
    typedef const __attribute__((address_space(4))) QueryState* const_qsp;
    const const_qsp qstate_k = (const_qsp)a.qstate;

    // the query group's merged coordinates h' = sqrt(m) mean hx across the lanes (group g in lane g), V = the scatter inside
    // the groups, 6 u ||hx||: kept across units while the group stays the same (the whole kernel when B <= NBG)
    int hgt[NBG];
    float Vq[NBG], hnq[NBG];
    int hgt_b0 = -1;
#pragma unroll
    for (int g = 0; g < NBG; ++g) { hgt[g] = 0; Vq[g] = hnq[g] = 0.0f; }

    // A survivor of the cheap test carries the (row, segment) index of the unit that listed it: at a unit's end only WHOLE
    // passes of the exact verification run (a pass costs the same for one survivor as for its `spp` -- eight at Foveal's
    // d = 34 -- and a unit lists one now and then), the rest waits for the next units'; the wave's last unit flushes.
    int ns = 0;                                              // survivors waiting in sl (wave-uniform)
    auto coords = [&](unsigned rsu, int& seg_start_e, int64_t& row_e) {
        const unsigned ri2 = fast_div(rsu, a.magic_nseg, (unsigned)a.nseg);
        seg_start_e = (int)(rsu - ri2 * (unsigned)a.nseg) * PSH_SEG;
        row_e = a.row0 + (int64_t)ri2 * a.row_stride;
    };
    // Exact verification of the listed survivors (window index | query << 12), rows across the lanes: lane (el, l)
    // runs the chains of row l and then of row d-1-l of survivor el (short and long support: equal work per lane),
    // 64 / ceil(d/2) survivors per pass; one lane per survivor then adds the d squares in row order.  A row's taps need
    // no matrix: c_i on [a_i, ktop), zero elsewhere -- the zero taps the dense chain visits inside its span of
    // 4-tap groups are visited too (fma(0, y, .) matters for non-finite y).
    // Exact verification of the listed survivors (window index | query << 12): the dense chains in the oracle's order
    // (a row's span of whole 4-tap groups from its first tap rounded down; c_i on [a_i, ktop), the zero taps of the span
    // visited too: fma(0, y, .) matters for non-finite y), then the d squares in row order by one lane per survivor.
    // The rows of a survivor are spread over `vnl` lanes by the plan -- longest row first onto the least loaded lane, so the
    // lanes' tap counts are even (Foveal(1.15, 0.9, 126): 865 taps, the longest row 115 -> 8 lanes, 8 survivors a pass
    // of 116 steps; a row pair per lane -- 17 lanes, 3 survivors, 136 steps -- was 2.7x the work per survivor).  Every lane
    // walks its list of rows, 4 taps a step, all lanes `vmax` taps.
    // `staged` (the call at the end of a unit, when E is no longer needed): the survivors' windows are first copied into
    // the wave's tile, as many as fit, with all their loads in flight together -- the chains then run at LDS latency.
    // Mid-unit (a full list: rare) a step reads its 4 samples from global memory.
    auto verify_list = [&](bool staged, bool whole_passes_only) {
        wave_lds_fence();
        const int Kst = (K + 3) & ~3;
        int spp = 64 / vnl;                              // survivors per pass
        spp = spp < PSH_PX_DLCAP / d ? spp : PSH_PX_DLCAP / d;
        const int fit = a.tile_floats / Kst;              // windows the tile can stage (5 at K = 252)
        if (staged && fit < spp) spp = fit;               // long windows: fewer survivors a pass rather than global reads
        if (spp < 1) { spp = 1; staged = false; }
        int nst = fit / spp * spp;                        // windows per staging batch: whole passes
        if (!staged) nst = 64;
        const int ns_all = ns;
        if (whole_passes_only) ns = ns / spp * spp;      // (the rest waits for the next units' survivors)
        const int nq4 = (Kst + 63) >> 6;
        const int sv = lane / vnl, ls = lane - sv * vnl;
        const int r0 = vstart[ls], r1 = vstart[ls + 1];
#pragma unroll 1
        for (int s0 = 0; s0 < ns; s0 += nst) {
            const int s1 = (s0 + nst) < ns ? (s0 + nst) : ns;
            if (staged) {
                wave_lds_fence();                        // the batch before this one has been read
#pragma unroll 1
                for (int sb = s0; sb < s1; sb += 4) {
                    float v[4][4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int su = (sb + u) < s1 ? (sb + u) : (s1 - 1);
                        int ss_u;
                        int64_t row_u;
                        coords((unsigned)slu[su], ss_u, row_u);
                        const float* yw = a.dataset + row_u * a.T + ss_u + (sl[su] & 4095);
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            int j = lane + 64 * q;
                            j = j < K ? j : K - 1;
                            if (q < nq4) v[u][q] = yw[j];
                        }
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int j = lane + 64 * q;
                            if (q < nq4 && sb + u < s1 && j < Kst) tile[(sb + u - s0) * Kst + j] = j < K ? v[u][q] : 0.0f;
                        }
                }
                wave_lds_fence();
            }
#pragma unroll 1
            for (int e0 = s0; e0 < s1; e0 += spp) {
                const bool lv = sv < spp && e0 + sv < s1;
                const int ent = lv ? sl[e0 + sv] : 0;
                const int pwin = ent & 4095, b = ent >> 12;
                int seg_start_e;
                int64_t row_e;
                coords(lv ? (unsigned)slu[e0 + sv] : 0u, seg_start_e, row_e);
                // the query's coordinates into the survivor's row of Dl (one round of loads for the whole pass)
                if (lv) {
                    const float* hxb = a.hx + (int64_t)b * d;
                    for (int i = ls; i < d; i += vnl) Dl[sv * d + i] = hxb[i];
                }
                wave_lds_fence();
                const float* ys = tile + (lv ? (e0 - s0 + sv) * Kst : 0);      // the staged window
                const float* yw = a.dataset + row_e * a.T + seg_start_e + pwin;
                int r = r0;
                int2 cur = (lv && r < r1) ? vrow[r] : make_int2(0, 0);
                int pos = 0;
                float hy = 0.0f;
#pragma unroll 1
                for (int step = 0; step < vmax; step += 4) {
                    const bool act = lv && r < r1;
                    const int lo = cur.x & 255, n4 = ((cur.x >> 8) & 127) << 2, ath = (cur.x >> 16) & 255;
                    const float c = __uint_as_float((unsigned)cur.y);
                    const bool cnz = (cur.y & 0x7fffffff) != 0;
                    const int j0 = lo + pos;                                    // a multiple of 4, below Kst while act
                    f32x4 y4 = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (staged) {
                        y4 = *reinterpret_cast<const f32x4*>(ys + (act ? j0 : 0));
                    } else if (act) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) y4[q] = (j0 + q < K) ? yw[j0 + q] : 0.0f;
                    }
                    float t = hy;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const bool on = (j0 + q >= ath) && (j0 + q < ktop);
                        t = cnz ? __builtin_fmaf(on ? c : 0.0f, y4[q], t) : t;   // (an empty row visits no tap)
                    }
                    hy = act ? t : hy;
                    pos += 4;
                    if (act && pos >= n4) {                                     // the row is done: D_i = hx_i - hy_i, next row
                        const int row = (cur.x >> 24) & 127;
                        Dl[sv * d + row] = __fsub_rn(Dl[sv * d + row], hy);
                        hy = 0.0f;
                        pos = 0;
                        ++r;
                        if (r < r1) cur = vrow[r];
                    }
                }
                wave_lds_fence();
                float ea = __uint_as_float(PSH_INF_BITS);
                bool hit = false;
                if (lv && ls == 0) {
                    ea = 0.0f;
                    for (int i = 0; i < d; ++i) { const float D = Dl[sv * d + i]; ea = __builtin_fmaf(D, D, ea); }
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
                    pend[slot] = u32x4{__float_as_uint(ea), (unsigned)(int)(row_e + a.r_offset), (unsigned)(seg_start_e + pwin), (unsigned)b};
                }
                npend += nh2;
            }
        }
        // the survivors that wait: to the front of the list
        const int left = ns_all - ns;
        int k0 = 0, k1 = 0;
        if (lane < left) { k0 = sl[ns + lane]; k1 = slu[ns + lane]; }
        wave_lds_fence();
        if (lane < left) { sl[lane] = k0; slu[lane] = k1; }
        ns = left;
        wave_lds_fence();                                // sl is refilled afterwards
    };

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
        float ymax = 0.0f, asum;
        {
            Stage st;
            stage_load<ALIGNED>(st, a.dataset + row * a.T, a.T, seg_start, nfloat, lane);
#pragma unroll
            for (int q = 0; q < PSH_NSTAGE; ++q) {          // max |y| of everything this segment reads (NaN ignored)
                if (q < PSH_NSTAGE - 1 || lane + 64 * q < ((nfloat + 3) >> 2)) {
                    ymax = fmaxf(ymax, fmaxf(fmaxf(fabsf(st.v[q][0]), fabsf(st.v[q][1])),
                                             fmaxf(fabsf(st.v[q][2]), fabsf(st.v[q][3]))));
                }
            }
            asum = prefix_store(st, tile, nfloat, lane);     // the tile holds E, not y
        }
        wave_lds_fence();
        ymax = wave_max_nonneg(ymax);
        // radius of the cheap embedding around the exact one: the exact chain's own rounding (ymax), the prefix sums'
        // (two entries per running sum, each within PSH_PX_SCAN_OPS u asum; cerr_p = 2 u ||c||_2 with its margin)
        const float err = __builtin_fmaf(ymax, cerr_y, asum * ((float)PSH_PX_SCAN_OPS * cerr_p));

        const int r_global = (int)(row + a.r_offset);
        const int q_begin = (int)qgi * a.q_per_group;
        const int q_end = (q_begin + a.q_per_group) < a.B ? (q_begin + a.q_per_group) : a.B;


        float Pk[PSH_L];                                     // E[t + ktop] of the lane's windows t = lane + 64 w
#pragma unroll
        for (int w = 0; w < PSH_L; ++w) Pk[w] = tile[lane + 64 * w + ktop];
        // admissible windows of the lane: seg_start + lane + 64 w < Tp
        int nv = (a.Tp - seg_start - lane + 63) >> 6;
        nv = nv < 0 ? 0 : (nv > PSH_L ? PSH_L : nv);
        const unsigned vmask = (1u << nv) - 1u;

        for (int b0 = q_begin; b0 < q_end; b0 += NBG) {
            const int nq = (q_end - b0) < NBG ? (q_end - b0) : NBG;
            if (b0 != hgt_b0) {                              // wave-uniform
                hgt_b0 = b0;
#pragma unroll
                for (int g = 0; g < NBG; ++g) {
                    float vs = 0.0f, sq = 0.0f;
                    hgt[g] = 0;
                    if (g < nq && lane < ngroups) {
                        const int4 ge = gtab[lane];
                        const float* hxb = a.hx + (int64_t)(b0 + g) * d;
                        float hj[4], sum = 0.0f;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            hj[j] = j < ge.w ? hxb[(ge.z >> (8 * j)) & 255] : 0.0f;
                            sum += hj[j];
                            sq = __builtin_fmaf(hj[j], hj[j], sq);
                        }
                        const float mean = sum / (float)ge.w;
#pragma unroll
                        for (int j = 0; j < 4; ++j) { const float dv = j < ge.w ? hj[j] - mean : 0.0f; vs = __builtin_fmaf(dv, dv, vs); }
                        const float rsm = ge.w == 1 ? 1.0f : (ge.w == 2 ? 0.70710678f : (ge.w == 3 ? 0.57735027f : 0.5f));
                        hgt[g] = (int)__float_as_uint(sum * rsm);
                    }
                    hq[g * 64 + lane] = __uint_as_float((unsigned)hgt[g]);   // the same coordinates by row, for the sparse second phase
#pragma unroll
                    for (int off = 32; off > 0; off >>= 1) { vs += __shfl_xor(vs, off, 64); sq += __shfl_xor(sq, off, 64); }
                    Vq[g] = vs * (1.0f + 1.0f / 256.0f);
                    hnq[g] = 6.0f * 5.9604645e-8f * 1.001f * __builtin_sqrtf(sq);
                }
                wave_lds_fence();
            }
            f32x2 acc[NBG][8];
#pragma unroll
            for (int g = 0; g < NBG; ++g)
#pragma unroll
                for (int w = 0; w < 8; ++w) acc[g][w] = f32x2{0.f, 0.f};
            // a (merged) row: its constant (as a pair) and first tap come out of the block's table -- one broadcast LDS read,
            // fetched a row ahead into the OTHER of two register sets (the loop takes two rows a turn) --, the query's
            // coordinate out of the lane that holds it
            const char* tile_b = reinterpret_cast<const char*>(tile + lane);
            // (NQ, the queries of this pass, is a compile-time constant of the loop: no branch inside a row)
            auto rows = [&](auto nq_c, int r_from, int r_to) {
                constexpr int NQ = decltype(nq_c)::value;
                auto row_step = [&](const int4& o, int i) {
                    const f32x2 c2 = f32x2{__uint_as_float((unsigned)o.x), __uint_as_float((unsigned)o.y)};
                    const float* pa = reinterpret_cast<const float*>(tile_b + o.z);
                    f32x2 S[8];                              // -S_i of the window pair: e = hx - c S = fma(c, -S, hx)
#pragma unroll
                    for (int w = 0; w < 8; ++w) S[w] = f32x2{pa[128 * w], pa[128 * w + 64]} - f32x2{Pk[2 * w], Pk[2 * w + 1]};
#pragma unroll
                    for (int g = 0; g < NQ; ++g) {
                        const float hv = __uint_as_float((unsigned)__builtin_amdgcn_readlane(hgt[g], i));
                        const f32x2 hx2 = f32x2{hv, hv};
                        f32x2 e[8];
#pragma unroll
                        for (int w = 0; w < 8; ++w) e[w] = __builtin_elementwise_fma(c2, S[w], hx2);
#pragma unroll
                        for (int w = 0; w < 8; ++w) acc[g][w] = __builtin_elementwise_fma(e[w], e[w], acc[g][w]);
                    }
                };
                // (fetching a row's E values while the row before it is computed -- two register sets, 768 threads for the
                //  registers -- measured slower, 1.92 against 1.57 ms: four waves per SIMD hide the LDS latency better)
                // Two rows a turn, ALWAYS: an odd row count ends on a blank entry (c' = 0, lane `ngroups` holds h' = 0: e = 0,
                // nothing is added; rtab has ngroups + 3 entries).  A separate copy of the row body for the odd row made the
                // compiler spill 15 registers around it.
                int4 oa = rtab[r_from];                       // (r_from is even: the rows of a phase come in pairs)
#pragma unroll 1
                for (int i = r_from; i < r_to; i += 2) {
                    const int4 ob = rtab[i + 1];
                    row_step(oa, i);
                    oa = rtab[i + 2];                        // (up to entry ngroups + 2: blank)
                    row_step(ob, i + 1);
                }
            };
            auto run_rows = [&](int r_from, int r_to) {
                if constexpr (NBG == 2) {
                    if (nq == 2) rows(std::integral_constant<int, 2>{}, r_from, r_to); else rows(std::integral_constant<int, 1>{}, r_from, r_to);
                } else {
                    switch (nq) {
                        case 1: rows(std::integral_constant<int, 1>{}, r_from, r_to); break;
                        case 2: rows(std::integral_constant<int, 2>{}, r_from, r_to); break;
                        case 3: rows(std::integral_constant<int, 3>{}, r_from, r_to); break;
                        case 4: rows(std::integral_constant<int, 4>{}, r_from, r_to); break;
                        case 5: rows(std::integral_constant<int, 5>{}, r_from, r_to); break;
                        default: rows(std::integral_constant<int, NBG>{}, r_from, r_to); break;
                    }
                }
            };
            // FILTER: the merged rows come heaviest first (the plan), and a partial sum of squares only grows -- after the
            // first r1 rows most windows are above the threshold already and need none of the others.  The few that are
            // not (a handful per segment) finish their rows as (window, row) pairs across the lanes; a segment with more of
            // them than the list holds (nothing may be rejected: non-finite data; or a threshold that rejects little)
            // runs the remaining rows for everybody, as before.
            const int r1 = MODE == PSH_MODE_FILTER ? r1_plan : ngroups;
            run_rows(0, r1);
            // windows of the lane at or below `thr` (or NaN) as a bit mask.  The common case -- none of the 16 -- costs a
            // minimum and two comparisons, not 16 (fminf drops a NaN; the sum of the non-negative values keeps it).
            auto below = [&](int g, float thr) -> unsigned {
                // (v_min3 on the values as they are: fminf() quiets every operand first, 16 more instructions per query and unit)
                const float m0 = min3f(acc[g][0][0], acc[g][0][1], acc[g][1][0]), m1 = min3f(acc[g][1][1], acc[g][2][0], acc[g][2][1]);
                const float m2 = min3f(acc[g][3][0], acc[g][3][1], acc[g][4][0]), m3 = min3f(acc[g][4][1], acc[g][5][0], acc[g][5][1]);
                const float m4 = min3f(acc[g][6][0], acc[g][6][1], acc[g][7][0]);
                const float mn = min3f(min3f(m0, m1, m2), min3f(m3, m4, acc[g][7][1]), __uint_as_float(PSH_INF_BITS));
                f32x2 sm2 = acc[g][0];
#pragma unroll
                for (int w = 1; w < 8; ++w) sm2 += acc[g][w];
                const float sm1 = sm2[0] + sm2[1];
                unsigned hm = 0u;
                if (__any(!(mn > thr) || !(sm1 == sm1))) {
                    // (a compare and an add-with-carry per window, hm = 2 hm + [!(acc > thr)], from the last window down: the
                    //  compiler's select + or3 form is 3.5 instructions per window)
#pragma unroll
                    for (int w = PSH_L - 1; w >= 0; --w)
                        asm volatile("v_cmp_ngt_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(hm) : "v"(acc[g][w >> 1][w & 1]), "v"(thr) : "vcc");
                    hm &= vmask;
                }
                return hm;
            };
            // survivors go to the wave's list; the whole wave verifies them together (verify_list)
            auto list_survivor = [&](bool has, int pwin, int b) {
                const unsigned long long sm = __ballot(has);
                if (!sm) return;
                const int ne = __popcll(sm);
                if (ns + ne > 64) verify_list(false, false);
                if (has) {
                    const int slot = ns + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(sm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)sm, 0u));
                    sl[slot] = pwin | (b << 12);
                    slu[slot] = (int)__builtin_amdgcn_readfirstlane((int)rs);
                }
                ns += ne;
            };
            unsigned done = 0u;                              // queries of the pass finished by the sparse second phase
            bool dense = false;
            if (MODE == PSH_MODE_FILTER && r1 < ngroups) {
#pragma unroll
                for (int g = 0; g < NBG; ++g) {
                    if (g >= nq || dense) continue;
                    const int b = b0 + g;
                    const float tau = __uint_as_float(qstate_k[b].tau2_bits);
                    const float st = __builtin_sqrtf(tau) * (1.0f + 1.0f / 32768.0f) + (err + hnq[g]);
                    const float thr = st * st * (1.0f + 1.0f / 16384.0f);
                    unsigned hm = below(g, thr);
                    // the lane's live windows (index, partial sum) into the wave's list
                    int nal = 0;
                    while (__any(hm != 0u)) {
                        const bool has = hm != 0u;
                        const int w = has ? (int)__builtin_ctz(hm) : 0;
                        hm &= hm - 1u;
                        const unsigned long long sm = __ballot(has);
                        const int ne = __popcll(sm);
                        if (nal + ne > PSH_PX_ALIVE) { dense = true; break; }
                        if (has) {
                            const int slot = nal + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(sm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)sm, 0u));
                            float pv = 0.0f;
#pragma unroll
                            for (int w2 = 0; w2 < PSH_L; ++w2) pv = (w2 == w) ? acc[g][w2 >> 1][w2 & 1] : pv;
                            alp[slot] = lane + 64 * w;
                            ala[slot] = pv;
                        }
                        nal += ne;
                    }
                    if (dense) continue;
                    done |= 1u << g;
                    if (nal == 0) continue;
                    wave_lds_fence();
                    // second phase: four live windows a turn, 16 lanes each; lane `sub` of a window takes the merged rows
                    // r1 + sub, r1 + sub + 16, ..; the 16 partial sums meet through four DPP rotations of the row
                    const int sub = lane & 15, grp = lane >> 4;
                    const float* hqg = hq + g * 64;
#pragma unroll 1
                    for (int e0 = 0; e0 < nal; e0 += 4) {
                        const bool lv = e0 + grp < nal;
                        const int pw = lv ? alp[e0 + grp] : 0;
                        const float Ek = tile[pw + ktop];
                        const char* tp = reinterpret_cast<const char*>(tile + pw);
                        float s2 = 0.0f;
#pragma unroll 1
                        for (int r = r1; r < ngroups; r += 16) {
                            const int rr = (r + sub) < ngroups ? (r + sub) : ngroups;   // (entry ngroups: blank, c' = 0, offset 0)
                            const int4 o = rtab[rr];
                            const float hv = (r + sub) < ngroups ? hqg[rr] : 0.0f;
                            const float S = *reinterpret_cast<const float*>(tp + o.z) - Ek;
                            const float e = __builtin_fmaf(__uint_as_float((unsigned)o.x), S, hv);
                            s2 = __builtin_fmaf(e, e, s2);
                        }
                        s2 += __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(s2), 0x128, 0xf, 0xf, false));   // row_ror:8
                        s2 += __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(s2), 0x124, 0xf, 0xf, false));   // row_ror:4
                        s2 += __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(s2), 0x122, 0xf, 0xf, false));   // row_ror:2
                        s2 += __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(s2), 0x121, 0xf, 0xf, false));   // row_ror:1
                        const float tot = (lv ? ala[e0 + grp] : 0.0f) + s2;
                        list_survivor(lv && sub == 0 && !(tot > thr), pw, b);
                    }
                    wave_lds_fence();                        // the list is refilled for the next query
                }
                if (dense) run_rows(r1, ngroups);
            }
#pragma unroll
            for (int g = 0; g < NBG; ++g) {
                const int b = b0 + g;
                if (g >= nq || ((done >> g) & 1u)) continue;
                const float errq = err + hnq[g];
                if (MODE == PSH_MODE_BOOT) {
                    // upper bound of the exact acc of the lane's (wave's) best window
                    float m = __uint_as_float(PSH_INF_BITS);
#pragma unroll
                    for (int w = 0; w < PSH_L; ++w) m = ((vmask >> w) & 1u) ? fminf(m, acc[g][w >> 1][w & 1]) : m;
                    if (a.boot_per_wave) {
#pragma unroll
                        for (int off = 32; off > 0; off >>= 1) m = fminf(m, __shfl_xor(m, off, 64));
                    }
                    const float su = __builtin_sqrtf(m + Vq[g]) * (1.0f + 1.0f / 32768.0f) + errq;
                    const float ub = su * su * (1.0f + 1.0f / 16384.0f);
                    if (a.boot_per_wave) {
                        if (lane == 0) a.minbuf[(int64_t)b * a.min_stride + (int64_t)rs] = ub;
                    } else {
                        a.minbuf[(int64_t)b * a.min_stride + (int64_t)rs * 64 + lane] = ub;
                    }
                } else {
                    const float tau = __uint_as_float(qstate_k[b].tau2_bits);
                    const float st = __builtin_sqrtf(tau) * (1.0f + 1.0f / 32768.0f) + errq;
                    const float thr = st * st * (1.0f + 1.0f / 16384.0f);
                    unsigned hm = below(g, thr);
                    while (__any(hm != 0u)) {
                        const bool has = hm != 0u;
                        const int w = has ? (int)__builtin_ctz(hm) : 0;
                        hm &= hm - 1u;
                        list_survivor(has, lane + 64 * w, b);
                    }
                }
            }
        }
        if (MODE == PSH_MODE_FILTER && ns >= 8) verify_list(true, true);     // (E is done with: the tile stages the windows; whole passes)
        wave_lds_fence();  // all lanes done with the tile before it is overwritten
    }
    if (MODE == PSH_MODE_FILTER) {
        if (ns > 0) verify_list(true, false);                // the survivors still waiting
        if (npend > 0) pend_flush(pend, npend, lcount, a, lane);
        __syncthreads();
        for (int q = (int)threadIdx.x; q < a.B; q += THREADS)
            a.bcount[(int64_t)q * PSH_MAX_BLOCKS + blockIdx.x] = lcount[q];
    }
}

#define PSH_PX_WIDE_THREADS 512
#define PSH_PX_WIDE_NBG 6
#define PSH_PX_NBG 2
#define PSH_PX_THREADS 1024

bool embed_px_supported(int tile_floats, int B, int d, int K, bool wide) {
    return d >= 1 && d <= PSH_EMB_MAX_D && K >= 1 && K <= 256 &&
           px_shmem_bytes(tile_floats, B, d, wide ? PSH_PX_WIDE_THREADS : PSH_PX_THREADS) <= PSH_LDS_BYTES;
}

template <bool ALIGNED, int MODE>
static hipError_t launch_px_mode(const ScanArgs& a, int grid, hipStream_t s) {
    if (a.emb_wide) {
        const size_t shmem = px_shmem_bytes(a.tile_floats, a.B, a.emb_d, PSH_PX_WIDE_THREADS);
        hipError_t e = hipFuncSetAttribute((const void*)embed_px_kernel<ALIGNED, MODE, PSH_PX_WIDE_THREADS, PSH_PX_WIDE_NBG>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((embed_px_kernel<ALIGNED, MODE, PSH_PX_WIDE_THREADS, PSH_PX_WIDE_NBG>), dim3(grid), dim3(PSH_PX_WIDE_THREADS), shmem, s, a);
        return hipGetLastError();
    }
    const size_t shmem = px_shmem_bytes(a.tile_floats, a.B, a.emb_d, PSH_PX_THREADS);
    hipError_t e = hipFuncSetAttribute((const void*)embed_px_kernel<ALIGNED, MODE, PSH_PX_THREADS, PSH_PX_NBG>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((embed_px_kernel<ALIGNED, MODE, PSH_PX_THREADS, PSH_PX_NBG>), dim3(grid), dim3(PSH_PX_THREADS), shmem, s, a);
    return hipGetLastError();
}

hipError_t launch_embed_px(const ScanArgs& a, int mode, bool aligned, int grid, hipStream_t s) {
    if (mode == PSH_MODE_BOOT) return aligned ? launch_px_mode<true, PSH_MODE_BOOT>(a, grid, s) : launch_px_mode<false, PSH_MODE_BOOT>(a, grid, s);
    return aligned ? launch_px_mode<true, PSH_MODE_FILTER>(a, grid, s) : launch_px_mode<false, PSH_MODE_FILTER>(a, grid, s);
}

}  // namespace psh
