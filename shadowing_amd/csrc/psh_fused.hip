// psh_fused.hip -- the whole single-query step in ONE launch (gfx950): bootstrap sample -> threshold -> full
// sliding-window scan -> selection, for Identity + RelativeMSE (reference path_shadowing.py:149-173, path_distance.py:
// 62-65), one block of 16 waves per CU.  Part of libpsh_hip.so; the scan itself is scan_mx_kernel's (psh_scan.hip):
// f16 rejection test on the matrix cores, exact fp32 chain for the survivors.
//
// Why: as four launches the step spent 40 % of its time outside the HBM-bound scan -- a latency-bound bootstrap
// launch, two ONE-block kernels (threshold, selection: 255 CUs idle for 35 us) and the gaps between them.  Here
//   A  every wave scans its one or two sampled segments (the cheap VALU upper bound of scan_kernel's bootstrap) and
//      publishes the segment minimum as an 8-byte {tag, value} granule;
//   B  every block sweeps ALL granules (the sweep is the grid barrier: data-tagged, no counter), stages the minima in
//      LDS and reads the admission level tau2 -- the rank-th smallest minimum, ~2k windows of the ensemble expected
//      below it -- off one 2048-bucket histogram; 256 CUs redo 10 us of one CU's work in 3 us and nobody waits for a
//      launch.  The f16 scale comes from the query and tau2 alone (see below), so nothing else is exchanged;
//   C  the scan; the first segment of every wave is already in flight during B.  Only windows with an exact
//      acc < tau2 are kept, in a per-block LDS list (a handful per block);
//   D  every block publishes its list (write-through stores + one tagged count), sweeps the 256 counts (second
//      barrier), loads all ~2k candidates and RANKS ITS OWN against them by counting: rank < k -> out[rank].  No sort,
//      no single-block tail.
// Everything that can go wrong -- fewer than k windows below the estimate, a block with more than PSH_FUSED_FRONT
// candidates (massive ties), a poll that times out because a block is not resident, a header that was never
// initialised -- raises PSH_STATUS_RETRY: the caller reruns the query through the separate launches
// (PSH_FLAG_NO_FUSE), which handle all of those.  Results that ARE returned are the exact top-k: every window below
// tau2 was found, and there were at least k of them.
//
// Inter-block visibility (MI355X_MICROARCH.md, "Workgroup dispatch ... visibility"): per-XCD L2s are not coherent and
// L1 is never refreshed, so every exchanged word is written with an agent-scope relaxed atomic store (sc1,
// write-through) and read with an agent-scope relaxed atomic load (sc1, L1 bypass); a count is published only after
// the publishing wave has drained its stores (s_waitcnt vmcnt(0)).  No fences, no device-scope RMW atomics.
//
// The f16 scale without a data maximum.  scan_mx_kernel scales by the largest |value| of the bootstrap rows and the
// query; its proof needs only (i) |x~| < 8 and (ii) tau~ <= 5120, so that a window holding a value beyond f16 range
// (|y~| > 255) has acc~ >= (255 - 8)^2 > tau~ and is rightly rejected when its t^ comes out +inf (NaN is kept).
// Both follow from the query and tau alone: scale = the largest power of two with max|x| * scale < 8 and
// tau2 * scale^2 <= 4096.
#include "psh_device.h"

namespace psh {

typedef unsigned long long u64;
typedef __attribute__((address_space(1))) u64 gu64;
typedef __attribute__((address_space(1))) unsigned gu32;

__device__ __forceinline__ void g_store(u64* p, u64 v) {
    __hip_atomic_store((gu64*)(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void g_store32(unsigned* p, unsigned v) {
    __hip_atomic_store((gu32*)(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// a word the HOST reads (a blocking caller's pinned block): write-through to system scope
__device__ __forceinline__ void sys_store32(unsigned* p, unsigned v) {
    __hip_atomic_store((gu32*)(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ u64 g_load(const u64* p) {
    return __hip_atomic_load((gu64*)(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Bulk reads of exchanged words: sc1 BUFFER loads (aux bit 4), which bypass L1 like the atomic loads but are ordinary
// loads to the compiler -- it keeps them all in flight and waits once.  (A relaxed agent-scope __hip_atomic_load is
// followed by s_waitcnt vmcnt(0) each: 8 granules per lane cost 8 serial round trips, 16 us per sweep.)
typedef unsigned u32x2v __attribute__((ext_vector_type(2)));
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
#define PSH_AUX_SC1 16
__device__ __forceinline__ __amdgpu_buffer_rsrc_t g_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ u64 g_load_b64(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    const u32x2v v = __builtin_amdgcn_raw_buffer_load_b64(r, (int)byte_off, 0, PSH_AUX_SC1);
    return ((u64)v[1] << 32) | (u64)v[0];
}

#define PSH_FUSED_FIXED_BYTES 4096   // control words, the block's front list, per-block counts, rank counters
#define PSH_FUSED_HIST 2048

// control words (ints at the start of LDS)
enum { C_FRONT = 0, C_NEXT = 1, C_BAIL = 2, C_KMIN = 3, C_KMAX = 4, C_NFINITE = 5, C_EDGE = 6, C_NTOTAL = 7, C_ANYOVF = 8,
       C_TAU2 = 9, C_THR2 = 10, C_SCALE = 11, C_XN = 12, C_EPOCH = 13, C_MAGIC_OK = 14 };

// HINTED: the caller's admission level (psh_profile.tau_hint) instead of the sample: no phase A, no first barrier, phase B
// only derives scale and threshold from the hint.  An instantiation of its own, so that the sampled launch compiles exactly
// as it did without it (as a run-time flag it cost the W = 20 form nine VGPRs and the unaligned one a spill).
// BLK: a blocking caller's launch (psh_shadow_blocking: FusedArgs::g_ds / done) -- the ranking gathers the winners' paths and the
// last blocks set the completion words; an instantiation of its own for the same reason.
template <int WT, bool ALIGNED, bool HINTED, bool BLK>
__global__ __launch_bounds__(PSH_SCAN_THREADS) void scan_fused_kernel(ScanArgs a, FusedArgs f) {
    static_assert(WT >= 0 && WT <= 33, "the shifted-query band must fit K = 64 (WT = 0: run-time W <= 33)");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NW = PSH_SCAN_THREADS / 64;
    const int lane = lane_id();
    const int tid = (int)threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    int* ctl = reinterpret_cast<int*>(smem);                                 // 32 control words
    u32x4* fl = reinterpret_cast<u32x4*>(ctl + 32);                           // PSH_FUSED_FRONT entries {acc|d bits, r, t, -}
    int* cnts = reinterpret_cast<int*>(fl + PSH_FUSED_FRONT);                 // PSH_FUSED_MAX_BLOCKS counts
    int* rankc = cnts + PSH_FUSED_MAX_BLOCKS;                                 // PSH_FUSED_FRONT rank counters
    int* offs = rankc + PSH_FUSED_FRONT;                                      // PSH_FUSED_MAX_BLOCKS + 1 prefix sums of the counts
    float* xs = reinterpret_cast<float*>(offs + PSH_FUSED_MAX_BLOCKS + 1);    // HINTED: the query's W <= 33 taps (see derive_levels)
    float* tiles = smem + PSH_FUSED_FIXED_BYTES / 4;
    float* tile = tiles + (size_t)wave * a.tile_floats;
    _Float16* ah0 = reinterpret_cast<_Float16*>(tiles + (size_t)NW * a.tile_floats);
    _Float16* a1 = ah0 + (size_t)wave * 2 * PSH_MX_NHALF;                     // y^
    _Float16* a2 = a1 + PSH_MX_NHALF;                                         // (y~^2)^
    // phase B scratch lives in the f16 region (unused until the scan): the staged minima and their histogram
    unsigned* keys = reinterpret_cast<unsigned*>(ah0);                        // PSH_FUSED_MAX_UNITS
    unsigned* hist = keys + PSH_FUSED_MAX_UNITS;                              // PSH_FUSED_HIST

    FusedHdr* hdr = f.hdr;
    // tuning aid (tools/fused_times.py, -DPSH_TUNING build only sets the pointer): phase boundaries of every wave 0
    auto stamp = [&](int i) { if (a.dbg_times && tid == 0) a.dbg_times[(size_t)blockIdx.x * 8 + i] = (unsigned long long)wall_clock64(); };
    stamp(0);
    if constexpr (BLK) { if (blockIdx.x == 0 && tid == 0 && f.done) sys_store32(f.done - 1, f.done_val); }   // diagnostics: "started" (psh.h PSH_SHADOW_OFF_STARTED)
    const long long t_start = wall_clock64();
    auto give_up = [&]() -> bool { return wall_clock64() - t_start > f.spin_ticks; };

    const int W = WT > 0 ? WT : a.W;
    const int nfloat = PSH_SEG + W - 1;
    // a blocking caller's query travels in the kernel arguments (FusedArgs::qv): read where they lie, in the kernarg segment --
    // explicit arguments start at its offset 0, `f` follows `a` at its own alignment
    constexpr size_t qv_off = (sizeof(ScanArgs) + alignof(FusedArgs) - 1) / alignof(FusedArgs) * alignof(FusedArgs) + offsetof(FusedArgs, qv);
    typedef const __attribute__((address_space(4))) char* const_i8p;
    const const_f32p x = BLK ? (const_f32p)((const_i8p)__builtin_amdgcn_kernarg_segment_ptr() + qv_off) : (const_f32p)a.queries;
    const unsigned n_rs = (unsigned)a.n_rows * (unsigned)a.nseg;
    // a block's share of the units.  Blocks are dealt to the XCDs round-robin (MI355X_MICROARCH.md: workgroup dispatch),
    // and the odd XCDs stream ~4 % slower than the even ones on every box measured (tools/fused_times.py: they end their
    // equal shares 2.5-4.5 us later, launch after launch): of the units of a pair of blocks (2j, 2j + 1) the even one
    // takes (256 + skew) / 512.  (Were the dealing different, the shares would merely be 2 % off.)
    unsigned u_lo, u_hi;
    {
        const unsigned pb = blockIdx.x & ~1u;
        const unsigned p_lo = (unsigned)(((u64)n_rs * pb) / gridDim.x);
        const unsigned p_hi = (unsigned)(((u64)n_rs * (pb + 2 < gridDim.x ? pb + 2 : gridDim.x)) / gridDim.x);
        const bool paired = pb + 1 < gridDim.x;
        const unsigned mid = paired ? p_lo + (unsigned)(((u64)(p_hi - p_lo) * (unsigned)(256 + f.xcd_skew)) >> 9) : p_hi;
        u_lo = (blockIdx.x & 1u) ? mid : p_lo;
        u_hi = (blockIdx.x & 1u) ? p_hi : mid;
    }

    // the first sampled segment of every wave is requested before anything else: the header check, the LDS set-up and
    // the block barrier below run under its HBM latency
    auto boot_load = [&](Stage& sx, unsigned uu) {
        const unsigned ri = fast_div(uu, a.magic_nseg, (unsigned)a.nseg);
        const unsigned sg = uu - ri * (unsigned)a.nseg;
        stage_load<ALIGNED>(sx, a.dataset + (f.boot_row0 + (int64_t)ri * f.boot_row_stride) * a.T, a.T, (int)sg * PSH_SEG, nfloat, lane);
    };
    constexpr bool hinted = HINTED;
    const unsigned nbu = hinted ? 0u : (unsigned)f.boot_units;
    Stage stb;
    unsigned ub = blockIdx.x * NW + (unsigned)wave;
    if (ub < nbu) boot_load(stb, ub);

    if (tid == 0) {
        const u64 mg = g_load(&hdr->magic);
        ctl[C_MAGIC_OK] = (mg == PSH_FUSED_MAGIC) ? 1 : 0;
        ctl[C_EPOCH] = (int)(unsigned)g_load(reinterpret_cast<const u64*>(&hdr->epoch));   // (epoch, pad[0]): low word
        ctl[C_FRONT] = 0; ctl[C_NEXT] = 0; ctl[C_BAIL] = 0;
        ctl[C_KMIN] = (int)0xffffffffu; ctl[C_KMAX] = 0; ctl[C_NFINITE] = 0; ctl[C_NTOTAL] = 0; ctl[C_ANYOVF] = 0;
    }
    for (int i = tid; i < PSH_FUSED_HIST; i += PSH_SCAN_THREADS) hist[i] = 0u;
    if (tid < PSH_FUSED_FRONT) rankc[tid] = 0;
    // HINTED: the query may sit in host memory (a blocking caller's pinned buffer, psh.h) -- one parallel fetch of its taps
    // here instead of 3 W dependent scalar loads over PCIe in the level derivation below (measured: +130 us per call)
    if constexpr (HINTED) { if (tid < (WT > 0 ? WT : a.W)) xs[tid] = BLK ? x[tid] : a.queries[tid]; }
    __syncthreads();
    if (!ctl[C_MAGIC_OK]) {            // a workspace psh_workspace_init never saw (or a run that gave up): separate launches
        if (blockIdx.x == 0 && tid == 0) f.status[0] = PSH_STATUS_RETRY_;
        if (blockIdx.x == 0) poison_results(f.out_d, f.out_idx, a.k, tid, PSH_SCAN_THREADS);
        return;
    }
    const unsigned epoch = (unsigned)ctl[C_EPOCH];
    const unsigned tagA = 2u * epoch, tagD = 2u * epoch + 1u;

    // ------------------------------------------------------------------ A: bootstrap sample
    if constexpr (!hinted) {
        constexpr bool CHEAP = (WT >= 17) && (WT <= 32);
        const unsigned stride = gridDim.x * NW;
        unsigned u = ub;
        Stage& st = stb;
        while (u < nbu) {
            const unsigned ri = fast_div(u, a.magic_nseg, (unsigned)a.nseg);
            const unsigned sg = u - ri * (unsigned)a.nseg;
            const int seg_start = (int)sg * PSH_SEG;
            stage_store(st, tile, nfloat, lane);
            wave_lds_fence();
            const unsigned un = u + stride;
            if (un < nbu) boot_load(st, un);
            const int t_lane = seg_start + PSH_L * lane;
            int nvalid = a.Tp - t_lane;
            nvalid = nvalid < 0 ? 0 : (nvalid > PSH_L ? PSH_L : nvalid);
            float acc[PSH_L];
            if constexpr (CHEAP) {
                // upper bounds of acc (scan_kernel's bootstrap): t_i = ny_i - 2 c_i, acc_i <= (nx + t_i + 2^-17 (nx + NY))(1 + 2^-19)
                constexpr int WX = CHEAP ? WT : 20;
                float xv[WX];
#pragma unroll
                for (int j = 0; j < WX; ++j) { xv[j] = x[j]; asm volatile("" : "+v"(xv[j])); }
                float NY;
                approx16<WX>(tile, lane, xv, acc, NY);
                float nx = 0.0f;
#pragma unroll
                for (int j = 0; j < WX; ++j) nx = __builtin_fmaf(xv[j], xv[j], nx);
                const float add = __builtin_fmaf(NY, 1.0f / 65536.0f, nx * (1.0f + 1.0f / 32768.0f));
#pragma unroll
                for (int i = 0; i < PSH_L; ++i) acc[i] = (acc[i] + add) * (1.0f + 1.0f / 65536.0f);
            } else {
                accumulate16<WT>(tile, lane, x, W, acc);
            }
            float m = __uint_as_float(PSH_INF_BITS);
#pragma unroll
            for (int i = 0; i < PSH_L; ++i) m = (i < nvalid) ? fminf(m, acc[i]) : m;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) m = fminf(m, __shfl_xor(m, off, 64));
            if (!(m == m)) m = __uint_as_float(PSH_INF_BITS);                  // NaN data: the segment carries no information
            if (lane == 0) g_store32(&hdr->minima[u], __float_as_uint(m));     // write-through; the block's flag follows the drain
            wave_lds_fence();
            u = un;
        }
    }
    // publish: every storing wave drains its write-through stores, then ONE flag per block
    if constexpr (!hinted) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) g_store(&hdr->aflag[blockIdx.x], ((u64)tagA << 32) | 1ull);
    }
    stamp(1);
    // the first segment of the scan is requested now: its HBM latency runs under phase B
    auto grab = [&]() -> unsigned {
        int v = 0;
        if (lane == 0) v = atomicAdd(&ctl[C_NEXT], 1);
        return u_lo + (unsigned)__builtin_amdgcn_readfirstlane(v);
    };
    auto decode = [&](unsigned uu, unsigned& ri, unsigned& sg) {
        ri = fast_div(uu, a.magic_nseg, (unsigned)a.nseg);
        sg = uu - ri * (unsigned)a.nseg;
    };
    auto load_unit = [&](Stage& sx, unsigned uu) {
        unsigned ri, sg;
        decode(uu, ri, sg);
        stage_load<ALIGNED>(sx, a.dataset + (a.row0 + (int64_t)ri * a.row_stride) * a.T, a.T, (int)sg * PSH_SEG, nfloat, lane);
    };
    Stage st;
    unsigned u = grab();
    if (u < u_hi) load_unit(st, u);

    // the admission level -> f16 scale and rejection threshold (one lane; phase B, sampled or hinted)
    auto derive_levels = [&](const float tau0) {
                bool armed = false;
                const const_f32p xk = x;                  // scalar loads: the query sits in the scalar cache since phase A
                auto xq = [&](int j) -> float { if constexpr (HINTED) return xs[j]; else return xk[j]; };
                const float s = sumsq8([&](int j) { return xq(j); }, W);
                const float xn = f.qnorm_in ? f.qnorm_in[0] : __builtin_sqrtf(s);
                unsigned qmaxbits = 0u;
                for (int j = 0; j < W; ++j) qmaxbits = max(qmaxbits, __float_as_uint(fabsf(xq(j))));
                if (tau0 > 0.0f && tau0 < __uint_as_float(PSH_INF_BITS) && qmaxbits < PSH_INF_BITS) {
                    // scale = 2^sexp: max|x| 2^sexp < 8 and tau0 4^sexp <= 4096 (exponents of the bit patterns: value in [2^(e-1), 2^e))
                    const int et = (int)((__float_as_uint(tau0) >> 23) & 255u) - 126;
                    int sexp = (12 - et) >= 0 ? (12 - et) / 2 : -((et - 12 + 1) / 2);
                    if (qmaxbits >= 0x00800000u) {
                        const int eq = (int)((qmaxbits >> 23) & 255u) - 126;
                        sexp = sexp < 3 - eq ? sexp : 3 - eq;
                    }
                    if (sexp <= 60 && sexp >= -60 && __float_as_uint(tau0) >= 0x00800000u) {
                        const float sc = __uint_as_float((unsigned)(127 + sexp) << 23);
                        double nxs = 0.0;
                        for (int j = 0; j < W; ++j) { const double vv = (double)xq(j) * (double)sc; nxs += vv * vv; }
                        const double am = 1.0 / 512.0, bm = 1.0 / 262144.0;
                        const double taus = (double)tau0 * (double)sc * (double)sc;
                        const double T = taus * (1.0 + 1.0 / 131072.0) * (1.0 + 2.0 * am) - nxs * (1.0 - 3.0 * am) * (1.0 - 1e-12) + bm;
                        float Tf = (float)T;
                        if ((double)Tf < T) Tf = __uint_as_float(Tf >= 0.0f ? __float_as_uint(Tf) + 1u : __float_as_uint(Tf) - 1u);
                        if (Tf == Tf && fabsf(Tf) < __uint_as_float(PSH_INF_BITS)) {
                            ctl[C_TAU2] = (int)__float_as_uint(tau0);
                            ctl[C_THR2] = (int)__float_as_uint(Tf);
                            ctl[C_SCALE] = (int)__float_as_uint(sc);
                            ctl[C_XN] = (int)__float_as_uint(xn);
                            armed = true;
                        }
                    }
                }
                if (!armed) ctl[C_BAIL] = 2;         // absurd magnitudes / a zero estimate: the separate launches cope
    };
    // ------------------------------------------------------------------ B: all minima -> tau2, scale, threshold
    if constexpr (hinted) {
        // (a hint that is not a positive finite number does not arm the launch: PSH_STATUS_RETRY, as for an absurd estimate)
        if (tid == 0) derive_levels(BLK ? f.hint_v : f.tau_hint[0]);
        __syncthreads();
        if (ctl[C_BAIL] != 0) {
            if (tid == 0) {
                f.status[0] = PSH_STATUS_RETRY_;
                if (blockIdx.x == 0) g_store(reinterpret_cast<u64*>(&hdr->epoch), (u64)(epoch + 1u));
            }
            if (blockIdx.x == 0) poison_results(f.out_d, f.out_idx, a.k, tid, PSH_SCAN_THREADS);
            return;
        }
    } else {
        const int nbu = f.boot_units;
        // first barrier: wave 0 sweeps the 256 block flags (2 KB; the other 15 waves do not add to the polling traffic)
        if (wave == 0) {
            const int nblk = (int)gridDim.x;
            const __amdgpu_buffer_rsrc_t rfl = g_rsrc(hdr->aflag, sizeof(hdr->aflag));
            for (;;) {
                bool ok = true;
                u64 xg[PSH_FUSED_MAX_BLOCKS / 64];
#pragma unroll
                for (int i = 0; i < PSH_FUSED_MAX_BLOCKS / 64; ++i) xg[i] = g_load_b64(rfl, (unsigned)(lane + 64 * i) * 8u);
#pragma unroll
                for (int i = 0; i < PSH_FUSED_MAX_BLOCKS / 64; ++i)
                    if (lane + 64 * i < nblk) ok = ok && ((unsigned)(xg[i] >> 32) == tagA);
                if (__all(ok)) break;
                if (give_up()) { if (lane == 0) ctl[C_BAIL] = 1; break; }
                __builtin_amdgcn_s_sleep(1);
                asm volatile("" ::: "memory");                                // the next pass re-reads memory
            }
        }
        __syncthreads();
        stamp(2);
        // every minimum, ONE 16-byte write-through-coherent load per thread (PSH_FUSED_MAX_UNITS = 4 x 1024)
        unsigned kmin = 0xffffffffu, kmax = 0u;
        int nfin = 0;
        {
            const __amdgpu_buffer_rsrc_t rmin = g_rsrc(hdr->minima, sizeof(hdr->minima));
            const u32x4v mv = __builtin_amdgcn_raw_buffer_load_b128(rmin, tid * 16, 0, PSH_AUX_SC1);
            *reinterpret_cast<u32x4v*>(keys + 4 * tid) = mv;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (4 * tid + e < nbu && mv[e] < PSH_INF_BITS) { kmin = mv[e] < kmin ? mv[e] : kmin; kmax = mv[e] > kmax ? mv[e] : kmax; ++nfin; }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const unsigned l2 = __shfl_xor(kmin, off, 64), h2 = __shfl_xor(kmax, off, 64);
            kmin = l2 < kmin ? l2 : kmin;
            kmax = h2 > kmax ? h2 : kmax;
            nfin += __shfl_xor(nfin, off, 64);
        }
        if (lane == 0) {
            atomicMin(reinterpret_cast<unsigned*>(&ctl[C_KMIN]), kmin);
            atomicMax(reinterpret_cast<unsigned*>(&ctl[C_KMAX]), kmax);
            atomicAdd(&ctl[C_NFINITE], nfin);
        }
        __syncthreads();
        const bool bail = ctl[C_BAIL] != 0 || ctl[C_NFINITE] < f.rank;
        if (bail) {
            // a poll that timed out disarms the header (the launches that follow take the separate kernels until
            // psh_workspace_init); a sample without `rank` finite minima is seen by every block alike -- all of them
            // have published and read the epoch by now, so block 0 may advance it: tags are never reused
            if (tid == 0) {
                f.status[0] = PSH_STATUS_RETRY_;
                if (ctl[C_BAIL]) g_store(&hdr->magic, 0ull);
                else if (blockIdx.x == 0) g_store(reinterpret_cast<u64*>(&hdr->epoch), (u64)(epoch + 1u));
            }
            if (blockIdx.x == 0) poison_results(f.out_d, f.out_idx, a.k, tid, PSH_SCAN_THREADS);
            return;
        }
        kmin = (unsigned)ctl[C_KMIN];
        kmax = (unsigned)ctl[C_KMAX];
        const unsigned range = kmax - kmin;
        const int hb = range ? 32 - __builtin_clz(range) : 0;                 // bits of the range
        const int shift = hb > 11 ? hb - 11 : 0;                              // (range >> shift) < 2048
        for (int g = tid; g < nbu; g += PSH_SCAN_THREADS) {
            const unsigned kb = keys[g];
            if (kb < PSH_INF_BITS) atomicAdd(&hist[(kb - kmin) >> shift], 1u);
        }
        __syncthreads();
        if (wave == 0) {
            constexpr int PER = PSH_FUSED_HIST / 64;
            unsigned h[PER];
            unsigned sl = 0;
#pragma unroll
            for (int q = 0; q < PER; ++q) { h[q] = hist[PER * lane + q]; sl += h[q]; }
            unsigned inc = sl;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const unsigned t2 = __shfl_up(inc, off, 64);
                if (lane >= off) inc += t2;
            }
            unsigned cum = inc - sl;
            const unsigned rk = (unsigned)f.rank;
            if (cum < rk && inc >= rk) {                                      // exactly one lane
                int bucket = PER * lane;
#pragma unroll
                for (int q = 0; q < PER; ++q) {
                    if (cum < rk && cum + h[q] >= rk) bucket = PER * lane + q;
                    cum += h[q];
                }
                // every minimum in buckets <= `bucket` is at or below the bucket's upper edge, and there are >= rank of them
                u64 edge = (u64)kmin + (((u64)bucket + 1ull) << shift) - 1ull;
                if (edge > (u64)kmax) edge = kmax;
                ctl[C_EDGE] = (int)(unsigned)edge;
            }
            wave_lds_fence();
            if (lane == 0) derive_levels(__uint_as_float((unsigned)ctl[C_EDGE]) * PSH_TAU_MARGIN);
        }
        __syncthreads();
        if (ctl[C_BAIL] != 0) {                      // not armed: the same verdict in every block (same minima, same query)
            if (tid == 0) {
                f.status[0] = PSH_STATUS_RETRY_;
                if (blockIdx.x == 0) g_store(reinterpret_cast<u64*>(&hdr->epoch), (u64)(epoch + 1u));
            }
            if (blockIdx.x == 0) poison_results(f.out_d, f.out_idx, a.k, tid, PSH_SCAN_THREADS);
            return;
        }
    }
    const float tau2 = __uint_as_float((unsigned)__builtin_amdgcn_readfirstlane(ctl[C_TAU2]));
    const float thr2 = __uint_as_float((unsigned)__builtin_amdgcn_readfirstlane(ctl[C_THR2]));
    const float scale = __uint_as_float((unsigned)__builtin_amdgcn_readfirstlane(ctl[C_SCALE]));
    const float xn = __uint_as_float((unsigned)__builtin_amdgcn_readfirstlane(ctl[C_XN]));

    stamp(3);
    // ------------------------------------------------------------------ C: the scan (scan_mx_kernel's loop)
    {   // the f16 arrays held the minima: every slot a segment does not write must be finite (0 * NaN poisons a row)
        unsigned* z = reinterpret_cast<unsigned*>(a1);
        for (int i = lane; i < PSH_MX_NHALF; i += 64) z[i] = 0u;              // 2 arrays x NHALF halves = NHALF dwords
    }
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
    wave_lds_fence();
    while (u < u_hi) {
        unsigned ri, sg;
        decode(u, ri, sg);
        const int64_t row = a.row0 + (int64_t)ri * a.row_stride;
        const int seg_start = (int)sg * PSH_SEG;
        const int r_global = (int)(row + a.r_offset);

        stage_store(st, tile, nfloat, lane);
        {
            const int nq = (nfloat + 3) >> 2;
#pragma unroll
            for (int q = 0; q < PSH_NSTAGE; ++q) {
                const int m = lane + 64 * q;
                if (q < PSH_NSTAGE - 1 || m < nq) {
                    const f32x4 v = st.v[q] * scale;
                    const f32x4 v2 = v * v;
                    *reinterpret_cast<f16x4*>(a1 + mx_half(4 * m)) = __builtin_convertvector(v, f16x4);
                    *reinterpret_cast<f16x4*>(a2 + mx_half(4 * m)) = __builtin_convertvector(v2, f16x4);
                }
            }
        }
        wave_lds_fence();
#ifdef PSH_TUNING
        // PSH_DBG bit 4: the launch's SKELETON -- sample, both barriers, level, ranking, and ONE unit per wave (the one
        // requested before the level is known) instead of the block's share of the ensemble (tools/fused_skeleton.py;
        // results are invalid: the ranking sees too few candidates and says RETRY)
        const unsigned un = (a.dbg & 16) ? u_hi : grab();
#else
        const unsigned un = grab();
#endif
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
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[s], bx[s], acc, 0, 0, 0);
        bool keep = false;                                 // NaN-safe: !(t^ > thr)
#pragma unroll
        for (int r = 0; r < 16; ++r) keep = keep || !(acc[r] > thr2);
        if (__any(keep)) {
            unsigned hm = 0u;
#pragma unroll
            for (int r = 0; r < 16; ++r) hm |= !(acc[r] > thr2) ? (1u << r) : 0u;
#pragma unroll 1
            for (int r = 0; r < 16; ++r) {
                const int p = 32 * ((r & 3) + 8 * (r >> 2) + 4 * hk) + m;      // C layout: row -> window
                bool hit = (((hm >> r) & 1u) != 0u) && (seg_start + p < a.Tp);
                if (!__ballot(hit)) continue;
                float v = 0.0f;
                if (hit) { if constexpr (WT > 0) v = exact_one<(WT > 0 ? WT : 20)>(tile, p, x); else v = exact_one_rt(tile, p, x, W); }
                hit = hit && (v < tau2);
                const unsigned long long mask = __ballot(hit);
                if (!mask) continue;
                int base = 0;
                if (lane == 0) base = atomicAdd(&ctl[C_FRONT], __popcll(mask));
                base = __builtin_amdgcn_readfirstlane(base);
                if (hit) {
                    const int slot = base + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
                    if (slot < PSH_FUSED_FRONT) fl[slot] = u32x4{__float_as_uint(v), (unsigned)r_global, (unsigned)(seg_start + p), 0u};
                }
            }
        }
        wave_lds_fence();  // all lanes done with the tile before it is overwritten
        u = un;
    }

    // ------------------------------------------------------------------ D: distributed selection
    stamp(4);
    __syncthreads();
    stamp(5);
    const int nfront = ctl[C_FRONT];
    const int mown = nfront < PSH_FUSED_FRONT ? nfront : PSH_FUSED_FRONT;
    if (wave == 0) {
        if (lane < mown) {                                   // distance bits in place of acc; publish {r | d, t}
            u32x4 e = fl[lane];
            e[0] = __float_as_uint(dist_from_acc(__uint_as_float(e[0]), xn));
            fl[lane] = e;
            u64* c = hdr->cand + ((size_t)blockIdx.x * PSH_FUSED_FRONT + lane) * 2;
            g_store(c, ((u64)e[1] << 32) | (u64)e[0]);
            g_store(c + 1, (u64)e[2]);
        }
        if (lane == 0) g_store(&hdr->blk2[blockIdx.x], ((u64)tagD << 32) | (u64)__float_as_uint(tau2));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the storing wave drains before the count is visible
        if (lane == 0)
            g_store(&hdr->blk[blockIdx.x], ((u64)tagD << 32) | (u64)(unsigned)mown | (nfront > PSH_FUSED_FRONT ? 0x80000000ull : 0ull));
        // second barrier: sweep the counts of all blocks
        const int nblk = (int)gridDim.x;
        unsigned cv[PSH_FUSED_MAX_BLOCKS / 64];
        const __amdgpu_buffer_rsrc_t rblk = g_rsrc(hdr->blk, sizeof(hdr->blk) + sizeof(hdr->blk2));   // blk, then blk2
        for (;;) {
            bool ok = true;
            u64 xg[PSH_FUSED_MAX_BLOCKS / 64];
#pragma unroll
            for (int i = 0; i < PSH_FUSED_MAX_BLOCKS / 64; ++i) xg[i] = g_load_b64(rblk, (unsigned)(lane + 64 * i) * 8u);
#pragma unroll
            for (int i = 0; i < PSH_FUSED_MAX_BLOCKS / 64; ++i) {
                const int b = lane + 64 * i;
                cv[i] = 0u;
                if (b < nblk) {
                    cv[i] = (unsigned)xg[i];
                    ok = ok && ((unsigned)(xg[i] >> 32) == tagD);
                }
            }
            if (__all(ok)) break;
            if (give_up()) { if (lane == 0) ctl[C_BAIL] = 1; break; }
            __builtin_amdgcn_s_sleep(4);
            asm volatile("" ::: "memory");
        }
        int tot = 0, ovf = 0;
        // every block must have admitted with the SAME tau2: "at least k candidates" proves the top-k complete only then
        // (the tags make a mixed view of the minima impossible; this is the belt to those braces)
        {
            u64 x2[PSH_FUSED_MAX_BLOCKS / 64];
#pragma unroll
            for (int i = 0; i < PSH_FUSED_MAX_BLOCKS / 64; ++i) x2[i] = g_load_b64(rblk, (unsigned)(PSH_FUSED_MAX_BLOCKS + lane + 64 * i) * 8u);
#pragma unroll
            for (int i = 0; i < PSH_FUSED_MAX_BLOCKS / 64; ++i) {
                const int b = lane + 64 * i;
                if (b < nblk && ((unsigned)(x2[i] >> 32) != tagD || (unsigned)x2[i] != __float_as_uint(tau2))) ovf = 1;
            }
        }
        // counts -> exclusive prefix offs[b] (block b = 4 lane + i: every lane owns four consecutive blocks)
        int c4[PSH_FUSED_MAX_BLOCKS / 64];
        int mine = 0;
#pragma unroll
        for (int i = 0; i < PSH_FUSED_MAX_BLOCKS / 64; ++i) {
            ovf |= (cv[i] >> 31) ? 1 : 0;
            const int b = lane + 64 * i;
            cnts[b] = b < nblk ? (int)(cv[i] & 0x7fffffffu) : 0;
        }
        wave_lds_fence();
#pragma unroll
        for (int i = 0; i < PSH_FUSED_MAX_BLOCKS / 64; ++i) { c4[i] = cnts[(PSH_FUSED_MAX_BLOCKS / 64) * lane + i]; mine += c4[i]; }
        int inc = mine;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int t2 = __shfl_up(inc, off, 64);
            if (lane >= off) inc += t2;
        }
        int run = inc - mine;
#pragma unroll
        for (int i = 0; i < PSH_FUSED_MAX_BLOCKS / 64; ++i) { offs[(PSH_FUSED_MAX_BLOCKS / 64) * lane + i] = run; run += c4[i]; }
        tot = __shfl(inc, 63, 64);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) ovf |= __shfl_xor(ovf, off, 64);
        if (lane == 0) { offs[PSH_FUSED_MAX_BLOCKS] = tot; ctl[C_NTOTAL] = tot; ctl[C_ANYOVF] = ovf; }
    }
    __syncthreads();
    stamp(6);
    if (ctl[C_BAIL] != 0) {
        if (tid == 0) { f.status[0] = PSH_STATUS_RETRY_; g_store(&hdr->magic, 0ull); }
        if (blockIdx.x == 0) poison_results(f.out_d, f.out_idx, a.k, tid, PSH_SCAN_THREADS);
        return;
    }
    const int ntotal = ctl[C_NTOTAL];
    constexpr int NE = 8;                                    // live candidates per thread the ranking holds: 8192 in all
    const bool good = !ctl[C_ANYOVF] && ntotal >= a.k && ntotal <= NE * PSH_SCAN_THREADS;
    if (good) {
        // the live candidates, compacted: candidate c = tid + 1024 i sits in slot c - offs[b] of the block b that owns
        // it (binary search over the 257 prefix sums in LDS); all loads of a thread in flight together
        const int ns = (ntotal + PSH_SCAN_THREADS - 1) / PSH_SCAN_THREADS;
        const __amdgpu_buffer_rsrc_t rc = g_rsrc(hdr->cand, sizeof(hdr->cand));
        u32x4v ent[NE];
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            ent[i] = u32x4v{0xffffffffu, 0xffffffffu, 0xffffffffu, 0u};     // past the end: above every real candidate
            if (i < ns) {                                    // block-uniform
                int c = tid + PSH_SCAN_THREADS * i;
                c = c < ntotal ? c : ntotal - 1;             // (clamped lanes re-read the last one: harmless, masked below)
                int lo = 0, hi = PSH_FUSED_MAX_BLOCKS;       // offs[lo] <= c < offs[hi]
#pragma unroll
                for (int st2 = 0; st2 < 8; ++st2) {
                    const int mid = (lo + hi) >> 1;
                    if (offs[mid] <= c) lo = mid; else hi = mid;
                }
                ent[i] = __builtin_amdgcn_raw_buffer_load_b128(rc, (int)((unsigned)(lo * PSH_FUSED_FRONT + (c - offs[lo])) * 16u), 0, PSH_AUX_SC1);
            }
        }
        unsigned kd[NE];
        u64 krt[NE];
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const bool live = i < ns && tid + PSH_SCAN_THREADS * i < ntotal;
            kd[i] = live ? ent[i][0] : 0xffffffffu;
            krt[i] = live ? (((u64)ent[i][1] << 32) | (u64)ent[i][2]) : ~0ull;
        }
        if (f.tbits >= 0) {
            // rank of an own candidate = candidates below it in (d, r, t), counted on the vector ALUs 8 own ones a turn
            // (rank_select_kernel's loop): (r, t) packs into 32 bits (the launcher checked), a 64-bit compare and an add
            // per pair, two counters to a register through a DPP wave sum.  (A ballot + popcount per pair keeps the scalar
            // unit of four waves busy: 2.7 of the phase's 5 us.)
            u64 key[NE];
#pragma unroll
            for (int i = 0; i < NE; ++i)
                key[i] = kd[i] == 0xffffffffu ? ~0ull : (((u64)kd[i] << 32) | (u64)((ent[i][1] << f.tbits) | ent[i][2]));
            for (int j0 = 0; j0 < mown; j0 += 8) {
                u64 ok[8];
                int c[8];
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) {
                    const u32x4 o = fl[j0 + jj < mown ? j0 + jj : 0];
                    ok[jj] = j0 + jj < mown ? (((u64)o[0] << 32) | (u64)((o[1] << f.tbits) | o[2])) : 0ull;
                    c[jj] = 0;
                }
#pragma unroll
                for (int i = 0; i < NE; ++i) {
                    if (i < ns) {
#pragma unroll
                        for (int jj = 0; jj < 8; ++jj) c[jj] += key[i] < ok[jj] ? 1 : 0;
                    }
                }
#pragma unroll
                for (int jj = 0; jj < 8; jj += 2) {
                    int v = c[jj] | (c[jj + 1] << 16);      // <= 8 per lane and counter
                    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);
                    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);
                    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);
                    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);
                    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, true);
                    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, true);
                    v = __builtin_amdgcn_readlane(v, 63);
                    if (lane == 0) {
                        if (j0 + jj < mown && (v & 0xffff)) atomicAdd(&rankc[j0 + jj], v & 0xffff);
                        if (j0 + jj + 1 < mown && (v >> 16)) atomicAdd(&rankc[j0 + jj + 1], v >> 16);
                    }
                }
            }
        } else
        for (int i2 = 0; i2 < mown; ++i2) {                 // rank of own candidate i2 = candidates below it in (d, r, t)
            const u32x4 o = fl[i2];
            const unsigned od = o[0];
            const u64 ort = ((u64)o[1] << 32) | (u64)o[2];
            int c = 0;
#pragma unroll
            for (int i = 0; i < NE; ++i) {
                if (i < ns) {
                    const bool less = kd[i] < od || (kd[i] == od && krt[i] < ort);
                    c += (int)__popcll(__ballot(less));
                }
            }
            if (lane == 0 && c) atomicAdd(&rankc[i2], c);
        }
        __syncthreads();
        if constexpr (BLK) {
            // a blocking caller: every winner of this block leaves with its path (shadow()'s gather, reference
            // path_shadowing.py:211-216: C x (W + h) samples from the ensemble) -- a wave per winner, everything as write-through
            // system-scope stores into the caller's pinned block (no gather launch, no copy, nothing left in an L2)
            for (int j = wave; j < mown; j += NW) {
                const int rk = rankc[j];
                if (rk >= a.k) continue;
                const u32x4 o = fl[j];
                if (lane < 3) sys_store32(lane == 0 ? reinterpret_cast<unsigned*>(f.out_d + rk) : reinterpret_cast<unsigned*>(f.out_idx + 2 * rk + (lane - 1)), o[lane]);
                const float* src = f.g_ds + ((int64_t)o[1] - a.r_offset) * f.g_C * f.g_T + (int64_t)o[2];
                unsigned* dst = reinterpret_cast<unsigned*>(f.g_out + (int64_t)rk * f.g_C * f.g_len);
                for (int c = 0; c < f.g_C; ++c)
                    for (int e = lane; e < f.g_len; e += 64) sys_store32(dst + c * f.g_len + e, __float_as_uint(src[(int64_t)c * f.g_T + e]));
            }
        } else if (tid < mown) {
            const int rk = rankc[tid];
            if (rk < a.k) {
                const u32x4 o = fl[tid];
                f.out_d[rk] = __uint_as_float(o[0]);
                f.out_idx[2 * rk + 0] = (int)o[1];
                f.out_idx[2 * rk + 1] = (int)o[2];
            }
        }
    }
    stamp(7);
    if (blockIdx.x == 0 && !good) poison_results(f.out_d, f.out_idx, a.k, tid, PSH_SCAN_THREADS);   // (`good` is the same in every block: none wrote a rank)
    if (blockIdx.x == 0 && tid == 0) {
        // RETRY is sticky: a block that gave up at the second barrier (its deadline passed while THIS block, dispatched
        // late, had not published yet) wrote RETRY, returned without its out[rank] rows and disarmed the header BEFORE
        // this point -- its give-up store follows its last failing poll by one round trip, this load follows this block's
        // publication by the candidate loads and the ranking (several round trips).  OK is only declared over a header
        // that is still armed.
        const bool armed = g_load(&hdr->magic) == PSH_FUSED_MAGIC;
        sys_store32(reinterpret_cast<unsigned*>(f.status), (good && armed) ? PSH_STATUS_OK_ : PSH_STATUS_RETRY_);   // (a blocking caller's word sits in host memory)
        if (f.total) f.total[0] = ntotal;
        if (a.qstate) {                                     // diagnostics / the separate launches' state, kept coherent
            QueryState q;
            q.xn = xn; q.tau_bits = __float_as_uint(tau2); q.n_valid = good ? a.k : 0; q.nx = 0.0f;
            q.thr_base = __uint_as_float(PSH_INF_BITS); q.mx_scale = scale; q.mx_thr = thr2;
            q.tau2_bits = __float_as_uint(tau2); q.mx_thr2 = thr2; q.mx8_P = q.mx8_L = q.mx8_k1 = 0.0f;
            a.qstate[0] = q;
        }
        // every block has read the epoch long ago (they all passed the first barrier): the next launch's tags
        g_store(reinterpret_cast<u64*>(&hdr->epoch), (u64)(epoch + 1u));
    }
    if constexpr (BLK) {
        // completion words of a blocking call: every wave's stores have been acknowledged (vmcnt) before its block takes a
        // ticket of its shard; the block that takes the shard's last ticket resets the counter for the next launch (launches on
        // a workspace are serialised) and tells the host.  The launches that return early (RETRY before the scan) set no word:
        // the host's wait falls back on the stream (psh_capi.hip).
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            const unsigned shard = blockIdx.x % (unsigned)f.done_shards;
            const unsigned in_shard = (gridDim.x - shard + (unsigned)f.done_shards - 1u) / (unsigned)f.done_shards;
            unsigned* cnt = &hdr->pad[4 + shard];
            const unsigned old = __hip_atomic_fetch_add((gu32*)cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (old + 1u == in_shard) {
                g_store32(cnt, 0u);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                sys_store32(f.done + shard, f.done_val);
            }
        }
    }
}

__global__ void fused_init_kernel(FusedHdr* hdr) {
    // tags of a fresh header can never match: epoch starts at 1 (tags 2, 3), the arrays are zeroed
    const size_t n = sizeof(FusedHdr) / 8;
    u64* p = reinterpret_cast<u64*>(hdr);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 0ull;
}
__global__ void fused_arm_kernel(FusedHdr* hdr) { hdr->epoch = 1u; hdr->pad[0] = 0u; hdr->magic = PSH_FUSED_MAGIC; }

hipError_t launch_fused_init(FusedHdr* hdr, hipStream_t s) {
    hipLaunchKernelGGL(fused_init_kernel, dim3(64), dim3(256), 0, s, hdr);
    hipLaunchKernelGGL(fused_arm_kernel, dim3(1), dim3(1), 0, s, hdr);       // after the zero fill (same stream)
    return hipGetLastError();
}

bool scan_fused_supported(int W) { return W >= 1 && W <= 33; }

size_t scan_fused_shmem_bytes(int tile_floats) {
    return (size_t)PSH_FUSED_FIXED_BYTES + (size_t)tile_floats * (PSH_SCAN_THREADS / 64) * sizeof(float)
           + (size_t)(PSH_SCAN_THREADS / 64) * 2 * PSH_MX_NHALF * sizeof(_Float16);
}

template <typename K>
static hipError_t launch_fused_k(K kernel, int grid, size_t shmem, hipStream_t s, const ScanArgs& a, const FusedArgs& f) {
    hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(PSH_SCAN_THREADS), shmem, s, a, f);
    return hipGetLastError();
}

template <bool HINTED, bool BLK>
static hipError_t launch_fused_hb(const ScanArgs& a, const FusedArgs& f, bool aligned, int grid, size_t shmem, hipStream_t s) {
    if (a.W == 20)
        return aligned ? launch_fused_k(scan_fused_kernel<20, true, HINTED, BLK>, grid, shmem, s, a, f)
                       : launch_fused_k(scan_fused_kernel<20, false, HINTED, BLK>, grid, shmem, s, a, f);
    return aligned ? launch_fused_k(scan_fused_kernel<0, true, HINTED, BLK>, grid, shmem, s, a, f)
                   : launch_fused_k(scan_fused_kernel<0, false, HINTED, BLK>, grid, shmem, s, a, f);
}

hipError_t launch_scan_fused(const ScanArgs& a, const FusedArgs& f, bool aligned, int grid, hipStream_t s) {
    const size_t shmem = scan_fused_shmem_bytes(a.tile_floats);
    const bool blk = f.g_ds != nullptr && f.done != nullptr;
    if (f.tau_hint) return blk ? launch_fused_hb<true, true>(a, f, aligned, grid, shmem, s) : launch_fused_hb<true, false>(a, f, aligned, grid, shmem, s);
    return blk ? launch_fused_hb<false, true>(a, f, aligned, grid, shmem, s) : launch_fused_hb<false, false>(a, f, aligned, grid, shmem, s);
}

}  // namespace psh
