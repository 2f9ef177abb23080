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
//   * one WAVE (64 lanes) owns a segment of 1024 consecutive windows of one row:
//     it streams the 4 KB (+ W-1 halo) with coalesced 16-byte loads, stages them
//     in a wave-private, bank-conflict-free LDS tile, and every lane walks 16
//     consecutive windows with a 16-register sliding window, so each dataset
//     element is fetched from HBM once and from LDS ~1.3 times.
//   * the per-window chain is kept in the reference's exact order (bit-exact
//     distances are what make indices bit-exact) -- no tree/shuffle reduction.
//   * selection never ranks on anything but the exact value: a cheap sample pass
//     gives a provable upper bound tau on the k-th smallest acc, the scan keeps
//     only windows below tau (a few thousand out of 1e8), and a one-block radix
//     select + bitonic sort orders the survivors by (d, r, t).
//   * no MFMA: the work is a streaming scan, not a contraction.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "psh_kernels.h"

namespace psh {

// ----------------------------------------------------------------------------------
// small helpers
// ----------------------------------------------------------------------------------
typedef const __attribute__((address_space(4))) float* const_f32p;  // scalar (SGPR) loads
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63u); }

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

// sum of squares in the order of ATen's contiguous last-dim norm reduce (see
// oracle/psh_oracle.c: psh_oracle_sumsq8): 8 lanes of fma over whole blocks of 8,
// lanes added left to right, tail: groups of 4 as rounded products added one by one,
// then a scalar fma chain for the last < 4.
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

__device__ __forceinline__ int hist_key(float acc, int base) {
    // log-spaced bins for free: sign(0) | 8 exponent bits | 7 mantissa bits
    int key = (int)(__float_as_uint(acc) >> 16) - base;
    key = key < 0 ? 0 : key;
    return key > (PSH_NBINS - 1) ? (PSH_NBINS - 1) : key;
}

// ----------------------------------------------------------------------------------
// K0: per-query preparation -- ||x||, histogram base, state reset
// ----------------------------------------------------------------------------------
__global__ void prep_kernel(PrepArgs a) {
    const int b = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (b >= a.B) return;
    const float* x = a.queries + (int64_t)b * a.W;
    const float s = sumsq8([&](int j) { return x[j]; }, a.W);
    a.qstate[b].xn = a.qnorm_in ? a.qnorm_in[b] : __builtin_sqrtf(s);
    // bins cover acc in [s/64, s*2^10): d in [0.125, 32)
    int base = (int)(__float_as_uint(s) >> 16) - 6 * 128;
    a.qstate[b].base = base < 0 ? 0 : base;
    a.qstate[b].tau = __uint_as_float(0x7f800000u);  // +inf until K2 lowers it
    a.qstate[b].n_valid = 0;
    a.counts[b] = 0;
    if (a.status) a.status[b] = PSH_STATUS_OK_;
}

__global__ void qnorm_kernel(const float* queries, int B, int W, float* out) {
    const int b = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (b >= B) return;
    const float* x = queries + (int64_t)b * W;
    out[b] = __builtin_sqrtf(sumsq8([&](int j) { return x[j]; }, W));
}

// ----------------------------------------------------------------------------------
// K1/K3: the sliding-window scan
// ----------------------------------------------------------------------------------
// Per-lane accumulation of the L=16 consecutive windows starting at logical tile
// index 16*lane.  win[s] holds y[16*lane + m] for the newest m = s (mod 16); at step
// j window i reads slot (i + j) & 15 and slot j & 15 is then refilled with y[.. + j + 16].
// The chain over j is strictly sequential per window: the reference's order.
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
#pragma unroll
                for (int i = 0; i < PSH_L; ++i) {
                    const float D = __fsub_rn(xj, win[(i + jj) & 15]);
                    acc[i] = __builtin_fmaf(D, D, acc[i]);
                }
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
#pragma unroll
                    for (int i = 0; i < PSH_L; ++i) {
                        const float D = __fsub_rn(xj, win[(i + jj) & 15]);
                        acc[i] = __builtin_fmaf(D, D, acc[i]);
                    }
                    win[jj] = nv[q];
                }
            }
        }
    }
}

// One-window-per-row edge case (T == W + h): the reference's numerator uses the
// 8-lane order instead of the sequential chain (oracle/psh_oracle.c).  Only lane
// window 0 of segment 0 exists; computed by every lane for its first window only.
__device__ inline float acc_single_window(const float* tile, int lane, const_f32p x, int W) {
    const int base = PSH_L * lane;
    return sumsq8([&](int j) { return __fsub_rn(x[j], tile[lds_pad(base + j)]); }, W);
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

template <bool ALIGNED>
__device__ __forceinline__ void stage_load(Stage& st, const float* __restrict__ row, int64_t T,
                                           int seg_start, int nfloat, int lane) {
    // row: first float of the row; floats [seg_start, seg_start + nfloat) wanted, clamped
    // to the row (the clamped tail only feeds inadmissible windows).
    if (ALIGNED) {
        const f32x4* src = reinterpret_cast<const f32x4*>(row + seg_start);
        const int last = (int)((T - seg_start) >> 2) - 1;  // last float4 inside the row
        const int nq = (nfloat + 3) >> 2;
#pragma unroll
        for (int q = 0; q < PSH_NSTAGE; ++q) {
            int m = lane + 64 * q;
            if (m < nq) {
                m = m > last ? last : m;
                st.v[q] = __builtin_nontemporal_load(src + m);
            }
        }
    } else {
        const int lastf = (int)(T - seg_start) - 1;
#pragma unroll
        for (int q = 0; q < PSH_NSTAGE; ++q) {
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
}

__device__ __forceinline__ void stage_store(const Stage& st, float* tile, int nfloat, int lane) {
    const int nq = (nfloat + 3) >> 2;
#pragma unroll
    for (int q = 0; q < PSH_NSTAGE; ++q) {
        const int m = lane + 64 * q;
        if (m < nq) *reinterpret_cast<f32x4*>(tile + lds_pad(4 * m)) = st.v[q];
    }
}

// MODE_SAMPLE: histogram of per-lane minima (a subset of the windows => its k-th
//              smallest is an upper bound of the global k-th smallest)
// MODE_FILTER: append windows with acc < tau
// MODE_ALL   : append every admissible window (exhaustive path)
template <int WT, bool ALIGNED, int MODE>
__global__ __launch_bounds__(PSH_SCAN_THREADS) void scan_kernel(ScanArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = lane_id();
    const int wave_in_block = (int)(threadIdx.x >> 6);
    float* tile = smem + (size_t)wave_in_block * a.tile_floats;

    const int W = WT > 0 ? WT : a.W;
    const int nfloat = PSH_SEG + W - 1;
    const int64_t n_rs = (int64_t)a.n_rows * a.nseg;      // (row, segment) units
    const int64_t n_units = n_rs * a.n_qgroups;
    const int64_t gw = (int64_t)blockIdx.x * (PSH_SCAN_THREADS / 64) + wave_in_block;
    const int64_t GW = (int64_t)gridDim.x * (PSH_SCAN_THREADS / 64);
    const const_f32p xq = (const_f32p)a.queries;

    Stage st;
    int64_t u = gw;
    if (u < n_units) {
        const int64_t rs = u % n_rs;
        const int64_t row = a.row0 + (rs / a.nseg) * a.row_stride;
        stage_load<ALIGNED>(st, a.dataset + row * a.T, a.T, (int)(rs % a.nseg) * PSH_SEG, nfloat, lane);
    }
    for (; u < n_units; u += GW) {
        const int64_t rs = u % n_rs;
        const int qg = (int)(u / n_rs);
        const int64_t row = a.row0 + (rs / a.nseg) * a.row_stride;
        const int seg_start = (int)(rs % a.nseg) * PSH_SEG;

        stage_store(st, tile, nfloat, lane);
        wave_lds_fence();
        {   // prefetch the next unit of this wave while this one is computed
            const int64_t un = u + GW;
            if (un < n_units) {
                const int64_t rsn = un % n_rs;
                const int64_t rown = a.row0 + (rsn / a.nseg) * a.row_stride;
                stage_load<ALIGNED>(st, a.dataset + rown * a.T, a.T, (int)(rsn % a.nseg) * PSH_SEG, nfloat, lane);
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
            float acc[PSH_L];
            if (a.Tp == 1) {
#pragma unroll
                for (int i = 0; i < PSH_L; ++i) acc[i] = 0.0f;
                acc[0] = acc_single_window(tile, lane, x, W);
            } else {
                accumulate16<WT>(tile, lane, x, W, acc);
            }

            if (MODE == PSH_MODE_SAMPLE) {
                float m;
                if (__all(nvalid == PSH_L)) {
                    m = min16(acc);
                } else {
                    m = __uint_as_float(0x7f800000u);
#pragma unroll
                    for (int i = 0; i < PSH_L; ++i) m = (i < nvalid) ? fminf(m, acc[i]) : m;
                }
                if (nvalid > 0) atomicAdd(a.hist + (int64_t)b * PSH_NBINS + hist_key(m, a.qstate[b].base), 1u);
            } else {
                const float tau = (MODE == PSH_MODE_FILTER) ? a.qstate[b].tau : 0.0f;
                bool any_hit;
                if (MODE == PSH_MODE_FILTER) any_hit = __any(min16(acc) < tau);
                else any_hit = true;
                if (any_hit) {  // rare in FILTER mode: ~1e-4 of the windows survive
                    const float xn = a.qstate[b].xn;
                    float* cd = a.cand_d + (int64_t)b * a.cap;
                    int2* crt = a.cand_rt + (int64_t)b * a.cap;
#pragma unroll
                    for (int i = 0; i < PSH_L; ++i) {
                        const bool hit = (i < nvalid) && (MODE == PSH_MODE_ALL || acc[i] < tau);
                        if (hit) {
                            const int pos = atomicAdd(a.counts + b, 1);
                            if (pos < a.cap) {
                                cd[pos] = dist_from_acc(acc[i], xn);
                                crt[pos] = make_int2(r_global, t_lane + i);
                            }
                        }
                    }
                }
            }
        }
        wave_lds_fence();  // all lanes done with the tile before it is overwritten
    }
}

// ----------------------------------------------------------------------------------
// K2: histogram of sample minima -> admission threshold tau (upper bin edge + margin)
// ----------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void threshold_kernel(ThresholdArgs a) {
    __shared__ unsigned part[256];
    __shared__ int found_bin;
    const int b = (int)blockIdx.x;
    const unsigned* h = a.hist + (int64_t)b * PSH_NBINS;
    constexpr int PER = PSH_NBINS / 256;
    unsigned loc[PER];
    unsigned s = 0;
#pragma unroll
    for (int i = 0; i < PER; ++i) { loc[i] = h[threadIdx.x * PER + i]; s += loc[i]; }
    part[threadIdx.x] = s;
    if (threadIdx.x == 0) found_bin = PSH_NBINS;  // "not found"
    __syncthreads();
    // exclusive prefix of the 256 partial sums (serial in one thread: 256 adds)
    if (threadIdx.x == 0) {
        unsigned run = 0;
        for (int i = 0; i < 256; ++i) { const unsigned v = part[i]; part[i] = run; run += v; }
    }
    __syncthreads();
    unsigned cum = part[threadIdx.x];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const unsigned before = cum;
        cum += loc[i];
        if (before < (unsigned)a.k && cum >= (unsigned)a.k) found_bin = (int)threadIdx.x * PER + i;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float tau = __uint_as_float(0x7f800000u);
        const int bin = found_bin;
        if (bin < PSH_NBINS - 1) {  // the last bin is the open-ended overflow bin
            // every sample value counted in bins <= bin is < edge
            const unsigned edge_bits = (unsigned)(a.qstate[b].base + bin + 1) << 16;
            if (edge_bits < 0x7f800000u) {
                // margin: a window at or above tau must have a strictly larger *distance*
                // than anything below the edge (sqrt and the division compress ulps)
                tau = __uint_as_float(edge_bits) * (1.0f + 1.0f / 262144.0f);
            }
        }
        a.qstate[b].tau = tau;
    }
}

// ----------------------------------------------------------------------------------
// K4: survivors -> k best by (d, r, t): MSB radix select on the 96-bit key, then an
// in-LDS bitonic sort of the selected k
// ----------------------------------------------------------------------------------
typedef unsigned __int128 u128;

__device__ __forceinline__ u128 key96(float d, int2 rt) {
    return ((u128)__float_as_uint(d) << 64) | ((u128)(unsigned)rt.x << 32) | (u128)(unsigned)rt.y;
}

__device__ __forceinline__ bool item_less(uint64_t x, uint64_t y, const int2* rt) {
    const unsigned dx = (unsigned)(x >> 32), dy = (unsigned)(y >> 32);
    if (dx != dy) return dx < dy;
    const unsigned sx = (unsigned)x, sy = (unsigned)y;
    if (sx == sy) return false;
    if (sx == 0xffffffffu) return false;   // padding sorts last
    if (sy == 0xffffffffu) return true;
    const int2 a = rt[sx], b = rt[sy];
    if (a.x != b.x) return (unsigned)a.x < (unsigned)b.x;
    return (unsigned)a.y < (unsigned)b.y;
}

__global__ __launch_bounds__(PSH_SELECT_THREADS) void select_kernel(SelectArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint64_t items[];   // kpad entries
    __shared__ unsigned hist[256];
    __shared__ u128 s_prefix;
    __shared__ int s_remaining, s_done, s_nsel;

    const int b = (int)blockIdx.x;
    const int tid = (int)threadIdx.x;
    const float* cd = a.cand_d + (int64_t)b * a.cand_stride;
    const int2* crt = a.cand_rt + (int64_t)b * a.cand_stride;
    int n = a.counts ? a.counts[b] : a.n_fixed;
    if (a.counts && n > a.cap) {
        n = a.cap;
        if (tid == 0 && a.status) a.status[b] = PSH_STATUS_OVERFLOW_;
    }
    int2* sel_rt = a.sel_rt + (int64_t)b * a.kpad;

    // merge inputs may carry padding entries (r < 0): only real candidates are ranked
    int n_real = n;
    if (a.skip_negative_rows) {
        if (tid == 0) s_nsel = 0;
        __syncthreads();
        int c = 0;
        for (int i = tid; i < n; i += PSH_SELECT_THREADS) c += (crt[i].x >= 0) ? 1 : 0;
        if (c) atomicAdd(&s_nsel, c);
        __syncthreads();
        n_real = s_nsel;
        __syncthreads();
    }
    const int need = a.k < n_real ? a.k : n_real;

    if (tid == 0) { s_prefix = 0; s_remaining = need; s_done = (need == n_real) ? 1 : 0; s_nsel = 0; }
    __syncthreads();

    // ---- radix select: find the truncated key below/at which exactly `need` candidates lie
    int p_done = -1;   // index of the last digit fixed in s_prefix
    if (!s_done && need > 0) {
        for (int p = 0; p < 12; ++p) {
            for (int i = tid; i < 256; i += PSH_SELECT_THREADS) hist[i] = 0;
            __syncthreads();
            const u128 prefix = s_prefix;
            const int sh_digit = 88 - 8 * p;
            for (int i = tid; i < n; i += PSH_SELECT_THREADS) {
                const int2 rt = crt[i];
                if (a.skip_negative_rows && rt.x < 0) continue;
                const u128 key = key96(cd[i], rt);
                const bool match = (p == 0) || ((key >> (sh_digit + 8)) == (prefix >> (sh_digit + 8)));
                if (match) atomicAdd(&hist[(unsigned)(key >> sh_digit) & 255u], 1u);
            }
            __syncthreads();
            if (tid == 0) {
                int rem = s_remaining;
                unsigned cum = 0;
                int bucket = 255;
                for (int i = 0; i < 256; ++i) {
                    if (cum + hist[i] >= (unsigned)rem) { bucket = i; break; }
                    cum += hist[i];
                }
                rem -= (int)cum;
                s_remaining = rem;
                s_prefix = prefix | ((u128)(unsigned)bucket << sh_digit);
                s_done = ((int)hist[bucket] == rem) ? 1 : 0;
            }
            __syncthreads();
            p_done = p;
            if (s_done) break;
        }
    }

    // ---- collect the selected candidates
    for (int i = tid; i < a.kpad; i += PSH_SELECT_THREADS) items[i] = ~0ull;
    __syncthreads();
    if (need > 0) {
        const u128 prefix = s_prefix;
        const int sh = (p_done < 0) ? 96 : (88 - 8 * p_done);
        for (int i = tid; i < n; i += PSH_SELECT_THREADS) {
            const int2 rt = crt[i];
            if (a.skip_negative_rows && rt.x < 0) continue;
            const float d = cd[i];
            const bool take = (sh >= 96) ? true : ((key96(d, rt) >> sh) <= (prefix >> sh));
            if (take) {
                const int slot = atomicAdd(&s_nsel, 1);
                if (slot < a.kpad) {
                    items[slot] = ((uint64_t)__float_as_uint(d) << 32) | (uint64_t)(unsigned)slot;
                    sel_rt[slot] = rt;
                }
            }
        }
    }
    __syncthreads();

    // ---- bitonic sort of kpad items by (d bits, r, t)
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
    __syncthreads();

    // ---- write out
    int nsel = s_nsel < need ? s_nsel : need;
    for (int i = tid; i < a.k; i += PSH_SELECT_THREADS) {
        float d = __uint_as_float(0x7f800000u);
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
}

// exhaustive path: seed the candidate buffer of the next chunk with the running best
__global__ void reseed_kernel(ReseedArgs a) {
    const int b = (int)blockIdx.y;
    const int nv = a.qstate[b].n_valid;
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i == 0) a.counts[b] = nv;
    if (i < nv) {
        a.cand_d[(int64_t)b * a.cap + i] = a.out_d[(int64_t)b * a.k + i];
        a.cand_rt[(int64_t)b * a.cap + i] =
            make_int2(a.out_idx[((int64_t)b * a.k + i) * 2], a.out_idx[((int64_t)b * a.k + i) * 2 + 1]);
    }
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
template <int WT, bool ALIGNED>
static hipError_t launch_scan_mode(const ScanArgs& a, int mode, int grid, size_t shmem, hipStream_t s) {
    switch (mode) {
        case PSH_MODE_SAMPLE:
            hipLaunchKernelGGL((scan_kernel<WT, ALIGNED, PSH_MODE_SAMPLE>), dim3(grid), dim3(PSH_SCAN_THREADS), shmem, s, a);
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

hipError_t launch_scan(const ScanArgs& a, int mode, bool aligned, int grid, hipStream_t s) {
    const size_t shmem = (size_t)a.tile_floats * (PSH_SCAN_THREADS / 64) * sizeof(float);
    if (a.W == 20) {
        return aligned ? launch_scan_mode<20, true>(a, mode, grid, shmem, s)
                       : launch_scan_mode<20, false>(a, mode, grid, shmem, s);
    }
    return aligned ? launch_scan_mode<0, true>(a, mode, grid, shmem, s)
                   : launch_scan_mode<0, false>(a, mode, grid, shmem, s);
}

hipError_t scan_blocks_per_cu(int W, bool aligned, size_t shmem, int* out) {
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

hipError_t launch_prep(const PrepArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(prep_kernel, dim3((a.B + 63) / 64), dim3(64), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_qnorm(const float* q, int B, int W, float* out, hipStream_t s) {
    hipLaunchKernelGGL(qnorm_kernel, dim3((B + 63) / 64), dim3(64), 0, s, q, B, W, out);
    return hipGetLastError();
}
hipError_t launch_threshold(const ThresholdArgs& a, int B, hipStream_t s) {
    hipLaunchKernelGGL(threshold_kernel, dim3(B), dim3(256), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_select(const SelectArgs& a, int B, hipStream_t s) {
    const size_t shmem = (size_t)a.kpad * sizeof(uint64_t);
    if (shmem > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)select_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(select_kernel, dim3(B), dim3(PSH_SELECT_THREADS), shmem, s, a);
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
