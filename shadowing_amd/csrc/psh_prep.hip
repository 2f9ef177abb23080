// psh_prep.hip -- non-finite samples in the ensemble, the way the reference treats them.
//
// The reference embeds with a conv1d whose kernel is zero-padded by the horizon (path_embedding.py:48-51, :129-132), and
// 0 * NaN = 0 * inf = NaN: every embedded coordinate of window t is NaN as soon as ONE sample of the conv's field
// y[r, :, t : t+K+h] -- the window, its h future samples, any channel -- is NaN or +-inf (probed on the reference:
// tests/golden/nan_in_ensemble_*.npz); torch.topk(largest=False) then ranks the window last (path_shadowing.py:165).
// The scans of this library judge a window by the taps [lo, hi) of its K samples that some kernel row's span covers
// (Identity: all of them; Foveal(.., 126) with its longest row of 115 samples: the last 115).  Both agree on an ensemble in
// which a non-finite sample at p has been written over [p - h - (K - hi), p + lo] as NaN: the covered taps of window t
// then hold a NaN iff the field [t, t+K+h) held a non-finite sample.  psh_count_nonfinite says whether an ensemble needs that (almost none does: one pass, once
// per resident ensemble), psh_smear_nonfinite builds the (R, T) rows the scan reads; paths are gathered from the
// original.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "psh_kernels.h"

namespace psh {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ bool nonfinite(float v) { return (__float_as_uint(v) & 0x7f800000u) == 0x7f800000u; }

__global__ __launch_bounds__(256) void count_nonfinite_kernel(const float* __restrict__ x, int64_t n, unsigned long long* out) {
    const int64_t n4 = n >> 2;
    const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
    unsigned c = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const f32x4 v = __builtin_nontemporal_load(x4 + i);
        c += nonfinite(v[0]) + nonfinite(v[1]) + nonfinite(v[2]) + nonfinite(v[3]);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) c += nonfinite(x[(n4 << 2) + threadIdx.x]);
    const unsigned long long m = __ballot(c != 0);
    if (m) {                                             // (rare: a wave that saw none does not touch the counter)
        unsigned tot = c;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) tot += __shfl_xor(tot, off, 64);
        if ((threadIdx.x & 63) == 0) atomicAdd(out, (unsigned long long)tot);
    }
}

hipError_t launch_count_nonfinite(const float* x, int64_t n, unsigned long long* out, hipStream_t s) {
    hipError_t e = hipMemsetAsync(out, 0, sizeof(unsigned long long), s);
    if (e != hipSuccess) return e;
    if (n <= 0) return hipSuccess;
    int64_t blocks = ((n >> 2) + 256 * 8 - 1) / (256 * 8);
    blocks = blocks < 1 ? 1 : (blocks > 4096 ? 4096 : blocks);
    hipLaunchKernelGGL(count_nonfinite_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x, n, out);
    return hipGetLastError();
}

// flags[r] = 1 if row r (any of its C channels) holds a non-finite sample, else 0: a wave per row.  What the embedded scans'
// callers split a dirty ensemble by (psh.h: psh_rows_nonfinite) -- clean rows keep the sampled scan and its rejection tests,
// the few dirty ones take the exhaustive dense chains, which meet a NaN the way the reference's conv does.
__global__ __launch_bounds__(256) void rows_nonfinite_kernel(const float* __restrict__ ds, int64_t R, int64_t row_len, int* __restrict__ flags) {
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    const int lane = (int)(threadIdx.x & 63);
    const float* row = ds + r * row_len;
    bool bad = false;
    for (int64_t p = lane; p < row_len; p += 64) bad = bad || nonfinite(row[p]);
    const unsigned long long m = __ballot(bad);
    if (lane == 0) flags[r] = m ? 1 : 0;
}

hipError_t launch_rows_nonfinite(const float* ds, int64_t R, int64_t row_len, int* flags, hipStream_t s) {
    if (R <= 0) return hipSuccess;
    hipLaunchKernelGGL(rows_nonfinite_kernel, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, s, ds, R, row_len, flags);
    return hipGetLastError();
}

// out[r, q] = NaN if any channel holds a non-finite sample in [q - fwd, q + back] (clipped to the row), else dataset[r, 0, q].
// A thread per output sample; the look-around stops at the first hit.  Only run for an ensemble that holds non-finite
// samples at all (psh_count_nonfinite), once per resident copy.
__global__ __launch_bounds__(256) void smear_nonfinite_kernel(const float* __restrict__ ds, int64_t R, int64_t C, int64_t T,
                                                              int back, int fwd, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R * T) return;
    const int64_t r = i / T, q = i - r * T;
    const int64_t lo = (q - fwd > 0) ? q - fwd : 0;
    const int64_t hi = (q + back < T - 1) ? q + back : T - 1;
    bool bad = false;
    for (int64_t c = 0; c < C && !bad; ++c) {
        const float* row = ds + (r * C + c) * T;
        for (int64_t p = lo; p <= hi; ++p)
            if (nonfinite(row[p])) { bad = true; break; }
    }
    out[i] = bad ? __uint_as_float(0x7fc00000u) : ds[r * C * T + q];
}

hipError_t launch_smear_nonfinite(const float* ds, int64_t R, int64_t C, int64_t T, int back, int fwd, float* out, hipStream_t s) {
    const int64_t n = R * T;
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(smear_nonfinite_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, ds, R, C, T, back, fwd, out);
    return hipGetLastError();
}

}  // namespace psh
