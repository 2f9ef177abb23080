// psh_predict.hip -- the reductions of predict_from_paths() on the device:
//   * moments_kernel   avg / std over the k shadowing paths of a (B, k, m) statistic with (B, k) weights
//                      (reference path_shadowing.py:245-252: proba.avg(values, axis=1), proba.std(values, axis=1));
//   * rv_kernel        the tutorial's statistic, realized variance per maturity of the out-context paths
//                      (reference shadowing/statistics.py:5-16: mean(x^2[..., :T]) * 252, its square root if vol).
// With these `predict(cuda=True, device_predict=True)` moves the (B, k) distances to the host (the installed averaging
// class turns them into weights there -- its formula is not restated) and the (B, m) moments back; the (B, k, C, W+h)
// paths and the (B, k, m) statistic never leave HBM.
// Arithmetic: the statistic's squares in fp32 as numpy computes them, every SUM in double (numpy: pairwise fp32 sums for
// the statistic, float64 for the moments), so the moments agree with the host path to ~1e-15 relative given the same
// statistic, and the statistic itself to an ulp or two of fp32.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "psh_kernels.h"

namespace psh {

// double sum over the 64 lanes of a wave (result in every lane)
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

#define PSH_MOM_THREADS 1024

// One block per (query b, chunk of up to PSH_MOM_THREADS columns).  The (k x m) slab of a query is walked with a stride that
// is a multiple of m, so a thread stays on ONE column i = t % m and the slab is read coalesced; per-thread partial sums in
// double, then the partials of a column are added in a fixed order (deterministic results).
__global__ __launch_bounds__(PSH_MOM_THREADS) void moments_kernel(MomentsArgs a) {
    __shared__ double part[PSH_MOM_THREADS];
    __shared__ double meanL[PSH_MOM_THREADS];
    const int b = (int)blockIdx.x;
    const int c0 = (int)blockIdx.y * PSH_MOM_THREADS;                       // first column of this chunk
    const int mc = (a.m - c0) < PSH_MOM_THREADS ? (a.m - c0) : PSH_MOM_THREADS;   // columns of this chunk
    const int rows_par = PSH_MOM_THREADS / mc;                               // path rows walked side by side
    const int t = (int)threadIdx.x;
    const bool live = t < rows_par * mc;
    const int i = live ? t % mc : 0, j0 = live ? t / mc : 0;
    const float* v = a.values + (int64_t)b * a.k * a.m + c0 + i;
    const double* w = a.weights ? a.weights + (int64_t)b * a.k : nullptr;
    const double wu = 1.0 / (double)a.k;                                     // uniform weights
    for (int pass = 0; pass < 2; ++pass) {
        double s = 0.0;
        const double mu = pass ? meanL[i] : 0.0;
        if (live)
            for (int j = j0; j < a.k; j += rows_par) {
                const double x = (double)v[(int64_t)j * a.m];
                const double ww = w ? w[j] : wu;
                const double e = x - mu;
                s += pass ? ww * e * e : ww * x;
            }
        part[t] = s;
        __syncthreads();
        if (t < mc) {
            double tot = 0.0;
            for (int r = 0; r < rows_par; ++r) tot += part[r * mc + t];
            if (pass == 0) {
                meanL[t] = tot;
                a.out_mean[(int64_t)b * a.m + c0 + t] = tot;
            } else {
                a.out_std[(int64_t)b * a.m + c0 + t] = sqrt(tot);
            }
        }
        __syncthreads();
    }
}

hipError_t launch_moments(const MomentsArgs& a, hipStream_t s) {
    dim3 grid((unsigned)a.B, (unsigned)((a.m + PSH_MOM_THREADS - 1) / PSH_MOM_THREADS));
    hipLaunchKernelGGL(moments_kernel, grid, dim3(PSH_MOM_THREADS), 0, s, a);
    return hipGetLastError();
}

// A wave per row: lane l squares samples l, l + 64, ... (fp32 products, as numpy's x ** 2) and adds each to the running
// sums of the maturities it lies below; a double wave sum per maturity.  Up to 8 maturities a pass.
#define PSH_RV_THREADS 256
__global__ __launch_bounds__(PSH_RV_THREADS) void rv_kernel(RvArgs a) {
    const int lane = (int)(threadIdx.x & 63);
    const int64_t row = (int64_t)blockIdx.x * (PSH_RV_THREADS / 64) + (threadIdx.x >> 6);
    if (row >= a.n_rows) return;
    const float* x = a.x + row * a.row_stride;
    for (int t0 = 0; t0 < a.nT; t0 += 8) {
        const int nt = (a.nT - t0) < 8 ? (a.nT - t0) : 8;
        int Tmax = 0;
        for (int q = 0; q < nt; ++q) Tmax = a.Ts[t0 + q] > Tmax ? a.Ts[t0 + q] : Tmax;
        double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int p = lane; p < Tmax; p += 64) {
            const float xv = x[p];
            const double x2 = (double)__fmul_rn(xv, xv);
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (q < nt && p < a.Ts[t0 + q]) acc[q] += x2;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            if (q >= nt) break;
            const double tot = wave_sum_f64(acc[q]);
            if (lane == 0) {
                float r = __fmul_rn((float)(tot / (double)a.Ts[t0 + q]), 252.0f);
                if (a.vol) r = sqrtf(r);
                a.out[row * a.nT + t0 + q] = r;
            }
        }
    }
}

hipError_t launch_realized_variance(const RvArgs& a, hipStream_t s) {
    const int64_t blocks = (a.n_rows + (PSH_RV_THREADS / 64) - 1) / (PSH_RV_THREADS / 64);
    hipLaunchKernelGGL(rv_kernel, dim3((unsigned)blocks), dim3(PSH_RV_THREADS), 0, s, a);
    return hipGetLastError();
}

}  // namespace psh
