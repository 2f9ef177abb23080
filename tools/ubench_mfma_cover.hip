// Does vector-ALU work of ONE wave run under the matrix-core time of ANOTHER wave of the same SIMD (gfx950)?
// 8-wave blocks, one per CU (two waves per SIMD).  Roles by wave parity within a SIMD pair (waves w and w + 4 share SIMD w % 4):
//   mode 0: every wave: N x v_mfma_i32_32x32x32_i8 back to back (independent tiles)                     -> matrix pipe alone
//   mode 1: every wave: N x 34 v_min3_i32 on its own registers                                          -> vector pipe alone
//   mode 2: waves 0-3 MFMAs only, waves 4-7 v_min3 only (same counts as modes 0 / 1 per wave)          -> do they overlap ACROSS waves?
//   mode 3: every wave: per turn 4 MFMAs then 34 v_min3 on the tiles just produced (the kernel's shape) -> what scan_mq8_kernel's loop does
//   mode 4: as 3, but the 34 v_min3 read OTHER registers than the MFMAs write (no dependency)          -> is it the dependency?
// hipcc --offload-arch=gfx950 -O3 -o tools/bin/ubench_mfma_cover tools/ubench_mfma_cover.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ int imin3(int a, int b, int c) { const int m = a < b ? a : b; return m < c ? m : c; }
__device__ __forceinline__ int tmin(const i32x16& t) {
    const int m0 = imin3(t[0], t[1], t[2]), m1 = imin3(t[3], t[4], t[5]), m2 = imin3(t[6], t[7], t[8]);
    const int m3 = imin3(t[9], t[10], t[11]), m4 = imin3(t[12], t[13], t[14]);
    return imin3(imin3(m0, m1, m2), imin3(m3, m4, t[15]), 0x7fffffff);
}
template <int MODE>
__global__ __launch_bounds__(512) void k(int* out, unsigned long long* cyc, int iters) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    i32x4 a[4], b;
    i32x16 c[4], acc[4], other[4];
    for (int g = 0; g < 4; ++g) {
        for (int i = 0; i < 4; ++i) a[g][i] = lane * 7 + g + i;
        for (int i = 0; i < 16; ++i) { c[g][i] = i + g; other[g][i] = lane + i * g; acc[g][i] = i; }
    }
    for (int i = 0; i < 4; ++i) b[i] = lane + i;
    int sink = 0;
    const bool do_m = MODE == 0 || MODE >= 3 || (MODE == 2 && wave < 4);
    const bool do_v = MODE == 1 || MODE >= 3 || (MODE == 2 && wave >= 4);
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        if (do_m) {
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[g], b, c[g], 0, 0, 0);
        }
        if (do_v) {
            int mn[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) mn[g] = tmin(MODE == 3 ? acc[g] : other[g]);
            const int m = imin3(imin3(mn[0], mn[1], mn[2]), mn[3], mn[3]);
            if (__builtin_amdgcn_ballot_w64(m < -1000000)) sink += m;
#pragma unroll
            for (int g = 0; g < 4; ++g) other[g][it & 15] += 1;      // keep the reads live and loop-variant
        }
        b[0] += 1;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    int s = sink;
    for (int g = 0; g < 4; ++g) s += acc[g][lane & 15] + other[g][3];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}
template <int MODE>
void run(int* out, unsigned long long* cyc, int iters, int grid) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(512), 0, 0, out, cyc, 100);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(512), 0, 0, out, cyc, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[8]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    printf("grid %3d mode %d: %.3f ms for %d turns -> %.1f ns per turn and block; s_memtime ticks per turn: wave0 %.1f wave4 %.1f\n", grid, MODE, ms, iters, 1e6 * ms / iters,
           (double)h[0] / iters, (double)h[4] / iters);
}
int main() {
    int* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
    const int iters = 200000;
    for (int grid : {1, 256}) { run<0>(out, cyc, iters, grid); run<1>(out, cyc, iters, grid); run<2>(out, cyc, iters, grid); run<3>(out, cyc, iters, grid); run<4>(out, cyc, iters, grid); }
    return 0;
}
