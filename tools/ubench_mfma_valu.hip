// Do the matrix cores and the vector ALUs overlap in the batched scan's group loop?  8 MFMAs (32x32x16 f16, 4 tiles x 2) +
// the 34 v_min3 / compare epilogue per group, two waves per SIMD as scan_mq_kernel runs.  Variants:
//   0  the loop as the kernel has it (MFMAs, then the epilogue on their results)
//   1  software-pipelined inside the wave: the MFMAs of group i+1 (second accumulator set) issued BEFORE the epilogue of group i
//   2  variant 0 with the second wave of every SIMD started half a group late
//   3  MFMAs only        4  epilogue only (on constant accumulators)
// hipcc --offload-arch=gfx950 -O3 -o tools/bin/ubench_mfma_valu tools/ubench_mfma_valu.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ float min3f(float a, float b, float c) { return __builtin_fminf(__builtin_fminf(a, b), c); }
__device__ __forceinline__ float tile_min16(const f32x16& t) {
    const float m0 = min3f(t[0], t[1], t[2]), m1 = min3f(t[3], t[4], t[5]), m2 = min3f(t[6], t[7], t[8]);
    const float m3 = min3f(t[9], t[10], t[11]), m4 = min3f(t[12], t[13], t[14]);
    return min3f(min3f(m0, m1, m2), min3f(m3, m4, t[15]), __builtin_inff());
}
template <int V>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, int iters, const _Float16* frag) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __shared__ _Float16 fr[64 * 8 * 2 * 32];
    for (int i = threadIdx.x; i < 64 * 8 * 2 * 32; i += 512) fr[i] = frag[i];
    __syncthreads();
    f16x8 fy[4][2];
    f32x16 ny[4];
    for (int g = 0; g < 4; ++g) { for (int h = 0; h < 2; ++h) for (int i = 0; i < 8; ++i) fy[g][h][i] = (_Float16)(0.01f * (lane + g + h + i)); for (int i = 0; i < 16; ++i) ny[g][i] = 1.0f + i; }
    float thr = -1e30f, sink = 0.0f;
    int hits = 0;
    if (V == 2 && (wave >= 4)) __builtin_amdgcn_s_sleep(4);            // ~256 cycles
    const unsigned long long t0 = clock64();
    f32x16 accA[4], accB[4];
    if (V == 1) {
        const f16x8 b0 = *reinterpret_cast<const f16x8*>(fr + lane * 8), b1 = *reinterpret_cast<const f16x8*>(fr + 64 * 8 + lane * 8);
        for (int g = 0; g < 4; ++g) accA[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fy[g][0], b0, ny[g], 0, 0, 0);
        for (int g = 0; g < 4; ++g) accA[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fy[g][1], b1, accA[g], 0, 0, 0);
    }
#pragma unroll 1
    for (int it = 0; it < iters; it += 2) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const _Float16* fp = fr + ((it + half) & 31) * 2 * 64 * 8 + lane * 8;
            const f16x8 b0 = *reinterpret_cast<const f16x8*>(fp), b1 = *reinterpret_cast<const f16x8*>(fp + 64 * 8);
            f32x16* cur = half ? accB : accA;          // epilogue reads these
            f32x16* nxt = half ? accA : accB;          // V == 1: next group's MFMAs go here first
            if (V == 0 || V == 2 || V == 3) {
#pragma unroll
                for (int g = 0; g < 4; ++g) cur[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fy[g][0], b0, ny[g], 0, 0, 0);
#pragma unroll
                for (int g = 0; g < 4; ++g) cur[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fy[g][1], b1, cur[g], 0, 0, 0);
            }
            if (V == 1) {
#pragma unroll
                for (int g = 0; g < 4; ++g) nxt[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fy[g][0], b0, ny[g], 0, 0, 0);
#pragma unroll
                for (int g = 0; g < 4; ++g) nxt[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fy[g][1], b1, nxt[g], 0, 0, 0);
            }
            if (V == 4 && it == 0) { for (int g = 0; g < 4; ++g) cur[g] = ny[g]; }
            if (V != 3) {
                float mn[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) mn[g] = tile_min16(cur[g]);
                if (__any(!(min3f(min3f(mn[0], mn[1], mn[2]), mn[3], mn[3]) > thr))) { ++hits; sink += mn[0]; }
                if (V == 4) thr += mn[1] * 1e-30f;
            } else {
                sink += cur[0][0] + cur[1][1] + cur[2][2] + cur[3][3];
            }
        }
    }
    const unsigned long long t1 = clock64();
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
    if (sink == 123.456f || hits == -1) out[threadIdx.x] = sink + accA[0][0] + accB[0][0];
}
template <int V> static void run(const char* name, int iters) {
    float* out; unsigned long long* cyc; _Float16* frag;
    hipMalloc(&out, 4096); hipMalloc(&cyc, 256 * 8 * 8); hipMalloc(&frag, 64 * 8 * 2 * 32 * 2);
    hipMemset(frag, 0, 64 * 8 * 2 * 32 * 2);
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k<V>, dim3(256), dim3(512), 0, 0, out, cyc, iters, frag); hipDeviceSynchronize(); }
    std::vector<unsigned long long> h(256 * 8); hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    // clock64 = 100 MHz wall clock on this part: ticks -> ns; report ns per group and per pair of waves
    printf("%-44s median %.1f ns per group per wave (min %.1f max %.1f) -> per SIMD (2 waves) %.1f ns per group\n", name,
           10.0 * h[h.size() / 2] / iters, 10.0 * h[0] / iters, 10.0 * h.back() / iters, 10.0 * h[h.size() / 2] / iters / 2);
}
int main() {
    const int iters = 20000;
    run<3>("3 MFMAs only (8 per group)", iters);
    run<4>("4 epilogue only", iters);
    run<0>("0 MFMAs then epilogue (as in the kernel)", iters);
    run<2>("2 same, second wave of a SIMD started late", iters);
    run<1>("1 software-pipelined inside the wave", iters);
    return 0;
}
