// Which matrix-core instructions let another wave's vector-ALU work run under them (gfx950, two waves per SIMD)?
// Per turn and wave: 64 accumulator registers' worth of MFMAs (KIND 0: 4 x i32_32x32x32_i8, 1: 4 x f32_32x32x16_f16 x 2 K-steps,
// 2: 16 x i32_16x16x64_i8, 3: 16 x f32_16x16x32_f16) and / or 34 v_min3 + a test.  mode 0 MFMA only, 1 VALU only, 2 waves 0-3 MFMA +
// waves 4-7 VALU, 3 both in every wave (MFMAs, then the minima of OTHER registers).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ int imin3(int a, int b, int c) { const int m = a < b ? a : b; return m < c ? m : c; }
template <int KIND, int MODE>
__global__ __launch_bounds__(512) void k(int* out, int iters) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    i32x4 ai[4], bi; f16x8 ah[4], bh;
    i32x16 ci[4], di[4]; f32x16 ch[4], dh[4];
    i32x4 c4[16], d4[16]; f32x4 e4[16], g4[16];
    int other[64];
    for (int g = 0; g < 4; ++g) { for (int i = 0; i < 4; ++i) ai[g][i] = lane * 7 + g + i; for (int i = 0; i < 8; ++i) ah[g][i] = (_Float16)(0.01f * (lane + g + i));
        for (int i = 0; i < 16; ++i) { ci[g][i] = i + g; ch[g][i] = i + g; di[g][i] = 0; dh[g][i] = 0; } }
    for (int g = 0; g < 16; ++g) for (int i = 0; i < 4; ++i) { c4[g][i] = g + i; d4[g][i] = 0; e4[g][i] = g + i; g4[g][i] = 0; }
    for (int i = 0; i < 4; ++i) bi[i] = lane + i;
    for (int i = 0; i < 8; ++i) bh[i] = (_Float16)(0.02f * (lane + i));
    for (int i = 0; i < 64; ++i) other[i] = lane * i + 3;
    int sink = 0;
    const bool do_m = MODE == 0 || MODE == 3 || (MODE == 2 && wave < 4);
    const bool do_v = MODE == 1 || MODE == 3 || (MODE == 2 && wave >= 4);
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        if (do_m) {
            if (KIND == 0) {
#pragma unroll
                for (int g = 0; g < 4; ++g) di[g] = __builtin_amdgcn_mfma_i32_32x32x32_i8(ai[g], bi, ci[g], 0, 0, 0);
            } else if (KIND == 1) {
#pragma unroll
                for (int g = 0; g < 4; ++g) dh[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[g], bh, ch[g], 0, 0, 0);
#pragma unroll
                for (int g = 0; g < 4; ++g) dh[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[g], bh, dh[g], 0, 0, 0);
            } else if (KIND == 2) {
#pragma unroll
                for (int g = 0; g < 16; ++g) d4[g] = __builtin_amdgcn_mfma_i32_16x16x64_i8(ai[g & 3], bi, c4[g], 0, 0, 0);
            } else {
#pragma unroll
                for (int g = 0; g < 16; ++g) g4[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[g & 3], bh, e4[g], 0, 0, 0);
            }
        }
        if (do_v) {
            int m = 0x7fffffff;
#pragma unroll
            for (int i = 0; i < 64; i += 2) m = imin3(m, other[i], other[i + 1]);
            if (__builtin_amdgcn_ballot_w64(m < -1000000)) sink += m;
            other[0] += 1; other[17] += it;
        }
        bi[0] += 1; bh[0] += (_Float16)0.001f;
    }
    int s = sink + other[5];
    for (int g = 0; g < 4; ++g) s += di[g][lane & 15] + (int)dh[g][lane & 15];
    for (int g = 0; g < 16; ++g) s += d4[g][lane & 3] + (int)g4[g][lane & 3];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
template <int KIND, int MODE>
float run(int* out, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<KIND, MODE>), dim3(256), dim3(512), 0, 0, out, 100);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<KIND, MODE>), dim3(256), dim3(512), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return 1e6f * ms / iters;
}
template <int KIND>
void kind(int* out, const char* name) {
    const int iters = 100000;
    const float m = run<KIND, 0>(out, iters), v = run<KIND, 1>(out, iters), x = run<KIND, 2>(out, iters), b = run<KIND, 3>(out, iters);
    printf("%-34s ns per turn: MFMA only %.1f, VALU only %.1f, split over the waves %.1f (overlapped: %.1f, added: %.1f), both in every wave %.1f (added: %.1f)\n",
           name, m, v, x, (m > v ? m : v) / 2, (m + v) / 2, b, m + v);
}
int main() {
    int* out; hipMalloc(&out, 256 * 512 * 4);
    kind<0>(out, "4 x v_mfma_i32_32x32x32_i8");
    kind<1>(out, "8 x v_mfma_f32_32x32x16_f16");
    kind<2>(out, "16 x v_mfma_i32_16x16x64_i8");
    kind<3>(out, "16 x v_mfma_f32_16x16x32_f16");
    return 0;
}
