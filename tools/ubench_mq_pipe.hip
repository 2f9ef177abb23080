// The batched scan's query-group loop (scan_mq_kernel): 8 x v_mfma_f32_32x32x16_f16 (4 tiles x 2 K halves, C = the window energies)
// and the 34 v_min3 + test of the tiles.  Does the epilogue of group G hide in the gaps between the MFMAs of group G + 1 when it
// is PLACED there (5 - 6 instructions per gap, __builtin_amdgcn_sched_barrier between the pieces), and what does that need --
// a second accumulator set (128 + 64 registers of tiles), so one wave per SIMD (256 threads, 512 registers) or two (512 threads)?
//   V = 0  as the kernel has it: the group's 8 MFMAs, then its epilogue             V = 1  MFMAs alone
//   V = 2  pipelined, placed: MFMA, one b-fragment / threshold read, 5 - 6 min3 of the other set, MFMA, ...
// hipcc --offload-arch=gfx950 -O3 -o tools/bin/ubench_mq_pipe tools/ubench_mq_pipe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ float min3f(float a, float b, float c) { return __builtin_fminf(__builtin_fminf(a, b), c); }
#define SB() __builtin_amdgcn_sched_barrier(0)
template <int V, int NT>
__global__ __launch_bounds__(NT) void k(float* out, unsigned long long* cyc, int iters, const _Float16* frag) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __shared__ __attribute__((aligned(16))) _Float16 fr[64 * 8 * 2 * 32];
    __shared__ float thrL[128];
    for (int i = threadIdx.x; i < 64 * 8 * 2 * 32; i += NT) fr[i] = frag[i];
    if (threadIdx.x < 128) thrL[threadIdx.x] = -1e30f;
    __syncthreads();
    f16x8 fy[4][2];
    f32x16 ny[4];
    for (int g = 0; g < 4; ++g) { for (int h = 0; h < 2; ++h) for (int i = 0; i < 8; ++i) fy[g][h][i] = (_Float16)(0.01f * (lane + g + h + i)); for (int i = 0; i < 16; ++i) ny[g][i] = 1.0f + i; }
    float sink = 0.0f;
    int hits = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (V == 0 || V == 1) {
#pragma unroll 1
        for (int it = 0; it < iters; ++it) {
            const _Float16* fp = fr + (it & 31) * 2 * 64 * 8 + lane * 8;
            const f16x8 b0 = *reinterpret_cast<const f16x8*>(fp), b1 = *reinterpret_cast<const f16x8*>(fp + 64 * 8);
            const float thr = thrL[(it & 31) * 4 + (lane >> 3 & 3)];
            f32x16 acc[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fy[g][0], b0, ny[g], 0, 0, 0);
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fy[g][1], b1, acc[g], 0, 0, 0);
            if (V == 0) {
                float mn[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x16& t = acc[g];
                    const float m0 = min3f(t[0], t[1], t[2]), m1 = min3f(t[3], t[4], t[5]), m2 = min3f(t[6], t[7], t[8]);
                    const float m3 = min3f(t[9], t[10], t[11]), m4 = min3f(t[12], t[13], t[14]);
                    mn[g] = min3f(min3f(m0, m1, m2), min3f(m3, m4, t[15]), __builtin_inff());
                }
                if (__any(!(min3f(min3f(mn[0], mn[1], mn[2]), mn[3], mn[3]) > thr))) { ++hits; sink += mn[0]; }
            } else {
                sink += acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
            }
        }
    } else {
        // two accumulator sets; a turn = group G's MFMAs into X with group G - 1's epilogue (set Y) in their gaps, then the roles swapped
        f32x16 X[4], Y[4];
        f16x8 bc0, bc1, bn0, bn1;
        float thr_y, thr_x;
        {
            const _Float16* fp = fr + lane * 8;
            bc0 = *reinterpret_cast<const f16x8*>(fp); bc1 = *reinterpret_cast<const f16x8*>(fp + 64 * 8);
            thr_y = thrL[lane >> 3 & 3];
#pragma unroll
            for (int g = 0; g < 4; ++g) Y[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fy[g][0], bc0, ny[g], 0, 0, 0);
#pragma unroll
            for (int g = 0; g < 4; ++g) Y[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fy[g][1], bc1, Y[g], 0, 0, 0);
            bc0 = bc1; bc1 = bc0;
        }
#define HALF(ACC, OLD, THR_NEW, THR_OLD, ITV)                                                                              \
        {                                                                                                                  \
            const _Float16* fp = fr + ((ITV) & 31) * 2 * 64 * 8 + lane * 8;                                                 \
            float m0, m1, m2, m3, m4, mn0, mn1, mn2, mn3;                                                                  \
            ACC[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fy[0][0], bc0, ny[0], 0, 0, 0); SB();                          \
            bn0 = *reinterpret_cast<const f16x8*>(fp);                                                                     \
            m0 = min3f(OLD[0][0], OLD[0][1], OLD[0][2]); m1 = min3f(OLD[0][3], OLD[0][4], OLD[0][5]); m2 = min3f(OLD[0][6], OLD[0][7], OLD[0][8]); \
            m3 = min3f(OLD[0][9], OLD[0][10], OLD[0][11]); m4 = min3f(OLD[0][12], OLD[0][13], OLD[0][14]); SB();           \
            ACC[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fy[1][0], bc0, ny[1], 0, 0, 0); SB();                          \
            bn1 = *reinterpret_cast<const f16x8*>(fp + 64 * 8);                                                            \
            mn0 = min3f(min3f(m0, m1, m2), min3f(m3, m4, OLD[0][15]), __builtin_inff());                                   \
            m0 = min3f(OLD[1][0], OLD[1][1], OLD[1][2]); m1 = min3f(OLD[1][3], OLD[1][4], OLD[1][5]); SB();                \
            ACC[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fy[2][0], bc0, ny[2], 0, 0, 0); SB();                          \
            THR_NEW = thrL[((ITV) & 31) * 4 + (lane >> 3 & 3)];                                                            \
            m2 = min3f(OLD[1][6], OLD[1][7], OLD[1][8]); m3 = min3f(OLD[1][9], OLD[1][10], OLD[1][11]); m4 = min3f(OLD[1][12], OLD[1][13], OLD[1][14]); \
            mn1 = min3f(min3f(m0, m1, m2), min3f(m3, m4, OLD[1][15]), __builtin_inff()); SB();                             \
            ACC[3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fy[3][0], bc0, ny[3], 0, 0, 0); SB();                          \
            m0 = min3f(OLD[2][0], OLD[2][1], OLD[2][2]); m1 = min3f(OLD[2][3], OLD[2][4], OLD[2][5]); m2 = min3f(OLD[2][6], OLD[2][7], OLD[2][8]); \
            m3 = min3f(OLD[2][9], OLD[2][10], OLD[2][11]); m4 = min3f(OLD[2][12], OLD[2][13], OLD[2][14]); SB();           \
            ACC[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fy[0][1], bc1, ACC[0], 0, 0, 0); SB();                         \
            mn2 = min3f(min3f(m0, m1, m2), min3f(m3, m4, OLD[2][15]), __builtin_inff());                                   \
            m0 = min3f(OLD[3][0], OLD[3][1], OLD[3][2]); m1 = min3f(OLD[3][3], OLD[3][4], OLD[3][5]); SB();                \
            ACC[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fy[1][1], bc1, ACC[1], 0, 0, 0); SB();                         \
            m2 = min3f(OLD[3][6], OLD[3][7], OLD[3][8]); m3 = min3f(OLD[3][9], OLD[3][10], OLD[3][11]); m4 = min3f(OLD[3][12], OLD[3][13], OLD[3][14]); \
            mn3 = min3f(min3f(m0, m1, m2), min3f(m3, m4, OLD[3][15]), __builtin_inff()); SB();                             \
            ACC[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fy[2][1], bc1, ACC[2], 0, 0, 0); SB();                         \
            const float mall = min3f(min3f(mn0, mn1, mn2), mn3, mn3);                                                      \
            const bool anyhit = __any(!(mall > THR_OLD)); SB();                                                            \
            ACC[3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fy[3][1], bc1, ACC[3], 0, 0, 0); SB();                         \
            bc0 = bn0; bc1 = bn1;                                                                                          \
            if (anyhit) { ++hits; sink += mn0; }                                                                           \
        }
#pragma unroll 1
        for (int it = 1; it < iters; it += 2) {
            HALF(X, Y, thr_x, thr_y, it + 1)
            HALF(Y, X, thr_y, thr_x, it + 2)
        }
        sink += X[0][0] + Y[1][1];
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) cyc[blockIdx.x * (NT / 64) + wave] = t1 - t0;
    if (sink == 123.456f || hits == -1) out[threadIdx.x] = sink;
}
template <int V, int NT> static void run(const char* name, int iters) {
    float* out; unsigned long long* cyc; _Float16* frag;
    hipMalloc(&out, 4096); hipMalloc(&cyc, 256 * 8 * 8); hipMalloc(&frag, 64 * 8 * 2 * 32 * 2);
    hipMemset(frag, 0, 64 * 8 * 2 * 32 * 2);
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL((k<V, NT>), dim3(256), dim3(NT), 0, 0, out, cyc, iters, frag); hipDeviceSynchronize(); }
    std::vector<unsigned long long> h(256 * NT / 64); hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double per_wave = (double)h[h.size() / 2] / iters;
    printf("%-52s %d waves/SIMD: %6.1f cycles per group and wave -> %6.1f per group and SIMD (floor 256)\n", name, NT / 256, per_wave, per_wave / (NT / 256));
}
int main() {
    const int iters = 20001;
    run<1, 256>("MFMAs alone", iters);                                 run<1, 512>("MFMAs alone", iters);
    run<0, 256>("8 MFMAs, then the epilogue (the kernel)", iters);     run<0, 512>("8 MFMAs, then the epilogue (the kernel)", iters);
    run<2, 256>("epilogue of G - 1 placed between the MFMAs of G", iters); run<2, 512>("epilogue of G - 1 placed between the MFMAs of G", iters);
    return 0;
}
