// The wavelet scan's product loop in isolation: 24 independent v_mfma_f32_16x16x32_f16 per step (2 M tiles x 12 kernel rows),
// their 12 + 2 fragments read from LDS with ds_read_b128 two groups ahead.  How many cycles does a step take per wave --
//   V = 0  MFMAs alone (fragments stay in registers)         V = 1  with the fragment reads, as the kernel issues them
//   V = 2  the reads alone                                    V = 3  MFMAs alone, accumulators updated in groups of 4 steps apart (24 -> 12 tiles)
// with one or two waves per SIMD (256 / 512 threads)?  Floor: 24 x 16 = 384 cycles per step and wave.
// hipcc --offload-arch=gfx950 -O3 -o tools/bin/ubench_emx_loop tools/ubench_emx_loop.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int V, int NT>
__global__ __launch_bounds__(NT) void k(float* out, unsigned long long* cyc, int iters) {
    extern __shared__ __attribute__((aligned(16))) _Float16 lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 32768; i += NT) lds[i] = (_Float16)(0.001f * (i & 255));
    __syncthreads();
    f32x4 C[2][12];
    for (int m = 0; m < 2; ++m) for (int i = 0; i < 12; ++i) C[m][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const _Float16* bp = lds + lane * 8;                    // conflict-free: 64 lanes x 16 bytes contiguous
    const _Float16* ap = lds + 16384 + wave * 1024 + lane * 8;
    auto ldB = [&](f16x8 (&f)[4], int g, int ks) {
#pragma unroll
        for (int r = 0; r < 4; ++r) f[r] = *reinterpret_cast<const f16x8*>(bp + (4 * g + r) * 1024 + 32 * (ks & 15));
    };
    auto mm = [&](const f16x8 (&f)[4], int g, const f16x8& a0, const f16x8& a1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            C[0][4 * g + r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, f[r], C[0][4 * g + r], 0, 0, 0);
            C[1][4 * g + r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, f[r], C[1][4 * g + r], 0, 0, 0);
        }
    };
    f16x8 F0[4], F1[4], F2[4];
    f16x8 a0 = *reinterpret_cast<const f16x8*>(ap), a1 = *reinterpret_cast<const f16x8*>(ap + 512);
    ldB(F0, 0, 0); ldB(F1, 1, 0); ldB(F2, 2, 0);
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int ks = 0; ks < ((V == 4 || V == 5) ? 0 : iters); ++ks) {
        if (V == 1 || V == 2) ldB(F2, 2, ks);
        __builtin_amdgcn_sched_barrier(0);
        if (V != 2) mm(F0, 0, a0, a1); else for (int r = 0; r < 4; ++r) asm volatile("" :: "v"(F0[r]));
        __builtin_amdgcn_sched_barrier(0);
        f16x8 n0 = a0, n1 = a1;
        if (V == 1 || V == 2) {
            n0 = *reinterpret_cast<const f16x8*>(ap + 32 * ((ks + 1) & 7));
            n1 = *reinterpret_cast<const f16x8*>(ap + 512 + 32 * ((ks + 1) & 7));
            ldB(F0, 0, ks + 1);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (V != 2) mm(F1, 1, a0, a1); else for (int r = 0; r < 4; ++r) asm volatile("" :: "v"(F1[r]));
        __builtin_amdgcn_sched_barrier(0);
        if (V == 1 || V == 2) ldB(F1, 1, ks + 1);
        __builtin_amdgcn_sched_barrier(0);
        if (V != 2) mm(F2, 2, a0, a1); else for (int r = 0; r < 4; ++r) asm volatile("" :: "v"(F2[r]));
        __builtin_amdgcn_sched_barrier(0);
        a0 = n0; a1 = n1;
    }
    if (V == 4 || V == 5) {
        // V = 4: a group's reads go into the buffer whose MFMAs were issued ONE GROUP EARLIER (not into the one just consumed)
        // V = 5: the same with every read issued between two MFMAs instead of in a block of 4 (+ 2)
#pragma unroll 1
        for (int ks = 0; ks < iters; ++ks) {
            f16x8 n0, n1;
            if (V == 4) {
                mm(F0, 0, a0, a1);
                __builtin_amdgcn_sched_barrier(0);
                ldB(F2, 2, ks);
                __builtin_amdgcn_sched_barrier(0);
                mm(F1, 1, a0, a1);
                __builtin_amdgcn_sched_barrier(0);
                n0 = *reinterpret_cast<const f16x8*>(ap + 32 * ((ks + 1) & 7));
                n1 = *reinterpret_cast<const f16x8*>(ap + 512 + 32 * ((ks + 1) & 7));
                ldB(F0, 0, ks + 1);
                __builtin_amdgcn_sched_barrier(0);
                mm(F2, 2, a0, a1);
                __builtin_amdgcn_sched_barrier(0);
                ldB(F1, 1, ks + 1);
                __builtin_amdgcn_sched_barrier(0);
            } else {
#define STEP(FM, gm, FL, gl, ksl) \
                _Pragma("unroll") for (int r = 0; r < 4; ++r) { \
                    C[0][4 * gm + r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, FM[r], C[0][4 * gm + r], 0, 0, 0); \
                    __builtin_amdgcn_sched_barrier(0); \
                    FL[r] = *reinterpret_cast<const f16x8*>(bp + (4 * gl + r) * 1024 + 32 * ((ksl) & 15)); \
                    __builtin_amdgcn_sched_barrier(0); \
                    C[1][4 * gm + r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, FM[r], C[1][4 * gm + r], 0, 0, 0); \
                    __builtin_amdgcn_sched_barrier(0); \
                }
                STEP(F0, 0, F2, 2, ks)
                n0 = *reinterpret_cast<const f16x8*>(ap + 32 * ((ks + 1) & 7));
                n1 = *reinterpret_cast<const f16x8*>(ap + 512 + 32 * ((ks + 1) & 7));
                __builtin_amdgcn_sched_barrier(0);
                STEP(F1, 1, F0, 0, ks + 1)
                STEP(F2, 2, F1, 1, ks + 1)
            }
            a0 = n0; a1 = n1;
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int m = 0; m < 2; ++m) for (int i = 0; i < 12; ++i) s += C[m][i][0] + C[m][i][1] + C[m][i][2] + C[m][i][3];
    out[blockIdx.x * NT + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * (NT / 64) + wave] = t1 - t0;
}
template <int V, int NT>
static void run(const char* name, float* out, unsigned long long* cyc, int iters) {
    hipFuncSetAttribute((const void*)k<V, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<V, NT>), dim3(256), dim3(NT), 65536, 0, out, cyc, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(256 * NT / 64);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    printf("%-44s %d waves/SIMD: %7.1f cycles per step and wave (median; min %.1f max %.1f)\n", name, NT / 256, (double)h[h.size() / 2] / iters,
           (double)h[0] / iters, (double)h.back() / iters);
}
int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
    const int iters = 2000;
    run<0, 256>("24 MFMAs", out, cyc, iters);          run<0, 512>("24 MFMAs", out, cyc, iters);
    run<1, 256>("24 MFMAs + 14 ds_read_b128", out, cyc, iters); run<1, 512>("24 MFMAs + 14 ds_read_b128", out, cyc, iters);
    run<4, 256>("... reads into the buffer of one group ago", out, cyc, iters); run<4, 512>("... reads into the buffer of one group ago", out, cyc, iters);
    run<5, 256>("... and one read between two MFMAs", out, cyc, iters); run<5, 512>("... and one read between two MFMAs", out, cyc, iters);
    run<2, 256>("14 ds_read_b128", out, cyc, iters);    run<2, 512>("14 ds_read_b128", out, cyc, iters);
    return 0;
}
