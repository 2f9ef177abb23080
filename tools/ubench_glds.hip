// The scan's streaming pattern (a wave owns a contiguous share, 5 KB units) with the loads going global -> VGPR (as the scan has
// it) or global -> LDS directly (global_load_lds_dwordx4: no registers, no ds_write pass), 1 - 3 units in flight per wave,
// 16 or 8 waves per CU.  Is the ceiling of ~6.7 TB/s a property of the register path?
// hipcc --offload-arch=gfx950 -O3 -o tools/bin/ubench_glds tools/ubench_glds.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int DEPTH>
__global__ __launch_bounds__(1024) void k_glds(const f32x4* __restrict__ src, float* out, int units_per_wave) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const size_t gw = (size_t)blockIdx.x * nw + wave;
    float* mine = lds + (size_t)wave * DEPTH * 1280;                    // DEPTH buffers of 5 KB
    const size_t base = gw * (size_t)units_per_wave * 320;
    auto issue = [&](int u) {
        const f32x4* p = src + base + (size_t)u * 320 + lane;
        float* dst = mine + (u % DEPTH) * 1280;
#pragma unroll
        for (int q = 0; q < 5; ++q)
            __builtin_amdgcn_global_load_lds(p + 64 * q, (__attribute__((address_space(3))) void*)(dst + 256 * q), 16, 0, 0);
    };
    float acc = 0.0f;
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) if (d < units_per_wave) issue(d);
    for (int u = 0; u < units_per_wave; ++u) {
        // wait for unit u: the DEPTH - 1 younger units may stay in flight
        if (DEPTH == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (DEPTH == 2) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        const float* b = mine + (u % DEPTH) * 1280;
        f32x4 c[5];
#pragma unroll
        for (int q = 0; q < 5; ++q) c[q] = *reinterpret_cast<const f32x4*>(b + 4 * lane + 256 * q);
#pragma unroll
        for (int q = 0; q < 5; ++q) acc += c[q][0] * c[q][1] + c[q][2] * c[q][3];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (u + DEPTH < units_per_wave) issue(u + DEPTH);
        else {                                                         // keep the counter arithmetic uniform at the tail
#pragma unroll
            for (int q = 0; q < 5; ++q) __builtin_amdgcn_global_load_lds(src + base + lane + 64 * q, (__attribute__((address_space(3))) void*)(mine + (u % DEPTH) * 1280 + 256 * q), 16, 0, 0);
        }
        for (int i = 0; i < 40; ++i) acc = acc * 1.0001f + 0.5f;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (acc == 12345.678f) out[0] = acc;
}
template <int DEPTH>
__global__ __launch_bounds__(1024) void k_reg(const f32x4* __restrict__ src, float* out, int units_per_wave) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const size_t gw = (size_t)blockIdx.x * nw + wave;
    const size_t base = gw * (size_t)units_per_wave * 320;
    f32x4 v[DEPTH][5];
    float acc = 0.0f;
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
#pragma unroll
        for (int q = 0; q < 5; ++q) v[d][q] = __builtin_nontemporal_load(src + base + (size_t)d * 320 + lane + 64 * q);
    for (int u = 0; u < units_per_wave; u += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            f32x4 c[5];
#pragma unroll
            for (int q = 0; q < 5; ++q) c[q] = v[d][q];
            const int un = u + d + DEPTH < units_per_wave ? u + d + DEPTH : 0;
#pragma unroll
            for (int q = 0; q < 5; ++q) v[d][q] = __builtin_nontemporal_load(src + base + (size_t)un * 320 + lane + 64 * q);
#pragma unroll
            for (int q = 0; q < 5; ++q) acc += c[q][0] * c[q][1] + c[q][2] * c[q][3];
            for (int i = 0; i < 40; ++i) acc = acc * 1.0001f + 0.5f;
        }
    }
    if (acc == 12345.678f) out[0] = acc + lds[lane];
}
template <typename K> static void run(const char* name, K kern, int waves, int depth, const f32x4* src, float* out, size_t lds) {
    const size_t n4 = ((size_t)512 << 20) / 16;
    const int upw = (int)(n4 / 320 / (256 * waves)) / 6 * 6;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    float best = 1e9f;
    for (int rep = 0; rep < 6; ++rep) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(kern, dim3(256), dim3(64 * waves), lds, 0, src, out, upw);
        (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (rep && ms < best) best = ms;
    }
    const double moved = (double)256 * waves * upw * 320 * 16;
    printf("%-22s %2d waves/CU, %d unit(s) in flight per wave: %7.1f us  %6.2f TB/s %s\n", name, waves, depth, best * 1e3, moved / (best * 1e-3) / 1e12,
           hipGetLastError() == hipSuccess ? "" : "(launch error)");
}
int main() {
    f32x4* src; float* out;
    if (hipMalloc(&src, (size_t)512 << 20) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) return 1;
    (void)hipMemset(src, 0, (size_t)512 << 20);
    const size_t big = 150 * 1024;
    run("global -> VGPR", k_reg<1>, 16, 1, src, out, big); run("global -> VGPR", k_reg<2>, 16, 2, src, out, big);
    run("global -> VGPR", k_reg<2>, 8, 2, src, out, big);  run("global -> VGPR", k_reg<3>, 8, 3, src, out, big);
    run("global -> LDS (DMA)", k_glds<1>, 16, 1, src, out, big); run("global -> LDS (DMA)", k_glds<2>, 8, 2, src, out, big);
    run("global -> LDS (DMA)", k_glds<3>, 8, 3, src, out, big);  run("global -> LDS (DMA)", k_glds<1>, 8, 1, src, out, big);
    run("global -> LDS (DMA)", k_glds<2>, 12, 2, src, out, big);
    return 0;
}
