// Does the memory-side cache (256 MB Infinity Cache) help a scan that streams the same ensemble again and again?  A cyclic pass
// over 512 MB thrashes any LRU-like cache of half that size; passes that ALTERNATE their direction start on what the pass before
// left behind.  16 waves per CU, one 5 KB unit in flight per wave (the scan's pattern), each wave a contiguous share.
//   size: 128 / 256 / 512 MB; order: every pass forward, or forward / backward alternating; loads: non-temporal (as the scan) or plain
// hipcc --offload-arch=gfx950 -O3 -o tools/bin/ubench_mall tools/ubench_mall.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <bool NT>
__global__ __launch_bounds__(1024) void stream_k(const f32x4* __restrict__ src, float* out, int units_per_wave, int reverse) {
    extern __shared__ float pad[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const size_t gw = (size_t)blockIdx.x * nw + wave;
    float acc = 0.0f;
    const size_t base = gw * (size_t)units_per_wave * 320;
    f32x4 v[5];
    auto load = [&](int u) {
        const int uu = reverse ? units_per_wave - 1 - u : u;
        const f32x4* p = src + base + (size_t)uu * 320;
#pragma unroll
        for (int q = 0; q < 5; ++q) v[q] = NT ? __builtin_nontemporal_load(p + lane + 64 * q) : p[lane + 64 * q];
    };
    load(0);
    for (int u = 0; u < units_per_wave; ++u) {
        f32x4 c[5];
#pragma unroll
        for (int q = 0; q < 5; ++q) c[q] = v[q];
        if (u + 1 < units_per_wave) load(u + 1);
#pragma unroll
        for (int q = 0; q < 5; ++q) acc += c[q][0] * c[q][1] + c[q][2] * c[q][3];
        for (int i = 0; i < 40; ++i) acc = acc * 1.0001f + 0.5f;
    }
    if (acc == 12345.678f) out[0] = acc + pad[lane];
}
int main() {
    const size_t maxb = (size_t)512 << 20;
    f32x4* src; float* out;
    if (hipMalloc(&src, maxb) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) return 1;
    (void)hipMemset(src, 0, maxb);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const size_t lds = 150 * 1024;
    (void)hipFuncSetAttribute((const void*)stream_k<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)stream_k<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int mb : {128, 256, 384, 512}) for (int nt = 1; nt >= 0; --nt) for (int alt = 0; alt < 2; ++alt) {
        const size_t n4 = ((size_t)mb << 20) / 16;
        const int upw = (int)(n4 / 320 / (256 * 16));
        const int passes = 40;
        for (int warm = 0; warm < 2; ++warm) {
            (void)hipEventRecord(e0, 0);
            for (int p = 0; p < passes; ++p) {
                if (nt) hipLaunchKernelGGL(stream_k<true>, dim3(256), dim3(1024), lds, 0, src, out, upw, alt ? (p & 1) : 0);
                else hipLaunchKernelGGL(stream_k<false>, dim3(256), dim3(1024), lds, 0, src, out, upw, alt ? (p & 1) : 0);
            }
            (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
        }
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        const double moved = (double)256 * 16 * upw * 320 * 16 * passes;
        printf("%3d MB, %-12s loads, passes %-22s: %7.1f us per pass  %6.2f TB/s\n", mb, nt ? "non-temporal" : "plain", alt ? "forward / backward" : "all forward",
               ms * 1e3 / passes, moved / (ms * 1e-3) / 1e12);
    }
    return 0;
}
