// How fast does the ensemble stream with W waves per CU, each holding ONE 4 KB unit in flight (the scan's access pattern:
// 5 non-temporal 16-byte loads per lane, a wave-private consumer), and does it survive a per-block pause (a block that
// spends `pause_us` not streaming before it starts, as a scan block does)?  LDS per block is padded so that exactly
// `blocks_per_cu` blocks of `waves` waves fit a CU.
// hipcc --offload-arch=gfx950 -O3 -o tools/bin/ubench_stream tools/ubench_stream.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void stream_k(const f32x4* __restrict__ src, size_t n4, float* out, int units_per_wave, int pause_ticks, int depth2) {
    extern __shared__ float pad[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const size_t gw = (size_t)blockIdx.x * nw + wave;
    if (pause_ticks > 0) { const long long t0 = wall_clock64(); while (wall_clock64() - t0 < pause_ticks) __builtin_amdgcn_s_sleep(8); }
    float acc = 0.0f;
    const size_t base = gw * (size_t)units_per_wave * 320;              // 320 float4 (5 KB incl. halo-ish) per unit
    f32x4 v[5], w[5];
    auto load = [&](f32x4 (&r)[5], int u) {
        const f32x4* p = src + (base + (size_t)u * 320) % (n4 - 320);
#pragma unroll
        for (int q = 0; q < 5; ++q) r[q] = __builtin_nontemporal_load(p + lane + 64 * q);
    };
    load(v, 0);
    if (depth2) load(w, 1);
    for (int u = 0; u < units_per_wave; ++u) {
        f32x4 c[5];
#pragma unroll
        for (int q = 0; q < 5; ++q) c[q] = (depth2 && (u & 1)) ? w[q] : v[q];
        if (depth2) { if (u + 2 < units_per_wave) { if (u & 1) load(w, u + 2); else load(v, u + 2); } }
        else if (u + 1 < units_per_wave) load(v, u + 1);
#pragma unroll
        for (int q = 0; q < 5; ++q) acc += c[q][0] * c[q][1] + c[q][2] * c[q][3];
        // ~the scan's per-unit work: a few hundred cycles of ALU
        for (int i = 0; i < 40; ++i) acc = acc * 1.0001f + 0.5f;
    }
    if (acc == 12345.678f) out[0] = acc + pad[lane];
}
int main() {
    const size_t bytes = (size_t)512 << 20;
    f32x4* src; float* out; hipMalloc(&src, bytes); hipMalloc(&out, 64); hipMemset(src, 0, bytes);
    const size_t n4 = bytes / 16;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    struct Cfg { int waves, bpc; } cfgs[] = {{16, 1}, {8, 2}, {20, 1}, {10, 2}, {24, 1}, {12, 2}, {32, 1}, {16, 2}};
    for (auto c : cfgs) for (int depth2 = 0; depth2 < 2; ++depth2) for (int pause_us : {0, 10}) {
        const int blocks = 256 * c.bpc, threads = 64 * c.waves;
        const size_t lds = (size_t)(160 * 1024 / c.bpc) - 1024;                      // exactly bpc blocks per CU
        const size_t units = n4 / 320;
        const int upw = (int)(units / ((size_t)blocks * c.waves));
        hipFuncSetAttribute((const void*)stream_k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(stream_k, dim3(blocks), dim3(threads), lds, 0, src, n4, out, upw, pause_us * 100, depth2);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (rep && ms < best) best = ms;
        }
        const double moved = (double)blocks * c.waves * upw * 320 * 16;
        printf("waves/CU %2d (%d x %2d)  units in flight per wave %d  pause %2d us: %7.1f us  %6.2f TB/s%s\n", c.waves * c.bpc, c.bpc, c.waves,
               depth2 + 1, pause_us, best * 1e3, moved / (best * 1e-3) / 1e12, hipGetLastError() == hipSuccess ? "" : "  (launch error)");
    }
    return 0;
}
