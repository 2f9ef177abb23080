// Isolated timing of the scan's hot blocks (no HBM traffic): how many SIMD cycles does one
// 1024-window segment cost in accumulate16<20> (exact) and approx16<20> (cheap test)?
#include "../shadowing_amd/csrc/psh_device.h"
#include <stdio.h>
using namespace psh;
template <int VARIANT>
__global__ __launch_bounds__(1024) void bench(const float* q, float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tile_floats = 1160;
    float* tile = smem + wave * tile_floats;
    for (int i = lane; i < tile_floats; i += 64) tile[i] = 0.01f * ((i * 7919 + wave) % 97) - 0.4f;
    __syncthreads();
    const const_f32p x = (const_f32p)q;
    float sum = 0.f;
    for (int it = 0; it < iters; ++it) {
        float acc[PSH_L];
        if (VARIANT == 0) { accumulate16<20>(tile, lane, x, 20, acc); }
        else { float NY; float xv[20]; for (int j = 0; j < 20; ++j) { xv[j] = x[j]; asm volatile("" : "+v"(xv[j])); } approx16<20>(tile, lane, xv, acc, NY); sum += NY; }
        float m = min16(acc);
        if (__any(m < -1e30f)) sum += m;    // keep the result live, never taken
        wave_lds_fence();
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
}
int main() {
    float hq[32]; for (int i = 0; i < 32; ++i) hq[i] = 0.01f * i - 0.1f;
    float *q, *out; hipMalloc(&q, 128); hipMemcpy(q, hq, 128, hipMemcpyHostToDevice); hipMalloc(&out, 4 * 1024 * 512);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 512; const size_t shmem = 16 * 1160 * 4;
    hipFuncSetAttribute((const void*)bench<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    hipFuncSetAttribute((const void*)bench<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    for (int v = 0; v < 2; ++v) for (int rep = 0; rep < 3; ++rep) {
        float ms;
        hipEventRecord(a);
        if (v == 0) hipLaunchKernelGGL(bench<0>, dim3(256), dim3(1024), shmem, 0, q, out, iters);
        else hipLaunchKernelGGL(bench<1>, dim3(256), dim3(1024), shmem, 0, q, out, iters);
        hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
        // 256 CUs x 16 waves x iters segments; per SIMD: 4 waves x iters
        double ns_per_seg_per_simd = ms * 1e6 / (4.0 * iters);
        printf("%s: %.3f ms  -> %.1f ns of SIMD time per segment (full scan of 131072 segments = %.1f us)\n",
               v ? "approx16<20>" : "accumulate16<20>", ms, ns_per_seg_per_simd, ns_per_seg_per_simd * 131072 / 1024 / 1e3);
    }
    return 0;
}
