// VALU issue-rate microbenchmark: is v_pk_{add,fma}_f32 faster than 2x scalar on gfx950?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));
#define ITERS 4096
// 16 independent chains, each step: D = x - y; acc = fma(D, D, acc)   (scalar form)
__global__ __launch_bounds__(256) void k_scalar(float* out, float x0) {
    float acc[16], y[16];
    for (int i = 0; i < 16; ++i) { acc[i] = 0.f; y[i] = threadIdx.x * 1e-3f + i; }
    float x = x0;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            float D;
            asm volatile("v_sub_f32 %0, %1, %2" : "=v"(D) : "s"(x), "v"(y[i]));
            asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(acc[i]) : "v"(D));
        }
    }
    float s = 0; for (int i = 0; i < 16; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_packed(float* out, float x0) {
    f2 acc[8], y[8];
    for (int i = 0; i < 8; ++i) { acc[i] = f2{0.f, 0.f}; y[i] = f2{threadIdx.x * 1e-3f + i, threadIdx.x * 2e-3f + i}; }
    f2 x = f2{x0, x0};
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            f2 D;
            asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(D) : "v"(x), "v"(y[i]));
            asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(acc[i]) : "v"(D));
        }
    }
    float s = 0; for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    float* out; hipMalloc(&out, 256 * 8192 * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int waves_per_simd = 1; waves_per_simd <= 8; waves_per_simd *= 2) {
        int blocks = 256 * waves_per_simd;   // 256 CUs x (4 waves/block = 1 wave per SIMD)
        for (int variant = 0; variant < 2; ++variant) {
            float ms = 0;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(a);
                if (variant == 0) hipLaunchKernelGGL(k_scalar, dim3(blocks), dim3(256), 0, 0, out, 1.5f);
                else hipLaunchKernelGGL(k_packed, dim3(blocks), dim3(256), 0, 0, out, 1.5f);
                hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
            }
            double laneops = (double)blocks * 256 * ITERS * 32.0;  // 16 sub + 16 fma per iter per lane (both variants)
            printf("%s waves/SIMD=%d: %.3f ms  %.2f T lane-ops/s (sub+fma counted as 2)\n",
                   variant ? "packed" : "scalar", waves_per_simd, ms, laneops / ms / 1e9);
        }
    }
    return 0;
}
