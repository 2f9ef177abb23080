// Does a ds_read_b128 at a 2-byte-aligned LDS address cost more than an aligned one on gfx950, for the access pattern the batched
// scan's B fragments would have if every query kept ONE zero-padded f16 copy (40 halves) instead of 8 shifted ones (256 halves)?
//   lane (n = lane & 31, hk = lane >> 5): query n >> 3 of the group, shift n & 7 -> 8 halves from half offset 7 - shift + 8 hk (+16)
// Variants: 0 the present layout (every lane its own aligned 16 bytes, 1 KB per fragment), 1 compact + unaligned reads,
//           2 compact layout, offsets forced to multiples of 8 halves (aligned reads of the same table: the cost of the pattern alone)
// hipcc --offload-arch=gfx950 -O3 -o tools/bin/ubench_lds_unaligned tools/ubench_lds_unaligned.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int V, int QSTRIDE>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, int iters) {
    __shared__ __attribute__((aligned(16))) _Float16 tab[32768];          // 64 KB
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 32768; i += 512) tab[i] = (_Float16)(0.001f * (i & 1023));
    __syncthreads();
    const int n = lane & 31, hk = lane >> 5, qsub = n >> 3, shift = n & 7;
    f16x8 acc0 = {0, 0, 0, 0, 0, 0, 0, 0}, acc1 = acc0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        const int G = it & 63;                                           // 64 groups of 4 queries
        const _Float16 *p0, *p1;
        if (V == 0) { p0 = tab + ((2 * G) * 64 + lane) * 8 % 32768; p1 = tab + ((2 * G + 1) * 64 + lane) * 8 % 32768; }
        else {
            int o = 7 - shift + 8 * hk;
            if (V == 2) o &= ~7;
            p0 = tab + (4 * G + qsub) * QSTRIDE + o;
            p1 = p0 + 16;
        }
        f16x8 b0, b1;
        __builtin_memcpy(&b0, p0, 16);
        __builtin_memcpy(&b1, p1, 16);
        acc0 += b0; acc1 += b1;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
    out[threadIdx.x] = (float)(acc0[0] + acc0[7] + acc1[3] + acc1[5] + acc0[1] + acc0[2] + acc1[6]);
}
template <int V, int QS> static void run(const char* name) {
    const int iters = 20000;
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 4096); hipMalloc(&cyc, 256 * 8 * 8);
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL((k<V, QS>), dim3(256), dim3(512), 0, 0, out, cyc, iters); hipDeviceSynchronize(); }
    std::vector<unsigned long long> h(256 * 8); hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    printf("%-64s %7.1f ticks per pair of fragment reads and wave (8 waves per CU)\n", name, (double)h[h.size() / 2] / iters);
}
int main() {
    run<0, 0>("0 present layout: aligned, 1 KB per fragment");
    run<1, 40>("1 compact (40 halves per query), unaligned ds_read_b128");
    run<1, 48>("1 compact (48 halves per query), unaligned");
    run<1, 64>("1 compact (64 halves per query), unaligned");
    run<2, 40>("2 compact (40), offsets rounded to 16 bytes (pattern only)");
    run<2, 64>("2 compact (64), offsets rounded to 16 bytes");
    return 0;
}
