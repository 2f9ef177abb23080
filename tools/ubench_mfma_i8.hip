// ubench_mfma_i8.hip -- the operand layout of v_mfma_i32_32x32x32_i8 on gfx950, checked against a host product:
//   lane (m = lane & 31, hk = lane >> 5) supplies A[m][16 hk + i], i = 0..15 (byte i of its four operand registers),
//   lane (n = lane & 31, hk = lane >> 5) supplies B[16 hk + i][n];
//   D[row][col = lane & 31], row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)   (the f16 32x32x16 map: dtype-independent)
// (the batched scan's 8-bit rejection test builds its Toeplitz operands on exactly this assumption)
//   hipcc --offload-arch=gfx950 -O2 tools/ubench_mfma_i8.hip -o /tmp/ubench_mfma_i8 && /tmp/ubench_mfma_i8
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
__global__ void k(const int8_t* A, const int8_t* B, const int* C, int* D) {
    const int lane = threadIdx.x & 63, n = lane & 31, hk = lane >> 5;
    i32x4 a, b;
    for (int r = 0; r < 4; ++r) {
        unsigned wa = 0, wb = 0;
        for (int i = 0; i < 4; ++i) {
            wa |= (unsigned)(uint8_t)A[n * 32 + 16 * hk + 4 * r + i] << (8 * i);
            wb |= (unsigned)(uint8_t)B[(16 * hk + 4 * r + i) * 32 + n] << (8 * i);
        }
        a[r] = (int)wa; b[r] = (int)wb;
    }
    i32x16 c;
    for (int r = 0; r < 16; ++r) c[r] = C[((r & 3) + 8 * (r >> 2) + 4 * hk) * 32 + n];
    c = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * hk) * 32 + n] = c[r];
}
int main() {
    int8_t hA[1024], hB[1024]; int hC[1024], hD[1024];
    srand(7);
    for (int i = 0; i < 1024; ++i) { hA[i] = (int8_t)(rand() % 255 - 127); hB[i] = (int8_t)(rand() % 255 - 127); hC[i] = rand() % 2000001 - 1000000; }
    int8_t *dA, *dB; int *dC, *dD;
    (void)hipMalloc(&dA, 1024); (void)hipMalloc(&dB, 1024); (void)hipMalloc(&dC, 4096); (void)hipMalloc(&dD, 4096);
    (void)hipMemcpy(dA, hA, 1024, hipMemcpyHostToDevice); (void)hipMemcpy(dB, hB, 1024, hipMemcpyHostToDevice); (void)hipMemcpy(dC, hC, 4096, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
    (void)hipMemcpy(hD, dD, 4096, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int m = 0; m < 32; ++m) for (int n = 0; n < 32; ++n) {
        int s = hC[m * 32 + n];
        for (int kk = 0; kk < 32; ++kk) s += (int)hA[m * 32 + kk] * (int)hB[kk * 32 + n];
        bad += s != hD[m * 32 + n];
    }
    printf("v_mfma_i32_32x32x32_i8 layout check: %d of 1024 outputs differ from the host product\n", bad);
    return bad != 0;
}
