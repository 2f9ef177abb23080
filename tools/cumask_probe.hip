// Which compute units does a CU-masked stream use on this part?  (tools; hipcc --offload-arch=gfx950 -o tools/bin/cumask_probe)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <set>
#include <vector>
__global__ void where(unsigned* out) {
    // HW_ID (hwreg 4): cu_id [11:8], sh_id [12], se_id [15:13];  XCC_ID (hwreg 20): [3:0]
    unsigned hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));
    unsigned xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (3 << 11));
    if (threadIdx.x == 0) out[blockIdx.x] = ((xcc & 15u) << 16) | (hw & 0xff00u);
    for (volatile int i = 0; i < 20000; ++i) {}
}
int main() {
    int ncu = 0; hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
    printf("CUs %d\n", ncu);
    unsigned* d; hipMalloc(&d, 4096 * 4);
    for (int clear : {0, 1, 8, 16, 32}) for (int from : {0, 1}) {
        const int words = (ncu + 31) / 32;
        std::vector<uint32_t> mask(words, 0xffffffffu);
        for (int b = 0; b < clear; ++b) { int bit = from ? (b * 32) % ncu + b / 8 : b; mask[bit / 32] &= ~(1u << (bit % 32)); }
        hipStream_t s; if (hipExtStreamCreateWithCUMask(&s, words, mask.data()) != hipSuccess) { printf("create failed\n"); continue; }
        hipLaunchKernelGGL(where, dim3(4096), dim3(1024), 65536, s, d);
        hipStreamSynchronize(s);
        std::vector<unsigned> h(4096); hipMemcpy(h.data(), d, 4096 * 4, hipMemcpyDeviceToHost);
        std::set<unsigned> cus; int perx[16] = {0}; std::set<unsigned> seen;
        for (unsigned v : h) if (seen.insert(v).second) perx[(v >> 16) & 15]++;
        printf("cleared %2d bits (%s): distinct CUs used %zu; per XCC:", clear, from ? "spread every 32nd" : "lowest", seen.size());
        for (int x = 0; x < 8; ++x) printf(" %d", perx[x]);
        printf("\n");
        hipStreamDestroy(s);
    }
    return 0;
}
