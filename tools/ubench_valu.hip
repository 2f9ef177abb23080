// VALU issue-rate microbenchmark for gfx950: which instruction mix reaches 1 wave64 VALU / 2 cycles / SIMD?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));
#define ITERS 2048
#define INIT float a0=0,a1=0,a2=0,a3=0,a4=0,a5=0,a6=0,a7=0; \
  float w0=threadIdx.x*1e-3f,w1=w0+1,w2=w0+2,w3=w0+3,w4=w0+4,w5=w0+5,w6=w0+6,w7=w0+7; float t0,t1,t2,t3,t4,t5,t6,t7;
#define FIN out[blockIdx.x*blockDim.x+threadIdx.x]=a0+a1+a2+a3+a4+a5+a6+a7;
// V0: 8 sub(s,v) then 8 fmac  (the scan kernel's block)
__global__ __launch_bounds__(256) void v0(float* out, float x) { INIT
  for (int it=0; it<ITERS; ++it) asm volatile(
   "v_sub_f32 %8,%16,%17\n v_sub_f32 %9,%16,%18\n v_sub_f32 %10,%16,%19\n v_sub_f32 %11,%16,%20\n v_sub_f32 %12,%16,%21\n v_sub_f32 %13,%16,%22\n v_sub_f32 %14,%16,%23\n v_sub_f32 %15,%16,%24\n"
   "v_fmac_f32 %0,%8,%8\n v_fmac_f32 %1,%9,%9\n v_fmac_f32 %2,%10,%10\n v_fmac_f32 %3,%11,%11\n v_fmac_f32 %4,%12,%12\n v_fmac_f32 %5,%13,%13\n v_fmac_f32 %6,%14,%14\n v_fmac_f32 %7,%15,%15"
   : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7),"=&v"(t0),"=&v"(t1),"=&v"(t2),"=&v"(t3),"=&v"(t4),"=&v"(t5),"=&v"(t6),"=&v"(t7)
   : "s"(x),"v"(w0),"v"(w1),"v"(w2),"v"(w3),"v"(w4),"v"(w5),"v"(w6),"v"(w7)); FIN }
// V1: same with x in a VGPR
__global__ __launch_bounds__(256) void v1(float* out, float x) { INIT float xv = x + threadIdx.x*0.f;
  for (int it=0; it<ITERS; ++it) asm volatile(
   "v_sub_f32 %8,%16,%17\n v_sub_f32 %9,%16,%18\n v_sub_f32 %10,%16,%19\n v_sub_f32 %11,%16,%20\n v_sub_f32 %12,%16,%21\n v_sub_f32 %13,%16,%22\n v_sub_f32 %14,%16,%23\n v_sub_f32 %15,%16,%24\n"
   "v_fmac_f32 %0,%8,%8\n v_fmac_f32 %1,%9,%9\n v_fmac_f32 %2,%10,%10\n v_fmac_f32 %3,%11,%11\n v_fmac_f32 %4,%12,%12\n v_fmac_f32 %5,%13,%13\n v_fmac_f32 %6,%14,%14\n v_fmac_f32 %7,%15,%15"
   : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7),"=&v"(t0),"=&v"(t1),"=&v"(t2),"=&v"(t3),"=&v"(t4),"=&v"(t5),"=&v"(t6),"=&v"(t7)
   : "v"(xv),"v"(w0),"v"(w1),"v"(w2),"v"(w3),"v"(w4),"v"(w5),"v"(w6),"v"(w7)); FIN }
// V2: 16 independent fmac with distinct sources (pure FMA rate)
__global__ __launch_bounds__(256) void v2(float* out, float x) { INIT
  for (int it=0; it<ITERS; ++it) asm volatile(
   "v_fmac_f32 %0,%8,%9\n v_fmac_f32 %1,%9,%10\n v_fmac_f32 %2,%10,%11\n v_fmac_f32 %3,%11,%12\n v_fmac_f32 %4,%12,%13\n v_fmac_f32 %5,%13,%14\n v_fmac_f32 %6,%14,%15\n v_fmac_f32 %7,%15,%8\n"
   "v_fmac_f32 %0,%8,%10\n v_fmac_f32 %1,%9,%11\n v_fmac_f32 %2,%10,%12\n v_fmac_f32 %3,%11,%13\n v_fmac_f32 %4,%12,%14\n v_fmac_f32 %5,%13,%15\n v_fmac_f32 %6,%14,%8\n v_fmac_f32 %7,%15,%9"
   : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7)
   : "v"(w0),"v"(w1),"v"(w2),"v"(w3),"v"(w4),"v"(w5),"v"(w6),"v"(w7)); FIN }
// V3: 16 v_sub only (VOP2, s + v)
__global__ __launch_bounds__(256) void v3(float* out, float x) { INIT
  for (int it=0; it<ITERS; ++it) asm volatile(
   "v_sub_f32 %0,%8,%0\n v_sub_f32 %1,%8,%1\n v_sub_f32 %2,%8,%2\n v_sub_f32 %3,%8,%3\n v_sub_f32 %4,%8,%4\n v_sub_f32 %5,%8,%5\n v_sub_f32 %6,%8,%6\n v_sub_f32 %7,%8,%7\n"
   "v_sub_f32 %0,%8,%0\n v_sub_f32 %1,%8,%1\n v_sub_f32 %2,%8,%2\n v_sub_f32 %3,%8,%3\n v_sub_f32 %4,%8,%4\n v_sub_f32 %5,%8,%5\n v_sub_f32 %6,%8,%6\n v_sub_f32 %7,%8,%7"
   : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "s"(x)); FIN }
// V4: packed: 4 pk_add (x - w) + 4 pk_fma per 8 windows
__global__ __launch_bounds__(256) void v4(float* out, float x) {
  f2 a0={0,0},a1={0,0},a2={0,0},a3={0,0}; float b=threadIdx.x*1e-3f; f2 w0={b,b+1},w1={b+2,b+3},w2={b+4,b+5},w3={b+6,b+7}; f2 t0,t1,t2,t3; f2 xx={x,x};
  for (int it=0; it<ITERS; ++it) { asm volatile(
   "v_pk_add_f32 %4,%8,%9 neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 %5,%8,%10 neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 %6,%8,%11 neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 %7,%8,%12 neg_lo:[0,1] neg_hi:[0,1]\n"
   "v_pk_fma_f32 %0,%4,%4,%0\n v_pk_fma_f32 %1,%5,%5,%1\n v_pk_fma_f32 %2,%6,%6,%2\n v_pk_fma_f32 %3,%7,%7,%3"
   : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"=&v"(t0),"=&v"(t1),"=&v"(t2),"=&v"(t3) : "v"(xx),"v"(w0),"v"(w1),"v"(w2),"v"(w3));
   asm volatile(
   "v_pk_add_f32 %4,%8,%9 neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 %5,%8,%10 neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 %6,%8,%11 neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 %7,%8,%12 neg_lo:[0,1] neg_hi:[0,1]\n"
   "v_pk_fma_f32 %0,%4,%4,%0\n v_pk_fma_f32 %1,%5,%5,%1\n v_pk_fma_f32 %2,%6,%6,%2\n v_pk_fma_f32 %3,%7,%7,%3"
   : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"=&v"(t0),"=&v"(t1),"=&v"(t2),"=&v"(t3) : "v"(xx),"v"(w0),"v"(w1),"v"(w2),"v"(w3)); }
  out[blockIdx.x*blockDim.x+threadIdx.x]=a0[0]+a0[1]+a1[0]+a1[1]+a2[0]+a2[1]+a3[0]+a3[1]; }
// V5: 8 v_sub then 8 v_fma (VOP3 3-operand form instead of v_fmac)
__global__ __launch_bounds__(256) void v5(float* out, float x) { INIT
  for (int it=0; it<ITERS; ++it) asm volatile(
   "v_sub_f32 %8,%16,%17\n v_sub_f32 %9,%16,%18\n v_sub_f32 %10,%16,%19\n v_sub_f32 %11,%16,%20\n v_sub_f32 %12,%16,%21\n v_sub_f32 %13,%16,%22\n v_sub_f32 %14,%16,%23\n v_sub_f32 %15,%16,%24\n"
   "v_fma_f32 %0,%8,%8,%0\n v_fma_f32 %1,%9,%9,%1\n v_fma_f32 %2,%10,%10,%2\n v_fma_f32 %3,%11,%11,%3\n v_fma_f32 %4,%12,%12,%4\n v_fma_f32 %5,%13,%13,%5\n v_fma_f32 %6,%14,%14,%6\n v_fma_f32 %7,%15,%15,%7"
   : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7),"=&v"(t0),"=&v"(t1),"=&v"(t2),"=&v"(t3),"=&v"(t4),"=&v"(t5),"=&v"(t6),"=&v"(t7)
   : "s"(x),"v"(w0),"v"(w1),"v"(w2),"v"(w3),"v"(w4),"v"(w5),"v"(w6),"v"(w7)); FIN }
typedef void (*kern_t)(float*, float);
int main() {
    float* out; hipMalloc(&out, 256 * 8192 * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    kern_t ks[] = {v0, v1, v2, v3, v4, v5};
    const char* names[] = {"V0 8sub(s,v)+8fmac", "V1 8sub(v,v)+8fmac", "V2 16 fmac distinct", "V3 16 sub(s,v)", "V4 packed 8pk_add+8pk_fma(=32)", "V5 8sub+8fma(VOP3)"};
    double per_iter[] = {16, 16, 16, 16, 32, 16};   // scalar-equivalent lane-ops per lane per iteration
    for (int v = 0; v < 6; ++v) for (int wps = 2; wps <= 8; wps *= 2) {
        int blocks = 256 * wps; float ms = 0, best = 1e9;
        for (int rep = 0; rep < 4; ++rep) {
            hipEventRecord(a); hipLaunchKernelGGL(ks[v], dim3(blocks), dim3(256), 0, 0, out, 1.5f); hipEventRecord(b);
            hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
        }
        double ops = (double)blocks * 256 * ITERS * per_iter[v];
        double instr_per_simd = (double)wps * ITERS * (v == 4 ? 16 : 16);
        printf("%-32s waves/SIMD=%d: %.3f ms  %.1f T lane-ops/s  (%.2f ns per wave-instr per SIMD)\n", names[v], wps, best, ops / best / 1e9, best * 1e6 / instr_per_simd);
    }
    return 0;
}
