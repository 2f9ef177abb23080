// Rate of v_dot2c_f32_bf16 / v_dot2_f32_bf16 / v_cvt_pk_bf16_f32 vs v_fmac_f32 on gfx950 (16 waves per CU).
#include <hip/hip_runtime.h>
#include <stdio.h>
#define ITERS 2048
#define INIT float a0=0,a1=0,a2=0,a3=0,a4=0,a5=0,a6=0,a7=0; unsigned w0=threadIdx.x*2654435761u,w1=w0+11,w2=w0+22,w3=w0+33,w4=w0+44,w5=w0+55,w6=w0+66,w7=w0+77;
#define FIN out[blockIdx.x*blockDim.x+threadIdx.x]=a0+a1+a2+a3+a4+a5+a6+a7;
__global__ __launch_bounds__(1024) void k_dot2c(float* out, unsigned x) { INIT
  for (int it=0; it<ITERS; ++it) asm volatile(
   "v_dot2c_f32_bf16 %0,%8,%9\n v_dot2c_f32_bf16 %1,%8,%10\n v_dot2c_f32_bf16 %2,%8,%11\n v_dot2c_f32_bf16 %3,%8,%12\n v_dot2c_f32_bf16 %4,%8,%13\n v_dot2c_f32_bf16 %5,%8,%14\n v_dot2c_f32_bf16 %6,%8,%15\n v_dot2c_f32_bf16 %7,%8,%16\n"
   "v_dot2c_f32_bf16 %0,%8,%10\n v_dot2c_f32_bf16 %1,%8,%11\n v_dot2c_f32_bf16 %2,%8,%12\n v_dot2c_f32_bf16 %3,%8,%13\n v_dot2c_f32_bf16 %4,%8,%14\n v_dot2c_f32_bf16 %5,%8,%15\n v_dot2c_f32_bf16 %6,%8,%16\n v_dot2c_f32_bf16 %7,%8,%9"
   : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "s"(x),"v"(w0),"v"(w1),"v"(w2),"v"(w3),"v"(w4),"v"(w5),"v"(w6),"v"(w7)); FIN }
__global__ __launch_bounds__(1024) void k_fmac(float* out, unsigned x) { INIT
  for (int it=0; it<ITERS; ++it) asm volatile(
   "v_fmac_f32 %0,%8,%9\n v_fmac_f32 %1,%8,%10\n v_fmac_f32 %2,%8,%11\n v_fmac_f32 %3,%8,%12\n v_fmac_f32 %4,%8,%13\n v_fmac_f32 %5,%8,%14\n v_fmac_f32 %6,%8,%15\n v_fmac_f32 %7,%8,%16\n"
   "v_fmac_f32 %0,%8,%10\n v_fmac_f32 %1,%8,%11\n v_fmac_f32 %2,%8,%12\n v_fmac_f32 %3,%8,%13\n v_fmac_f32 %4,%8,%14\n v_fmac_f32 %5,%8,%15\n v_fmac_f32 %6,%8,%16\n v_fmac_f32 %7,%8,%9"
   : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "s"(x),"v"(w0),"v"(w1),"v"(w2),"v"(w3),"v"(w4),"v"(w5),"v"(w6),"v"(w7)); FIN }
__global__ __launch_bounds__(1024) void k_cvt(float* out, unsigned x) { INIT float f0=threadIdx.x*1e-3f,f1=f0+1,f2=f0+2,f3=f0+3;
  unsigned r0,r1,r2,r3,r4,r5,r6,r7;
  for (int it=0; it<ITERS; ++it) { asm volatile(
   "v_cvt_pk_bf16_f32 %0,%8,%9\n v_cvt_pk_bf16_f32 %1,%9,%10\n v_cvt_pk_bf16_f32 %2,%10,%11\n v_cvt_pk_bf16_f32 %3,%11,%8\n v_cvt_pk_bf16_f32 %4,%8,%10\n v_cvt_pk_bf16_f32 %5,%9,%11\n v_cvt_pk_bf16_f32 %6,%10,%8\n v_cvt_pk_bf16_f32 %7,%11,%9\n"
   "v_cvt_pk_bf16_f32 %0,%8,%9\n v_cvt_pk_bf16_f32 %1,%9,%10\n v_cvt_pk_bf16_f32 %2,%10,%11\n v_cvt_pk_bf16_f32 %3,%11,%8\n v_cvt_pk_bf16_f32 %4,%8,%10\n v_cvt_pk_bf16_f32 %5,%9,%11\n v_cvt_pk_bf16_f32 %6,%10,%8\n v_cvt_pk_bf16_f32 %7,%11,%9"
   : "=&v"(r0),"=&v"(r1),"=&v"(r2),"=&v"(r3),"=&v"(r4),"=&v"(r5),"=&v"(r6),"=&v"(r7) : "v"(f0),"v"(f1),"v"(f2),"v"(f3)); a0 += __uint_as_float(r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7) * 0.f; f0 += 1e-7f; } FIN }
__global__ __launch_bounds__(1024) void k_fmac_v(float* out, unsigned x) { INIT float xv = __uint_as_float(x) + threadIdx.x * 0.f;
  for (int it=0; it<ITERS; ++it) asm volatile(
   "v_fmac_f32 %0,%8,%9\n v_fmac_f32 %1,%8,%10\n v_fmac_f32 %2,%8,%11\n v_fmac_f32 %3,%8,%12\n v_fmac_f32 %4,%8,%13\n v_fmac_f32 %5,%8,%14\n v_fmac_f32 %6,%8,%15\n v_fmac_f32 %7,%8,%16\n"
   "v_fmac_f32 %0,%8,%10\n v_fmac_f32 %1,%8,%11\n v_fmac_f32 %2,%8,%12\n v_fmac_f32 %3,%8,%13\n v_fmac_f32 %4,%8,%14\n v_fmac_f32 %5,%8,%15\n v_fmac_f32 %6,%8,%16\n v_fmac_f32 %7,%8,%9"
   : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(xv),"v"(w0),"v"(w1),"v"(w2),"v"(w3),"v"(w4),"v"(w5),"v"(w6),"v"(w7)); FIN }
__global__ __launch_bounds__(1024) void k_dot2c_v(float* out, unsigned x) { INIT unsigned xv = x + threadIdx.x * 0u;
  for (int it=0; it<ITERS; ++it) asm volatile(
   "v_dot2c_f32_bf16 %0,%8,%9\n v_dot2c_f32_bf16 %1,%8,%10\n v_dot2c_f32_bf16 %2,%8,%11\n v_dot2c_f32_bf16 %3,%8,%12\n v_dot2c_f32_bf16 %4,%8,%13\n v_dot2c_f32_bf16 %5,%8,%14\n v_dot2c_f32_bf16 %6,%8,%15\n v_dot2c_f32_bf16 %7,%8,%16\n"
   "v_dot2c_f32_bf16 %0,%8,%10\n v_dot2c_f32_bf16 %1,%8,%11\n v_dot2c_f32_bf16 %2,%8,%12\n v_dot2c_f32_bf16 %3,%8,%13\n v_dot2c_f32_bf16 %4,%8,%14\n v_dot2c_f32_bf16 %5,%8,%15\n v_dot2c_f32_bf16 %6,%8,%16\n v_dot2c_f32_bf16 %7,%8,%9"
   : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(xv),"v"(w0),"v"(w1),"v"(w2),"v"(w3),"v"(w4),"v"(w5),"v"(w6),"v"(w7)); FIN }
// the exact block with x in a VGPR: 8 sub(v,v) + 8 fmac
__global__ __launch_bounds__(1024) void k_subfmac_v(float* out, unsigned x) { INIT float xv = __uint_as_float(x) + threadIdx.x * 0.f; float t0,t1,t2,t3,t4,t5,t6,t7;
  float f0=__uint_as_float(w0&0x3fffffff),f1=f0+1,f2=f0+2,f3=f0+3,f4=f0+4,f5=f0+5,f6=f0+6,f7=f0+7;
  for (int it=0; it<ITERS; ++it) asm volatile(
   "v_sub_f32 %8,%16,%17\n v_sub_f32 %9,%16,%18\n v_sub_f32 %10,%16,%19\n v_sub_f32 %11,%16,%20\n v_sub_f32 %12,%16,%21\n v_sub_f32 %13,%16,%22\n v_sub_f32 %14,%16,%23\n v_sub_f32 %15,%16,%24\n"
   "v_fmac_f32 %0,%8,%8\n v_fmac_f32 %1,%9,%9\n v_fmac_f32 %2,%10,%10\n v_fmac_f32 %3,%11,%11\n v_fmac_f32 %4,%12,%12\n v_fmac_f32 %5,%13,%13\n v_fmac_f32 %6,%14,%14\n v_fmac_f32 %7,%15,%15"
   : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7),"=&v"(t0),"=&v"(t1),"=&v"(t2),"=&v"(t3),"=&v"(t4),"=&v"(t5),"=&v"(t6),"=&v"(t7)
   : "v"(xv),"v"(f0),"v"(f1),"v"(f2),"v"(f3),"v"(f4),"v"(f5),"v"(f6),"v"(f7)); FIN }
__global__ __launch_bounds__(1024) void k_subfmac_s(float* out, unsigned x) { INIT float xs = __uint_as_float(x); float t0,t1,t2,t3,t4,t5,t6,t7;
  float f0=__uint_as_float(w0&0x3fffffff),f1=f0+1,f2=f0+2,f3=f0+3,f4=f0+4,f5=f0+5,f6=f0+6,f7=f0+7;
  for (int it=0; it<ITERS; ++it) asm volatile(
   "v_sub_f32 %8,%16,%17\n v_sub_f32 %9,%16,%18\n v_sub_f32 %10,%16,%19\n v_sub_f32 %11,%16,%20\n v_sub_f32 %12,%16,%21\n v_sub_f32 %13,%16,%22\n v_sub_f32 %14,%16,%23\n v_sub_f32 %15,%16,%24\n"
   "v_fmac_f32 %0,%8,%8\n v_fmac_f32 %1,%9,%9\n v_fmac_f32 %2,%10,%10\n v_fmac_f32 %3,%11,%11\n v_fmac_f32 %4,%12,%12\n v_fmac_f32 %5,%13,%13\n v_fmac_f32 %6,%14,%14\n v_fmac_f32 %7,%15,%15"
   : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7),"=&v"(t0),"=&v"(t1),"=&v"(t2),"=&v"(t3),"=&v"(t4),"=&v"(t5),"=&v"(t6),"=&v"(t7)
   : "s"(xs),"v"(f0),"v"(f1),"v"(f2),"v"(f3),"v"(f4),"v"(f5),"v"(f6),"v"(f7)); FIN }
typedef void (*kern_t)(float*, unsigned);
int main() {
    float* out; hipMalloc(&out, 1024 * 512 * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    kern_t ks[] = {k_fmac, k_fmac_v, k_dot2c, k_dot2c_v, k_cvt, k_subfmac_s, k_subfmac_v}; const char* names[] = {"v_fmac_f32 (s,v)", "v_fmac_f32 (v,v)", "v_dot2c_f32_bf16 (s,v)", "v_dot2c_f32_bf16 (v,v)", "v_cvt_pk_bf16_f32", "8sub(s,v)+8fmac", "8sub(v,v)+8fmac"};
    for (int v = 0; v < 7; ++v) { float ms = 0, best = 1e9;
        for (int rep = 0; rep < 4; ++rep) { hipEventRecord(a); hipLaunchKernelGGL(ks[v], dim3(256), dim3(1024), 0, 0, out, 0x3f803f80u); hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms; }
        printf("%-26s 4 waves/SIMD: %.3f ms -> %.2f ns per wave-instr per SIMD\n", names[v], best, best * 1e6 / (4.0 * ITERS * 16)); }
    // numerics probe of dot2c: does it round products / flush denormals?
    return 0;
}
