// Micro-benchmark of f64 / conversion latency and issue rate on gfx950 (one wave, then 2 waves per SIMD).
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ubench_f64.hip -o /tmp/ubench && /tmp/ubench
#include <hip/hip_runtime.h>
#pragma clang diagnostic ignored "-Wunused-value"
#pragma clang diagnostic ignored "-Wunused-result"
#include <cstdio>
#define N 4096
template <int MODE> __global__ void k(double *out, long long *cyc, double a, double b) {
  double x0 = a + threadIdx.x, x1 = b, x2 = a * 2, x3 = b * 3, x4 = a * 5, x5 = b * 7, x6 = a * 11, x7 = b * 13;
  float f = (float)a;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < N; ++i) {
    if (MODE == 0) { x0 = x0 + b; }                                              // dependent add
    if (MODE == 1) { x0 = x0 * b; }                                              // dependent mul
    if (MODE == 2) { x0 = __builtin_fma(x0, b, a); }                             // dependent fma
    if (MODE == 3) { f = (float)((double)f + b); }                               // the ordered-sum step
    if (MODE == 4) { x0 += b; x1 += b; x2 += b; x3 += b; x4 += b; x5 += b; x6 += b; x7 += b; } // 8 indep adds
    if (MODE == 5) { float g = (float)x0; x0 = (double)g + b; }                  // cvt pair + add
    if (MODE == 6) { x0 = (double)(float)x0; }                                   // cvt pair only
    if (MODE == 7) { x0 = x0 * b; x1 = x1 * b; x2 = x2 * b; x3 = x3 * b; x4 *= b; x5 *= b; x6 *= b; x7 *= b; }
    if (MODE == 8) { f = f + (float)b; }                                         // dependent f32 add
    if (MODE == 9) { x0 = (double)(float)x0; x1 = (double)(float)x1; x2 = (double)(float)x2; x3 = (double)(float)x3; } // 4 indep cvt pairs
    asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(f));
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + f;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int MODE> void run(const char *name, int ops, int threads) {
  double *out; long long *cyc, h;
  hipMalloc(&out, 8 * 1024 * 256); hipMalloc(&cyc, 8);
  k<MODE><<<1, threads>>>(out, cyc, 1.000001, 1.0000001);
  k<MODE><<<1, threads>>>(out, cyc, 1.000001, 1.0000001);
  hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-28s threads %4d: %7.2f cycles/iter  (%5.2f per op)\n", name, threads, (double)h / N, (double)h / N / ops);
  hipFree(out); hipFree(cyc);
}
int main() {
  for (int th : {64, 512, 1024}) {
    run<0>("dep v_add_f64", 1, th); run<1>("dep v_mul_f64", 1, th); run<2>("dep v_fma_f64", 1, th);
    run<3>("ordered-sum step (3 ops)", 3, th); run<5>("cvt,cvt,add via double", 3, th);
    run<6>("dep cvt f64->f32->f64", 2, th); run<9>("4 indep cvt pairs", 8, th);
    run<4>("8 indep v_add_f64", 8, th); run<7>("8 indep v_mul_f64", 8, th); run<8>("dep v_add_f32", 1, th);
  }
  return 0;
}
