// Aggregate f64 instruction rate of the whole chip (hipEvent timing): 1024 blocks x 256 threads,
// each thread runs N iterations of 8 independent ops of one kind.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -w tools/ubench_rate.hip -o tools/ubench_rate.bin
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE> __global__ void k(double *out, int n, double a, double b) {
  double x0 = a + threadIdx.x, x1 = b, x2 = a * 2, x3 = b * 3, x4 = a * 5, x5 = b * 7, x6 = a * 11, x7 = b * 13;
  for (int i = 0; i < n; ++i) {
    if (MODE == 0) { x0 += b; x1 += b; x2 += b; x3 += b; x4 += b; x5 += b; x6 += b; x7 += b; }
    if (MODE == 1) { x0 *= b; x1 *= b; x2 *= b; x3 *= b; x4 *= b; x5 *= b; x6 *= b; x7 *= b; }
    if (MODE == 2) { x0 = __builtin_fma(x0, b, a); x1 = __builtin_fma(x1, b, a); x2 = __builtin_fma(x2, b, a); x3 = __builtin_fma(x3, b, a);
                     x4 = __builtin_fma(x4, b, a); x5 = __builtin_fma(x5, b, a); x6 = __builtin_fma(x6, b, a); x7 = __builtin_fma(x7, b, a); }
    asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
template <int MODE> void run(const char *name, int blocks, int threads) {
  double *out; hipMalloc(&out, 8ull * blocks * threads);
  const int n = 20000;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  k<MODE><<<blocks, threads>>>(out, 100, 1.000001, 1.0000001);
  hipEventRecord(a); k<MODE><<<blocks, threads>>>(out, n, 1.000001, 1.0000001); hipEventRecord(b);
  hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b);
  double winstr = (double)blocks * threads / 64 * n * 8;
  printf("%-10s blocks %5d x %4d threads: %8.3f ms  -> %6.2f T wave-instr*64/s (lane-ops), %5.2f T wave-instr... per SIMD-cycle@2.4GHz: %.3f\n",
         name, blocks, threads, ms, winstr * 64 / ms / 1e9, winstr / ms / 1e9, winstr / (ms * 1e-3) / (1024.0 * 2.4e9));
  hipFree(out);
}
int main() {
  for (int th : {256, 512, 1024}) {
    run<0>("add_f64", 1024, th); run<1>("mul_f64", 1024, th); run<2>("fma_f64", 1024, th);
  }
  return 0;
}
