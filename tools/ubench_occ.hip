// SUPERSEDED by tools/gen_ubench_issue.py (round 4): the percentages this tool prints include two things that are
// not issue limits — a taken branch per 32-64 instructions (~150 cycles each) and the tail in which the waves that
// were served first (the oldest wave of a SIMD gets 96 % of the slots) have finished and the others run alone.
// Measured for a fixed time instead of a fixed amount of work, two waves per SIMD issue f64 at 99.5 % of the rate.
// How much of the f64 VALU issue rate can W waves per SIMD reach?  One workgroup per CU (a 100 KB
// LDS allocation keeps a second one out), 64 * 4 * W threads, every wave runs N iterations of C
// independent v_fma_f64 chains (+ optionally LDS round trips between the bursts, like a kernel that
// exchanges data through LDS).  Prints cycles per instruction per SIMD at the measured clock.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -w tools/ubench_occ.hip -o tools/ubench_occ.bin
#include <hip/hip_runtime.h>
#include <cstdio>

template <int CH, int LDSOPS>
__global__ void k(double *out, int n, double a, double b, long long *clk) {
  extern __shared__ double lds[];
  double x[CH];
#pragma unroll
  for (int i = 0; i < CH; ++i) x[i] = a * (i + 1) + threadIdx.x;
  const long long t0 = clock64();
  for (int it = 0; it < n; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int i = 0; i < CH; ++i) x[i] = __builtin_fma(x[i], b, a);
    }
    if (LDSOPS) {  /* a write -> read exchange through this wave's private slice, LDSOPS b128 each way */
      double *p = lds + (threadIdx.x >> 6) * 1024 + (threadIdx.x & 63) * 2;
#pragma unroll
      for (int q = 0; q < LDSOPS; ++q) { p[q * 128] = x[(2 * q) % CH]; p[q * 128 + 1] = x[(2 * q + 1) % CH]; }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      double *g = lds + (threadIdx.x >> 6) * 1024 + ((threadIdx.x + 17) & 63) * 2;
#pragma unroll
      for (int q = 0; q < LDSOPS; ++q) { x[(2 * q) % CH] += g[q * 128]; x[(2 * q + 1) % CH] += g[q * 128 + 1]; }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
#pragma unroll
    for (int i = 0; i < CH; ++i) asm volatile("" : "+v"(x[i]));
  }
  const long long t1 = clock64();
  double s = 0;
#pragma unroll
  for (int i = 0; i < CH; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

template <int CH, int LDSOPS> void run(int waves_per_simd) {
  const int threads = 256 * waves_per_simd, blocks = 256, n = 4000;
  double *out; long long *clk; hipMalloc(&out, 8ull * blocks * threads); hipMalloc(&clk, 8);
  hipFuncSetAttribute(reinterpret_cast<const void *>(k<CH, LDSOPS>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  k<CH, LDSOPS><<<blocks, threads, 100 * 1024>>>(out, 50, 1.000001, 0.9999999, clk);
  float best = 1e9f;
  for (int r = 0; r < 3; ++r) {
    hipEventRecord(a); k<CH, LDSOPS><<<blocks, threads, 100 * 1024>>>(out, n, 1.000001, 0.9999999, clk); hipEventRecord(b);
    hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); best = ms < best ? ms : best;
  }
  long long c; hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
  const double instr_per_simd = (double)n * 4 * CH * waves_per_simd;   /* f64 fma wave-instructions per SIMD */
  // clock64 ticks at 100 MHz on this part: use the event time and report at the clock that makes 4 waves x 8 chains = 4.0
  printf("waves/SIMD %d, chains %2d, lds ops %2d: %7.3f ms  -> %.3f us per 1000 fma-instr per SIMD  (%.1f G wave-instr/s chip)\n",
         waves_per_simd, CH, LDSOPS, best, best * 1e3 / (instr_per_simd / 1000.0), instr_per_simd * 1024 / (best * 1e-3) / 1e9);
  hipFree(out); hipFree(clk);
}
int main() {
  for (int w = 1; w <= 4; ++w) { run<8, 0>(w); run<16, 0>(w); }
  for (int w = 1; w <= 4; ++w) { run<16, 8>(w); }
  return 0;
}
