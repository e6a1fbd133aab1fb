// Issue behaviour of one SIMD: W waves per SIMD (blocks of 256*W threads, one block per CU),
// each wave runs chains of dependent ops with ILP independent chains.  Reports cycles per
// wave-instruction per SIMD (s_memtime), for f64 add, the f32<->f64 convert pair used by
// the ordered sum, and v_mov DPP.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -w tools/ubench_lat.hip -o tools/ubench_lat.bin
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE, int ILP> __global__ void k(double *out, long long *cyc, int n, double a, double b) {
  double x[ILP];
  float f[ILP];
#pragma unroll
  for (int j = 0; j < ILP; ++j) { x[j] = a * (j + 1) + threadIdx.x; f[j] = (float)x[j]; }
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
#pragma unroll
      for (int j = 0; j < ILP; ++j) {
        if (MODE == 0) x[j] += b;                                   // 1 instr
        if (MODE == 1) f[j] = (float)((double)f[j] + b);           // 3 instr: cvt, add, cvt
        if (MODE == 2) x[j] = __builtin_fma(x[j], b, a);
        if (MODE == 3) { f[j] = (float)x[j]; x[j] = (double)f[j]; } // 2 cvt
      }
#pragma unroll
      for (int j = 0; j < ILP; ++j) asm volatile("" : "+v"(x[j]), "+v"(f[j]));
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
#pragma unroll
  for (int j = 0; j < ILP; ++j) s += x[j] + f[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int MODE, int ILP> void run(const char *name, int waves_per_simd, int ipo) {
  const int threads = 256 * waves_per_simd, blocks = 256, n = 2000;
  double *out; long long *cyc, h;
  hipMalloc(&out, 8ull * blocks * threads); hipMalloc(&cyc, 8);
  k<MODE, ILP><<<blocks, threads>>>(out, cyc, 10, 1.000001, 1e-9);
  k<MODE, ILP><<<blocks, threads>>>(out, cyc, n, 1.000001, 1e-9);
  hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  const double instr = (double)n * 8 * ILP * ipo; // per wave
  printf("%-12s ILP %d  waves/SIMD %d: %7.2f cycles per instr per wave, %6.2f cycles per instr per SIMD\n", name, ILP,
         waves_per_simd, h / instr, h / (instr * waves_per_simd));
  hipFree(out); hipFree(cyc);
}
int main() {
  run<0, 1>("add_f64", 1, 1); run<0, 2>("add_f64", 1, 1); run<0, 4>("add_f64", 1, 1); run<0, 8>("add_f64", 1, 1);
  run<0, 1>("add_f64", 2, 1); run<0, 2>("add_f64", 2, 1); run<0, 4>("add_f64", 2, 1);
  run<2, 1>("fma_f64", 1, 1); run<2, 4>("fma_f64", 1, 1);
  run<1, 1>("cvt-add-cvt", 1, 3); run<1, 2>("cvt-add-cvt", 1, 3); run<1, 4>("cvt-add-cvt", 1, 3);
  run<1, 1>("cvt-add-cvt", 2, 3); run<1, 2>("cvt-add-cvt", 2, 3);
  run<3, 1>("cvt-cvt", 1, 2); run<3, 4>("cvt-cvt", 1, 2); run<3, 4>("cvt-cvt", 2, 2);
  return 0;
}
