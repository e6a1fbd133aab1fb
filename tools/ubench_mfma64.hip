// Can gfx950 run v_mfma_f64_16x16x4_f64 beside f64 VALU work?  One workgroup of 512 threads per CU
// (two waves per SIMD, as k_env_windows3), hipEvent timing of:
//   valu   every wave: N x 16 independent v_fma_f64
//   mfma   every wave: N x 4 independent v_mfma_f64_16x16x4_f64
//   same   every wave: N x (16 v_fma_f64 + M v_mfma) interleaved in one instruction stream
//   split  waves 0-3 the VALU loop, waves 4-7 the MFMA loop (one of each per SIMD)
// If the matrix pipe is its own unit, `same` and `split` cost max(valu, mfma), not the sum.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -w tools/ubench_mfma64.hip -o tools/ubench_mfma64.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

#define VALU16()                                                                                    \
  x0 = __builtin_fma(x0, b, a); x1 = __builtin_fma(x1, b, a); x2 = __builtin_fma(x2, b, a);         \
  x3 = __builtin_fma(x3, b, a); x4 = __builtin_fma(x4, b, a); x5 = __builtin_fma(x5, b, a);         \
  x6 = __builtin_fma(x6, b, a); x7 = __builtin_fma(x7, b, a); x8 = __builtin_fma(x8, b, a);         \
  x9 = __builtin_fma(x9, b, a); xa = __builtin_fma(xa, b, a); xb = __builtin_fma(xb, b, a);         \
  xc = __builtin_fma(xc, b, a); xd = __builtin_fma(xd, b, a); xe = __builtin_fma(xe, b, a);         \
  xf = __builtin_fma(xf, b, a);

template <int MODE, int M> __global__ __launch_bounds__(512) void k(double *out, int n, double a, double b) {
  double x0 = a + threadIdx.x, x1 = b, x2 = a * 2, x3 = b * 3, x4 = a * 5, x5 = b * 7, x6 = a * 11, x7 = b * 13;
  double x8 = a * 17, x9 = b * 19, xa = a * 23, xb = b * 29, xc = a * 31, xd = b * 37, xe = a * 41, xf = b * 43;
  d4 c0 = {a, b, a, b}, c1 = c0 * 2.0, c2 = c0 * 3.0, c3 = c0 * 5.0;
  const double ma = a * 1e-3 * (threadIdx.x & 15), mb = b * 1e-3;
  const int wave = threadIdx.x >> 6;
  const bool do_valu = MODE == 0 || MODE == 2 || (MODE == 3 && wave < 4) || (MODE == 4 && (wave & 1) == 0);
  const bool do_mfma = MODE == 1 || MODE == 2 || (MODE == 3 && wave >= 4) || (MODE == 4 && (wave & 1) == 1);
  if (MODE == 2) {
    for (int i = 0; i < n; ++i) {
      if (M >= 1) c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(ma, mb, c0, 0, 0, 0);
      VALU16();
      if (M >= 2) c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ma, mb, c1, 0, 0, 0);
      if (M >= 3) c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(ma, mb, c2, 0, 0, 0);
      if (M >= 4) c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(ma, mb, c3, 0, 0, 0);
      asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
      asm volatile("" : "+v"(x8), "+v"(x9), "+v"(xa), "+v"(xb), "+v"(xc), "+v"(xd), "+v"(xe), "+v"(xf));
    }
  } else if (do_valu) {
    for (int i = 0; i < n; ++i) {
      VALU16();
      asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
      asm volatile("" : "+v"(x8), "+v"(x9), "+v"(xa), "+v"(xb), "+v"(xc), "+v"(xd), "+v"(xe), "+v"(xf));
    }
  } else if (do_mfma) {
    for (int i = 0; i < n; ++i) {
      c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(ma, mb, c0, 0, 0, 0);
      if (M >= 2) c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ma, mb, c1, 0, 0, 0);
      if (M >= 3) c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(ma, mb, c2, 0, 0, 0);
      if (M >= 4) c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(ma, mb, c3, 0, 0, 0);
    }
  }
  d4 c = c0 + c1 + c2 + c3;
  out[blockIdx.x * blockDim.x + threadIdx.x] =
      x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + x8 + x9 + xa + xb + xc + xd + xe + xf + c.x + c.y + c.z + c.w;
}

template <int MODE, int M> float run(const char *name, int blocks) {
  double *out; hipMalloc(&out, 8ull * blocks * 512);
  const int n = 20000;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  k<MODE, M><<<blocks, 512>>>(out, 100, 1.000001, 0.9999999);
  float best = 1e9f;
  for (int r = 0; r < 3; ++r) {
    hipEventRecord(a); k<MODE, M><<<blocks, 512>>>(out, n, 1.000001, 0.9999999); hipEventRecord(b);
    hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); best = ms < best ? ms : best;
  }
  // cycles per loop iteration per wave at 2.4 GHz
  printf("%-28s blocks %4d: %8.3f ms  = %7.1f cycles@2.4GHz per iteration\n", name, blocks, best, best * 1e-3 * 2.4e9 / n);
  hipFree(out);
  return best;
}
int main() {
  for (int blocks : {256, 512}) {
    run<0, 4>("valu: 16 fma", blocks);
    run<1, 1>("mfma: 1", blocks);
    run<1, 2>("mfma: 2", blocks);
    run<1, 4>("mfma: 4", blocks);
    run<2, 1>("same wave: 16 fma + 1 mfma", blocks);
    run<2, 2>("same wave: 16 fma + 2 mfma", blocks);
    run<2, 4>("same wave: 16 fma + 4 mfma", blocks);
    run<3, 1>("split w0-3 valu / w4-7 mfma1", blocks);
    run<3, 2>("split w0-3 valu / w4-7 mfma2", blocks);
    run<4, 1>("split even valu / odd mfma1", blocks);
  }
  return 0;
}
