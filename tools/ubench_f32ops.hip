// Issue rate of the f32 / integer instructions on the input side of k_freq_frames, whole chip
// (hipEvent timing): 2048 blocks x 256 threads, each thread N iterations of 8 independent ops.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -w tools/ubench_f32ops.hip -o tools/ubench_f32ops.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE> __global__ void k(float *out, int n, float a, float b, int ia) {
  float x[8]; int y[8]; f2 p[8];
  for (int j = 0; j < 8; ++j) { x[j] = a * (j + 1) + threadIdx.x; y[j] = ia * (j + 3) + threadIdx.x; p[j] = (f2){x[j], x[j] + 1}; }
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (MODE == 0) x[j] = x[j] + b;                               // v_add_f32
      if (MODE == 1) x[j] = __builtin_truncf(x[j]);                 // v_trunc_f32
      if (MODE == 2) { x[j] = (float)y[j]; }                        // v_cvt_f32_i32
      if (MODE == 3) p[j] = p[j] * (f2){b, b};                      // v_pk_mul_f32
      if (MODE == 4) p[j] = __builtin_elementwise_fma(p[j], (f2){b, b}, (f2){a, a});  // v_pk_fma_f32
      if (MODE == 5) y[j] = (int)(short)(y[j] & 0xFFFF) + (int)(short)((unsigned)y[j] >> 16);  // v_add_u32_sdwa
      if (MODE == 6) x[j] = __builtin_fmaf(x[j], b, a);             // v_fma_f32
      if (MODE == 7) y[j] = (int)x[j];                              // v_cvt_i32_f32
      if (MODE == 8) x[j] = __builtin_rintf(x[j]);                  // v_rndne_f32
    }
    asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]));
    asm volatile("" : "+v"(y[0]), "+v"(y[1]), "+v"(y[2]), "+v"(y[3]), "+v"(y[4]), "+v"(y[5]), "+v"(y[6]), "+v"(y[7]));
    asm volatile("" : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]));
  }
  float s = 0;
  for (int j = 0; j < 8; ++j) s += x[j] + (float)y[j] + p[j].x + p[j].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE> void run(const char *name) {
  const int blocks = 2048, threads = 256, n = 20000;
  float *out; hipMalloc(&out, 4ull * blocks * threads);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  k<MODE><<<blocks, threads>>>(out, 100, 1.000001f, 1.0000001f, 3);
  hipEventRecord(a); k<MODE><<<blocks, threads>>>(out, n, 1.000001f, 1.0000001f, 3); hipEventRecord(b);
  hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b);
  double winstr = (double)blocks * threads / 64 * n * 8;
  printf("%-16s %8.3f ms  %6.3f T wave-instr/s  -> %.2f SIMD-cycles per instruction @2.4 GHz\n", name, ms,
         winstr / ms / 1e9, (ms * 1e-3) * 1024.0 * 2.4e9 / winstr);
  hipFree(out);
}
int main() {
  run<0>("v_add_f32"); run<6>("v_fma_f32"); run<1>("v_trunc_f32"); run<8>("v_rndne_f32"); run<2>("v_cvt_f32_i32"); run<7>("v_cvt_i32_f32");
  run<3>("v_pk_mul_f32"); run<4>("v_pk_fma_f32"); run<5>("v_add_u32_sdwa");
  return 0;
}
