// What does v_mfma_f64_16x16x4_f64 compute, bit for bit?  D = A(16x4) * B(4x16) + C on random operands of mixed
// magnitude, compared on the host with the candidate evaluation orders: an fma chain over k = 0..3 starting from C,
// the same chain k = 3..0, and products summed first and C added last.  Prints how many of the outputs each
// candidate reproduces exactly.  Also states the operand layout used (and checks it with an asymmetric B).
// Build: hipcc --offload-arch=gfx950 -O2 -ffp-contract=off tools/mfma_f64_check.hip -o tools/mfma_f64_check.bin
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

typedef double double4v __attribute__((ext_vector_type(4)));

__global__ void k(const double *A, const double *B, const double *C, double *D, int trials) {
  const int lane = threadIdx.x, i = lane & 15, kk = lane >> 4;
  for (int t = 0; t < trials; ++t) {
    const double *a = A + t * 64, *b = B + t * 64, *c = C + t * 256;
    double *dd = D + t * 256;
    /* A[i][k] in lane i + 16 k; B[k][j] in lane j + 16 k; C/D[row][col]: col = lane & 15, row = (lane >> 4) + 4 r */
    double4v acc;
    for (int r = 0; r < 4; ++r) acc[r] = c[(kk + 4 * r) * 16 + i];
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i * 4 + kk], b[kk * 16 + i], acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) dd[(kk + 4 * r) * 16 + i] = acc[r];
  }
}

int main() {
  const int T = 4096;
  std::mt19937_64 rng(12345);
  std::uniform_real_distribution<double> u(-1.0, 1.0);
  std::vector<double> A(T * 64), B(T * 64), C(T * 256), D(T * 256);
  auto rnd = [&](int spread) { return std::ldexp(u(rng), (int)(rng() % (2 * spread + 1)) - spread); };
  for (auto &x : A) x = rnd(20);
  for (auto &x : B) x = rnd(20);
  for (auto &x : C) x = rnd(30);
  double *dA, *dB, *dC, *dD;
  hipMalloc(&dA, A.size() * 8); hipMalloc(&dB, B.size() * 8); hipMalloc(&dC, C.size() * 8); hipMalloc(&dD, D.size() * 8);
  hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice);
  hipMemcpy(dB, B.data(), B.size() * 8, hipMemcpyHostToDevice);
  hipMemcpy(dC, C.data(), C.size() * 8, hipMemcpyHostToDevice);
  k<<<1, 64>>>(dA, dB, dC, dD, T);
  if (hipMemcpy(D.data(), dD, D.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) { printf("launch failed\n"); return 1; }
  long n = 0, fwd = 0, bwd = 0, sumfirst = 0, unfused = 0, tree = 0;
  for (int t = 0; t < T; ++t)
    for (int i = 0; i < 16; ++i)
      for (int j = 0; j < 16; ++j) {
        const double *a = &A[t * 64 + i * 4], c = C[t * 256 + i * 16 + j], got = D[t * 256 + i * 16 + j];
        double b[4];
        for (int q = 0; q < 4; ++q) b[q] = B[t * 64 + q * 16 + j];
        double f = c, g = c, s = 0, un = c;
        for (int q = 0; q < 4; ++q) { f = std::fma(a[q], b[q], f); un = un + a[q] * b[q]; }
        for (int q = 3; q >= 0; --q) g = std::fma(a[q], b[q], g);
        for (int q = 0; q < 4; ++q) s = std::fma(a[q], b[q], s);
        const double tr = std::fma(a[1], b[1], a[0] * b[0]) + std::fma(a[3], b[3], a[2] * b[2]) + c;
        ++n;
        fwd += !memcmp(&got, &f, 8); bwd += !memcmp(&got, &g, 8);
        const double sf = s + c; sumfirst += !memcmp(&got, &sf, 8);
        unfused += !memcmp(&got, &un, 8); tree += !memcmp(&got, &tr, 8);
      }
  printf("outputs %ld: fma chain k=0..3 from C: %ld   k=3..0 from C: %ld   products first, then + C: %ld   "
         "unfused k=0..3: %ld   pairwise tree: %ld\n", n, fwd, bwd, sumfirst, unfused, tree);
  return 0;
}
