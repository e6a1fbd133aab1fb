// Host-side unit test of bliss_amd/csrc/bl_fft.h: runs the exact lane code of the
// device transform (16 lanes emulated sequentially per phase) and compares the
// power spectrum with a long-double DFT.  Build: g++ -O2 -std=c++17 -ffp-contract=off
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../bliss_amd/csrc/bl_fft.h"

template <typename T> static double run(unsigned seed) {
  std::vector<bl_c2<T>> tw256(256), tw512(256), xch(BL_FFT_XCH_ELEMS), par(BL_FFT_PAR_ELEMS);
  const double pi = 3.14159265358979323846;
  for (int e = 0; e < 256; ++e) {
    const int ex = ((e & 15) * (e >> 4)) & 255;  // tw256 is laid out [k1][n0] -> exponent n0*k1
    tw256[e].re = (T)cos(2 * pi * ex / 256); tw256[e].im = (T)-sin(2 * pi * ex / 256);
    tw512[e].re = (T)cos(2 * pi * e / 512); tw512[e].im = (T)-sin(2 * pi * e / 512);
  }
  std::vector<double> x(512);
  srand(seed);
  for (auto &v : x) v = (rand() / (double)RAND_MAX - 0.5) * 2000.0;
  T re[16][16], im[16][16];
  for (int n0 = 0; n0 < 16; ++n0)
    for (int m1 = 0; m1 < 16; ++m1) {
      int m = 16 * m1 + n0;
      re[n0][m1] = (T)x[2 * m]; im[n0][m1] = (T)x[2 * m + 1];
    }
  for (int l = 0; l < 16; ++l) bl_fft512_phaseA<T>(l, re[l], im[l], tw256.data(), xch.data());
  for (int l = 0; l < 16; ++l) bl_fft512_phaseB<T>(l, re[l], im[l], xch.data(), par.data());
  std::vector<double> pw(257, -1.0);
  for (int l = 0; l < 16; ++l) {
    T own[8], mir[8], mid;
    bl_fft512_phaseC<T>(l, re[l], im[l], tw512.data(), par.data(), own, mir, mid);
    for (int k0 = 0; k0 < 8; ++k0) {
      pw[l + 16 * k0] = own[k0];
      pw[256 - l - 16 * k0] = mir[k0];
    }
    if (l == 0) pw[128] = mid;
  }
  double worst = 0;
  for (int k = 0; k <= 256; ++k) {
    long double sr = 0, si = 0;
    for (int n = 0; n < 512; ++n) {
      long double a = -2.0L * 3.14159265358979323846264338327950288L * k * n / 512.0L;
      sr += (T)x[n] * cosl(a); si += (T)x[n] * sinl(a);
    }
    long double ref = sr * sr + si * si;
    double rel = (double)fabsl((long double)pw[k] - ref) / (double)(ref + 1e-30L);
    // compare relative to the spectrum's mean level too (bins near zero)
    if (rel > worst && ref > 1e-3) worst = rel;
  }
  return worst;
}

int main() {
  double w64 = 0, w32 = 0;
  for (unsigned s = 1; s <= 5; ++s) { w64 = fmax(w64, run<double>(s)); w32 = fmax(w32, run<float>(s)); }
  printf("worst rel err f64 %.3e  f32 %.3e\n", w64, w32);
  if (!(w64 < 1e-11) || !(w32 < 2e-3)) { printf("FAIL\n"); return 1; }
  printf("OK\n");
  return 0;
}
