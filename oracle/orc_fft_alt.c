/*
 * orc_fft_alt.c — TEST INFRASTRUCTURE (see bliss_oracle.h).
 *
 * Two more implementations of the two DFTs orc_fft.c restates, structurally unlike it and unlike each other, so
 * that a test can show what the results owe to the choice of FFT (the reference's are FFTW3 and libavcodec,
 * neither of which can be run here):
 *   variant 1  recursive decimation-in-time radix-4 (one radix-2 level: 512 = 4^4 * 2) on the UNPACKED input — a
 *              512-point complex transform of (x, 0), no even/odd packing, no real-input split, other twiddles in
 *              another order;
 *   variant 2  the defining sum X_k = sum_n x_n exp(-2 pi i k n / 512), term by term: the f64 transform with long
 *              double products, sums and twiddles (cosl / sinl), rounded to double once; the f32 transform in
 *              double, rounded to float once — what every FFT of that precision approximates.
 * orc_set_fft_variant() selects what orc_rdft512_f32 / orc_r2c512_f64 (and therefore orc_frequency / orc_envelope)
 * run; 0, the default: f64 the packed radix-2 of orc_fft.c, f32 libavcodec's operation order (orc_fft_lavc.c, also
 * variant 3); 4: the f32 packed radix-2 of orc_fft.c, the default until round 6.
 * Used by tests/test_fft_independence.py and tools/fft_independence.py only.
 */
#include <math.h>
#include <string.h>
#include "bliss_oracle.h"

#define N 512

static int g_variant = 0;
void orc_set_fft_variant(int v) { g_variant = (v >= 0 && v <= 4) ? v : 0; }
int orc_fft_variant(void) { return g_variant; }

static int g_init = 0;
static double g_c[N], g_s[N];        /* exp(-2 pi i k / 512) */
static float g_cf[N], g_sf[N];
static long double g_cl[N], g_sl[N];

static void init_tables(void) {
  if (g_init) return;
  const long double pil = 3.14159265358979323846264338327950288L;
  for (int k = 0; k < N; ++k) {
    g_cl[k] = cosl(2.0L * pil * k / N);
    g_sl[k] = -sinl(2.0L * pil * k / N);
    g_c[k] = (double)g_cl[k];
    g_s[k] = (double)g_sl[k];
    g_cf[k] = (float)g_c[k];
    g_sf[k] = (float)g_s[k];
  }
  g_init = 1;
}

/* y[0..n) = DFT_n of x[0], x[stride], ..., out of place; n a power of two <= 512 */
#define DEFINE_R4(NAME, T, CT, ST)                                                              \
  static void NAME(int n, int stride, const T *xr, const T *xi, T *yr, T *yi) {                 \
    if (n == 1) { yr[0] = xr[0]; yi[0] = xi[0]; return; }                                       \
    if (n == 2) {                                                                               \
      T ar = xr[0], ai = xi[0], br = xr[stride], bi = xi[stride];                               \
      yr[0] = ar + br; yi[0] = ai + bi; yr[1] = ar - br; yi[1] = ai - bi;                       \
      return;                                                                                   \
    }                                                                                           \
    const int m = n / 4, tw = N / n;                                                            \
    for (int j = 0; j < 4; ++j)                                                                 \
      NAME(m, 4 * stride, xr + j * stride, xi + j * stride, yr + j * m, yi + j * m);            \
    for (int k = 0; k < m; ++k) {                                                               \
      const T ar = yr[k], ai = yi[k];                                                           \
      const T w1r = CT[(k * tw) % N], w1i = ST[(k * tw) % N];                                   \
      const T w2r = CT[(2 * k * tw) % N], w2i = ST[(2 * k * tw) % N];                           \
      const T w3r = CT[(3 * k * tw) % N], w3i = ST[(3 * k * tw) % N];                           \
      const T br = yr[k + m] * w1r - yi[k + m] * w1i, bi = yr[k + m] * w1i + yi[k + m] * w1r;   \
      const T cr = yr[k + 2 * m] * w2r - yi[k + 2 * m] * w2i;                                   \
      const T ci = yr[k + 2 * m] * w2i + yi[k + 2 * m] * w2r;                                   \
      const T dr = yr[k + 3 * m] * w3r - yi[k + 3 * m] * w3i;                                   \
      const T di = yr[k + 3 * m] * w3i + yi[k + 3 * m] * w3r;                                   \
      const T s0r = ar + cr, s0i = ai + ci, s1r = ar - cr, s1i = ai - ci;                       \
      const T s2r = br + dr, s2i = bi + di, s3r = br - dr, s3i = bi - di;                       \
      yr[k] = s0r + s2r;         yi[k] = s0i + s2i;                                             \
      yr[k + m] = s1r + s3i;     yi[k + m] = s1i - s3r;     /* a - i b - c + i d */             \
      yr[k + 2 * m] = s0r - s2r; yi[k + 2 * m] = s0i - s2i;                                     \
      yr[k + 3 * m] = s1r - s3i; yi[k + 3 * m] = s1i + s3r; /* a + i b - c - i d */             \
    }                                                                                           \
  }

DEFINE_R4(r4_f64, double, g_c, g_s)
DEFINE_R4(r4_f32, float, g_cf, g_sf)

void orc_alt_r2c512_f64(int variant, const double *in, double *re, double *im) {
  init_tables();
  if (variant == 1) {
    double xi[N], yr[N], yi[N];
    memset(xi, 0, sizeof xi);
    r4_f64(N, 1, in, xi, yr, yi);
    for (int k = 0; k <= N / 2; ++k) { re[k] = yr[k]; im[k] = yi[k]; }
  } else {
    for (int k = 0; k <= N / 2; ++k) {
      long double sr = 0, si = 0;
      for (int n = 0; n < N; ++n) {
        const int e = (k * n) % N;
        sr += (long double)in[n] * g_cl[e];
        si += (long double)in[n] * g_sl[e];
      }
      re[k] = (double)sr;
      im[k] = (k == 0 || k == N / 2) ? 0.0 : (double)si;
    }
  }
}

/* FFmpeg RDFT packed layout, in place (see orc_rdft512_f32) */
void orc_alt_rdft512_f32(int variant, float *x) {
  init_tables();
  float yr[N], yi[N];
  if (variant == 1) {
    float xi[N];
    memset(xi, 0, sizeof xi);
    r4_f32(N, 1, x, xi, yr, yi);
  } else {
    for (int k = 0; k <= N / 2; ++k) {
      double sr = 0, si = 0;
      for (int n = 0; n < N; ++n) {
        const int e = (k * n) % N;
        sr += (double)x[n] * g_c[e];
        si += (double)x[n] * g_s[e];
      }
      yr[k] = (float)sr;
      yi[k] = (float)si;
    }
  }
  x[0] = yr[0];
  x[1] = yr[N / 2];
  for (int k = 1; k < N / 2; ++k) { x[2 * k] = yr[k]; x[2 * k + 1] = yi[k]; }
}
