/*
 * orc_fft.c — TEST INFRASTRUCTURE (see bliss_oracle.h).
 *
 * The reference calls two third-party real FFTs that are absent from
 * /root/reference and from this image:
 *   - libavcodec av_rdft_init(9, DFT_R2C) / av_rdft_calc  (f32, 512 points),
 *     call sites ref src/frequency_sort.c:65,83 — version unpinned;
 *   - FFTW3 fftw_plan_dft_r2c_1d(512, FFTW_ESTIMATE) / fftw_execute (f64),
 *     call sites ref src/tempo_atk_sort.c:94,141 — version unpinned.
 * Both compute the unnormalised forward DFT X_k = sum_n x_n exp(-2 pi i k n/N).
 * They are restated here as a radix-2 complex FFT of N/2 points on the
 * even/odd packed input followed by the standard real-input split, evaluated
 * in the same precision as the library being replaced (f32 resp. f64).
 * Parity is anchored on the reference's goldens (tests/test_analyze.c:30-35),
 * which these reproduce to within the test's own 1e-5.
 * Since round 6 the f32 transform the oracle RUNS is orc_fft_lavc.c (libavcodec's
 * own operation order: every golden value to the last printed digit); the f32
 * radix-2 here stays as variant 4, the f64 one is the default f64 transform.
 */
#include <math.h>
#include "bliss_oracle.h"

#define HALF 256 /* complex points */
#define LOGH 8

static int g_init = 0;
static int g_rev[HALF];
static double g_cw[HALF / 2], g_sw[HALF / 2]; /* exp(-2 pi i k/256) */
static double g_cu[HALF], g_su[HALF];         /* exp(-2 pi i k/512) */
static float g_cwf[HALF / 2], g_swf[HALF / 2];
static float g_cuf[HALF], g_suf[HALF];

static void init_tables(void) {
  if (g_init) return;
  for (int i = 0; i < HALF; ++i) {
    int r = 0;
    for (int b = 0; b < LOGH; ++b)
      if (i & (1 << b)) r |= 1 << (LOGH - 1 - b);
    g_rev[i] = r;
  }
  const double pi = 3.14159265358979323846;
  for (int k = 0; k < HALF / 2; ++k) {
    g_cw[k] = cos(2.0 * pi * k / HALF);
    g_sw[k] = -sin(2.0 * pi * k / HALF);
    g_cwf[k] = (float)g_cw[k];
    g_swf[k] = (float)g_sw[k];
  }
  for (int k = 0; k < HALF; ++k) {
    g_cu[k] = cos(2.0 * pi * k / (2 * HALF));
    g_su[k] = -sin(2.0 * pi * k / (2 * HALF));
    g_cuf[k] = (float)g_cu[k];
    g_suf[k] = (float)g_su[k];
  }
  g_init = 1;
}

#define DEFINE_CFFT(NAME, T, CW, SW)                                        \
  static void NAME(T *zr, T *zi) {                                          \
    for (int i = 0; i < HALF; ++i) {                                        \
      int j = g_rev[i];                                                     \
      if (j > i) {                                                          \
        T t = zr[i]; zr[i] = zr[j]; zr[j] = t;                              \
        t = zi[i]; zi[i] = zi[j]; zi[j] = t;                                \
      }                                                                     \
    }                                                                       \
    for (int len = 2; len <= HALF; len <<= 1) {                             \
      int half = len >> 1, step = HALF / len;                               \
      for (int base = 0; base < HALF; base += len) {                        \
        for (int k = 0; k < half; ++k) {                                    \
          T wr = CW[k * step], wi = SW[k * step];                           \
          int a = base + k, b = a + half;                                   \
          T tr = zr[b] * wr - zi[b] * wi;                                   \
          T ti = zr[b] * wi + zi[b] * wr;                                   \
          zr[b] = zr[a] - tr; zi[b] = zi[a] - ti;                           \
          zr[a] = zr[a] + tr; zi[a] = zi[a] + ti;                           \
        }                                                                   \
      }                                                                     \
    }                                                                       \
  }

DEFINE_CFFT(cfft256_f32, float, g_cwf, g_swf)
DEFINE_CFFT(cfft256_f64, double, g_cw, g_sw)

/* f32, in place, FFmpeg RDFT packed output:
 * x[0]=Re X0, x[1]=Re X256, x[2k]=Re Xk, x[2k+1]=Im Xk (k=1..255). */
void orc_rdft512_f32(float *x) {
  /* default (0) and 3: libavcodec's operation order (orc_fft_lavc.c) — the one f32 DFT under which the oracle prints
   * all ten golden values of ref tests/test_analyze.c:30-35,62-68 to the last digit; 1, 2: the cross-checks of
   * orc_fft_alt.c; 4: the packed radix-2 below, the default until round 6 */
  const int variant = orc_fft_variant();
  if (variant == 0 || variant == 3) { orc_lavc_rdft512_f32(x); return; }
  if (variant == 1 || variant == 2) { orc_alt_rdft512_f32(variant, x); return; }
  float zr[HALF], zi[HALF];
  init_tables();
  for (int m = 0; m < HALF; ++m) { zr[m] = x[2 * m]; zi[m] = x[2 * m + 1]; }
  cfft256_f32(zr, zi);
  x[0] = zr[0] + zi[0];
  x[1] = zr[0] - zi[0];
  for (int k = 1; k < HALF; ++k) {
    int nk = HALF - k;
    float er = 0.5f * (zr[k] + zr[nk]), ei = 0.5f * (zi[k] - zi[nk]);
    float orr = 0.5f * (zi[k] + zi[nk]), oi = -0.5f * (zr[k] - zr[nk]);
    float wr = g_cuf[k], wi = g_suf[k];
    x[2 * k] = er + (orr * wr - oi * wi);
    x[2 * k + 1] = ei + (orr * wi + oi * wr);
  }
}

/* f64, out of place: re[k], im[k] for k = 0..256 (FFTW r2c layout). */
void orc_r2c512_f64(const double *in, double *re, double *im) {
  if (orc_fft_variant() == 1 || orc_fft_variant() == 2) { orc_alt_r2c512_f64(orc_fft_variant(), in, re, im); return; }
  double zr[HALF], zi[HALF];
  init_tables();
  for (int m = 0; m < HALF; ++m) { zr[m] = in[2 * m]; zi[m] = in[2 * m + 1]; }
  cfft256_f64(zr, zi);
  re[0] = zr[0] + zi[0]; im[0] = 0.0;
  re[HALF] = zr[0] - zi[0]; im[HALF] = 0.0;
  for (int k = 1; k < HALF; ++k) {
    int nk = HALF - k;
    double er = 0.5 * (zr[k] + zr[nk]), ei = 0.5 * (zi[k] - zi[nk]);
    double orr = 0.5 * (zi[k] + zi[nk]), oi = -0.5 * (zr[k] - zr[nk]);
    double wr = g_cu[k], wi = g_su[k];
    re[k] = er + (orr * wr - oi * wi);
    im[k] = ei + (orr * wi + oi * wr);
  }
}
