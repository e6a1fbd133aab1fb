/*
 * orc_fft_lavc.c — TEST INFRASTRUCTURE (see bliss_oracle.h).
 *
 * FFT variant 3: the 512-point f32 real DFT the way libavcodec's C code evaluates it.
 *
 * The reference's frequency analysis calls av_rdft_init(9, DFT_R2C) / av_rdft_calc (ref src/frequency_sort.c:65,83).
 * libavcodec is absent from /root/reference and from this image and the reference pins no version; what is
 * restated here — from the PUBLISHED algorithm, not from a source file in reach — is the operation order of the
 * generic C path of FFmpeg 0.7 ... 4.4 (libavcodec/fft_template.c + rdft.c, FFT_FLOAT):
 *   * cosine tables ff_cos_N[i] = (float)cos(i * 2 pi / N) for i <= N/4, mirrored tab[N/2 - i] = tab[i]; the
 *     "sine" of a pass is read from the mirrored half of the same table (wim = wre + N/4, indexed downwards);
 *   * split-radix decimation in time on the bit-permuted input (split_radix_permutation), in place:
 *     fft4 / fft8 / fft16 written out, fft(n) = fft(n/2) on the first half, fft(n/4) on the third and the fourth
 *     quarter, then pass(): one radix-4-shaped butterfly per k with the twiddle applied as
 *     t1 = a2.re * wre - a2.im * (-wim), t2 = a2.re * (-wim) + a2.im * wre (CMUL), every product and sum rounded to
 *     float (no fused operations: the x86-64 baseline the reference targets has none);
 *   * the real-input post-pass of rdft_calc_c for DFT_R2C: ev / od from (data[i1], data[n - i1]) with k1 = k2 = 0.5,
 *     odsum = od * (tcos + i tsin) in the "negative sin" form, tsin = tcos + n/4.
 * Which libavcodec build produced the reference's golden values (tests/test_analyze.c:30-35) is unknown; x86 builds
 * normally run hand-written SSE/AVX versions of the same butterflies.  tests/test_fft_independence.py records where
 * `frequency` of the reference's recording lands under this variant and under the other three.
 * f32 only: orc_alt_r2c512_f64(3, ...) is variant 0's transform.
 */
#include <math.h>
#include <string.h>
#include "bliss_oracle.h"

typedef struct { float re, im; } lavc_cpx;

#define NC 256 /* complex points of the 512-point real transform */

static int g_init = 0;
static float g_cos16[8], g_cos32[16], g_cos64[32], g_cos128[64], g_cos256[128], g_cos512[256];
static unsigned short g_revtab[NC];
static float g_sqrthalf;

static void init_cos(float *tab, int m) {
  const double freq = 2 * 3.14159265358979323846 / m;
  for (int i = 0; i <= m / 4; ++i) tab[i] = (float)cos(i * freq);
  for (int i = 1; i < m / 4; ++i) tab[m / 2 - i] = tab[i];
}

static int split_radix_permutation(int i, int n, int inverse) {
  if (n <= 2) return i & 1;
  int m = n >> 1;
  if (!(i & m)) return split_radix_permutation(i, m, inverse) * 2;
  m >>= 1;
  if (inverse == !(i & m)) return split_radix_permutation(i, m, inverse) * 4 + 1;
  return split_radix_permutation(i, m, inverse) * 4 - 1;
}

static void init_tables(void) {
  if (g_init) return;
  init_cos(g_cos16, 16); init_cos(g_cos32, 32); init_cos(g_cos64, 64);
  init_cos(g_cos128, 128); init_cos(g_cos256, 256); init_cos(g_cos512, 512);
  for (int i = 0; i < NC; ++i) g_revtab[-split_radix_permutation(i, NC, 0) & (NC - 1)] = (unsigned short)i;
  g_sqrthalf = (float)0.70710678118654752440;
  g_init = 1;
}

#define BF(x, y, a, b) do { x = (a) - (b); y = (a) + (b); } while (0)
#define CMUL(dre, dim, are, aim, bre, bim) do {      \
    (dre) = (are) * (bre) - (aim) * (bim);           \
    (dim) = (are) * (bim) + (aim) * (bre);           \
  } while (0)

#define BUTTERFLIES(a0, a1, a2, a3) {                \
    BF(t3, t5, t5, t1);                              \
    BF(a2.re, a0.re, a0.re, t5);                     \
    BF(a3.im, a1.im, a1.im, t3);                     \
    BF(t4, t6, t2, t6);                              \
    BF(a3.re, a1.re, a1.re, t4);                     \
    BF(a2.im, a0.im, a0.im, t6);                     \
  }
#define TRANSFORM(a0, a1, a2, a3, wre, wim) {        \
    CMUL(t1, t2, a2.re, a2.im, wre, -(wim));         \
    CMUL(t5, t6, a3.re, a3.im, wre, (wim));          \
    BUTTERFLIES(a0, a1, a2, a3)                      \
  }
#define TRANSFORM_ZERO(a0, a1, a2, a3) {             \
    t1 = a2.re; t2 = a2.im; t5 = a3.re; t6 = a3.im;  \
    BUTTERFLIES(a0, a1, a2, a3)                      \
  }

/* z[0 .. 8n), w[1 .. 2n) */
static void pass(lavc_cpx *z, const float *wre, unsigned n) {
  float t1, t2, t3, t4, t5, t6;
  const int o1 = 2 * n, o2 = 4 * n, o3 = 6 * n;
  const float *wim = wre + o1;
  n--;
  TRANSFORM_ZERO(z[0], z[o1], z[o2], z[o3]);
  TRANSFORM(z[1], z[o1 + 1], z[o2 + 1], z[o3 + 1], wre[1], wim[-1]);
  do {
    z += 2; wre += 2; wim -= 2;
    TRANSFORM(z[0], z[o1], z[o2], z[o3], wre[0], wim[0]);
    TRANSFORM(z[1], z[o1 + 1], z[o2 + 1], z[o3 + 1], wre[1], wim[-1]);
  } while (--n);
}

static void fft4(lavc_cpx *z) {
  float t1, t2, t3, t4, t5, t6, t7, t8;
  BF(t3, t1, z[0].re, z[1].re);
  BF(t8, t6, z[3].re, z[2].re);
  BF(z[2].re, z[0].re, t1, t6);
  BF(t4, t2, z[0].im, z[1].im);
  BF(t7, t5, z[2].im, z[3].im);
  BF(z[3].im, z[1].im, t4, t8);
  BF(z[3].re, z[1].re, t3, t7);
  BF(z[2].im, z[0].im, t2, t5);
}

static void fft8(lavc_cpx *z) {
  float t1, t2, t3, t4, t5, t6;
  fft4(z);
  BF(t1, z[5].re, z[4].re, -z[5].re);
  BF(t2, z[5].im, z[4].im, -z[5].im);
  BF(t5, z[7].re, z[6].re, -z[7].re);
  BF(t6, z[7].im, z[6].im, -z[7].im);
  BUTTERFLIES(z[0], z[2], z[4], z[6]);
  TRANSFORM(z[1], z[3], z[5], z[7], g_sqrthalf, g_sqrthalf);
}

static void fft16(lavc_cpx *z) {
  float t1, t2, t3, t4, t5, t6;
  const float cos_16_1 = g_cos16[1], cos_16_3 = g_cos16[3];
  fft8(z);
  fft4(z + 8);
  fft4(z + 12);
  TRANSFORM_ZERO(z[0], z[4], z[8], z[12]);
  TRANSFORM(z[2], z[6], z[10], z[14], g_sqrthalf, g_sqrthalf);
  TRANSFORM(z[1], z[5], z[9], z[13], cos_16_1, cos_16_3);
  TRANSFORM(z[3], z[7], z[11], z[15], cos_16_3, cos_16_1);
}

#define DECL_FFT(n, n2, n4, tab)       \
  static void fft##n(lavc_cpx *z) {    \
    fft##n2(z);                        \
    fft##n4(z + n4 * 2);               \
    fft##n4(z + n4 * 3);               \
    pass(z, tab, n4 / 2);              \
  }
DECL_FFT(32, 16, 8, g_cos32)
DECL_FFT(64, 32, 16, g_cos64)
DECL_FFT(128, 64, 32, g_cos128)
DECL_FFT(256, 128, 64, g_cos256)

/* in place; FFmpeg's packed RDFT output: x[0] = Re X0, x[1] = Re X256, x[2k] = Re Xk, x[2k+1] = Im Xk */
void orc_lavc_rdft512_f32(float *data) {
  init_tables();
  const int n = 512;
  lavc_cpx tmp[NC], *z = (lavc_cpx *)data;
  for (int j = 0; j < NC; ++j) tmp[g_revtab[j]] = z[j]; /* fft_permute */
  memcpy(z, tmp, sizeof tmp);
  fft256(z);                                            /* fft_calc */
  const float k1 = 0.5f, k2 = 0.5f;                    /* k2 = 0.5 - inverse */
  const float *tcos = g_cos512, *tsin = g_cos512 + (n >> 2);
  lavc_cpx ev, od, odsum;
  ev.re = data[0];
  data[0] = ev.re + data[1];
  data[1] = ev.re - data[1];
  int i;
  for (i = 1; i < (n >> 2); i++) {
    const int i1 = 2 * i, i2 = n - i1;
    ev.re = k1 * (data[i1] + data[i2]);
    od.im = k2 * (data[i2] - data[i1]);
    ev.im = k1 * (data[i1 + 1] - data[i2 + 1]);
    od.re = k2 * (data[i1 + 1] + data[i2 + 1]);
    odsum.re = od.re * tcos[i] + od.im * tsin[i];      /* DFT_R2C: negative_sin */
    odsum.im = od.im * tcos[i] - od.re * tsin[i];
    data[i1] = ev.re + odsum.re;
    data[i1 + 1] = ev.im + odsum.im;
    data[i2] = ev.re - odsum.re;
    data[i2 + 1] = odsum.im - ev.im;
  }
  data[2 * i + 1] = -data[2 * i + 1];                   /* sign_convention = -1 */
}
