/*
 * bliss_oracle.c — TEST INFRASTRUCTURE (see bliss_oracle.h), not product code.
 *
 * Plain-C restatement of the arithmetic of the reference's per-song analysis:
 * every function cites the reference lines whose operation order, operand
 * types and rounding points it follows.  Compile WITHOUT floating-point
 * contraction (-ffp-contract=off, the Makefile does) — the reference is built
 * -std=c99 on x86-64 (CMakeLists.txt:22), i.e. SSE2 arithmetic, no FMA, no
 * excess precision.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "bliss_oracle.h"

/* ------------------------------------------------------------------------ */
/* ref src/helpers.c:30-37 — int32 accumulator (wraps), C truncating divide  */
int orc_mean(const int16_t *pcm, int n) {
  uint32_t acc = 0; /* unsigned arithmetic == two's-complement wrap of the int */
  for (int i = 0; i < n; ++i) acc += (uint32_t)(int32_t)pcm[i];
  return (int32_t)acc / n;
}

/* ref src/helpers.c:39-49 — v*v is an int32 product (wraps), summed in int64 */
int orc_variance(const int16_t *pcm, int n, int mean) {
  int64_t acc = 0;
  for (int i = 0; i < n; ++i) {
    int32_t v = (int32_t)pcm[i] - mean;
    int32_t sq = (int32_t)((uint32_t)v * (uint32_t)v);
    acc += sq;
  }
  return (int)(acc / n);
}

/* ------------------------------------------------------------------------ */
/* ref src/amplitude_sort.c:12-80                                            */
#define HIST_BINS 65536
#define HIST_PASSES 300 /* loop runs g = 0..300, i.e. 301 passes (:41) */
#define INT_LO (32767 - 1000)
#define INT_HI (32767 + 1000)

float orc_amplitude(const int16_t *pcm, int n, orc_result *r) {
  float *h = (float *)calloc(HIST_BINS, sizeof(float));
  float *s = (float *)calloc(HIST_BINS, sizeof(float));
  int start = 0, end = n - 1;
  while (pcm[start] == 0) ++start; /* :26-27 (undefined on all-zero input) */
  while (pcm[end] == 0) --end;     /* :29-31 */
  for (int i = start; i <= end; ++i) h[(int)pcm[i] + 32768] += 1; /* :33-39 */

  for (int g = 0; g <= HIST_PASSES; ++g) { /* :41-59 */
    s[0] = h[0];
    s[1] = (float)(1. / 4. * (h[0] + (2 * h[1]) + h[2]));
    s[2] = (float)(1. / 9. * (h[0] + (2 * h[1]) + (3 * h[2]) + (2 * h[3]) + h[4]));
    for (int i = 3; i < HIST_BINS - 5; ++i) {
      /* f32 left-to-right sum, then * (double)(1/27), rounded to f32 on store */
      float acc = h[i - 3] + (3 * h[i - 2]);
      acc = acc + (6 * h[i - 1]);
      acc = acc + (7 * h[i]);
      acc = acc + (6 * h[i + 1]);
      acc = acc + (3 * h[i + 2]);
      acc = acc + h[i + 3];
      s[i] = (float)(1. / 27. * acc);
    }
    for (int i = 3; i < HIST_BINS - 5; ++i) h[i] = s[i];
  }
  /* :62-66 — float /= int (int converted to float), float *= double, fabs */
  float denom = (float)(start - end);
  float integral = 0;
  for (int i = INT_LO; i <= INT_HI; ++i) { /* :69-71, f32 sequential sum */
    float v = s[i] / denom;
    v = (float)(v * 100.);
    v = (float)fabs(v);
    integral += v;
  }
  free(h);
  free(s);
  if (r) { r->start = start; r->end = end; r->hist_integral = integral; }
  return -0.2f * integral + 6.0f; /* :79 */
}

/* ------------------------------------------------------------------------ */
/* ref src/frequency_sort.c:20-140                                           */
float orc_frequency(const int16_t *pcm, int n, int channels, orc_result *r) {
  enum { W = 512 };
  float hann[W], x[W], ps[W / 2 + 1];
  for (int i = 0; i < W; ++i) /* :40-42: float*(float - double) -> double -> f32 */
    hann[i] = (float)(.5f * (1.0f - cos(2 * M_PI * i / (W - 1))));
  for (int i = 0; i <= W / 2; ++i) ps[i] = 0.0f;
  int n_frames = (n / channels) / W; /* :50 */
  for (int f = 0; f < n_frames; ++f) {
    const int16_t *p = pcm + (size_t)f * W * channels;
    if (channels == 2) { /* :69-75 int add, C truncating /2, f32 multiply */
      for (int d = 0; d < W; ++d)
        x[d] = (float)(((int)p[2 * d] + (int)p[2 * d + 1]) / 2) * hann[d];
    } else { /* :76-80 */
      for (int d = 0; d < W; ++d) x[d] = (float)p[d] * hann[d];
    }
    orc_rdft512_f32(x); /* :83 */
    ps[0] = x[0] * x[0]; /* :86-87 overwritten every frame, never read later */
    for (int d = 1; d < W / 2; ++d) { /* :88-93 f32 accumulate in frame order */
      float re = x[2 * d], im = x[2 * d + 1];
      float raw = (re * re) + (im * im);
      ps[d] += raw;
    }
  }
  float peak = 0;
  for (int d = 1; d <= W / 2; ++d) { /* :97-102 (ps[256] stays 0) */
    ps[d] = (float)sqrt(ps[d] / W);
    peak = (float)fmax(ps[d], peak);
  }
  for (int d = 1; d <= W / 2; ++d) /* :105-107 */
    ps[d] = (float)(20 * log10(ps[d] / peak) - 3);
  float b[5] = {0, 0, 0, 0, 0};
  b[0] = (ps[2] + ps[4]) / 2; /* :110-112 */
  b[1] = (ps[6] + ps[8]) / 2;
  for (int i = 10; i <= 60; ++i) b[2] += ps[i]; /* :114-117, divisor 50 */
  b[2] /= 50;
  for (int i = 61; i <= 118; ++i) b[3] += ps[i]; /* :119-122, divisor 57 */
  b[3] /= 57;
  for (int i = 119; i <= 234; ++i) b[4] += ps[i]; /* :124-127, divisor 115 */
  b[4] /= 115;
  float sum = b[4] + b[3] + b[2] - b[0] - b[1]; /* :129 */
  if (r) { r->n_frames = n_frames; r->freq_peak = peak; }
  return (float)((1. / 3.) * sum + 68. / 3.); /* :139 */
}

/* ------------------------------------------------------------------------ */
/* ref src/tempo_atk_sort.c:19-40                                            */
void orc_rect_filter(double *out, const double *in, int n, int width) {
  int half = (int)round(width / 2.);
  double run = 0;
  for (int k = 0; k < width; ++k) run += in[k];
  for (int k = 0; k < n - width; ++k) {
    out[k + half - 1] = run;
    run -= in[k];
    run += in[k + width];
  }
  for (int k = n - width; k < n; ++k) out[n - half] += in[k];
  for (int k = 0; k < n; ++k) out[k] /= width;
}

/* constants: literal digits of ref include/bandpass_coeffs.h:1-7,484-492 */
static const double FIR17[17] = {
    -0.0023470, 0.0044613, -0.0114627, 0.0226382, -0.0405147, 0.0580037,
    -0.0779167, 0.0882711, 0.9065095,  0.0882711, -0.0779167, 0.0580037,
    -0.0405147, 0.0226382, -0.0114627, 0.0044613, -0.0023470};
static const double BUT_B[7] = {1.9510e-05, 1.1706e-04, 2.9266e-04, 3.9021e-04,
                                2.9266e-04, 1.1706e-04, 1.9510e-05};
static const double BUT_A[7] = {1.00000, -4.59007, 8.91034, -9.34191,
                                5.56998, -1.78845, 0.24136};

/* ref src/tempo_atk_sort.c:42-296 */
void orc_envelope(const int16_t *pcm, int n, uint64_t duration, orc_result *r,
                  float *energies) {
  enum { W = 512, HOP = 256 };
  int nb_frames = (n - (n % W)) * 2 / W; /* :63-64 */
  int iter = (n - n % W) - W;            /* :66-67 */
  int mean = orc_mean(pcm, n);           /* :101-103 */
  int var = orc_variance(pcm, n, mean);
  double md = (double)mean / 32768;      /* :105-107 */
  double vd = (double)var / 32768;
  vd /= 32768;

  double *xs = (double *)malloc((size_t)n * sizeof(double));
  for (int i = 0; i < n; ++i) xs[i] = ((double)pcm[i] / 32768 - md) / vd; /* :109-114 */

  double *filt = (double *)calloc((size_t)nb_frames, sizeof(double));
  double in[W], re[W / 2 + 1], im[W / 2 + 1];
  int n_windows = 0;
  for (int b = 0; b < iter; b += HOP) { /* :120 */
    for (int j = 0; j < W; ++j) {       /* :123-138, delay line zeroed per window */
      /* tap m reads xs[b+j-m], zero before the window start */
#define TAP(m) ((j - (m)) >= 0 ? xs[b + j - (m)] : 0.0)
      double y = 0;
      for (int k = 7; k > 0; --k) y += FIR17[k] * (TAP(k) + TAP(16 - k));
      y += TAP(8) * FIR17[8];
      y += FIR17[0] * (TAP(0) + TAP(16));
#undef TAP
      in[j] = y;
    }
    orc_r2c512_f64(in, re, im); /* :141 */
    float sum_fft = 0;          /* :142-149: float += double, rounds every step */
    for (int k = 0; k <= W / 2; ++k) {
      double p = re[k] * re[k] + im[k] * im[k];
      sum_fft += p;
    }
    filt[n_windows] += sum_fft; /* :150-151 */
    ++n_windows;
  }
  free(xs);
  if (energies)
    for (int j = 0; j < nb_frames; ++j) energies[j] = (float)filt[j];

  /* Part 2 — ref :184-233 */
  int N = 2 * nb_frames;
  double *t1 = (double *)calloc((size_t)N, sizeof(double));
  double *t2 = (double *)calloc((size_t)N, sizeof(double));
  double *wa = (double *)calloc((size_t)N, sizeof(double));
  double *ss = (double *)calloc((size_t)N, sizeof(double));
  float mu = 100.0f, lambda = 0.8f; /* :170-171 */
  for (int j = 0; j < nb_frames; ++j) { /* :186-190 */
    t1[2 * j] = log(1 + mu * filt[j]) / log(1 + mu);
    t1[2 * j + 1] = 0;
  }
  double xr[7] = {0}, yr[7] = {0}, y = 0;
  for (int j = 0; j < N; ++j) { /* :201-218 */
    for (int k = 6; k > 0; --k) { xr[k] = xr[k - 1]; yr[k] = yr[k - 1]; }
    xr[0] = t1[j];
    yr[0] = y;
    double d = 0, c = 0;
    for (int k = 0; k < 7; ++k) d += BUT_B[k] * xr[k];
    for (int k = 1; k < 7; ++k) c += BUT_A[k] * yr[k - 1];
    y = (d - c) / BUT_A[0];
    t2[j] = y;
  }
  t1[0] = t2[0]; /* :221-226 */
  for (int j = 1; j < N; ++j) {
    double dj = t2[j] - t2[j - 1];
    t1[j] = dj > 0 ? dj : 0;
  }
  for (int j = 0; j < N; ++j) /* :229-232 — (1-lambda), lambda*172 are f32 */
    wa[j] = (1 - lambda) * t2[j] + lambda * 172 * t1[j] / 10;

  double atk_sum = 0; /* :246-248 */
  for (int j = 0; j < N - 1; ++j) atk_sum += wa[j];

  /* Part 3 — ref :259-284 */
  for (int j = 0; j < N - 1; ++j) ss[j] += wa[j];
  if (N >= 20) {
    orc_rect_filter(wa, ss, N, 19); /* :267 out = wa keeps its old edge cells */
    memset(ss, 0, (size_t)N * sizeof(double));
    orc_rect_filter(ss, wa, N, 19); /* :270 */
  }
  float epsilon = 0.000001f; /* :275 */
  int beat = 0;
  double margin = 1e300;
  for (int j = 1; j < N - 1; ++j) { /* :277-280 */
    double dl = ss[j] - ss[j - 1], dr = ss[j] - ss[j + 1];
    if (dl > epsilon && dr > epsilon) beat++;
    /* distance of this decision from flipping (exactly flat cells, e.g. the
     * zero edge cells the box filter leaves, carry no information) */
    if (dl == 0 && dr == 0) continue;
    double m;
    if (dl > epsilon && dr > epsilon) m = fmin(dl, dr) - epsilon;
    else if (dl > epsilon) m = epsilon - dr;
    else if (dr > epsilon) m = epsilon - dl;
    else m = fmax(epsilon - dl, epsilon - dr);
    if (m < margin) margin = m;
  }
  double tempo = 4 * (float)beat / (float)duration - 30.4; /* :283 */
  double atk = -1.74 * atk_sum * 10000 / n + 58.3;          /* :284 */
  free(t1); free(t2); free(wa); free(ss); free(filt);
  if (r) {
    r->tempo = (float)tempo;
    r->attack = (float)atk;
    r->mean = mean; r->variance = var;
    r->nb_frames = nb_frames; r->n_windows = n_windows;
    r->beat = beat; r->atk_sum = atk_sum; r->min_peak_margin = margin;
  }
}

/* ------------------------------------------------------------------------ */
/* ref src/analyze.c:40-80 (after a successful decode)                       */
int orc_analyze_pcm(const int16_t *pcm, int n, int channels, uint64_t duration,
                    orc_result *r) {
  memset(r, 0, sizeof(*r));
  r->amplitude = orc_amplitude(pcm, n, r);
  r->frequency = orc_frequency(pcm, n, channels, r);
  orc_envelope(pcm, n, duration, r, NULL);
  float rating = (float)(fmax(r->tempo, 0) + r->amplitude + r->frequency +
                         fmax(r->attack, 0)); /* :68-72 double sum -> float */
  r->force = rating;
  r->calm_or_loud = rating > 0 ? 0 : (rating < 0 ? 1 : 2); /* :73-79 */
  return r->calm_or_loud;
}

/* ref src/analyze.c:96-100 — every operand f32; gcc emits sqrtf-equivalent */
float orc_distance(const float a[4], const float b[4]) {
  float d0 = a[0] - b[0], d1 = a[1] - b[1], d2 = a[2] - b[2], d3 = a[3] - b[3];
  float s = d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
  return (float)sqrt(s);
}

/* ref src/analyze.c:135-140 — f32 dot and norms, sqrt in double, f32 divide?
 * No: sqrt() returns double, so the product of the two square roots and the
 * division are double; the result is rounded to f32 on assignment. */
float orc_cosine(const float a[4], const float b[4]) {
  float dot = a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3];
  float na = a[0] * a[0] + a[1] * a[1] + a[2] * a[2] + a[3] * a[3];
  float nb = b[0] * b[0] + b[1] * b[1] + b[2] * b[2] + b[3] * b[3];
  return (float)(dot / (sqrt(na) * sqrt(nb)));
}

/* the all-pairs form of the same expression, for the kernels' matrix (test infrastructure, as everything here) */
void orc_cosine_matrix(const float *vecs, int n, float *out) {
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j)
      out[(size_t)i * n + j] = orc_cosine(vecs + 4 * i, vecs + 4 * j);
}

void orc_distance_matrix(const float *vecs, int n, float *out) {
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j)
      out[(size_t)i * n + j] = orc_distance(vecs + 4 * i, vecs + 4 * j);
}
