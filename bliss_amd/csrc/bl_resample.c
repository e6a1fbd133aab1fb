/*
 * bl_resample.c — rate conversion to 22 050 Hz for bl_audio_decode().
 *
 * The reference hands every file that is not already 22 050 Hz s16 to libswresample with
 * default options and an out layout of stereo / s16 (ref src/decode.c:317-346, 379-401), so
 * what its analyzers see for a 44.1 or 48 kHz file is libswresample's output.  libswresample is
 * not in the reference tree; this file restates its published default algorithm (FFmpeg 4.x,
 * libswresample/resample.c, resample_template.c, rematrix.c, audioconvert.c):
 *
 *   - filter bank: a Kaiser(beta = 9)-windowed sinc, cutoff 0.97 of the lower Nyquist,
 *     ceil(32 / factor) taps made even, designed in double, normalised by the DC gain of phase 0,
 *     one row per phase of out_rate / in_rate in lowest terms (exact_rational: 147 phases for
 *     48 kHz, one for 44.1 kHz), or 1 024 truncated phases when that ratio has more; with an
 *     even phase count the upper half of the bank mirrors the lower;
 *   - position: integer phase arithmetic (dst_incr / src_incr as av_reduce leaves them), no
 *     interpolation between phases;
 *   - edges: the input is reflected about its first sample at the start and, on flush,
 *     about its end ((min(left, taps) + 1) / 2 samples); the number of output samples follows;
 *   - sources of at most 16 bits are converted in s16 with Q15 coefficients and an int32
 *     accumulator (exact integer arithmetic); wider sources as float (sample * 2^-31, float
 *     coefficients, eight strided partial sums with fused multiply-add combined pairwise — the
 *     order of libswresample's AVX2/FMA3 kernel, which is what x86-64 machines of the last
 *     decade run — then lrintf(v * 32768) clipped);
 *   - a mono source is up-mixed before the conversion with gain 1/sqrt(2) on both channels.
 *
 * Pin: the float path and the mono up-mix reproduce, bit for bit, the MD5 digests that ref
 * tests/test_decode.c:35-36,55-56 hold for the converted audio/song_s32.flac and
 * audio/song_s32_mono.flac (48 kHz, 24 bit -> 22 050 Hz s16; tests/test_ingest.py), and the
 * features of the converted song_s32.flac match ref tests/test_analyze.c:62-68.  The s16 path
 * shares the filter design and the position arithmetic; its Q15 rounding and integer
 * accumulation have no reference vector (parity unpinned for that half).
 */
#include "bl_resample.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "bliss.h"

#define RS_FILTER_SIZE 32
#define RS_PHASE_SHIFT 10
#define RS_CUTOFF 0.97
#define RS_KAISER_BETA 9.0


/* modified Bessel function of the first kind, order 0 (power series; the window only needs it
 * to double precision, the coefficients are rounded to float / Q15 afterwards) */
static double bessel_i0(double x) {
  double v = 1, last = 0, t = 1;
  x = x * x / 4;
  for (int i = 1; v != last && i < 500; ++i) {
    last = v;
    t *= x / ((double)i * (double)i);
    v += t;
  }
  return v;
}

static int64_t gcd64(int64_t a, int64_t b) {
  while (b) { int64_t t = a % b; a = b; b = t; }
  return a;
}

void bl_rs_plan_free(bl_rs_plan *f) {
  free(f->fbank);
  free(f->ibank);
  f->fbank = NULL;
  f->ibank = NULL;
}

int bl_rs_plan_geometry(bl_rs_plan *f, int out_rate, int in_rate) {
  memset(f, 0, sizeof *f);
  if (out_rate <= 0 || in_rate <= 0) return -1;
  double factor = (double)out_rate * RS_CUTOFF / (double)in_rate;
  if (factor > 1.0) factor = 1.0;
  int phase_count = 1 << RS_PHASE_SHIFT;
  { /* exact_rational (the default): when out_rate / in_rate in lowest terms has no more than
     * 1 024 phases, use exactly those — 147 for 48 kHz, 1 for 44.1 kHz */
    const int64_t g0 = gcd64(out_rate, in_rate);
    if (out_rate / g0 <= phase_count) phase_count = (int)(out_rate / g0);
  }
  int taps = (int)ceil(RS_FILTER_SIZE / factor);
  if (taps < 1) taps = 1;
  if (taps > 1) taps = (taps + 1) & ~1;
  const int alloc = (taps + 15) & ~15; /* rows padded with zero coefficients */
  f->phase_count = phase_count;
  f->taps = taps;
  f->alloc = alloc;

  /* out_rate / (in_rate * phase_count) in lowest terms, then scaled up like the library does
   * (only the ratio matters for the positions; kept for fidelity of the integer stepping) */
  int64_t num = out_rate, den = (int64_t)in_rate * phase_count;
  const int64_t g = gcd64(num, den);
  num /= g;
  den /= g;
  if (den > INT32_MAX / 2) return -1;
  while (den < (1 << 20) && num < (1 << 20)) { den *= 2; num *= 2; }
  f->src_incr = (int)num;
  f->dst_incr = (int)den;
  f->dst_incr_div = (int)(den / num);
  f->dst_incr_mod = (int)(den % num);
  return 0;
}

int bl_rs_plan_build(bl_rs_plan *f, int out_rate, int in_rate, int want_float) {
  if (bl_rs_plan_geometry(f, out_rate, in_rate)) return -1;
  f->is_float = want_float != 0;
  const int phase_count = f->phase_count, taps = f->taps, alloc = f->alloc;
  double factor = (double)out_rate * RS_CUTOFF / (double)in_rate;
  if (factor > 1.0) factor = 1.0;

  const size_t rows = (size_t)phase_count + 1;
  if (want_float) f->fbank = (float *)calloc(rows * alloc + 16, sizeof(float));
  else f->ibank = (int16_t *)calloc(rows * alloc + 16, sizeof(int16_t));
  double *tab = (double *)malloc(sizeof(double) * (size_t)(taps + 1));
  if ((!f->fbank && !f->ibank) || !tab) { free(tab); bl_rs_plan_free(f); return -1; }

  const int center = (taps - 1) / 2;
  const int ph_nb = phase_count % 2 ? phase_count : phase_count / 2 + 1;
  double norm = 0;
  for (int ph = 0; ph < ph_nb; ++ph) {
    /* no low-pass needed (factor 1): the sine is taken once per phase and its sign alternated */
    double s = factor == 1.0 ? sin(M_PI * ph / phase_count) * (center & 1 ? 1 : -1) : 0;
    for (int i = 0; i < taps; ++i, s = -s) {
      const double x = M_PI * ((double)(i - center) - (double)ph / phase_count) * factor;
      double y = x == 0 ? 1.0 : factor == 1.0 ? s / x : sin(x) / x;
      const double w = 2.0 * x / (factor * taps * M_PI);
      const double r = 1 - w * w;
      y *= bessel_i0(RS_KAISER_BETA * sqrt(r > 0 ? r : 0));
      tab[i] = y;
      if (!ph) norm += y;
    }
    if (want_float) {
      float *row = f->fbank + (size_t)ph * alloc, *mir = f->fbank + (size_t)(phase_count - ph) * alloc;
      for (int i = 0; i < taps; ++i) row[i] = (float)(tab[i] / norm);
      if (phase_count % 2 == 0)
        for (int i = 0; i < taps; ++i) mir[taps - 1 - i] = row[i];
    } else {
      int16_t *row = f->ibank + (size_t)ph * alloc, *mir = f->ibank + (size_t)(phase_count - ph) * alloc;
      for (int i = 0; i < taps; ++i) {
        long q = lrintf((float)(tab[i] * 32768 / norm));
        row[i] = (int16_t)(q > 32767 ? 32767 : q < -32768 ? -32768 : q);
      }
      if (phase_count % 2 == 0)
        for (int i = 0; i < taps; ++i) mir[taps - 1 - i] = row[i];
    }
  }
  free(tab);
  return 0;
}

/* sum of src[i] * filt[i] over `alloc` (a multiple of 8; the coefficients beyond the taps are
 * zero) in the order of libswresample's AVX/FMA3 kernel: lane j of 8 accumulates the taps
 * j, j + 8, ... with fused multiply-adds, then (l0+l4 + l2+l6) + (l1+l5 + l3+l7). */
static float dot_float_portable(const float *src, const float *filt, int alloc) {
  float l[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = 0; i < alloc; i += 8)
    for (int j = 0; j < 8; ++j) l[j] = fmaf(src[i + j], filt[i + j], l[j]);
  const float a0 = l[0] + l[4], a1 = l[1] + l[5], a2 = l[2] + l[6], a3 = l[3] + l[7];
  return (a0 + a2) + (a1 + a3);
}

#if defined(__x86_64__)
#include <immintrin.h>
__attribute__((target("avx2,fma"))) static float dot_float_fma3(const float *src, const float *filt,
                                                                int alloc) {
  __m256 acc = _mm256_setzero_ps();
  for (int i = 0; i < alloc; i += 8)
    acc = _mm256_fmadd_ps(_mm256_loadu_ps(src + i), _mm256_loadu_ps(filt + i), acc);
  __m128 a = _mm_add_ps(_mm256_castps256_ps128(acc), _mm256_extractf128_ps(acc, 1));
  __m128 b = _mm_add_ps(a, _mm_movehl_ps(a, a));
  return _mm_cvtss_f32(_mm_add_ss(b, _mm_shuffle_ps(b, b, 1)));
}

/* Q15: int32 accumulator that wraps like the library's (paddd) */
__attribute__((target("avx2"))) static int32_t dot_s16_avx2(const int16_t *src, const int16_t *filt,
                                                            int alloc) {
  __m256i acc = _mm256_setzero_si256();
  for (int i = 0; i < alloc; i += 16)
    acc = _mm256_add_epi32(acc, _mm256_madd_epi16(_mm256_loadu_si256((const __m256i *)(src + i)),
                                                  _mm256_loadu_si256((const __m256i *)(filt + i))));
  __m128i a = _mm_add_epi32(_mm256_castsi256_si128(acc), _mm256_extracti128_si256(acc, 1));
  a = _mm_add_epi32(a, _mm_shuffle_epi32(a, 0x4E));
  a = _mm_add_epi32(a, _mm_shuffle_epi32(a, 0xB1));
  return _mm_cvtsi128_si32(a);
}
#endif

static int32_t dot_s16_portable(const int16_t *src, const int16_t *filt, int alloc) {
  uint32_t v = 0;
  for (int i = 0; i < alloc; ++i) v += (uint32_t)((int32_t)src[i] * (int32_t)filt[i]);
  return (int32_t)v;
}

static int have_fma3(void) {
#if defined(__x86_64__)
  static int cached = -1; /* every thread computes the same value: relaxed atomics are all it needs */
  int c = __atomic_load_n(&cached, __ATOMIC_RELAXED);
  if (c < 0) {
    c = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma");
    __atomic_store_n(&cached, c, __ATOMIC_RELAXED);
  }
  return c;
#else
  return 0;
#endif
}

static inline int16_t clip16(long v) { return (int16_t)(v > 32767 ? 32767 : v < -32768 ? -32768 : v); }

/* Output n reads the taps that start at ext position w_n = w0 + floor(n * dst_incr /
 * (src_incr * phase_count)) (ext = taps reflected samples, then the input).  Before the flush
 * every n with w_n <= frames is produced; the flush appends (min(left, taps) + 1) / 2
 * reflected samples and the same rule applies once more. */
size_t bl_rs_out_frames(const bl_rs_plan *f, size_t frames, size_t *refl_out) {
  const uint64_t L = (uint64_t)f->taps, w0 = L - (L - 1) / 2;
  if (frames < L + 1) { /* the library waits for taps + 1 samples before it produces any */
    if (refl_out) *refl_out = 0;
    return 0;
  }
  const uint64_t A = (uint64_t)f->dst_incr, B = (uint64_t)f->src_incr * (uint64_t)f->phase_count;
  const uint64_t n1 = (((uint64_t)frames - w0 + 1) * B - 1) / A + 1;
  const uint64_t w1 = w0 + n1 * A / B;
  const uint64_t left = L + frames - w1;
  const uint64_t refl = ((left < L ? left : L) + 1) / 2;
  if (refl_out) *refl_out = (size_t)refl;
  return (size_t)((((uint64_t)frames + refl - w0 + 1) * B - 1) / A + 1);
}

/* One channel.  `ext` holds taps mirrored samples, the n input samples, and room for the
 * flush reflection (+ zeroed slack for the padded taps); returns the number of output samples
 * (written to out[0], out[stride], ...). */
static size_t rs_run(const bl_rs_plan *f, void *ext, int is_float, size_t n, int16_t *out, size_t stride,
                     size_t limit) {
  const int L = f->taps, A = f->alloc;
  const int fast = have_fma3();
  float *xf = (float *)ext;
  int16_t *xi = (int16_t *)ext;
  /* reflect about the first sample: ext[L - k] = ext[L + k] */
  for (int k = 1; k <= L; ++k) {
    if (is_float) xf[L - k] = xf[L + k];
    else xi[L - k] = xi[L + k];
  }
  size_t avail = (size_t)L + n, w = (size_t)(L - (L - 1) / 2), produced = 0;
  int index = 0, frac = 0;
  for (int pass = 0; pass < 2; ++pass) {
    while (w + (size_t)L <= avail) {
      if (produced >= limit) return limit + 1; /* the closed form and the stepping disagree */
      if (is_float) {
        const float *fr = f->fbank + (size_t)index * A;
        float v;
#if defined(__x86_64__)
        if (fast) v = dot_float_fma3(xf + w, fr, A);
        else
#endif
          v = dot_float_portable(xf + w, fr, A);
        out[produced * stride] = clip16(lrintf(v * 32768.0f));
      } else {
        const int16_t *ir = f->ibank + (size_t)index * A;
        uint32_t v;
#if defined(__x86_64__)
        if (fast) v = (uint32_t)dot_s16_avx2(xi + w, ir, A);
        else
#endif
          v = (uint32_t)dot_s16_portable(xi + w, ir, A);
        out[produced * stride] = clip16((int32_t)(v + (1u << 14)) >> 15);
      }
      ++produced;
      frac += f->dst_incr_mod;
      index += f->dst_incr_div;
      if (frac >= f->src_incr) { frac -= f->src_incr; ++index; }
      w += (size_t)(index / f->phase_count);
      index %= f->phase_count;
    }
    if (pass == 0) { /* flush: reflect about the end */
      const size_t left = avail > w ? avail - w : 0;
      const size_t refl = ((left < (size_t)L ? left : (size_t)L) + 1) / 2;
      for (size_t j = 0; j < refl; ++j) {
        if (is_float) xf[avail + j] = xf[avail - j - 1];
        else xi[avail + j] = xi[avail - j - 1];
      }
      avail += refl;
    }
  }
  return produced;
}

int bl_resample_to_stereo_s16(const void *in, int in_is_s32, size_t frames, int channels, int in_rate,
                              int out_rate, int16_t **out, size_t *out_frames) {
  *out = NULL;
  *out_frames = 0;
  if (!in || channels < 1 || channels > 2 || in_rate <= 0 || out_rate <= 0) return BL_UNEXPECTED;
  bl_rs_plan f;
  if (bl_rs_plan_build(&f, out_rate, in_rate, in_is_s32)) return BL_UNEXPECTED;
  const int L = f.taps;
  const size_t bound = bl_rs_out_frames(&f, frames, NULL);
  if (bound == 0) {
    bl_rs_plan_free(&f);
    return BL_UNEXPECTED;
  }
  const size_t ext_len = (size_t)L + frames + (size_t)L + 32;
  const size_t esz = in_is_s32 ? sizeof(float) : sizeof(int16_t);
  void *ext = calloc(ext_len, esz);
  if (!ext) { bl_rs_plan_free(&f); return BL_UNEXPECTED; }
  int16_t *o = (int16_t *)malloc(bound * 2 * sizeof(int16_t));
  if (!o) { free(ext); bl_rs_plan_free(&f); return BL_UNEXPECTED; }

  size_t produced = 0;
  for (int c = 0; c < 2; ++c) {
    if (channels == 1 && c == 1) { /* both output channels of an up-mixed mono source are equal */
      for (size_t i = 0; i < produced; ++i) o[2 * i + 1] = o[2 * i];
      break;
    }
    memset(ext, 0, ext_len * esz);
    if (in_is_s32) {
      const int32_t *p = (const int32_t *)in;
      float *x = (float *)ext + L;
      const float g = (float)M_SQRT1_2;
      for (size_t i = 0; i < frames; ++i) {
        float v;
        if (in_is_s32 == 2) { /* float source: the sample as it is (not-a-numbers count as silence) */
          memcpy(&v, &p[i * (size_t)channels + (size_t)c], sizeof v);
          if (!(v == v) || v > 4.0f || v < -4.0f) v = v > 0 ? 4.0f : (v < 0 ? -4.0f : 0.0f);
        } else {
          v = (float)p[i * (size_t)channels + (size_t)c] * (1.0f / 2147483648.0f);
        }
        if (channels == 1) v *= g;
        x[i] = v;
      }
    } else {
      const int16_t *p = (const int16_t *)in;
      int16_t *x = (int16_t *)ext + L;
      for (size_t i = 0; i < frames; ++i) {
        int32_t v = p[i * (size_t)channels + (size_t)c];
        if (channels == 1) v = (v * 23170 + 16384) >> 15; /* Q15 1/sqrt(2) */
        x[i] = (int16_t)v;
      }
    }
    produced = rs_run(&f, ext, in_is_s32, frames, o + c, 2, bound);
    if (produced != bound) { free(o); free(ext); bl_rs_plan_free(&f); return BL_UNEXPECTED; }
  }
  free(ext);
  bl_rs_plan_free(&f);
  *out = o;
  *out_frames = produced;
  return BL_OK;
}
