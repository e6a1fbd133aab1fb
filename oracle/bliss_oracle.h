/*
 * bliss_oracle.h — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C99) of the arithmetic of the reference's per-song
 * analysis path, used only by tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py as the checker for the HIP path.  Nothing under
 * bliss_amd/ may include, link or call this.
 *
 * Pinning: orc_* reproduce the five goldens of the reference's
 * tests/test_analyze.c:30-35 on audio/song.flac (tests/test_oracle_golden.py).
 * The reference itself is UNBUILDABLE in this image (needs libavformat,
 * libavcodec/avfft, libswresample and FFTW3 headers+libs, none installed;
 * CMakeLists.txt:5-9), so there is no oracle/_ref.  The two third-party FFTs
 * (libavcodec av_rdft_*, FFTW3 r2c; neither vendored nor version-pinned by the
 * reference) are restated as plain DFT-equivalent FFTs of the same precision
 * (f32 / f64) — see orc_fft.c.
 */
#ifndef BLISS_ORACLE_H_
#define BLISS_ORACLE_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* every intermediate the parity tests compare, one record per song */
typedef struct orc_result {
  /* force vector, order of ref include/bliss.h:26-31 */
  float tempo, amplitude, frequency, attack;
  float force;      /* ref src/analyze.c:68-72 */
  int calm_or_loud; /* ref src/analyze.c:73-79 */
  /* integer quantities that must be bit-exact */
  int start, end;      /* ref src/amplitude_sort.c:26-31 */
  int mean, variance;  /* ref src/helpers.c:30-49 */
  int n_frames;        /* ref src/frequency_sort.c:50 */
  int nb_frames;       /* ref src/tempo_atk_sort.c:63-64 */
  int n_windows;       /* number of FIR+FFT windows actually run (:120) */
  int beat;            /* ref src/tempo_atk_sort.c:277-280 */
  /* diagnostics */
  double atk_sum;          /* ref src/tempo_atk_sort.c:246-248 */
  double min_peak_margin;  /* min | |ss[j]-ss[j±1]| - eps | over all decisions */
  float hist_integral;     /* ref src/amplitude_sort.c:69-71 */
  float freq_peak;         /* ref src/frequency_sort.c:101 */
} orc_result;

int orc_mean(const int16_t *pcm, int n);
int orc_variance(const int16_t *pcm, int n, int mean);
float orc_amplitude(const int16_t *pcm, int n, orc_result *r);
float orc_frequency(const int16_t *pcm, int n, int channels, orc_result *r);
/* energies: optional out array of nb_frames floats (window energies, f32) */
void orc_envelope(const int16_t *pcm, int n, uint64_t duration, orc_result *r,
                  float *energies);
void orc_rect_filter(double *out, const double *in, int n, int width);
/* whole per-song path after decode: ref src/analyze.c:40-80 */
int orc_analyze_pcm(const int16_t *pcm, int n, int channels, uint64_t duration,
                    orc_result *r);
float orc_distance(const float a[4], const float b[4]);
float orc_cosine(const float a[4], const float b[4]);
void orc_distance_matrix(const float *vecs, int n, float *out);
void orc_cosine_matrix(const float *vecs, int n, float *out);

/* FFTs (orc_fft.c) */
void orc_rdft512_f32(float *x);                 /* in-place, FFmpeg packed layout */
void orc_r2c512_f64(const double *in, double *re, double *im); /* k = 0..256 */
/* which implementation the two above run: 0 the defaults — f64: the packed radix-2 of orc_fft.c; f32: libavcodec's
 * split-radix operation order, orc_fft_lavc.c (every committed golden since round 6) —, 1 recursive radix-4 on the
 * unpacked complex input, 2 the defining sum in extended precision (orc_fft_alt.c), 3 = the f32 default by name,
 * 4 the f32 packed radix-2 of orc_fft.c (the f32 default until round 6); 3 and 4 leave the f64 transform at 0's */
void orc_set_fft_variant(int v);
int orc_fft_variant(void);
void orc_alt_r2c512_f64(int variant, const double *in, double *re, double *im);
void orc_alt_rdft512_f32(int variant, float *x);
void orc_lavc_rdft512_f32(float *x);            /* orc_fft_lavc.c: libavcodec fft_template.c / rdft.c operation order */

/* integer-only synthetic PCM (orc_synth.c); identical bytes on every machine */
int16_t orc_synth_sample(uint32_t seed, uint32_t rate, uint32_t channels, uint32_t i);
void orc_synth_fill(int16_t *pcm, uint32_t n, uint32_t seed, uint32_t rate,
                    uint32_t channels);

#ifdef __cplusplus
}
#endif
#endif
