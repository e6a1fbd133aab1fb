/*
 * bliss.h — drop-in public header of the MI355X-native bliss hot path.
 *
 * Every declaration below replaces the declaration of the same name in the
 * reference's include/bliss.h (cited per item as "ref include/bliss.h:LINE").
 * Struct layouts, return codes and prototypes are identical, so a caller
 * compiled against the reference header links against libbliss_amd.so
 * unchanged.  Unlike the reference header this one does not pull in
 * <libavformat/avformat.h> / <libavutil/md5.h> (ref include/bliss.h:5-6):
 * nothing in the API surface needs them.
 *
 * The analysis itself (bl_amplitude_sort, bl_frequency_sort, bl_envelope_sort,
 * bl_analyze's force assembly, bl_distance*) runs as hand-written HIP kernels
 * on gfx950; there is no CPU fallback — entry points fail with BL_UNEXPECTED
 * (and a message on stderr) when no HIP device is usable.
 */
#ifndef BL_BLISS_H_
#define BL_BLISS_H_

/* what callers of the reference header get transitively from libavformat's headers and use
 * without including it themselves (ref examples/analyze.c:40 PRId64, examples/detect-gapless.c:28,38
 * PRId16 / fabs): kept, so that such sources compile unchanged */
#include <inttypes.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#ifdef __cplusplus
extern "C" {
#endif
/* the library itself is built with -fvisibility=hidden: what these headers declare is its whole export list */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ref include/bliss.h:12 */
#define BL_VERSION 1.2

/* ref include/bliss.h:20-24 */
#define BL_LOUD 0
#define BL_CALM 1
#define BL_UNKNOWN 2
#define BL_UNEXPECTED -2
#define BL_OK 0

/* ref include/bliss.h:26-31 — 16 bytes, field order tempo/amplitude/frequency/attack */
struct force_vector_s {
  float tempo;
  float amplitude;
  float frequency;
  float attack;
};

/* ref include/bliss.h:34-37 */
struct envelope_result_s {
  float tempo;
  float attack;
};

/* ref include/bliss.h:39-47 — kept only so that sizeof/ABI of the header match */
struct thread_result_s {
  struct bl_song const *const song;
  float result;
};

struct thread_envelope_result_s {
  struct bl_song const *const song;
  struct envelope_result_s *results;
};

/* ref include/bliss.h:49-67 — 120 bytes on x86-64 (offsets pinned by tests) */
struct bl_song {
  float force;
  struct force_vector_s force_vector;
  int8_t *sample_array;
  int channels;
  int nSamples;
  int sample_rate;
  int bitrate;
  int nb_bytes_per_sample;
  int calm_or_loud;
  int resampled;
  uint64_t duration;
  char *filename;
  char *artist;
  char *title;
  char *album;
  char *tracknumber;
  char *genre;
};

/* ref include/bliss.h:80-81 / src/analyze.c:33-86 */
int bl_analyze(char const *const filename, struct bl_song *current_song);

/* ref include/bliss.h:99-103 / src/analyze.c:105-125 */
float bl_distance_file(char const *const filename1, char const *const filename2,
                       struct bl_song *song1, struct bl_song *song2);

/* ref include/bliss.h:116-118 / src/analyze.c:88-103 */
float bl_distance(struct force_vector_s v_song1, struct force_vector_s v_song2);

/* ref include/bliss.h:136-140 / src/analyze.c:145-167 */
float bl_cosine_similarity_file(char const *const filename1,
                                char const *const filename2,
                                struct bl_song *song1, struct bl_song *song2);

/* ref include/bliss.h:151-153 / src/analyze.c:127-143 */
float bl_cosine_similarity(struct force_vector_s v_song1,
                           struct force_vector_s v_song2);

/* ref include/bliss.h:184-185 / src/tempo_atk_sort.c:42-296 */
void bl_envelope_sort(struct bl_song const *const song,
                      struct envelope_result_s *result);

/* ref include/bliss.h:200 / src/amplitude_sort.c:12-80 */
float bl_amplitude_sort(struct bl_song const *const song);

/* ref include/bliss.h:217 / src/frequency_sort.c:20-140 */
float bl_frequency_sort(struct bl_song const *const song);

/* ref include/bliss.h:234-235 / src/decode.c:27-213 (host ingest) */
int bl_audio_decode(char const *const filename, struct bl_song *const song);

/* ref include/bliss.h:247 / src/helpers.c:3-13 */
void bl_free_song(struct bl_song *const song);

/* ref include/bliss.h:254 / src/helpers.c:25-28 */
float bl_version(void);

/* ref include/bliss.h:262 / src/helpers.c:15-23 */
void bl_initialize_song(struct bl_song *const song);

/* ref include/bliss.h:270 / src/helpers.c:30-37 */
int bl_mean(int16_t *sample_array, int nSamples);

/* ref include/bliss.h:278 / src/helpers.c:39-49 */
int bl_variance(int16_t *sample_array, int nSamples, int mean);

/* ref include/bliss.h:289-290 / src/tempo_atk_sort.c:19-40 */
void bl_rectangular_filter(double *sample_array_out, double *sample_array_in,
                           int nSamples, int smooth_width);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif

#endif /* BL_BLISS_H_ */
