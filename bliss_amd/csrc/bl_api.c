/*
 * bl_api.c — the reference's C API (include/bliss.h) on top of the HIP path.
 *
 * Host code stays in C, as in the reference; every analysis result comes from
 * the kernels in bl_kernels.hip through the thin launch layer declared in
 * bl_device.h.  There is no CPU implementation of the analysis in this
 * library: without a usable HIP device the analysis entry points print a
 * message and return BL_UNEXPECTED (or the float conversion of it, as the
 * reference does for bl_distance_file, ref src/analyze.c:123-124).
 *
 * Three scalar helpers are plain host arithmetic, as in the reference:
 * bl_distance / bl_cosine_similarity of ONE pair (a 4-float expression; through
 * a 1 x 1 kernel launch and a blocking copy a call cost ~20-30 us, so the
 * reference's 10^8-call loop, SURVEY.md section 6, would have taken the better
 * part of an hour) and bl_rectangular_filter (a serial running sum).  Their
 * all-pairs / streaming forms — bl_amd_distance_matrix_*, the box filters inside
 * the envelope tail — stay on the GPU, and the tests hold the scalar functions
 * against those kernels bit for bit.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "bl_device.h"
#include "bliss.h"

/* ref src/helpers.c:15-23 */
void bl_initialize_song(struct bl_song *const song) {
  song->artist = NULL;
  song->title = NULL;
  song->album = NULL;
  song->tracknumber = NULL;
  song->sample_array = NULL;
  song->filename = NULL;
  song->genre = NULL;
}

/* ref src/helpers.c:3-13 */
void bl_free_song(struct bl_song *const song) {
  free(song->artist);
  free(song->title);
  free(song->album);
  free(song->tracknumber);
  free(song->sample_array);
  free(song->filename);
  free(song->genre);
  bl_initialize_song(song);
}

/* ref src/helpers.c:25-28 */
float bl_version(void) {
  printf("Using bliss analyzer version %0.1f.\n", BL_VERSION);
  return (float)BL_VERSION;
}

static int song_usable(struct bl_song const *const song) {
  return song && song->sample_array && song->nSamples >= 5120 &&
         (song->channels == 1 || song->channels == 2);
}

static int run_song(struct bl_song const *const song, int what, bl_amd_song_result *res) {
  if (!song_usable(song)) {
    fprintf(stderr, "bliss_amd: song not analysable (need >= 5120 s16 samples, 1 or 2 channels)\n");
    return BL_UNEXPECTED;
  }
  /* duration only enters the tempo score (ref src/tempo_atk_sort.c:283); a clip shorter
   * than one second (duration 0: division by zero in the reference) is rejected when the
   * envelope analysis is asked for, like in the batch entry points */
  uint64_t duration = song->duration;
  if (!(what & 4) && duration == 0) duration = 1;
  return bld_analyze_one_host((const int16_t *)song->sample_array, song->nSamples, song->channels,
                              duration, what, res);
}

/* ref include/bliss.h:200 / src/amplitude_sort.c:12-80 */
float bl_amplitude_sort(struct bl_song const *const song) {
  bl_amd_song_result r;
  if (run_song(song, 1, &r) != BL_OK) return (float)BL_UNEXPECTED;
  return r.v.amplitude;
}

/* ref include/bliss.h:217 / src/frequency_sort.c:20-140 */
float bl_frequency_sort(struct bl_song const *const song) {
  bl_amd_song_result r;
  if (run_song(song, 2, &r) != BL_OK) return (float)BL_UNEXPECTED;
  return r.v.frequency;
}

/* ref include/bliss.h:184-185 / src/tempo_atk_sort.c:42-296 */
void bl_envelope_sort(struct bl_song const *const song, struct envelope_result_s *result) {
  bl_amd_song_result r;
  if (run_song(song, 4, &r) != BL_OK) {
    result->tempo = (float)BL_UNEXPECTED;
    result->attack = (float)BL_UNEXPECTED;
    return;
  }
  result->tempo = r.v.tempo;
  result->attack = r.v.attack;
}

/* ref include/bliss.h:80-81 / src/analyze.c:33-86 */
int bl_analyze(char const *const filename, struct bl_song *current_song) {
  if (bl_audio_decode(filename, current_song) == BL_OK) {
    bl_amd_song_result r;
    if (run_song(current_song, 7, &r) != BL_OK || r.status != BL_OK) {
      fprintf(stderr, "Couldn't analyse song on the HIP device\n");
      return BL_UNEXPECTED;
    }
    current_song->force_vector = r.v;      /* ref :63-66 */
    current_song->force = r.force;         /* ref :68-72 */
    current_song->calm_or_loud = r.calm_or_loud; /* ref :73-79 */
    return current_song->calm_or_loud;
  }
  fprintf(stderr, "Couldn't decode song\n"); /* ref :83 */
  return BL_UNEXPECTED;
}

/* ref include/bliss.h:116-118 / src/analyze.c:88-103: every operand f32, sums left to
 * right, sqrt correctly rounded (this file is compiled with -ffp-contract=off, like the
 * reference's -std=c99 build: no fused multiply-add).  Same bits as k_pairwise<false>. */
float bl_distance(struct force_vector_s v_song1, struct force_vector_s v_song2) {
  const float d0 = v_song1.tempo - v_song2.tempo;
  const float d1 = v_song1.amplitude - v_song2.amplitude;
  const float d2 = v_song1.frequency - v_song2.frequency;
  const float d3 = v_song1.attack - v_song2.attack;
  float s = d0 * d0;
  s = s + d1 * d1;
  s = s + d2 * d2;
  s = s + d3 * d3;
  return sqrtf(s);
}

/* ref include/bliss.h:151-153 / src/analyze.c:127-143: f32 dot product and squared norms,
 * double sqrt, product and quotient.  Same bits as k_pairwise<true>. */
float bl_cosine_similarity(struct force_vector_s v_song1, struct force_vector_s v_song2) {
  const float a[4] = {v_song1.tempo, v_song1.amplitude, v_song1.frequency, v_song1.attack};
  const float b[4] = {v_song2.tempo, v_song2.amplitude, v_song2.frequency, v_song2.attack};
  float dot = a[0] * b[0], na = a[0] * a[0], nb = b[0] * b[0];
  for (int k = 1; k < 4; ++k) {
    dot = dot + a[k] * b[k];
    na = na + a[k] * a[k];
    nb = nb + b[k] * b[k];
  }
  return (float)((double)dot / (sqrt((double)na) * sqrt((double)nb)));
}

/* ref include/bliss.h:99-103 / src/analyze.c:105-125 */
float bl_distance_file(char const *const filename1, char const *const filename2,
                       struct bl_song *song1, struct bl_song *song2) {
  if ((bl_analyze(filename1, song1) != BL_UNEXPECTED) &&
      (bl_analyze(filename2, song2) != BL_UNEXPECTED))
    return bl_distance(song1->force_vector, song2->force_vector);
  return BL_UNEXPECTED;
}

/* ref include/bliss.h:136-140 / src/analyze.c:145-167 */
float bl_cosine_similarity_file(char const *const filename1, char const *const filename2,
                                struct bl_song *song1, struct bl_song *song2) {
  if ((bl_analyze(filename1, song1) != BL_UNEXPECTED) &&
      (bl_analyze(filename2, song2) != BL_UNEXPECTED))
    return bl_cosine_similarity(song1->force_vector, song2->force_vector);
  return BL_UNEXPECTED;
}

/* ref include/bliss.h:270 / src/helpers.c:30-37 */
int bl_mean(int16_t *sample_array, int nSamples) {
  int mean = 0;
  if (bld_mean_variance_host(sample_array, nSamples, 0, 0, &mean, NULL) != BL_OK) return BL_UNEXPECTED;
  return mean;
}

/* ref include/bliss.h:278 / src/helpers.c:39-49 */
int bl_variance(int16_t *sample_array, int nSamples, int mean) {
  int var = 0;
  if (bld_mean_variance_host(sample_array, nSamples, 1, mean, NULL, &var) != BL_OK) return BL_UNEXPECTED;
  return var;
}

/* ref include/bliss.h:289-290 / src/tempo_atk_sort.c:19-40: running sum of `smooth_width`
 * inputs written at the window centre, the last window ADDED to out[n - half], then every
 * cell divided by the width — cells the loop never writes keep their previous contents / width.
 * The hot path uses the streaming form of the same arithmetic (bl_box19 in bl_tail.h). */
void bl_rectangular_filter(double *sample_array_out, double *sample_array_in, int nSamples,
                           int smooth_width) {
  if (!sample_array_out || !sample_array_in || smooth_width <= 0 || smooth_width > nSamples) {
    fprintf(stderr, "bliss_amd: bl_rectangular_filter: need 0 < smooth_width <= nSamples\n");
    return;
  }
  const double *lag = sample_array_in, *lead = sample_array_in + smooth_width;
  const double *const end = sample_array_in + nSamples;
  double run = 0;
  for (const double *p = lag; p < lead; ++p) run += *p;
  const int half = (int)round(smooth_width / 2.);
  double *centre = sample_array_out + half - 1;
  while (lead < end) { /* slide: drop the oldest input, then take the next one */
    *centre++ = run;
    run -= *lag++;
    run += *lead++;
  }
  /* lag == &in[n - width]: the last window is added to whatever out[n - half] held.  For an odd
   * width that is the cell `centre` has reached; for an even one it is the cell after it (the
   * reference names the index, ref :32-33), so the index is named here too. */
  double *const last = sample_array_out + (nSamples - half);
  for (; lag < end; ++lag) *last += *lag;
  for (int k = 0; k < nSamples; ++k) sample_array_out[k] /= smooth_width;
}
