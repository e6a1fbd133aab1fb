/*
 * bliss_amd.h — batch / device-resident C-ABI of libbliss_amd.so.
 *
 * The reference has no batch API: its only corpus loop is the sequential
 * `for file: bl_song(file)` of python/examples/make_m3u_playlist.py:51-72 and the
 * per-pair bl_distance of src/analyze.c:88-103.  These entry points are the
 * batched form of exactly that path (bl_analyze's analyzers after decode,
 * ref src/analyze.c:40-80, and bl_distance / bl_cosine_similarity over all
 * pairs) for callers that hold many decoded songs.  Plain pointers and sizes
 * only; `stream` is a hipStream_t passed as void* (NULL = default stream).
 * Pointers named d_* are device pointers, h_* host pointers.
 *
 * All functions return BL_OK (0) or BL_UNEXPECTED (-2); there is no CPU
 * fallback: without a usable HIP device they fail and print to stderr.
 */
#ifndef BLISS_AMD_H_
#define BLISS_AMD_H_

#include <stddef.h>
#include <stdint.h>
#include "bliss.h"

#ifdef __cplusplus
extern "C" {
#endif

/* One decoded song inside a PCM arena: what the analyzers read from
 * struct bl_song (ref include/bliss.h:49-67): sample_array, nSamples,
 * channels, duration. */
typedef struct bl_amd_song_desc {
  uint64_t pcm_offset; /* int16 elements from the arena base; multiple of 8 */
  int32_t n_samples;   /* interleaved sample count (bl_song.nSamples), >= 5120 */
  int32_t channels;    /* 1 or 2 */
  uint64_t duration;   /* whole seconds (bl_song.duration), > 0 */
} bl_amd_song_desc;

/* Per-song output: the force vector plus every integer intermediate the
 * reference computes on the way (bit-exact quantities of SURVEY.md §8a). */
typedef struct bl_amd_song_result {
  struct force_vector_s v; /* tempo, amplitude, frequency, attack */
  float force;             /* ref src/analyze.c:68-72 */
  int32_t calm_or_loud;    /* BL_LOUD / BL_CALM / BL_UNKNOWN, ref :73-79 */
  int32_t status;          /* BL_OK, or BL_UNEXPECTED for an input the
                              reference leaves undefined (all-zero PCM, ...) */
  int32_t start, end;      /* ref src/amplitude_sort.c:26-31 */
  int32_t mean, variance;  /* ref src/helpers.c:30-49 */
  int32_t n_frames;        /* ref src/frequency_sort.c:50 */
  int32_t nb_frames;       /* ref src/tempo_atk_sort.c:63-64 */
  int32_t n_windows;       /* FIR+FFT windows run, ref :120 */
  int32_t beat;            /* ref src/tempo_atk_sort.c:277-280 */
  float hist_integral;     /* ref src/amplitude_sort.c:69-71 */
  float freq_peak;         /* ref src/frequency_sort.c:101 */
  double atk_sum;          /* ref src/tempo_atk_sort.c:246-248 */
} bl_amd_song_result;

/* Select / initialise the HIP device used by this process (default 0). */
int bl_amd_init(int device);
/* Number of visible HIP devices (0 if none / no runtime). */
int bl_amd_device_count(void);

/* Analyse n_songs songs whose PCM already sits in device memory.
 * d_pcm: arena base; h_desc: host array of n_songs descriptors;
 * d_results: device array of n_songs results (written asynchronously on
 * `stream`).  Scratch comes from an internal, growing device workspace that all calls
 * share: batches enqueued on different streams are ordered on the device (each waits
 * for the previous one to finish with the workspace), they do not overlap. */
int bl_amd_analyze_batch_device(const int16_t *d_pcm, const bl_amd_song_desc *h_desc,
                                int n_songs, bl_amd_song_result *d_results, void *stream);

/* Same from host memory: stages PCM through pinned buffers with
 * hipMemcpyAsync overlapped against the kernels of the previous wave of
 * songs, then copies the results back.  Blocking. */
int bl_amd_analyze_batch_host(const int16_t *const *h_pcm, const int32_t *n_samples,
                              const int32_t *channels, const uint64_t *duration, int n_songs,
                              bl_amd_song_result *h_results);

/* Rows [row_begin, row_begin + n_rows) of the N x N bl_distance matrix
 * (ref src/analyze.c:96-100 for every pair).  d_out: n_rows * n floats. */
int bl_amd_distance_matrix_device(const struct force_vector_s *d_vecs, int n, int row_begin,
                                  int n_rows, float *d_out, void *stream);
/* Same for bl_cosine_similarity (ref src/analyze.c:135-140). */
int bl_amd_cosine_matrix_device(const struct force_vector_s *d_vecs, int n, int row_begin,
                                int n_rows, float *d_out, void *stream);
/* Host-pointer conveniences (blocking). */
int bl_amd_distance_matrix_host(const struct force_vector_s *h_vecs, int n, float *h_out);
int bl_amd_cosine_matrix_host(const struct force_vector_s *h_vecs, int n, float *h_out);

/* Seeded playlist (ref python/examples/make_m3u_playlist.py:62-72): d_dist[j] =
 * bl_distance(vecs[seed_index], vecs[j]) and d_order = the song indices by increasing
 * distance (stable: ties by index).  d_order: n int32, d_dist: n floats. */
int bl_amd_playlist_device(const struct force_vector_s *d_vecs, int n, int seed_index,
                           int32_t *d_order, float *d_dist, void *stream);
int bl_amd_playlist_host(const struct force_vector_s *h_vecs, int n, int seed_index,
                         int32_t *h_order, float *h_dist /* may be NULL */);

/* Integer-only synthetic PCM (the benchmark corpus of BASELINE.json),
 * generated in place on the device: song i = seed_base + i, written at
 * h_desc[i].pcm_offset.  Byte-identical to oracle/orc_synth.c. */
int bl_amd_synth_pcm_device(int16_t *d_pcm, const bl_amd_song_desc *h_desc, int n_songs,
                            uint32_t seed_base, uint32_t sample_rate, void *stream);

/* Per-kernel device time, measured with hipEvents on the launch stream while
 * profiling is on (bench.py's roofline leg).  name is one of "pcm_scan",
 * "amp_finish", "freq_frames", "freq_finish", "env_windows", "env_tail",
 * "distance"; returns accumulated milliseconds and the launch count since the
 * last reset, or -1 for an unknown name. */
void bl_amd_profile(int enable);
void bl_amd_profile_reset(void);
double bl_amd_profile_ms(const char *name, int *launches);

/* Diagnostic: per-window envelope energies (the reference's filtered_array,
 * ref src/tempo_atk_sort.c:150) of the most recent batch, songs concatenated with
 * nb_frames slots each (the last two of a song are never written).  Copies up to
 * max_elems floats to h_out; returns the number copied, 0 if none, -1 on error. */
long long bl_amd_last_energies(float *h_out, long long max_elems);

/* Releases the workspace, streams and pinned staging buffers. */
void bl_amd_shutdown(void);

#ifdef __cplusplus
}
#endif
#endif /* BLISS_AMD_H_ */
