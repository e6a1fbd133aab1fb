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
/* the library itself is built with -fvisibility=hidden: what these headers declare is its whole export list */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
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

/* Select the HIP device the plain entry points below use ON THE CALLING THREAD and create
 * its default context (workspace, internal streams).  The first call of the process also
 * sets the process default, which threads that never called bl_amd_init use (device 0 if
 * nobody did).  One process can drive several GPUs: one thread per device. */
int bl_amd_init(int device);
/* Number of visible HIP devices (0 if none / no runtime). */
int bl_amd_device_count(void);

/* Explicit contexts: one device, an own scratch workspace, own internal streams and pinned
 * staging.  Calls on one context are ordered; different contexts — also on the same device —
 * are independent and may be driven from different host threads concurrently. */
typedef struct bl_amd_ctx bl_amd_ctx;
int bl_amd_ctx_create(int device, bl_amd_ctx **out);
void bl_amd_ctx_destroy(bl_amd_ctx *ctx);
int bl_amd_ctx_device(const bl_amd_ctx *ctx);

/* Analyse n_songs songs whose PCM already sits in device memory.
 * d_pcm: arena base; h_desc: host array of n_songs descriptors (copied before the call
 * returns); d_results: device array of n_songs results, written asynchronously on `stream`.
 * The call only enqueues work (descriptor upload from pinned memory, kernels); it does not
 * wait for the device unless the workspace has to grow or four earlier batches of the
 * context are still in flight.  Scratch comes from the context's growing workspace: batches
 * of one context enqueued on different streams are ordered on the device (each waits for the
 * previous one to finish with the workspace); batches of different contexts overlap. */
int bl_amd_analyze_batch_device(const int16_t *d_pcm, const bl_amd_song_desc *h_desc,
                                int n_songs, bl_amd_song_result *d_results, void *stream);
int bl_amd_ctx_analyze_batch_device(bl_amd_ctx *ctx, const int16_t *d_pcm,
                                    const bl_amd_song_desc *h_desc, int n_songs,
                                    bl_amd_song_result *d_results, void *stream);

/* Same from host memory: stages PCM through pinned buffers with
 * hipMemcpyAsync overlapped against the kernels of the previous wave of
 * songs, then copies the results back.  Blocking. */
int bl_amd_analyze_batch_host(const int16_t *const *h_pcm, const int32_t *n_samples,
                              const int32_t *channels, const uint64_t *duration, int n_songs,
                              bl_amd_song_result *h_results);
int bl_amd_ctx_analyze_batch_host(bl_amd_ctx *ctx, const int16_t *const *h_pcm,
                                  const int32_t *n_samples, const int32_t *channels,
                                  const uint64_t *duration, int n_songs,
                                  bl_amd_song_result *h_results);
/* The reference's corpus loop — `for file: bl_analyze(file, &song)` (ref python/examples/
 * make_m3u_playlist.py:51-72, examples/analyze.c:17) — as one call.  Files are decoded
 * (bl_audio_decode) on n_threads host threads (0 = one per hardware thread, at most 32) that run a
 * bounded number of files ahead; the decoded songs go to the GPU in file order, wave by wave,
 * through the pinned-staging path of bl_amd_analyze_batch_host, so decoding, transfer and
 * analysis overlap.  songs[i] (caller-owned, uninitialised is fine) is filled exactly as
 * bl_analyze(filenames[i], &songs[i]) fills it — release each with bl_free_song; with
 * keep_pcm == 0 the sample_array is freed (and NULL) once the song has been analysed.  With keep_pcm != 0 every decoded
 * sample_array stays allocated until the caller frees it: the library bounds its decoders' read-ahead (3 GiB of PCM
 * not yet analysed), not the total, which is the caller's — a corpus that does not fit in host memory has to be
 * analysed with keep_pcm == 0 or in slices.
 * codes (optional, n_files ints) receives what bl_analyze would have returned for the file:
 * BL_LOUD / BL_CALM / BL_UNKNOWN or BL_UNEXPECTED.  Returns the number of files analysed, or
 * BL_UNEXPECTED if the device path itself failed.  Blocking. */
int bl_amd_analyze_files(const char *const *filenames, int n_files, struct bl_song *songs, int *codes,
                         int n_threads, int keep_pcm);

/* How host buffers reach the device: BL_AMD_HOST_STAGED copies them into the library's pinned
 * double buffers on several host threads (default); BL_AMD_HOST_REGISTERED pins the caller's
 * buffers in place with hipHostRegister for the duration of the call (free() stays valid,
 * SURVEY.md section 8b) and copies each song straight into the arena.  Also settable with
 * the environment variable BL_AMD_HOST_MODE=staged|registered. */
#define BL_AMD_HOST_STAGED 0
#define BL_AMD_HOST_REGISTERED 1
int bl_amd_set_host_transfer(int mode);

/* 32-bit sources (BASELINE configs[4] "s16/s32"): h_pcm[i] holds n_samples[i] interleaved
 * int32 samples; they reach the hot path as s16 through an arithmetic >> 16 — the same-rate
 * S32 -> S16 conversion the reference gets from libswresample (ref src/decode.c:323-346,
 * 388-392; third-party arithmetic, parity unpinned).  The narrowing happens while the songs
 * are staged, so the PCIe link only carries s16. */
int bl_amd_analyze_batch_host_s32(const int32_t *const *h_pcm, const int32_t *n_samples,
                                  const int32_t *channels, const uint64_t *duration, int n_songs,
                                  bl_amd_song_result *h_results);
/* Host batches at another sample rate (a 44.1 or 48 kHz collection decoded by the caller):
 * every song is at `sample_rate` Hz, int16 or (pcm_is_s32) int32 left-justified, n_samples[i]
 * interleaved samples at that rate.  Each wave of songs is converted on the device between its
 * transfer and its analysis (bl_amd_resample_batch_device's arithmetic); the results describe
 * the converted songs (nb_frames etc. at 22 050 Hz stereo).  sample_rate == 22 050 is the plain
 * host batch.  Blocking. */
int bl_amd_analyze_batch_host_rate(const void *const *h_pcm, int pcm_is_s32, const int32_t *n_samples,
                                   const int32_t *channels, const uint64_t *duration, int n_songs,
                                   int sample_rate, bl_amd_song_result *h_results);
/* The same narrowing for a device-resident int32 buffer: d_out[i] = (int16)(d_in[i] >> 16). */
int bl_amd_narrow_s32_device(const int32_t *d_in, int16_t *d_out, size_t n, void *stream);

/* Rate conversion to the analyzers' 22 050 Hz stereo s16 — the arithmetic bl_audio_decode()
 * applies to a file at another rate (a restatement of libswresample's default resampler that
 * reproduces the digests of ref tests/test_decode.c:35-36,55-56), for
 * callers that bring their own decoder.  PARITY: only the path for sources wider than 16 bits is
 * pinned on the reference's digests; the 16-bit (Q15) path — what a 44.1 kHz s16 collection goes
 * through — is parity unpinned (s16 path): no reference vector covers its rounding, and every
 * throughput figure quoted for it is a figure for this restatement.
 * `in`: interleaved frames of 1 or 2 channels at in_rate
 * Hz, int16, or (in_is_s32 = 1) int32 left-justified, or — host form only — (in_is_s32 = 2)
 * float with full scale +-1.  A mono source comes out as two equal
 * channels at gain 1/sqrt(2), as the reference's out layout does.
 *   bl_amd_resample_out_frames: output frames for `frames` of input (0: shorter than the filter);
 *   bl_amd_resample_host: *out is malloc'd (free() it), 2 * *out_frames int16;
 *   bl_amd_resample_batch_device: songs resident in HBM, one call for the batch; song i reads
 *     h_desc[i].frames frames at d_in + in_offset (elements of the input type) and writes
 *     2 * bl_amd_resample_out_frames(frames, in_rate) int16 at d_out + out_offset, which is what
 *     bl_amd_analyze_batch_device takes (pcm_offset = out_offset, n_samples = 2 * out frames).
 *     Asynchronous on `stream`; bit-identical to the host form. */
typedef struct bl_amd_resample_desc {
  uint64_t in_offset;  /* elements from d_in; even for stereo */
  uint64_t out_offset; /* int16 elements from d_out; even (multiple of 8 to feed the analysis) */
  int32_t frames;      /* input frames */
  int32_t channels;    /* 1 or 2 */
} bl_amd_resample_desc;
size_t bl_amd_resample_out_frames(size_t frames, int in_rate);
int bl_amd_resample_host(const void *in, int in_is_s32, size_t frames, int channels, int in_rate,
                         int16_t **out, size_t *out_frames);
int bl_amd_resample_batch_device(const void *d_in, int in_is_s32, const bl_amd_resample_desc *h_desc,
                                 int n_songs, int in_rate, int16_t *d_out, void *stream);
int bl_amd_ctx_resample_batch_device(bl_amd_ctx *ctx, const void *d_in, int in_is_s32,
                                     const bl_amd_resample_desc *h_desc, int n_songs, int in_rate,
                                     int16_t *d_out, void *stream);

/* Batch-of-songs mode across the GPUs of one node (BASELINE configs[2]): the corpus is
 * sharded by song over the ranks listed in `devices` (one host thread and one context per
 * rank: contiguous blocks for equal lengths, longest-first greedy by sample count
 * otherwise), each rank analyses its shard through the host-batch path, the 16-byte force
 * vectors are all-gathered (flags: BL_AMD_MULTI_GATHER_RCCL = ncclAllGather over xGMI,
 * librccl loaded on first use; BL_AMD_MULTI_GATHER_PEER = direct peer copies, which also
 * permits several ranks on one device) and rank r computes rows [r N / W, (r+1) N / W) of
 * the N x N bl_distance matrix in the caller's song order.  h_results: n_songs records in
 * caller order.  h_matrix: NULL, or n_songs * n_songs floats that receive the row blocks.
 * Blocking. */
#define BL_AMD_MULTI_GATHER_RCCL 0
#define BL_AMD_MULTI_GATHER_PEER 1
int bl_amd_analyze_corpus_multi(const int16_t *const *h_pcm, const int32_t *n_samples,
                                const int32_t *channels, const uint64_t *duration, int n_songs,
                                const int *devices, int n_devices, int flags,
                                bl_amd_song_result *h_results, float *h_matrix);

/* The same for a corpus that is already RESIDENT in the GPUs' memory (configs[2] proper: 8 192
 * three-minute songs are 260 GB per GPU — they are decoded, converted or generated into each
 * GPU's HBM in waves, never held in host memory at once).  One bl_amd_shard per rank: the arena
 * on that rank's device and its songs; the corpus order is shard-major (all songs of shard 0,
 * then shard 1, ...).  Every rank analyses its arena where it lies (a host thread and a context
 * per rank), the force vectors are all-gathered as above, and rank r computes the rows of its
 * own songs against all N.  d_results (optional, on the shard's device): the shard's records;
 * d_rows (optional, on the shard's device): n_songs x N floats, the shard's row block, which
 * stays in HBM.  h_results (optional): N records in corpus order.  h_matrix (optional): N x N
 * floats.  Blocking.  The reference's corpus loop this replaces: python/examples/
 * make_m3u_playlist.py:51-72 (analyse every file, then distances from the vectors). */
typedef struct bl_amd_shard {
  int32_t device;                  /* HIP device the arena lives on */
  int32_t n_songs;
  const int16_t *d_pcm;            /* arena base, on `device` */
  const bl_amd_song_desc *h_desc;  /* host array of n_songs descriptors */
  bl_amd_song_result *d_results;   /* NULL, or n_songs records on `device` */
  float *d_rows;                   /* NULL, or n_songs * N floats on `device` */
} bl_amd_shard;
int bl_amd_analyze_corpus_multi_device(const bl_amd_shard *shards, int n_shards, int flags,
                                       bl_amd_song_result *h_results, float *h_matrix);

/* bl_audio_decode() follows the reference in always presenting 22 050 Hz PCM to the
 * analyzers (ref src/decode.c:7-9,317-346): a file at another rate, or wider than 16 bits at
 * another rate, goes through a restatement of libswresample's default converter and comes out
 * as 22 050 Hz stereo s16 (resampled = 1).  allow != 0 (also BL_AMD_ALLOW_NATIVE_RATE=1)
 * switches that off: the file is handed over at its own rate, narrowed to s16 only, and its
 * force vector is not comparable with the reference's. */
void bl_amd_decode_allow_native_rate(int allow);
/* Integrity check of the FLAC decoder behind bl_audio_decode: decodes `filename` and compares
 * the MD5 of the decoded samples at their native width (before the narrowing to s16) with the
 * signature of the unencoded audio in the file's STREAMINFO block.  1 = match, 0 = mismatch,
 * BL_UNEXPECTED = not decodable.  computed / stored (16 bytes each) may be NULL. */
int bl_amd_flac_verify(const char *filename, uint8_t computed[16], uint8_t stored[16]);

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

/* Self-test of the square root inside the bl_distance kernels (bliss_amd/csrc/bl_sqrt.h): runs it
 * over every f32 bit pattern on the device and compares with the correctly rounded root,
 * (float)sqrt((double)s).  counts[0] = values in the domain of the five-instruction form,
 * counts[1] = its mismatches, counts[2] = mismatches of the fallback (the compiler's correctly
 * rounded sqrtf) over all 2^32 patterns.  Both must be 0 (tests/test_gpu_parity.py). */
int bl_amd_selftest_sqrt(uint64_t counts[3]);
/* Same for the cosine matrix's guarded quotient (bl_cos.h): at least `triples` pseudo-random (dot, |a|^2, |b|^2)
 * on the device against the plain expression of ref src/analyze.c:135-140.  counts: [0] triples tried, [1] taken by
 * the fast path, [2] fast results that differ from the plain expression (must be 0), [3] largest difference of
 * the two double quotients in ulp (provable bound < 6, guard 16), [4] triples within 64 ulp of a float rounding boundary,
 * [5] of those, how many an unguarded fast path would have got wrong. */
int bl_amd_selftest_cos(uint64_t counts[6], uint64_t triples);

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

/* Arithmetic of the envelope kernel's 17-tap FIR (ref src/tempo_atk_sort.c:123-138):
 *   0  the reference's order, every product and sum rounded separately: window energies
 *      bit-identical to the reference arithmetic;
 *   1  the same sum with the products folded in by fused multiply-adds;
 *   2  (default) as 1, and the normalisation of ref :109-114 folded into the taps.
 * 1 and 2 differ from 0 by a few 1e-16 of an output's largest partial sum — what a different
 * FFT library behind it already does — and are 9 % / 12 % faster.  Measured on 2.3 billion windows
 * of 38 912 songs: 10 f32 window energies move, by one ulp, no integer and no feature changes;
 * expected `beat` changes per three-minute song 4e-10.  mode -1 = follow the
 * environment variable BL_AMD_FIR_FUSED, else the default.  Process-wide. */
int bl_amd_set_fir_mode(int mode);
int bl_amd_fir_mode(void);

/* Per-kernel device time, measured with hipEvents on the launch stream while
 * profiling is on (bench.py's roofline leg).  name is one of "pcm_scan", "freq_scan",
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

/* Releases every default context (workspaces, streams, pinned staging) and the multi-device
 * state.  Explicit contexts are released by bl_amd_ctx_destroy. */
void bl_amd_shutdown(void);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* BLISS_AMD_H_ */
