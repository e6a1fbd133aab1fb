/*
 * bl_launch.h — internal seam between the kernel translation unit (bl_kernels.hip: device
 * code + launch geometry) and the runtime (bl_runtime.hip: contexts, workspaces, streams,
 * the C-ABI of include/bliss_amd.h; bl_multi.hip: the multi-device corpus path).
 * C++ only, not installed.
 */
#ifndef BL_LAUNCH_H_
#define BL_LAUNCH_H_

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "bl_device.h"
#include "bl_fft.h"

/* ---- device-side records ------------------------------------------------- */

struct bl_dsong {
  unsigned long long pcm_off;  /* int16 elements from the arena base */
  unsigned long long duration; /* seconds */
  long long env_off;           /* first slot of this song in the per-window arrays */
  int n, channels;
  int n_frames;  /* (n / channels) / 512              ref frequency_sort.c:50 */
  int nb_frames; /* 2 * floor(n / 512)                ref tempo_atk_sort.c:63-64 */
  int n_windows; /* nb_frames - 2 windows of hop 256  ref tempo_atk_sort.c:66-67,120 */
  int reserved0;
  int out_idx;   /* result slot = position in the caller's order (records are length-sorted) */
  int reserved1;
};

struct bl_dstats {
  unsigned long long sum;   /* two's-complement sum of all samples */
  unsigned long long sumsq; /* sum of squares */
  unsigned first;           /* first index with a non-zero sample */
  int last;                 /* last index with a non-zero sample */
  int mean, variance;
  double vprime; /* variance * 2^-15 */
  double rcp;    /* RN(1 / (2 vprime)) */
  double rcp_lo; /* 1 / (2 vprime) - rcp, see bl_norm */
  int wrap_pass; /* 1: variance must come from k_variance_wrap */
  int status;
  long long wrap_acc; /* accumulator of k_variance_wrap */
  double firc[9];     /* RN(c_m / (2 vprime)), m = 0..8: the FIR taps with the normalisation folded in
                       * (BL_AMD_FIR_FUSED=2 only) */
};

typedef bl_c2<double> c2d;
typedef bl_c2<float> c2f;

struct bl_tables {
  const c2d *tw256_d, *tw512_d;
  const float *hann;
  const c2f *lv_tw;   /* bl_fft_lavc.h: [LV_TW_SLOTS][16 lanes] (cos, "sin") pairs of libavcodec's tables */
  float lv_leafc[4];  /* sqrthalf, cos_16[1], cos_16[3] */
  double log101;
};

/* ---- per-kernel timing (bench.py's roofline leg) ---------------------------- */

enum { PK_SCAN, PK_AMP, PK_FREQ, PK_FREQ_FIN, PK_ENV, PK_TAIL, PK_DIST, PK_FREQ_SCAN, PK_COUNT };

/* called by the launchers around a kernel when profiling is on: begin = 1 before the
 * launch, 0 after it, on the stream the kernel is launched on */
typedef void (*blk_mark_fn)(void *user, int kernel_id, hipStream_t stream, int begin);

/* ---- launchers (all asynchronous on the given stream, current device) ------ */

size_t blk_tables_bytes(void);
void blk_tables_fill_host(unsigned char *h);           /* twiddles + Hann, double math on the host */
bl_tables blk_tables_bind(const void *d_mem);          /* pointers into the device copy */
int blk_configure_device(void);                        /* dynamic-LDS attributes, once per device */

struct blk_analyze_args {
  const int16_t *pcm;        /* arena base */
  const bl_dsong *songs;     /* device, n_songs records */
  bl_dstats *stats;          /* device scratch */
  unsigned *hist;            /* device scratch, BL_HIST_BINS per song (zeroed by the launcher) */
  float *spectrum;           /* device scratch, 256 per song */
  float *energies;           /* device scratch, one per envelope slot */
  double *lc;                /* device scratch, one per envelope slot */
  bl_amd_song_result *results;
  int n_songs, max_n, what, n_cu;
  int n_head = 0, max_n_rest = 0; /* mixed lengths: the first n_head (longest) songs get their own window launch */
  bl_tables tb;
  hipStream_t stream, side;  /* side == nullptr: envelope tail on `stream` */
  hipEvent_t ev_env, ev_tail;
  hipStream_t side2 = nullptr; /* mixed lengths: the long songs' tail (nullptr: no separate launch for them) */
  hipEvent_t ev_head = nullptr, ev_tail2 = nullptr;
  blk_mark_fn mark;          /* may be nullptr */
  void *mark_user;
};
int blk_analyze(const blk_analyze_args &a);

int blk_synth(hipStream_t s, int16_t *pcm, const bl_dsong *d_songs, int n_songs, int max_n,
              int n_cu, unsigned seed_base, unsigned rate);
int blk_pairwise(hipStream_t s, const struct force_vector_s *d_vecs, int n, int row_begin,
                 int n_rows, float *d_out, bool cosine, blk_mark_fn mark, void *mark_user);
/* exhaustive self-test of bl_sqrt.h over f32 bit patterns [first, first + count): d_counts[0..2] +=
 * values in the fast domain, mismatches of the fast root, mismatches of the compiler's sqrtf */
int blk_sqrt_sweep(hipStream_t s, unsigned long long first, unsigned long long count,
                   unsigned long long *d_counts, int n_cu);
int blk_cos_sweep(hipStream_t s, unsigned long long seed, int per_thread, unsigned long long *d_counts, int n_cu);
int blk_playlist(hipStream_t s, const struct force_vector_s *d_vecs, int n, int seed_index,
                 int32_t *d_order, float *d_dist);
/* out[i] = (int16)(in[i] >> 16): the same-rate S32 -> S16 narrowing */
int blk_narrow_s32(hipStream_t s, const int32_t *d_in, int16_t *d_out, size_t n, int n_cu);
/* out[order[i]] = in[i] for 16-byte force vectors (shard-major -> caller order) */
int blk_scatter_vecs(hipStream_t s, const struct force_vector_s *d_in, const int32_t *d_order,
                     struct force_vector_s *d_out, int n);
/* force vectors of a result array, in result order */
int blk_extract_vecs(hipStream_t s, const bl_amd_song_result *d_res, struct force_vector_s *d_out,
                     int n);
/* bl_mean / bl_variance helpers: one song described by d_songs[0] */
int blk_scan_one(hipStream_t s, const int16_t *pcm, const bl_dsong *d_songs, bl_dstats *d_stats,
                 unsigned *d_hist, int n, int n_cu);
int blk_variance_wrap_one(hipStream_t s, const int16_t *pcm, const bl_dsong *d_songs,
                          bl_dstats *d_stats, int n, int n_cu);

/* ---- device rate converter (bl_rs_kernels.hip) ------------------------------ */

#define BL_RS_MAX_DEVICES 16

struct bl_rs_dsong {
  unsigned long long in_off;  /* elements (int16 or int32) from the input base */
  unsigned long long out_off; /* int16 elements from the output base, even */
  int frames, channels;       /* input frames, 1 | 2 */
  int out_frames, refl;       /* bl_rs_out_frames() */
};

struct bl_rs_geom {
  int phase_count, taps, taps8, alloc, w0, span, tiles_per_wg;
  unsigned long long src_incr, dst_incr;
};

/* geometry + LDS budget of a plan (bl_resample.h); BL_UNEXPECTED when a tile's input span
 * does not fit the LDS (input rates far above 192 kHz) */
int blk_resample_geom(int phase_count, int taps, int alloc, int src_incr, int dst_incr, bl_rs_geom *g,
                      size_t *lds_bytes, int *bank_in_lds);
/* d_bank: phase_count rows of `alloc` 32-bit elements (float, or the Q15 coefficients as int) */
int blk_resample(hipStream_t s, const void *d_in, int in_is_s32, const bl_rs_dsong *d_songs, int n_songs,
                 int max_out_frames, const void *d_bank, const bl_rs_geom &g, size_t lds_bytes,
                 int bank_in_lds, int16_t *d_out);

#endif /* BL_LAUNCH_H_ */
