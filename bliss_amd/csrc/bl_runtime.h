/*
 * bl_runtime.h — internal: the per-device context behind include/bliss_amd.h, shared by
 * bl_runtime.hip (single-device C-ABI) and bl_multi.hip (multi-device corpus path).
 * C++ only, not installed.
 */
#ifndef BL_RUNTIME_H_
#define BL_RUNTIME_H_

#include <hip/hip_runtime.h>

#include <mutex>
#include <vector>

#include "bl_launch.h"

#define BL_HIP_CHECK(expr)                                                              \
  do {                                                                                  \
    hipError_t e_ = (expr);                                                             \
    if (e_ != hipSuccess) {                                                             \
      fprintf(stderr, "bliss_amd: %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(e_), \
              __FILE__, __LINE__);                                                      \
      return BL_UNEXPECTED;                                                             \
    }                                                                                   \
  } while (0)

#define BL_MAX_DEVICES 16
#define BL_GROUP_SONGS_MAX 32768 /* gridDim.y of the (blocks, songs) launch grids */
#define BL_PIN_SLOTS 4

struct bl_buf {
  void *p = nullptr;
  size_t cap = 0;
};

struct bl_pin_slot { /* pinned staging of small host->device records */
  void *p = nullptr;
  size_t cap = 0;
  hipEvent_t ev = nullptr;
  bool busy = false;
};

/* One device, one workspace, one set of internal streams.  Calls on one context are
 * ordered (host mutex for the enqueue, an event for the shared workspace on the device);
 * different contexts — on the same device or on different ones — are independent. */
struct bl_amd_ctx {
  std::mutex mu;
  int device = 0;
  int n_cu = 256;
  int group_songs = BL_GROUP_SONGS_MAX;
  hipStream_t side = nullptr; /* envelope tail runs here, beside the frequency pass */
  hipEvent_t ev_env = nullptr, ev_tail = nullptr;
  hipStream_t side2 = nullptr; /* mixed lengths: the tail of the long songs, under the window kernel of the rest */
  hipEvent_t ev_head = nullptr, ev_tail2 = nullptr;
  hipEvent_t ev_ws = nullptr; /* end of the last launch group that used the workspace */
  bool ws_used = false;
  long long last_env_total = 0;
  bl_tables tb{};
  void *tables_mem = nullptr;
  bl_buf songs, stats, hist, spectrum, energies, lc, results, misc;
  bl_pin_slot ring[BL_PIN_SLOTS];
  int ring_next = 0;
  /* profiling */
  bool prof = false;
  struct Ev { int k; hipEvent_t a, b; };
  std::vector<Ev> events;
  std::vector<Ev> open; /* begun, not yet ended */
  double prof_ms[PK_COUNT] = {0};
  int prof_n[PK_COUNT] = {0};
  /* host-batch staging: two waves in flight */
  void *pinned[2] = {nullptr, nullptr};
  size_t pinned_cap[2] = {0, 0};
  bl_buf arena[2];
  bl_buf arena22[2]; /* converted (22 050 Hz) songs of a wave whose input is at another rate */
  hipStream_t streams[2] = {nullptr, nullptr};
  std::vector<void *> registered[2]; /* host ranges pinned in place for wave k */
  /* multi-device corpus path (bl_multi.hip): this rank's vectors, the gathered blocks, the
   * vectors in output order, the order table and the row block — grown on demand, kept */
  bl_buf mx_my, mx_gath, mx_all, mx_order, mx_rows;
  /* device rate converter: the plan of the last (input rate, sample kind) stays uploaded */
  bl_buf rs_songs, rs_bank;
  int rs_rate = 0, rs_kind = -1, rs_bank_lds = 0;
  bl_rs_geom rs_geom{};
  size_t rs_lds = 0;
  int rs_taps = 0, rs_phases = 0, rs_src_incr = 0, rs_dst_incr = 0;
};

/* bl_runtime.hip */
int blr_ensure(bl_buf &b, size_t bytes);
/* thread's default context (device chosen by bl_amd_init, default 0); nullptr + message on failure */
bl_amd_ctx *blr_default_ctx(void);
int blr_analyze_device(bl_amd_ctx *c, const int16_t *d_pcm, const bl_amd_song_desc *h_desc, int n_songs,
                       bl_amd_song_result *d_results, hipStream_t stream, int what);
/* host-memory batch on one context; pcm_is_s32: h_pcm[i] points at int32 samples that are
 * narrowed with >> 16 while they are staged.  in_rate: 0 / 22 050 = as is; another rate = the songs
 * are converted on the device first (wide sources then travel as int32).  d_res_out (optional) receives the device
 * pointer of the results (valid until the context's next host batch). */
int blr_analyze_host(bl_amd_ctx *c, const void *const *h_pcm, int pcm_is_s32, const int32_t *n_samples,
                     const int32_t *channels, const uint64_t *duration, int n_songs, int in_rate,
                     bl_amd_song_result *h_results, bl_amd_song_result **d_res_out);

#endif /* BL_RUNTIME_H_ */
