/*
 * bl_runtime.hip — contexts, workspaces, streams and the single-device C-ABI of
 * include/bliss_amd.h.  Host code only (compiled by hipcc for the HIP runtime API); every
 * kernel lives in bl_kernels.hip and is reached through the launchers of bl_launch.h.
 *
 * Contexts.  A bl_amd_ctx owns one device's scratch workspace, internal streams and pinned
 * staging.  The plain entry points (bl_amd_analyze_batch_device, ...) use the calling
 * thread's default context: the device chosen with bl_amd_init() on that thread, else the
 * process default (the first bl_amd_init of the process, else device 0).  One process can
 * therefore drive several GPUs from several threads, and explicit contexts
 * (bl_amd_ctx_create) give independent workspaces on one device.
 *
 * Asynchrony.  bl_amd_analyze_batch_device() only enqueues: descriptors go through a ring of
 * pinned slots (hipMemcpyAsync from pinned memory does not block), launch groups of more than
 * 32 768 songs follow each other on the stream, and the shared workspace is handed from one
 * batch to the next by an event, not by a host synchronisation.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <thread>

#include "bl_runtime.h"
#include "bl_resample.h"

namespace {

std::mutex g_mu;                       /* guards the default-context table */
bl_amd_ctx *g_default[BL_MAX_DEVICES]; /* lazily created, one per device */
std::atomic<int> g_process_device{-1};
thread_local int tl_device = -1;
std::atomic<int> g_host_mode{-1}; /* -1: from BL_AMD_HOST_MODE or staged */

/* makes the context's device current for the duration of a call and puts the caller's back:
 * the current device is per-thread state shared with whoever else uses HIP in this thread
 * (torch, the caller's own code) */
struct DevGuard {
  int prev = -1;
  bool changed = false;
  bool ok = true;
  explicit DevGuard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) { prev = -1; (void)hipGetLastError(); }
    if (prev != dev) {
      ok = hipSetDevice(dev) == hipSuccess;
      changed = ok && prev >= 0;
    }
  }
  ~DevGuard() {
    if (changed) (void)hipSetDevice(prev);
  }
};

void release_buf(bl_buf &b) {
  if (b.p) (void)hipFree(b.p);
  b.p = nullptr;
  b.cap = 0;
}

int ctx_init(bl_amd_ctx *c, int device) {
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0) {
    fprintf(stderr, "bliss_amd: no HIP device available (%s); this library has no CPU path\n",
            e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    return BL_UNEXPECTED;
  }
  if (device < 0 || device >= count || device >= BL_MAX_DEVICES) {
    fprintf(stderr, "bliss_amd: device %d out of range (%d visible)\n", device, count);
    return BL_UNEXPECTED;
  }
  DevGuard dg(device);
  if (!dg.ok) return BL_UNEXPECTED;
  hipDeviceProp_t prop;
  BL_HIP_CHECK(hipGetDeviceProperties(&prop, device));
  c->device = device;
  c->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  std::vector<unsigned char> h(blk_tables_bytes());
  blk_tables_fill_host(h.data());
  BL_HIP_CHECK(hipMalloc(&c->tables_mem, h.size()));
  BL_HIP_CHECK(hipMemcpy(c->tables_mem, h.data(), h.size(), hipMemcpyHostToDevice));
  c->tb = blk_tables_bind(c->tables_mem);
  if (blk_configure_device() != BL_OK) return BL_UNEXPECTED;
  {
    /* songs per launch group; lowered by the tests to exercise the multi-group path */
    const char *gs = getenv("BL_AMD_GROUP_SONGS");
    int g = gs ? atoi(gs) : BL_GROUP_SONGS_MAX;
    c->group_songs = g < 1 ? 1 : (g > BL_GROUP_SONGS_MAX ? BL_GROUP_SONGS_MAX : g);
  }
  BL_HIP_CHECK(hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking));
  BL_HIP_CHECK(hipStreamCreateWithFlags(&c->side2, hipStreamNonBlocking));
  BL_HIP_CHECK(hipEventCreateWithFlags(&c->ev_env, hipEventDisableTiming));
  BL_HIP_CHECK(hipEventCreateWithFlags(&c->ev_tail, hipEventDisableTiming));
  BL_HIP_CHECK(hipEventCreateWithFlags(&c->ev_head, hipEventDisableTiming));
  BL_HIP_CHECK(hipEventCreateWithFlags(&c->ev_tail2, hipEventDisableTiming));
  BL_HIP_CHECK(hipEventCreateWithFlags(&c->ev_ws, hipEventDisableTiming));
  for (int k = 0; k < BL_PIN_SLOTS; ++k)
    BL_HIP_CHECK(hipEventCreateWithFlags(&c->ring[k].ev, hipEventDisableTiming));
  return BL_OK;
}

void prof_collect(bl_amd_ctx *c) {
  for (auto &e : c->events) {
    float ms = 0;
    if (hipEventSynchronize(e.b) == hipSuccess && hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess) {
      c->prof_ms[e.k] += ms;
      c->prof_n[e.k] += 1;
    }
    (void)hipEventDestroy(e.a);
    (void)hipEventDestroy(e.b);
  }
  c->events.clear();
}

void unregister_wave(bl_amd_ctx *c, int k) {
  for (void *p : c->registered[k]) (void)hipHostUnregister(p);
  c->registered[k].clear();
}

void ctx_release(bl_amd_ctx *c) {
  DevGuard dg(c->device);
  if (!dg.ok) return;
  (void)hipDeviceSynchronize();
  prof_collect(c);
  bl_buf *bufs[] = {&c->songs,   &c->stats,   &c->hist, &c->spectrum, &c->energies, &c->lc,
                    &c->results, &c->misc,    &c->arena[0], &c->arena[1], &c->arena22[0], &c->arena22[1],
                    &c->rs_songs, &c->rs_bank, &c->mx_my, &c->mx_gath, &c->mx_all, &c->mx_order, &c->mx_rows};
  for (bl_buf *b : bufs) release_buf(*b);
  for (int k = 0; k < 2; ++k) {
    unregister_wave(c, k);
    if (c->pinned[k]) (void)hipHostFree(c->pinned[k]);
    c->pinned[k] = nullptr;
    c->pinned_cap[k] = 0;
    if (c->streams[k]) (void)hipStreamDestroy(c->streams[k]);
    c->streams[k] = nullptr;
  }
  for (int k = 0; k < BL_PIN_SLOTS; ++k) {
    if (c->ring[k].p) (void)hipHostFree(c->ring[k].p);
    if (c->ring[k].ev) (void)hipEventDestroy(c->ring[k].ev);
    c->ring[k] = bl_pin_slot();
  }
  if (c->tables_mem) (void)hipFree(c->tables_mem);
  c->tables_mem = nullptr;
  if (c->side) (void)hipStreamDestroy(c->side);
  if (c->side2) (void)hipStreamDestroy(c->side2);
  if (c->ev_env) (void)hipEventDestroy(c->ev_env);
  if (c->ev_tail) (void)hipEventDestroy(c->ev_tail);
  if (c->ev_head) (void)hipEventDestroy(c->ev_head);
  if (c->ev_tail2) (void)hipEventDestroy(c->ev_tail2);
  c->side2 = nullptr;
  c->ev_head = c->ev_tail2 = nullptr;
  if (c->ev_ws) (void)hipEventDestroy(c->ev_ws);
  c->side = nullptr;
  c->ev_env = c->ev_tail = c->ev_ws = nullptr;
  c->ws_used = false;
  c->rs_rate = 0;
  c->rs_kind = -1;
}

int current_device(void) {
  if (tl_device >= 0) return tl_device;
  const int d = g_process_device.load();
  return d >= 0 ? d : 0;
}

void mark_cb(void *user, int k, hipStream_t s, int begin) {
  bl_amd_ctx *c = static_cast<bl_amd_ctx *>(user);
  if (!c->prof) return;
  if (begin) {
    bl_amd_ctx::Ev e{k, nullptr, nullptr};
    if (hipEventCreate(&e.a) != hipSuccess || hipEventCreate(&e.b) != hipSuccess) return;
    (void)hipEventRecord(e.a, s);
    c->open.push_back(e);
  } else {
    for (size_t i = c->open.size(); i-- > 0;)
      if (c->open[i].k == k) {
        (void)hipEventRecord(c->open[i].b, s);
        c->events.push_back(c->open[i]);
        c->open.erase(c->open.begin() + (long)i);
        break;
      }
  }
}

/* pinned slot of at least `bytes`; blocks only when BL_PIN_SLOTS batches are still in flight */
int ring_get(bl_amd_ctx *c, size_t bytes, bl_pin_slot **out) {
  bl_pin_slot &s = c->ring[c->ring_next];
  c->ring_next = (c->ring_next + 1) % BL_PIN_SLOTS;
  if (s.busy) {
    BL_HIP_CHECK(hipEventSynchronize(s.ev));
    s.busy = false;
  }
  if (s.cap < bytes) {
    if (s.p) (void)hipHostFree(s.p);
    s.p = nullptr;
    s.cap = 0;
    const size_t cap = bytes + bytes / 4 + 4096;
    BL_HIP_CHECK(hipHostMalloc(&s.p, cap, hipHostMallocDefault));
    s.cap = cap;
  }
  *out = &s;
  return BL_OK;
}

int validate_desc(const bl_amd_song_desc &d, int i) {
  if (d.n_samples < 5120 || (d.channels != 1 && d.channels != 2) || d.duration == 0 ||
      (d.pcm_offset & 7)) {
    fprintf(stderr,
            "bliss_amd: song %d rejected (n_samples=%d channels=%d duration=%llu offset=%llu): "
            "need n_samples >= 5120, channels 1|2, duration > 0, offset %% 8 == 0\n",
            i, d.n_samples, d.channels, (unsigned long long)d.duration,
            (unsigned long long)d.pcm_offset);
    return BL_UNEXPECTED;
  }
  return BL_OK;
}

/* host mirror of the per-song geometry (integer work of ref tempo_atk_sort.c:63-67,
 * frequency_sort.c:50) for one launch group, written to `out` (pinned) */
void fill_group(const bl_amd_song_desc *desc, int n_songs, bl_dsong *out, long long &env_total,
                int &max_n) {
  env_total = 0;
  max_n = 0;
  for (int i = 0; i < n_songs; ++i) {
    const bl_amd_song_desc &d = desc[i];
    bl_dsong &s = out[i];
    s.pcm_off = d.pcm_offset;
    s.duration = d.duration;
    s.n = d.n_samples;
    s.channels = d.channels;
    s.n_frames = (d.n_samples / d.channels) / 512;
    s.nb_frames = (d.n_samples - (d.n_samples % 512)) * 2 / 512;
    s.n_windows = s.nb_frames - 2;
    s.env_off = env_total;
    s.reserved0 = 0;
    s.out_idx = i;
    s.reserved1 = 0;
    env_total += s.nb_frames;
    if (d.n_samples > max_n) max_n = d.n_samples;
  }
  /* Mixed-length corpora: process the records longest first.  The 64 songs that share a
   * wave of k_env_tail then have similar lengths (its straight-line steady-state path is
   * wave-uniform), and the long songs do not straggle at the end of the per-song grids.
   * Results go back to the caller's order through out_idx; scratch offsets keep the
   * caller's order too.  Equal lengths: the order is left alone. */
  bool mixed = false;
  for (int i = 1; i < n_songs && !mixed; ++i) mixed = out[i].n != out[0].n;
  if (mixed)
    std::stable_sort(out, out + n_songs, [](const bl_dsong &a, const bl_dsong &b) { return a.n > b.n; });
}

} // namespace

int blr_ensure(bl_buf &b, size_t bytes) {
  if (bytes <= b.cap) return BL_OK;
  if (b.p) {
    /* growth is rare (capacity is kept); work in flight may still use the old block */
    BL_HIP_CHECK(hipDeviceSynchronize());
    BL_HIP_CHECK(hipFree(b.p));
    b.p = nullptr;
    b.cap = 0;
  }
  size_t cap = bytes + bytes / 8 + 4096;
  BL_HIP_CHECK(hipMalloc(&b.p, cap));
  b.cap = cap;
  return BL_OK;
}

bl_amd_ctx *blr_default_ctx(void) {
  const int dev = current_device();
  if (dev < 0 || dev >= BL_MAX_DEVICES) return nullptr;
  std::lock_guard<std::mutex> lk(g_mu);
  if (!g_default[dev]) {
    bl_amd_ctx *c = new bl_amd_ctx();
    if (ctx_init(c, dev) != BL_OK) {
      ctx_release(c);
      delete c;
      return nullptr;
    }
    g_default[dev] = c;
  }
  return g_default[dev];
}

/* Enqueue the analysis of n_songs device-resident songs on `stream` (caller holds c->mu and
 * has made c->device current).  Nothing here waits for the device. */
int blr_analyze_device(bl_amd_ctx *c, const int16_t *d_pcm, const bl_amd_song_desc *h_desc, int n_songs,
                       bl_amd_song_result *d_results, hipStream_t stream, int what) {
  for (int i = 0; i < n_songs; ++i)
    if (validate_desc(h_desc[i], i) != BL_OK) return BL_UNEXPECTED;
  const int G = c->group_songs;
  const int n_groups = (n_songs + G - 1) / G;
  /* all descriptors of the call in one pinned slot, one asynchronous copy */
  bl_pin_slot *slot = nullptr;
  if (ring_get(c, sizeof(bl_dsong) * (size_t)n_songs, &slot) != BL_OK) return BL_UNEXPECTED;
  bl_dsong *hs = static_cast<bl_dsong *>(slot->p);
  std::vector<long long> env_total(n_groups);
  std::vector<int> max_n(n_groups);
  long long env_max = 0;
  for (int gi = 0; gi < n_groups; ++gi) {
    const int b = gi * G, cnt = std::min(G, n_songs - b);
    fill_group(h_desc + b, cnt, hs + b, env_total[gi], max_n[gi]);
    env_max = std::max(env_max, env_total[gi]);
  }
  const int gmax = std::min(G, n_songs);
  /* the workspace is shared by every call on this context: a batch enqueued on another
   * stream waits (on the device) for the previous user; the mutex only orders the enqueues */
  if (c->ws_used) BL_HIP_CHECK(hipStreamWaitEvent(stream, c->ev_ws, 0));
  if (blr_ensure(c->songs, sizeof(bl_dsong) * (size_t)n_songs) != BL_OK) return BL_UNEXPECTED;
  if (blr_ensure(c->stats, sizeof(bl_dstats) * (size_t)gmax) != BL_OK) return BL_UNEXPECTED;
  if (blr_ensure(c->hist, sizeof(unsigned) * BL_HIST_BINS * (size_t)gmax) != BL_OK) return BL_UNEXPECTED;
  if (blr_ensure(c->spectrum, sizeof(float) * 256 * (size_t)gmax) != BL_OK) return BL_UNEXPECTED;
  if (blr_ensure(c->energies, sizeof(float) * (size_t)env_max) != BL_OK) return BL_UNEXPECTED;
  if (blr_ensure(c->lc, sizeof(double) * (size_t)env_max) != BL_OK) return BL_UNEXPECTED;
  bl_dsong *d_songs = static_cast<bl_dsong *>(c->songs.p);
  BL_HIP_CHECK(hipMemcpyAsync(d_songs, hs, sizeof(bl_dsong) * (size_t)n_songs, hipMemcpyHostToDevice,
                              stream));
  BL_HIP_CHECK(hipEventRecord(slot->ev, stream));
  slot->busy = true;
  for (int gi = 0; gi < n_groups; ++gi) {
    const int b = gi * G, cnt = std::min(G, n_songs - b);
    blk_analyze_args a;
    a.pcm = d_pcm;
    a.songs = d_songs + b;
    a.stats = static_cast<bl_dstats *>(c->stats.p);
    a.hist = static_cast<unsigned *>(c->hist.p);
    a.spectrum = static_cast<float *>(c->spectrum.p);
    a.energies = static_cast<float *>(c->energies.p);
    a.lc = static_cast<double *>(c->lc.p);
    a.results = d_results + b;
    a.n_songs = cnt;
    a.max_n = max_n[gi];
    /* a mixed-length group (sorted longest first by fill_group): the songs longer than a quarter of the longest
     * go first, in whole waves of the tail kernel, if both parts stay wide enough to fill the chip */
    if (cnt >= 1024 && hs[b].n != hs[b + cnt - 1].n) {
      int k = 0;
      while (k < cnt && hs[b + k].n > max_n[gi] / 4) ++k;
      k = (k + 63) / 64 * 64;
      if (k >= 256 && k <= cnt - 256) {
        a.n_head = k;
        a.max_n_rest = hs[b + k].n;
      }
    }
    a.what = what;
    a.n_cu = c->n_cu;
    a.tb = c->tb;
    a.stream = stream;
#ifdef BL_AMD_MEASURE
    a.side = getenv("BL_AMD_NO_SIDE") ? nullptr : c->side; /* measurement builds only: serialise the tail */
#else
    a.side = c->side;
#endif
    a.ev_env = c->ev_env;
    a.ev_tail = c->ev_tail;
    a.side2 = a.side ? c->side2 : nullptr;
    a.ev_head = c->ev_head;
    a.ev_tail2 = c->ev_tail2;
    a.mark = c->prof ? mark_cb : nullptr;
    a.mark_user = c;
    if (blk_analyze(a) != BL_OK) return BL_UNEXPECTED;
    c->last_env_total = env_total[gi];
  }
  BL_HIP_CHECK(hipEventRecord(c->ev_ws, stream));
  c->ws_used = true;
  return BL_OK;
}

/* ---- device rate conversion (bl_rs_kernels.hip), shared by the C-ABI and the host batch ---- */
#define BL_RS_OUT_RATE 22050 /* ref src/decode.c:7 SAMPLE_RATE */

/* (re)build and upload the plan of (in_rate, kind); caller holds c->mu, device is current */
static int rs_prepare(bl_amd_ctx *c, int in_rate, int in_is_s32) {
  if (c->rs_rate == in_rate && c->rs_kind == in_is_s32) return BL_OK;
  bl_rs_plan p;
  if (bl_rs_plan_build(&p, BL_RS_OUT_RATE, in_rate, in_is_s32)) {
    fprintf(stderr, "bliss_amd: no conversion plan for %d Hz\n", in_rate);
    return BL_UNEXPECTED;
  }
  bl_rs_geom g;
  size_t lds = 0;
  int bank_lds = 0;
  if (blk_resample_geom(p.phase_count, p.taps, p.alloc, p.src_incr, p.dst_incr, &g, &lds, &bank_lds) != BL_OK) {
    fprintf(stderr, "bliss_amd: %d Hz is beyond the device converter (input span of a tile exceeds the LDS)\n",
            in_rate);
    bl_rs_plan_free(&p);
    return BL_UNEXPECTED;
  }
  /* 32-bit elements on the device: float as is, Q15 widened */
  const size_t elems = (size_t)p.phase_count * (size_t)p.alloc;
  std::vector<int32_t> host(elems);
  if (in_is_s32) memcpy(host.data(), p.fbank, elems * sizeof(float));
  else for (size_t i = 0; i < elems; ++i) host[i] = p.ibank[i];
  bl_rs_plan_free(&p);
  /* a plan change is rare; kernels of the previous plan may still be reading the old bank */
  BL_HIP_CHECK(hipDeviceSynchronize());
  c->rs_rate = 0;
  c->rs_kind = -1;
  if (blr_ensure(c->rs_bank, elems * 4) != BL_OK) return BL_UNEXPECTED;
  BL_HIP_CHECK(hipMemcpy(c->rs_bank.p, host.data(), elems * 4, hipMemcpyHostToDevice));
  c->rs_geom = g;
  c->rs_lds = lds;
  c->rs_bank_lds = bank_lds;
  c->rs_taps = p.taps;
  c->rs_phases = p.phase_count;
  c->rs_src_incr = p.src_incr;
  c->rs_dst_incr = p.dst_incr;
  c->rs_rate = in_rate;
  c->rs_kind = in_is_s32;
  return BL_OK;
}

/* Enqueue the conversion of n_songs device-resident songs on `s` (caller holds c->mu and has
 * made c->device current). */
int blr_resample_device(bl_amd_ctx *c, const void *d_in, int in_is_s32, const bl_amd_resample_desc *h_desc,
                        int n_songs, int in_rate, int16_t *d_out, hipStream_t s) {
  in_is_s32 = in_is_s32 != 0;
  if (rs_prepare(c, in_rate, in_is_s32) != BL_OK) return BL_UNEXPECTED;
  bl_rs_plan geo;
  memset(&geo, 0, sizeof geo);
  geo.taps = c->rs_taps;
  geo.phase_count = c->rs_phases;
  geo.src_incr = c->rs_src_incr;
  geo.dst_incr = c->rs_dst_incr;
  bl_pin_slot *slot = nullptr;
  if (ring_get(c, sizeof(bl_rs_dsong) * (size_t)n_songs, &slot) != BL_OK) return BL_UNEXPECTED;
  bl_rs_dsong *hs = static_cast<bl_rs_dsong *>(slot->p);
  for (int i = 0; i < n_songs; ++i) {
    const bl_amd_resample_desc &d = h_desc[i];
    size_t refl = 0;
    const size_t of = d.frames > 0 ? bl_rs_out_frames(&geo, (size_t)d.frames, &refl) : 0;
    if (of == 0 || of > (size_t)INT32_MAX / 2 || (d.channels != 1 && d.channels != 2) ||
        (d.out_offset & 1) || (d.channels == 2 && (d.in_offset & 1))) {
      fprintf(stderr,
              "bliss_amd: resample: song %d rejected (frames=%d channels=%d in_offset=%llu out_offset=%llu): "
              "need frames > filter length (%d), channels 1|2, even offsets\n",
              i, d.frames, d.channels, (unsigned long long)d.in_offset, (unsigned long long)d.out_offset,
              c->rs_taps);
      return BL_UNEXPECTED;
    }
    hs[i].in_off = d.in_offset;
    hs[i].out_off = d.out_offset;
    hs[i].frames = d.frames;
    hs[i].channels = d.channels;
    hs[i].out_frames = (int)of;
    hs[i].refl = (int)refl;
  }
  if (c->ws_used) BL_HIP_CHECK(hipStreamWaitEvent(s, c->ev_ws, 0));
  if (blr_ensure(c->rs_songs, sizeof(bl_rs_dsong) * (size_t)n_songs) != BL_OK) return BL_UNEXPECTED;
  bl_rs_dsong *d_songs = static_cast<bl_rs_dsong *>(c->rs_songs.p);
  BL_HIP_CHECK(hipMemcpyAsync(d_songs, hs, sizeof(bl_rs_dsong) * (size_t)n_songs, hipMemcpyHostToDevice, s));
  BL_HIP_CHECK(hipEventRecord(slot->ev, s));
  slot->busy = true;
  const int G = BL_GROUP_SONGS_MAX;
  for (int b = 0; b < n_songs; b += G) {
    const int cnt = std::min(G, n_songs - b);
    int max_out = 0;
    for (int i = 0; i < cnt; ++i) max_out = std::max(max_out, hs[b + i].out_frames);
    if (blk_resample(s, d_in, in_is_s32, d_songs + b, cnt, max_out, c->rs_bank.p, c->rs_geom, c->rs_lds,
                     c->rs_bank_lds, d_out) != BL_OK)
      return BL_UNEXPECTED;
  }
  BL_HIP_CHECK(hipEventRecord(c->ev_ws, s));
  c->ws_used = true;
  return BL_OK;
}


/* ---- host-memory batch: pinned staging, copy/compute overlap on 2 streams ---- */
#ifndef BL_STAGE_THREADS
#define BL_STAGE_THREADS 8 /* host threads of the staging copy (BL_AMD_STAGE_THREADS overrides) */
#endif

static int stage_threads(void) {
  static const int n = [] { /* initialised once, thread-safe (C++11 function-local static) */
    const char *e = getenv("BL_AMD_STAGE_THREADS");
    const int v = e ? atoi(e) : BL_STAGE_THREADS;
    return v < 1 ? 1 : (v > 64 ? 64 : v);
  }();
  return n;
}

static int host_mode(void) {
  int m = g_host_mode.load();
  if (m < 0) {
    const char *e = getenv("BL_AMD_HOST_MODE");
    m = (e && !strcmp(e, "registered")) ? BL_AMD_HOST_REGISTERED : BL_AMD_HOST_STAGED;
    g_host_mode.store(m);
  }
  return m;
}

int blr_analyze_host(bl_amd_ctx *c, const void *const *h_pcm, int pcm_is_s32, const int32_t *n_samples,
                     const int32_t *channels, const uint64_t *duration, int n_songs, int in_rate,
                     bl_amd_song_result *h_results, bl_amd_song_result **d_res_out) {
  /* songs at another rate than the analyzers' are converted on the device, wave by wave, between
   * the transfer and the analysis; wide sources then travel as int32 (the converter's float path
   * needs all their bits) */
  const bool convert = in_rate > 0 && in_rate != BL_RS_OUT_RATE;
  bl_rs_plan geo;
  memset(&geo, 0, sizeof geo);
  if (convert && bl_rs_plan_geometry(&geo, BL_RS_OUT_RATE, in_rate)) return BL_UNEXPECTED;
  const size_t esz = (convert && pcm_is_s32) ? 4 : 2; /* bytes per staged sample */
  /* reject bad descriptors before anything is staged (a negative length would otherwise
   * become a ~2^64-byte copy) */
  for (int i = 0; i < n_songs; ++i) {
    if (!h_pcm[i]) {
      fprintf(stderr, "bliss_amd: song %d rejected: NULL sample pointer\n", i);
      return BL_UNEXPECTED;
    }
    bl_amd_song_desc d;
    d.pcm_offset = 0; d.n_samples = n_samples[i]; d.channels = channels[i]; d.duration = duration[i];
    if (convert) { /* what the analyzers will see: stereo at 22 050 Hz */
      if (n_samples[i] <= 0 || (channels[i] != 1 && channels[i] != 2) || n_samples[i] % channels[i]) {
        fprintf(stderr, "bliss_amd: song %d rejected (n_samples=%d channels=%d)\n", i, n_samples[i], channels[i]);
        return BL_UNEXPECTED;
      }
      const size_t of = bl_rs_out_frames(&geo, (size_t)(n_samples[i] / channels[i]), nullptr);
      d.n_samples = of > (size_t)INT32_MAX / 2 ? -1 : (int32_t)(2 * of);
      d.channels = 2;
    }
    if (validate_desc(d, i) != BL_OK) return BL_UNEXPECTED;
  }
  for (int k = 0; k < 2; ++k)
    if (!c->streams[k]) BL_HIP_CHECK(hipStreamCreateWithFlags(&c->streams[k], hipStreamNonBlocking));
  if (blr_ensure(c->results, sizeof(bl_amd_song_result) * (size_t)n_songs) != BL_OK) return BL_UNEXPECTED;
  bl_amd_song_result *d_res = static_cast<bl_amd_song_result *>(c->results.p);
  const bool in_place = esz == 2 && !pcm_is_s32 && host_mode() == BL_AMD_HOST_REGISTERED;

  /* waves of songs of at most WAVE_BYTES of PCM each (one song may exceed it): large enough
   * that the ~20 ms latency of the serial envelope tail (paid once per wave) stays below
   * the wave's transfer time, small enough for two pinned and two device buffers */
  const size_t WAVE_BYTES = (size_t)2 << 30;
  int begin = 0, wave = 0;
  int rc = BL_OK;
  /* The scratch workspace is per context, so the waves' kernels follow each other (event chain
   * inside blr_analyze_device); the copies of wave w+1 overlap the kernels of wave w.  The host
   * only ever waits for a wave's TRANSFER (its pinned buffer is free again), never for its
   * kernels: the device arena is reused in stream order.  Waiting for the kernels put the
   * ~20 ms latency of the serial envelope tail between two transfers (44 instead of 55 GB/s). */
  hipEvent_t done[2] = {nullptr, nullptr};
  for (int k = 0; k < 2 && rc == BL_OK; ++k) /* every exit below goes through the cleanup loop */
    if (hipEventCreateWithFlags(&done[k], hipEventDisableTiming) != hipSuccess) {
      fprintf(stderr, "bliss_amd: hipEventCreateWithFlags failed (%s:%d)\n", __FILE__, __LINE__);
      done[k] = nullptr;
      rc = BL_UNEXPECTED;
    }
  bool used[2] = {false, false};
  while (begin < n_songs && rc == BL_OK) {
    const int k = wave & 1;
    size_t elems = 0, elems22 = 0;
    int end = begin;
    std::vector<bl_amd_song_desc> desc;   /* the staged songs (native rate when converting) */
    std::vector<bl_amd_song_desc> desc22; /* converting: the songs as the analysis sees them */
    std::vector<bl_amd_resample_desc> rdesc;
    while (end < n_songs) {
      const size_t need = ((size_t)n_samples[end] + 7) & ~(size_t)7;
      if (end > begin && (elems + need) * esz > WAVE_BYTES) break;
      bl_amd_song_desc d;
      d.pcm_offset = elems; d.n_samples = n_samples[end]; d.channels = channels[end];
      d.duration = duration[end];
      desc.push_back(d);
      if (convert) {
        const size_t frames = (size_t)(n_samples[end] / channels[end]);
        const size_t of = bl_rs_out_frames(&geo, frames, nullptr);
        bl_amd_resample_desc r;
        r.in_offset = elems; r.out_offset = elems22; r.frames = (int32_t)frames; r.channels = channels[end];
        rdesc.push_back(r);
        bl_amd_song_desc d2;
        d2.pcm_offset = elems22; d2.n_samples = (int32_t)(2 * of); d2.channels = 2; d2.duration = duration[end];
        desc22.push_back(d2);
        elems22 += (2 * of + 7) & ~(size_t)7;
      }
      elems += need;
      ++end;
    }
    const size_t bytes = elems * esz + 64;
    const bool trace = getenv("BL_AMD_HOST_TRACE") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_a = now();
    if (used[k]) { /* pinned buffer k is free again once the transfer of wave-2 has finished */
      if (hipEventSynchronize(done[k]) != hipSuccess) { rc = BL_UNEXPECTED; break; }
      unregister_wave(c, k);
    }
    const double t_b = now();
    if (blr_ensure(c->arena[k], bytes) != BL_OK) { rc = BL_UNEXPECTED; break; }
    hipStream_t s = c->streams[k];
    int16_t *d_arena = static_cast<int16_t *>(c->arena[k].p);
    if (in_place) {
      /* pin the caller's buffers where they are (hipHostRegister keeps free() valid for the
       * caller) and copy each song straight to its place in the arena */
      for (size_t i = 0; i < desc.size() && rc == BL_OK; ++i) {
        void *hp = const_cast<void *>(h_pcm[begin + i]);
        const size_t nb = (size_t)desc[i].n_samples * 2;
        if (hipHostRegister(hp, nb, hipHostRegisterDefault) == hipSuccess) c->registered[k].push_back(hp);
        else (void)hipGetLastError(); /* already registered / unpinnable: the copy still works, staged by the runtime */
        if (hipMemcpyAsync(d_arena + desc[i].pcm_offset, hp, nb, hipMemcpyHostToDevice, s) != hipSuccess)
          rc = BL_UNEXPECTED;
      }
      if (rc != BL_OK) break;
    } else {
      if (c->pinned_cap[k] < bytes) {
        if (c->pinned[k]) (void)hipHostFree(c->pinned[k]);
        c->pinned[k] = nullptr; c->pinned_cap[k] = 0;
        if (hipHostMalloc(&c->pinned[k], bytes, hipHostMallocDefault) != hipSuccess) { rc = BL_UNEXPECTED; break; }
        c->pinned_cap[k] = bytes;
      }
      unsigned char *stage = static_cast<unsigned char *>(c->pinned[k]);
      /* staging copy on several host threads: one thread moves ~10 GB/s, the link takes more.
       * 32-bit sources are narrowed here (>> 16), so the link only ever carries s16. */
      const int n_thr = (int)std::min<size_t>((size_t)stage_threads(), std::max<size_t>(1, elems * esz / ((size_t)32 << 20)));
      auto copy_range = [&](int t) {
        for (size_t i = (size_t)t; i < desc.size(); i += (size_t)n_thr) {
          const size_t n = (size_t)desc[i].n_samples, padded = (n + 7) & ~(size_t)7;
          if (esz == 4) { /* wide source on its way to the converter: all 32 bits */
            int32_t *dst = reinterpret_cast<int32_t *>(stage) + desc[i].pcm_offset;
            memcpy(dst, h_pcm[begin + i], n * 4);
            for (size_t z = n; z < padded; ++z) dst[z] = 0;
            continue;
          }
          int16_t *dst = reinterpret_cast<int16_t *>(stage) + desc[i].pcm_offset;
          if (pcm_is_s32) {
            const int32_t *src = static_cast<const int32_t *>(h_pcm[begin + i]);
            for (size_t j = 0; j < n; ++j) dst[j] = (int16_t)(src[j] >> 16);
          } else {
            memcpy(dst, h_pcm[begin + i], n * 2);
          }
          for (size_t z = n; z < padded; ++z) dst[z] = 0;
        }
      };
      std::vector<std::thread> pool;
      for (int t = 1; t < n_thr; ++t) pool.emplace_back(copy_range, t);
      copy_range(0);
      for (auto &th : pool) th.join();
      if (trace) fprintf(stderr, "wave %d: wait %.1f ms, stage copy %.1f ms (%zu MB, %d threads)\n", wave,
                         1e3 * (t_b - t_a), 1e3 * (now() - t_b), elems * esz >> 20, n_thr);
      if (hipMemcpyAsync(d_arena, stage, elems * esz, hipMemcpyHostToDevice, s) != hipSuccess) { rc = BL_UNEXPECTED; break; }
    }
    if (hipEventRecord(done[k], s) != hipSuccess) { rc = BL_UNEXPECTED; break; }
    const int16_t *d_pcm = d_arena;
    const bl_amd_song_desc *wave_desc = desc.data();
    if (convert) { /* native-rate arena -> 22 050 Hz arena of the same wave slot, same stream */
      if (blr_ensure(c->arena22[k], elems22 * 2 + 64) != BL_OK) { rc = BL_UNEXPECTED; break; }
      int16_t *d22 = static_cast<int16_t *>(c->arena22[k].p);
      if (blr_resample_device(c, d_arena, esz == 4, rdesc.data(), (int)rdesc.size(), in_rate, d22, s) != BL_OK) {
        rc = BL_UNEXPECTED;
        break;
      }
      d_pcm = d22;
      wave_desc = desc22.data();
    }
    if (blr_analyze_device(c, d_pcm, wave_desc, (int)desc.size(), d_res + begin, s, 7) != BL_OK) {
      rc = BL_UNEXPECTED;
      break;
    }
    used[k] = true;
    begin = end;
    ++wave;
  }
  for (int k = 0; k < 2; ++k) {
    if (c->streams[k] && hipStreamSynchronize(c->streams[k]) != hipSuccess) rc = BL_UNEXPECTED;
    unregister_wave(c, k);
    if (done[k]) (void)hipEventDestroy(done[k]);
  }
  if (rc == BL_OK && h_results &&
      hipMemcpy(h_results, d_res, sizeof(bl_amd_song_result) * (size_t)n_songs, hipMemcpyDeviceToHost) != hipSuccess)
    rc = BL_UNEXPECTED;
  if (d_res_out) *d_res_out = d_res;
  return rc;
}

/* ========================================================================= */
/* C-ABI                                                                      */

extern "C" {

int bl_amd_device_count(void) {
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess) return 0;
  return count;
}

int bl_amd_init(int device) {
  const int prev = tl_device;
  tl_device = device;
  bl_amd_ctx *c = blr_default_ctx();
  if (!c) {
    tl_device = prev;
    return BL_UNEXPECTED;
  }
  int expect = -1;
  g_process_device.compare_exchange_strong(expect, device);
  return BL_OK;
}

int bld_ready(void) { return blr_default_ctx() ? BL_OK : BL_UNEXPECTED; }

int bl_amd_ctx_create(int device, bl_amd_ctx **out) {
  if (!out) return BL_UNEXPECTED;
  *out = nullptr;
  bl_amd_ctx *c = new bl_amd_ctx();
  if (ctx_init(c, device) != BL_OK) {
    ctx_release(c);
    delete c;
    return BL_UNEXPECTED;
  }
  *out = c;
  return BL_OK;
}

void bl_amd_ctx_destroy(bl_amd_ctx *ctx) {
  if (!ctx) return;
  {
    std::lock_guard<std::mutex> lk(ctx->mu);
    ctx_release(ctx);
  }
  delete ctx;
}

int bl_amd_ctx_device(const bl_amd_ctx *ctx) { return ctx ? ctx->device : BL_UNEXPECTED; }

void bl_amd_profile(int enable) {
  bl_amd_ctx *c = blr_default_ctx();
  if (!c) return;
  std::lock_guard<std::mutex> lk(c->mu);
  c->prof = enable != 0;
}

void bl_amd_profile_reset(void) {
  bl_amd_ctx *c = blr_default_ctx();
  if (!c) return;
  std::lock_guard<std::mutex> lk(c->mu);
  DevGuard dg(c->device);
  prof_collect(c);
  for (int k = 0; k < PK_COUNT; ++k) { c->prof_ms[k] = 0; c->prof_n[k] = 0; }
}

double bl_amd_profile_ms(const char *name, int *launches) {
  static const char *const kProfNames[PK_COUNT] = {"pcm_scan",    "amp_finish", "freq_frames", "freq_finish",
                                                   "env_windows", "env_tail",   "distance",    "freq_scan"};
  if (launches) *launches = 0;
  bl_amd_ctx *c = blr_default_ctx();
  if (!c || !name) return -1.0;
  std::lock_guard<std::mutex> lk(c->mu);
  DevGuard dg(c->device);
  prof_collect(c);
  for (int k = 0; k < PK_COUNT; ++k)
    if (!strcmp(name, kProfNames[k])) {
      if (launches) *launches = c->prof_n[k];
      return c->prof_ms[k];
    }
  return -1.0;
}

int bl_amd_ctx_analyze_batch_device(bl_amd_ctx *ctx, const int16_t *d_pcm, const bl_amd_song_desc *h_desc,
                                    int n_songs, bl_amd_song_result *d_results, void *stream) {
  if (!ctx || n_songs <= 0 || !d_pcm || !h_desc || !d_results) return BL_UNEXPECTED;
  std::lock_guard<std::mutex> lk(ctx->mu);
  DevGuard dg(ctx->device);
  if (!dg.ok) return BL_UNEXPECTED;
  return blr_analyze_device(ctx, d_pcm, h_desc, n_songs, d_results, static_cast<hipStream_t>(stream), 7);
}

int bl_amd_analyze_batch_device(const int16_t *d_pcm, const bl_amd_song_desc *h_desc, int n_songs,
                                bl_amd_song_result *d_results, void *stream) {
  return bl_amd_ctx_analyze_batch_device(blr_default_ctx(), d_pcm, h_desc, n_songs, d_results, stream);
}

int bl_amd_synth_pcm_device(int16_t *d_pcm, const bl_amd_song_desc *h_desc, int n_songs,
                            uint32_t seed_base, uint32_t sample_rate, void *stream) {
  if (n_songs <= 0 || !d_pcm || !h_desc) return BL_UNEXPECTED;
  bl_amd_ctx *c = blr_default_ctx();
  if (!c) return BL_UNEXPECTED;
  std::lock_guard<std::mutex> lk(c->mu);
  DevGuard dg(c->device);
  hipStream_t s = static_cast<hipStream_t>(stream);
  for (int i = 0; i < n_songs; ++i)
    if (validate_desc(h_desc[i], i) != BL_OK) return BL_UNEXPECTED;
  /* the descriptor block of the workspace is shared with the analysis: same hand-over */
  if (c->ws_used) BL_HIP_CHECK(hipStreamWaitEvent(s, c->ev_ws, 0));
  if (blr_ensure(c->songs, sizeof(bl_dsong) * (size_t)n_songs) != BL_OK) return BL_UNEXPECTED;
  bl_pin_slot *slot = nullptr;
  if (ring_get(c, sizeof(bl_dsong) * (size_t)n_songs, &slot) != BL_OK) return BL_UNEXPECTED;
  bl_dsong *hs = static_cast<bl_dsong *>(slot->p);
  const int G = BL_GROUP_SONGS_MAX;
  std::vector<int> max_n((n_songs + G - 1) / G);
  for (int b = 0, gi = 0; b < n_songs; b += G, ++gi) {
    long long env_total;
    fill_group(h_desc + b, std::min(G, n_songs - b), hs + b, env_total, max_n[gi]);
  }
  bl_dsong *d_songs = static_cast<bl_dsong *>(c->songs.p);
  BL_HIP_CHECK(hipMemcpyAsync(d_songs, hs, sizeof(bl_dsong) * (size_t)n_songs, hipMemcpyHostToDevice, s));
  BL_HIP_CHECK(hipEventRecord(slot->ev, s));
  slot->busy = true;
  for (int b = 0, gi = 0; b < n_songs; b += G, ++gi)
    if (blk_synth(s, d_pcm, d_songs + b, std::min(G, n_songs - b), max_n[gi], c->n_cu,
                  seed_base + (uint32_t)b, sample_rate) != BL_OK)
      return BL_UNEXPECTED;
  BL_HIP_CHECK(hipEventRecord(c->ev_ws, s));
  c->ws_used = true;
  return BL_OK;
}

static int matrix_device(const struct force_vector_s *d_vecs, int n, int row_begin, int n_rows,
                         float *d_out, void *stream, bool cosine) {
  if (n <= 0 || n_rows <= 0 || row_begin < 0 || row_begin + n_rows > n || !d_vecs || !d_out)
    return BL_UNEXPECTED;
  bl_amd_ctx *c = blr_default_ctx();
  if (!c) return BL_UNEXPECTED;
  std::lock_guard<std::mutex> lk(c->mu);
  DevGuard dg(c->device);
  return blk_pairwise(static_cast<hipStream_t>(stream), d_vecs, n, row_begin, n_rows, d_out, cosine,
                      c->prof ? mark_cb : nullptr, c);
}

int bl_amd_distance_matrix_device(const struct force_vector_s *d_vecs, int n, int row_begin,
                                  int n_rows, float *d_out, void *stream) {
  return matrix_device(d_vecs, n, row_begin, n_rows, d_out, stream, false);
}

int bl_amd_cosine_matrix_device(const struct force_vector_s *d_vecs, int n, int row_begin,
                                int n_rows, float *d_out, void *stream) {
  return matrix_device(d_vecs, n, row_begin, n_rows, d_out, stream, true);
}

int bl_amd_playlist_device(const struct force_vector_s *d_vecs, int n, int seed_index,
                           int32_t *d_order, float *d_dist, void *stream) {
  if (n <= 0 || seed_index < 0 || seed_index >= n || !d_vecs || !d_order || !d_dist)
    return BL_UNEXPECTED;
  bl_amd_ctx *c = blr_default_ctx();
  if (!c) return BL_UNEXPECTED;
  DevGuard dg(c->device);
  return blk_playlist(static_cast<hipStream_t>(stream), d_vecs, n, seed_index, d_order, d_dist);
}

int bl_amd_playlist_host(const struct force_vector_s *h_vecs, int n, int seed_index,
                         int32_t *h_order, float *h_dist) {
  if (n <= 0 || !h_vecs || !h_order) return BL_UNEXPECTED;
  bl_amd_ctx *c = blr_default_ctx();
  if (!c) return BL_UNEXPECTED;
  DevGuard dg(c->device);
  void *dv = nullptr, *dord = nullptr, *dd = nullptr;
  int rc = BL_UNEXPECTED;
  if (hipMalloc(&dv, sizeof(struct force_vector_s) * (size_t)n) == hipSuccess &&
      hipMalloc(&dord, sizeof(int32_t) * (size_t)n) == hipSuccess &&
      hipMalloc(&dd, sizeof(float) * (size_t)n) == hipSuccess &&
      hipMemcpy(dv, h_vecs, sizeof(struct force_vector_s) * (size_t)n, hipMemcpyHostToDevice) == hipSuccess &&
      bl_amd_playlist_device(static_cast<struct force_vector_s *>(dv), n, seed_index,
                             static_cast<int32_t *>(dord), static_cast<float *>(dd), nullptr) == BL_OK &&
      hipMemcpy(h_order, dord, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost) == hipSuccess &&
      (!h_dist || hipMemcpy(h_dist, dd, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost) == hipSuccess))
    rc = BL_OK;
  if (dv) (void)hipFree(dv);
  if (dord) (void)hipFree(dord);
  if (dd) (void)hipFree(dd);
  return rc;
}

int bl_amd_selftest_sqrt(uint64_t counts[3]) {
  if (!counts) return BL_UNEXPECTED;
  bl_amd_ctx *c = blr_default_ctx();
  if (!c) return BL_UNEXPECTED;
  DevGuard dg(c->device);
  unsigned long long *d = nullptr;
  BL_HIP_CHECK(hipMalloc(&d, 3 * sizeof(unsigned long long)));
  int rc = BL_UNEXPECTED;
  unsigned long long h[3] = {0, 0, 0};
  if (hipMemset(d, 0, sizeof h) == hipSuccess &&
      blk_sqrt_sweep(nullptr, 0ull, 1ull << 32, d, c->n_cu) == BL_OK &&   /* every f32 bit pattern */
      hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost) == hipSuccess) {
    for (int k = 0; k < 3; ++k) counts[k] = h[k];
    rc = BL_OK;
  }
  (void)hipFree(d);
  return rc;
}

int bl_amd_selftest_cos(uint64_t counts[6], uint64_t triples) {
  if (!counts) return BL_UNEXPECTED;
  bl_amd_ctx *c = blr_default_ctx();
  if (!c) return BL_UNEXPECTED;
  DevGuard dg(c->device);
  unsigned long long *d = nullptr;
  BL_HIP_CHECK(hipMalloc(&d, 6 * sizeof(unsigned long long)));
  int rc = BL_UNEXPECTED;
  unsigned long long h[6] = {0, 0, 0, 0, 0, 0};
  const unsigned long long threads = (unsigned long long)c->n_cu * 8 * 256;
  const int per_thread = (int)std::min<unsigned long long>((triples / 8 + threads - 1) / threads, 1u << 24);
  if (hipMemset(d, 0, sizeof h) == hipSuccess &&
      blk_cos_sweep(nullptr, 0x5eedc05ull, std::max(per_thread, 1), d, c->n_cu) == BL_OK &&
      hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost) == hipSuccess) {
    for (int k = 0; k < 6; ++k) counts[k] = h[k];
    rc = BL_OK;
  }
  (void)hipFree(d);
  return rc;
}

static int matrix_host(const struct force_vector_s *h_vecs, int n, float *h_out, bool cosine) {
  if (n <= 0 || !h_vecs || !h_out) return BL_UNEXPECTED;
  bl_amd_ctx *c = blr_default_ctx();
  if (!c) return BL_UNEXPECTED;
  DevGuard dg(c->device);
  void *dv = nullptr, *dout = nullptr;
  BL_HIP_CHECK(hipMalloc(&dv, sizeof(struct force_vector_s) * (size_t)n));
  if (hipMalloc(&dout, sizeof(float) * (size_t)n * n) != hipSuccess) { (void)hipFree(dv); return BL_UNEXPECTED; }
  int rc = BL_UNEXPECTED;
  if (hipMemcpy(dv, h_vecs, sizeof(struct force_vector_s) * (size_t)n, hipMemcpyHostToDevice) == hipSuccess &&
      matrix_device(static_cast<struct force_vector_s *>(dv), n, 0, n, static_cast<float *>(dout),
                    nullptr, cosine) == BL_OK &&
      hipMemcpy(h_out, dout, sizeof(float) * (size_t)n * n, hipMemcpyDeviceToHost) == hipSuccess)
    rc = BL_OK;
  (void)hipFree(dv);
  (void)hipFree(dout);
  return rc;
}

int bl_amd_distance_matrix_host(const struct force_vector_s *h_vecs, int n, float *h_out) {
  return matrix_host(h_vecs, n, h_out, false);
}
int bl_amd_cosine_matrix_host(const struct force_vector_s *h_vecs, int n, float *h_out) {
  return matrix_host(h_vecs, n, h_out, true);
}

int bl_amd_set_host_transfer(int mode) {
  if (mode != BL_AMD_HOST_STAGED && mode != BL_AMD_HOST_REGISTERED) return BL_UNEXPECTED;
  g_host_mode.store(mode);
  return BL_OK;
}

int bl_amd_ctx_analyze_batch_host(bl_amd_ctx *ctx, const int16_t *const *h_pcm, const int32_t *n_samples,
                                  const int32_t *channels, const uint64_t *duration, int n_songs,
                                  bl_amd_song_result *h_results) {
  if (!ctx || n_songs <= 0 || !h_pcm || !n_samples || !channels || !duration || !h_results)
    return BL_UNEXPECTED;
  std::lock_guard<std::mutex> lk(ctx->mu);
  DevGuard dg(ctx->device);
  if (!dg.ok) return BL_UNEXPECTED;
  return blr_analyze_host(ctx, reinterpret_cast<const void *const *>(h_pcm), 0, n_samples, channels,
                          duration, n_songs, 0, h_results, nullptr);
}

int bl_amd_analyze_batch_host(const int16_t *const *h_pcm, const int32_t *n_samples,
                              const int32_t *channels, const uint64_t *duration, int n_songs,
                              bl_amd_song_result *h_results) {
  return bl_amd_ctx_analyze_batch_host(blr_default_ctx(), h_pcm, n_samples, channels, duration, n_songs,
                                       h_results);
}

/* The corpus loop of ref python/examples/make_m3u_playlist.py:51-72 / examples/analyze.c:17 as one
 * call: `for file: bl_analyze(file, &song)`.  Decoding (bl_audio_decode: FLAC / WAV readers, rate
 * converter) runs on n_threads host threads that stay a bounded number of files ahead; the calling
 * thread takes the decoded songs in file order, wave by wave, through the pinned-staging host path
 * (blr_analyze_host), so that the next wave is being decoded while this one is transferred and
 * analysed.  Every songs[i] ends up exactly as bl_analyze(filenames[i], &songs[i]) leaves it. */
int bl_amd_analyze_files(const char *const *filenames, int n_files, struct bl_song *songs, int *codes,
                         int n_threads, int keep_pcm) {
  if (n_files <= 0 || !filenames || !songs) return BL_UNEXPECTED;
  bl_amd_ctx *c = blr_default_ctx();
  if (!c) return BL_UNEXPECTED;
  if (n_threads <= 0) n_threads = (int)std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 32u);
  n_threads = std::min(n_threads, n_files);
  const size_t WAVE_BYTES = (size_t)1 << 30; /* decoded PCM per wave */
  /* The decoders run ahead of the wave being analysed by at most AHEAD_BYTES of decoded PCM (and AHEAD_FILES
   * files): a byte bound, because a file bound alone lets 768 ten-minute songs (40 GB) pile up on the host.
   * The file the consumer is waiting for is always taken, whatever the budget says, so the two cannot deadlock. */
  const size_t AHEAD_BYTES = 3 * WAVE_BYTES; /* bounds the READ-AHEAD only: with keep_pcm the caller keeps every decoded
                                               * sample_array (include/bliss_amd.h), and that total is the caller's */
  const int WAVE_FILES = 512, AHEAD_FILES = 768; /* AHEAD_FILES >= WAVE_FILES: a wave never waits for a file the decoders may not take */

  std::mutex mu;
  std::condition_variable cv;
  std::vector<signed char> state((size_t)n_files, 0); /* 0 pending, 1 decoded, -1 failed */
  int next = 0, limit = std::min(n_files, AHEAD_FILES); /* decoders take indices below `limit` */
  int want = 0;          /* the file the consumer needs next */
  size_t ahead_bytes = 0; /* decoded PCM not yet released by the consumer */
  bool stop = false;
  auto decoder = [&] {
    for (;;) {
      int i;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return stop || next >= n_files || (next < limit && (ahead_bytes < AHEAD_BYTES || next <= want)); });
        if (stop || next >= n_files) return;
        i = next++;
      }
      /* bl_audio_decode initialises every field of songs[i] itself and fails on a NULL name: the caller's
       * structs may be uninitialised (ref tests/test_analyze.c:27-28), so it must run for every index */
      const int rc = bl_audio_decode(filenames[i], &songs[i]);
      if (rc != BL_OK) fprintf(stderr, "Couldn't decode song\n"); /* ref src/analyze.c:83 */
      {
        std::lock_guard<std::mutex> lk(mu);
        state[(size_t)i] = rc == BL_OK ? 1 : -1;
        if (rc == BL_OK && songs[i].sample_array) ahead_bytes += (size_t)songs[i].nSamples * 2;
      }
      cv.notify_all();
    }
  };
  std::vector<std::thread> pool;
  for (int t = 0; t < n_threads; ++t) pool.emplace_back(decoder);

  int done = 0, ok_count = 0, rc_all = BL_OK;
  std::vector<const void *> pcm;
  std::vector<int32_t> ns, ch;
  std::vector<uint64_t> du;
  std::vector<int> idx;
  std::vector<bl_amd_song_result> res;
  while (done < n_files) {
    /* the next wave: files in order until the byte or file budget is reached; wait for each decode */
    pcm.clear(); ns.clear(); ch.clear(); du.clear(); idx.clear();
    size_t bytes = 0;
    int e = done;
    while (e < n_files && (int)(e - done) < WAVE_FILES && bytes < WAVE_BYTES) {
      {
        std::unique_lock<std::mutex> lk(mu);
        if (state[(size_t)e] == 0) {
          want = e;
          cv.notify_all();
          cv.wait(lk, [&] { return state[(size_t)e] != 0; });
        }
      }
      if (codes) codes[e] = BL_UNEXPECTED;
      if (state[(size_t)e] == 1) {
        const struct bl_song &sg = songs[e];
        /* what run_song() of bl_api.c and validate_desc() accept; anything else is reported per file */
        if (sg.sample_array && sg.nSamples >= 5120 && (sg.channels == 1 || sg.channels == 2) && sg.duration > 0) {
          pcm.push_back(sg.sample_array); ns.push_back(sg.nSamples); ch.push_back(sg.channels);
          du.push_back(sg.duration); idx.push_back(e);
          bytes += (size_t)sg.nSamples * 2;
        } else {
          fprintf(stderr, "bliss_amd: %s not analysable (need >= 5120 s16 samples, 1 or 2 channels, >= 1 s)\n",
                  filenames[e]);
        }
      }
      ++e;
    }
    { /* let the decoders run ahead of the wave that is about to be analysed */
      std::lock_guard<std::mutex> lk(mu);
      limit = std::min(n_files, e + AHEAD_FILES);
    }
    cv.notify_all();
    if (!idx.empty()) {
      res.assign(idx.size(), bl_amd_song_result());
      int rc;
      {
        std::lock_guard<std::mutex> lk(c->mu);
        DevGuard dg(c->device);
        rc = dg.ok ? blr_analyze_host(c, pcm.data(), 0, ns.data(), ch.data(), du.data(), (int)idx.size(), 0,
                                      res.data(), nullptr)
                   : BL_UNEXPECTED;
      }
      if (rc != BL_OK) rc_all = BL_UNEXPECTED;
      for (size_t k = 0; k < idx.size(); ++k) {
        struct bl_song &sg = songs[idx[k]];
        if (rc == BL_OK && res[k].status == BL_OK) {
          sg.force_vector = res[k].v;            /* ref src/analyze.c:63-66 */
          sg.force = res[k].force;               /* ref :68-72 */
          sg.calm_or_loud = res[k].calm_or_loud; /* ref :73-79 */
          if (codes) codes[idx[k]] = res[k].calm_or_loud;
          ++ok_count;
        } else {
          fprintf(stderr, "Couldn't analyse song on the HIP device\n");
        }
      }
    }
    size_t released = 0;
    for (int i = done; i < e; ++i) {
      if (state[(size_t)i] == 1 && songs[i].sample_array) released += (size_t)songs[i].nSamples * 2;
      if (!keep_pcm) {
        free(songs[i].sample_array);
        songs[i].sample_array = NULL;
      }
    }
    {
      std::lock_guard<std::mutex> lk(mu);
      ahead_bytes -= std::min(released, ahead_bytes);
    }
    cv.notify_all();
    done = e;
  }
  {
    std::lock_guard<std::mutex> lk(mu);
    stop = true;
  }
  cv.notify_all();
  for (auto &t : pool) t.join();
  return rc_all == BL_OK ? ok_count : BL_UNEXPECTED;
}

int bl_amd_analyze_batch_host_s32(const int32_t *const *h_pcm, const int32_t *n_samples,
                                  const int32_t *channels, const uint64_t *duration, int n_songs,
                                  bl_amd_song_result *h_results) {
  if (n_songs <= 0 || !h_pcm || !n_samples || !channels || !duration || !h_results) return BL_UNEXPECTED;
  bl_amd_ctx *c = blr_default_ctx();
  if (!c) return BL_UNEXPECTED;
  std::lock_guard<std::mutex> lk(c->mu);
  DevGuard dg(c->device);
  return blr_analyze_host(c, reinterpret_cast<const void *const *>(h_pcm), 1, n_samples, channels,
                          duration, n_songs, 0, h_results, nullptr);
}

int bl_amd_analyze_batch_host_rate(const void *const *h_pcm, int pcm_is_s32, const int32_t *n_samples,
                                   const int32_t *channels, const uint64_t *duration, int n_songs,
                                   int sample_rate, bl_amd_song_result *h_results) {
  if (n_songs <= 0 || !h_pcm || !n_samples || !channels || !duration || !h_results || sample_rate <= 0)
    return BL_UNEXPECTED;
  bl_amd_ctx *c = blr_default_ctx();
  if (!c) return BL_UNEXPECTED;
  std::lock_guard<std::mutex> lk(c->mu);
  DevGuard dg(c->device);
  if (!dg.ok) return BL_UNEXPECTED;
  return blr_analyze_host(c, h_pcm, pcm_is_s32 != 0, n_samples, channels, duration, n_songs, sample_rate,
                          h_results, nullptr);
}

int bl_amd_narrow_s32_device(const int32_t *d_in, int16_t *d_out, size_t n, void *stream) {
  if (!d_in || !d_out) return BL_UNEXPECTED;
  if (n == 0) return BL_OK;
  bl_amd_ctx *c = blr_default_ctx();
  if (!c) return BL_UNEXPECTED;
  DevGuard dg(c->device);
  return blk_narrow_s32(static_cast<hipStream_t>(stream), d_in, d_out, n, c->n_cu);
}

/* ---- rate conversion (bl_resample.c on the host, bl_rs_kernels.hip on the device) ---- */

size_t bl_amd_resample_out_frames(size_t frames, int in_rate) {
  bl_rs_plan p;
  if (bl_rs_plan_geometry(&p, BL_RS_OUT_RATE, in_rate)) return 0;
  return bl_rs_out_frames(&p, frames, nullptr);
}

int bl_amd_resample_host(const void *in, int in_is_s32, size_t frames, int channels, int in_rate,
                         int16_t **out, size_t *out_frames) {
  if (!out || !out_frames) return BL_UNEXPECTED;
  return bl_resample_to_stereo_s16(in, in_is_s32, frames, channels, in_rate, BL_RS_OUT_RATE, out, out_frames);
}

int bl_amd_ctx_resample_batch_device(bl_amd_ctx *c, const void *d_in, int in_is_s32,
                                     const bl_amd_resample_desc *h_desc, int n_songs, int in_rate,
                                     int16_t *d_out, void *stream) {
  if (!c || !d_in || !d_out || !h_desc || n_songs <= 0 || in_rate <= 0 || (in_is_s32 != 0 && in_is_s32 != 1))
    return BL_UNEXPECTED; /* float sources: host form only */
  std::lock_guard<std::mutex> lk(c->mu);
  DevGuard dg(c->device);
  if (!dg.ok) return BL_UNEXPECTED;
  return blr_resample_device(c, d_in, in_is_s32, h_desc, n_songs, in_rate, d_out, static_cast<hipStream_t>(stream));
}

int bl_amd_resample_batch_device(const void *d_in, int in_is_s32, const bl_amd_resample_desc *h_desc,
                                 int n_songs, int in_rate, int16_t *d_out, void *stream) {
  return bl_amd_ctx_resample_batch_device(blr_default_ctx(), d_in, in_is_s32, h_desc, n_songs, in_rate,
                                          d_out, stream);
}


/* ---- helpers behind the reference-API shims of bl_api.c ---- */
int bld_analyze_one_host(const int16_t *h_pcm, int n, int channels, uint64_t duration, int what,
                         bl_amd_song_result *res) {
  if (!h_pcm || !res) return BL_UNEXPECTED;
  bl_amd_ctx *c = blr_default_ctx();
  if (!c) return BL_UNEXPECTED;
  std::lock_guard<std::mutex> lk(c->mu);
  DevGuard dg(c->device);
  if (!c->streams[0]) BL_HIP_CHECK(hipStreamCreateWithFlags(&c->streams[0], hipStreamNonBlocking));
  hipStream_t s = c->streams[0];
  const size_t elems = ((size_t)n + 7) & ~(size_t)7;
  if (blr_ensure(c->arena[0], elems * 2 + 64) != BL_OK) return BL_UNEXPECTED;
  if (blr_ensure(c->results, sizeof(bl_amd_song_result)) != BL_OK) return BL_UNEXPECTED;
  if (c->ws_used) BL_HIP_CHECK(hipStreamWaitEvent(s, c->ev_ws, 0));
  BL_HIP_CHECK(hipMemcpyAsync(c->arena[0].p, h_pcm, (size_t)n * 2, hipMemcpyHostToDevice, s));
  BL_HIP_CHECK(hipMemsetAsync(c->results.p, 0, sizeof(bl_amd_song_result), s));
  bl_amd_song_desc d;
  d.pcm_offset = 0; d.n_samples = n; d.channels = channels; d.duration = duration;
  if (blr_analyze_device(c, static_cast<const int16_t *>(c->arena[0].p), &d, 1,
                         static_cast<bl_amd_song_result *>(c->results.p), s, what) != BL_OK)
    return BL_UNEXPECTED;
  BL_HIP_CHECK(hipMemcpyAsync(res, c->results.p, sizeof(bl_amd_song_result), hipMemcpyDeviceToHost, s));
  BL_HIP_CHECK(hipStreamSynchronize(s));
  return BL_OK;
}

int bld_mean_variance_host(const int16_t *h_pcm, int n, int have_mean, int mean_in, int *mean_out,
                           int *variance_out) {
  if (!h_pcm || n <= 0) return BL_UNEXPECTED;
  bl_amd_ctx *c = blr_default_ctx();
  if (!c) return BL_UNEXPECTED;
  std::lock_guard<std::mutex> lk(c->mu);
  DevGuard dg(c->device);
  if (!c->streams[0]) BL_HIP_CHECK(hipStreamCreateWithFlags(&c->streams[0], hipStreamNonBlocking));
  hipStream_t s = c->streams[0];
  const size_t elems = ((size_t)n + 7) & ~(size_t)7;
  if (blr_ensure(c->arena[0], elems * 2 + 64) != BL_OK) return BL_UNEXPECTED;
  if (blr_ensure(c->songs, sizeof(bl_dsong)) != BL_OK) return BL_UNEXPECTED;
  if (blr_ensure(c->stats, sizeof(bl_dstats)) != BL_OK) return BL_UNEXPECTED;
  if (blr_ensure(c->hist, sizeof(unsigned) * BL_HIST_BINS) != BL_OK) return BL_UNEXPECTED;
  /* same workspace as the batches: wait for the previous user on the device, hand it on after */
  if (c->ws_used) BL_HIP_CHECK(hipStreamWaitEvent(s, c->ev_ws, 0));
  BL_HIP_CHECK(hipMemcpyAsync(c->arena[0].p, h_pcm, (size_t)n * 2, hipMemcpyHostToDevice, s));
  bl_pin_slot *slot = nullptr;
  if (ring_get(c, sizeof(bl_dsong), &slot) != BL_OK) return BL_UNEXPECTED;
  bl_dsong *hs = static_cast<bl_dsong *>(slot->p);
  memset(hs, 0, sizeof *hs);
  hs->n = n; hs->channels = 1; hs->duration = 1;
  BL_HIP_CHECK(hipMemcpyAsync(c->songs.p, hs, sizeof *hs, hipMemcpyHostToDevice, s));
  BL_HIP_CHECK(hipEventRecord(slot->ev, s));
  slot->busy = true;
  bl_dstats *d_stats = static_cast<bl_dstats *>(c->stats.p);
  const bl_dsong *d_songs = static_cast<const bl_dsong *>(c->songs.p);
  const int16_t *d_pcm = static_cast<const int16_t *>(c->arena[0].p);
  if (blk_scan_one(s, d_pcm, d_songs, d_stats, static_cast<unsigned *>(c->hist.p), n, c->n_cu) != BL_OK)
    return BL_UNEXPECTED;
  bl_dstats st;
  BL_HIP_CHECK(hipMemcpyAsync(&st, d_stats, sizeof st, hipMemcpyDeviceToHost, s));
  BL_HIP_CHECK(hipStreamSynchronize(s));
  /* ref helpers.c:30-37 */
  const int mean = have_mean ? mean_in : (int)(unsigned)(st.sum & 0xFFFFFFFFull) / n;
  if (mean_out) *mean_out = mean;
  if (variance_out) {
    /* always the exact wrapping form (ref helpers.c:39-49) for the stand-alone helper */
    st.mean = mean; st.wrap_pass = 1; st.wrap_acc = 0;
    BL_HIP_CHECK(hipMemcpyAsync(d_stats, &st, sizeof st, hipMemcpyHostToDevice, s));
    if (blk_variance_wrap_one(s, d_pcm, d_songs, d_stats, n, c->n_cu) != BL_OK) return BL_UNEXPECTED;
    BL_HIP_CHECK(hipMemcpyAsync(&st, d_stats, sizeof st, hipMemcpyDeviceToHost, s));
    BL_HIP_CHECK(hipStreamSynchronize(s));
    *variance_out = (int)(st.wrap_acc / n);
  }
  BL_HIP_CHECK(hipEventRecord(c->ev_ws, s));
  c->ws_used = true;
  return BL_OK;
}

/* diagnostic: the per-window energies (ref tempo_atk_sort.c:150, filtered_array) of the
 * most recent launch group, songs concatenated, nb_frames slots per song (last two unused) */
long long bl_amd_last_energies(float *h_out, long long max_elems) {
  bl_amd_ctx *c = blr_default_ctx();
  if (!c) return -1;
  std::lock_guard<std::mutex> lk(c->mu);
  DevGuard dg(c->device);
  if (!c->energies.p || c->last_env_total <= 0) return 0;
  const long long n = c->last_env_total < max_elems ? c->last_env_total : max_elems;
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  if (hipMemcpy(h_out, c->energies.p, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost) != hipSuccess)
    return -1;
  return n;
}

void bl_multi_shutdown(void); /* bl_multi.hip */

void bl_amd_shutdown(void) {
  bl_multi_shutdown();
  std::lock_guard<std::mutex> lk(g_mu);
  for (int d = 0; d < BL_MAX_DEVICES; ++d) {
    if (!g_default[d]) continue;
    {
      std::lock_guard<std::mutex> lk2(g_default[d]->mu);
      ctx_release(g_default[d]);
    }
    delete g_default[d];
    g_default[d] = nullptr;
  }
}

} /* extern "C" */
