/*
 * bl_multi.hip — the batch-of-songs mode across the GPUs of one node, from C
 * (bl_amd_analyze_corpus_multi and bl_amd_analyze_corpus_multi_device of include/bliss_amd.h).
 *
 * Songs are independent (ref src/analyze.c:33-86 keeps no cross-song state), so the corpus is
 * sharded by song.  One host thread and one context per rank (a rank = one entry of `devices`,
 * or one bl_amd_shard).  Two ways in:
 *   host corpus      the library shards (contiguous blocks for equal lengths, longest-processing-
 *                    time-first on the sample count otherwise) and every rank runs the ordinary
 *                    host-batch path (pinned staging, PCIe-bound) on its shard;
 *   resident corpus  the caller has placed each rank's songs in that rank's HBM (BASELINE
 *                    configs[2]: 8 192 three-minute songs = 260 GB per GPU cannot come from
 *                    host memory in one piece); every rank analyses its arena where it lies.
 * The only exchange is one all-gather of the 16-byte force vectors:
 *   BL_AMD_MULTI_GATHER_RCCL  ncclAllGather over xGMI, librccl loaded at first use (dlopen:
 *                             the single-device library does not depend on it, and the four
 *                             prototypes used are declared here, so neither does the build);
 *   BL_AMD_MULTI_GATHER_PEER  every rank copies its block straight into every peer's buffer
 *                             (hipMemcpyPeerAsync) — 128 KiB per rank at 8 192 songs is
 *                             latency-bound, a 7-peer write is as good as a ring, and it also
 *                             allows two ranks on one device (how the tests run N = 2 on a
 *                             1-GPU box).
 * Then rank r computes its row block of the N x N bl_distance matrix in its own HBM and, if
 * asked, copies it to the host matrix.  No other collective.  The exchange buffers belong to
 * the rank's context and only ever grow: a call allocates nothing once they are large enough.
 */
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <queue>
#include <thread>

#include "bl_runtime.h"

extern "C" int bl_amd_ctx_create(int device, bl_amd_ctx **out);
extern "C" void bl_amd_ctx_destroy(bl_amd_ctx *ctx);

namespace {

/* The part of RCCL's C API this file uses (rccl/rccl.h: ncclComm_t is an opaque pointer,
 * ncclSuccess = 0, ncclFloat = 7), declared here so that the library builds where the RCCL
 * headers are absent; the symbols come from dlopen at first use. */
typedef struct ncclComm *ncclComm_t;
typedef int ncclResult_t;
typedef int ncclDataType_t;
enum { BL_NCCL_SUCCESS = 0, BL_NCCL_FLOAT = 7 };

struct Barrier {
  std::mutex m;
  std::condition_variable cv;
  int n, waiting = 0, gen = 0;
  explicit Barrier(int nn) : n(nn) {}
  void wait() {
    std::unique_lock<std::mutex> lk(m);
    const int g = gen;
    if (++waiting == n) {
      waiting = 0;
      ++gen;
      cv.notify_all();
    } else {
      cv.wait(lk, [&] { return gen != g; });
    }
  }
};

struct Rccl {
  void *h = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  bool load() {
    if (h) return true;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *n : names)
      if ((h = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
    if (!h) {
      fprintf(stderr, "bliss_amd: cannot load librccl (%s); use BL_AMD_MULTI_GATHER_PEER\n", dlerror());
      return false;
    }
    CommInitAll = reinterpret_cast<decltype(CommInitAll)>(dlsym(h, "ncclCommInitAll"));
    CommDestroy = reinterpret_cast<decltype(CommDestroy)>(dlsym(h, "ncclCommDestroy"));
    AllGather = reinterpret_cast<decltype(AllGather)>(dlsym(h, "ncclAllGather"));
    GetErrorString = reinterpret_cast<decltype(GetErrorString)>(dlsym(h, "ncclGetErrorString"));
    if (!CommInitAll || !CommDestroy || !AllGather || !GetErrorString) {
      fprintf(stderr, "bliss_amd: librccl lacks an expected symbol\n");
      return false;
    }
    return true;
  }
};

struct MultiState {
  std::mutex mu;
  bl_amd_ctx *ctx[BL_MAX_DEVICES] = {nullptr};
  Rccl rccl;
  std::vector<int> comm_devices;
  std::vector<ncclComm_t> comms;
  void drop_comms() {
    for (ncclComm_t c : comms)
      if (c) (void)rccl.CommDestroy(c);
    comms.clear();
    comm_devices.clear();
  }
};
MultiState g_multi;

struct RankShared { /* what the ranks publish to each other */
  struct force_vector_s *d_gath[BL_MAX_DEVICES];
};

/* what every rank of one call has in common */
struct Call {
  int world, flags, n_songs;
  int m;                               /* padded songs per rank */
  const std::vector<int32_t> *order;   /* world * m entries: position in the output order, or -1 */
  const int *devices;                  /* device of rank r */
  const int *row0, *rows;              /* rank r computes matrix rows [row0[r], row0[r] + rows[r]) */
  float *h_matrix;                     /* may be NULL */
  Barrier *bar;
  std::atomic<int> *failed;
  RankShared *shared;
};

/* The exchange every rank runs once its shard's results are in d_res (cnt records, on the
 * rank's device, complete on stream s): extract the vectors, all-gather, bring them into the
 * output order, compute the row block (into d_rows_out if the caller gave one, else into the
 * context's buffer when a host matrix is wanted).  `ok_in` = 0 joins the barriers of a rank
 * that has already failed.  Returns with the stream idle. */
void rank_exchange(const Call &c, int rank, bl_amd_ctx *ctx, ncclComm_t comm, hipStream_t s,
                   const bl_amd_song_result *d_res, int cnt, float *d_rows_out, int ok_in) {
  int ok = ok_in;
  const int device = c.devices[rank], n = c.n_songs, W = c.world;
  const size_t blk = sizeof(struct force_vector_s) * (size_t)c.m;
  const size_t ord_bytes = sizeof(int32_t) * (size_t)c.m * W;
  const int my_rows = c.rows[rank], row0 = c.row0[rank];
  const bool want_rows = my_rows > 0 && (d_rows_out || c.h_matrix);
  if (ok)
    ok = blr_ensure(ctx->mx_my, blk) == BL_OK && blr_ensure(ctx->mx_gath, blk * W) == BL_OK &&
         blr_ensure(ctx->mx_all, sizeof(struct force_vector_s) * (size_t)n) == BL_OK &&
         blr_ensure(ctx->mx_order, ord_bytes) == BL_OK &&
         (!want_rows || d_rows_out || blr_ensure(ctx->mx_rows, sizeof(float) * (size_t)my_rows * n) == BL_OK);
  struct force_vector_s *d_my = static_cast<struct force_vector_s *>(ctx->mx_my.p);
  struct force_vector_s *d_gath = static_cast<struct force_vector_s *>(ctx->mx_gath.p);
  struct force_vector_s *d_all = static_cast<struct force_vector_s *>(ctx->mx_all.p);
  int32_t *d_order = static_cast<int32_t *>(ctx->mx_order.p);
  float *d_rows = d_rows_out ? d_rows_out : static_cast<float *>(ctx->mx_rows.p);
  if (ok)
    ok = hipMemsetAsync(d_my, 0, blk, s) == hipSuccess &&
         (cnt == 0 || blk_extract_vecs(s, d_res, d_my, cnt) == BL_OK) &&
         hipMemcpyAsync(d_order, c.order->data(), ord_bytes, hipMemcpyHostToDevice, s) == hipSuccess &&
         hipStreamSynchronize(s) == hipSuccess; /* one sync: the peers are about to read / write */
  c.shared->d_gath[rank] = d_gath;
  if (!ok) c.failed->store(1);
  c.bar->wait(); /* every rank has its vectors and has published its gather buffer */
  const bool go = c.failed->load() == 0;
  if (go) {
    if (c.flags & BL_AMD_MULTI_GATHER_PEER) {
      for (int p = 0; p < W && ok; ++p)
        ok = hipMemcpyPeerAsync(c.shared->d_gath[p] + (size_t)rank * c.m, c.devices[p], d_my, device, blk,
                                s) == hipSuccess;
      ok = ok && hipStreamSynchronize(s) == hipSuccess; /* my block has landed everywhere */
    } else {
      /* stream-ordered: the scatter and the row block below simply follow on s */
      const ncclResult_t r = g_multi.rccl.AllGather(d_my, d_gath, (size_t)c.m * 4, BL_NCCL_FLOAT, comm, s);
      if (r != BL_NCCL_SUCCESS) {
        fprintf(stderr, "bliss_amd: ncclAllGather failed on rank %d: %s\n", rank, g_multi.rccl.GetErrorString(r));
        ok = 0;
      }
    }
    if (!ok) c.failed->store(1);
  }
  if (c.flags & BL_AMD_MULTI_GATHER_PEER) c.bar->wait(); /* all blocks have landed everywhere */
  if (go && ok) {
    ok = blk_scatter_vecs(s, d_gath, d_order, d_all, c.m * W) == BL_OK;
    if (ok && want_rows) {
      ok = blk_pairwise(s, d_all, n, row0, my_rows, d_rows, false, nullptr, nullptr) == BL_OK;
      if (ok && c.h_matrix)
        ok = hipMemcpyAsync(c.h_matrix + (size_t)row0 * n, d_rows, sizeof(float) * (size_t)my_rows * n,
                            hipMemcpyDeviceToHost, s) == hipSuccess;
    }
    ok = ok && hipStreamSynchronize(s) == hipSuccess;
    if (!ok) c.failed->store(1);
  }
  (void)hipStreamSynchronize(s);
  c.bar->wait(); /* nobody returns (and lets a later call reuse its buffers) while a peer may still write */
}

/* the rank's internal stream 0, created on first use */
hipStream_t rank_stream(bl_amd_ctx *ctx) {
  if (!ctx->streams[0] && hipStreamCreateWithFlags(&ctx->streams[0], hipStreamNonBlocking) != hipSuccess)
    return nullptr;
  return ctx->streams[0];
}

struct HostJob {
  const std::vector<int> *mine; /* caller indices of this rank's songs */
  const int16_t *const *h_pcm;
  const int32_t *n_samples, *channels;
  const uint64_t *duration;
  bl_amd_song_result *h_results;
};

void rank_main_host(const Call *c, int rank, bl_amd_ctx *ctx, ncclComm_t comm, HostJob j) {
  int ok = hipSetDevice(c->devices[rank]) == hipSuccess;
  const int cnt = (int)j.mine->size();
  std::vector<const void *> pcm(cnt);
  std::vector<int32_t> ns(cnt), ch(cnt);
  std::vector<uint64_t> du(cnt);
  std::vector<bl_amd_song_result> res(cnt);
  for (int i = 0; i < cnt; ++i) {
    const int s = (*j.mine)[i];
    pcm[i] = j.h_pcm[s]; ns[i] = j.n_samples[s]; ch[i] = j.channels[s]; du[i] = j.duration[s];
  }
  std::unique_lock<std::mutex> lk(ctx->mu);
  bl_amd_song_result *d_res = nullptr;
  if (ok && cnt > 0)
    ok = blr_analyze_host(ctx, pcm.data(), 0, ns.data(), ch.data(), du.data(), cnt, 0, res.data(), &d_res) == BL_OK;
  if (ok)
    for (int i = 0; i < cnt; ++i) j.h_results[(*j.mine)[i]] = res[i];
  hipStream_t s = ok ? rank_stream(ctx) : nullptr;
  rank_exchange(*c, rank, ctx, comm, s, d_res, cnt, nullptr, ok && s);
}

void rank_main_device(const Call *c, int rank, bl_amd_ctx *ctx, ncclComm_t comm, const bl_amd_shard *sh,
                      bl_amd_song_result *h_results_block) {
  int ok = hipSetDevice(sh->device) == hipSuccess;
  std::unique_lock<std::mutex> lk(ctx->mu);
  hipStream_t s = ok ? rank_stream(ctx) : nullptr;
  ok = ok && s;
  bl_amd_song_result *d_res = sh->d_results;
  if (ok && !d_res && sh->n_songs > 0) { /* the caller keeps no device copy: the context's buffer */
    ok = blr_ensure(ctx->results, sizeof(bl_amd_song_result) * (size_t)sh->n_songs) == BL_OK;
    d_res = static_cast<bl_amd_song_result *>(ctx->results.p);
  }
  if (ok && sh->n_songs > 0)
    ok = blr_analyze_device(ctx, sh->d_pcm, sh->h_desc, sh->n_songs, d_res, s, 7) == BL_OK;
  if (ok && h_results_block && sh->n_songs > 0)
    ok = hipMemcpyAsync(h_results_block, d_res, sizeof(bl_amd_song_result) * (size_t)sh->n_songs,
                        hipMemcpyDeviceToHost, s) == hipSuccess;
  rank_exchange(*c, rank, ctx, comm, s, d_res, sh->n_songs, sh->d_rows, ok);
}

/* contexts (one per rank, kept between calls) and, for the RCCL gather, the communicators */
int prepare_ranks(const int *devices, int W, int flags, std::vector<ncclComm_t> &comms) {
  for (int r = 0; r < W; ++r) {
    if (g_multi.ctx[r] && g_multi.ctx[r]->device != devices[r]) {
      bl_amd_ctx_destroy(g_multi.ctx[r]);
      g_multi.ctx[r] = nullptr;
    }
    if (!g_multi.ctx[r] && bl_amd_ctx_create(devices[r], &g_multi.ctx[r]) != BL_OK) return BL_UNEXPECTED;
  }
  comms.assign(W, nullptr);
  if (flags & BL_AMD_MULTI_GATHER_PEER) return BL_OK;
  std::vector<int> devs(devices, devices + W), sorted = devs;
  std::sort(sorted.begin(), sorted.end());
  if (std::adjacent_find(sorted.begin(), sorted.end()) != sorted.end()) {
    fprintf(stderr, "bliss_amd: RCCL gather needs distinct devices per rank; use BL_AMD_MULTI_GATHER_PEER\n");
    return BL_UNEXPECTED;
  }
  if (!g_multi.rccl.load()) return BL_UNEXPECTED;
  if (g_multi.comm_devices != devs) {
    g_multi.drop_comms();
    g_multi.comms.assign(W, nullptr);
    const ncclResult_t r = g_multi.rccl.CommInitAll(g_multi.comms.data(), W, devs.data());
    if (r != BL_NCCL_SUCCESS) {
      fprintf(stderr, "bliss_amd: ncclCommInitAll failed: %s\n", g_multi.rccl.GetErrorString(r));
      g_multi.comms.clear();
      return BL_UNEXPECTED;
    }
    g_multi.comm_devices = devs;
  }
  comms = g_multi.comms;
  return BL_OK;
}

} // namespace

extern "C" {

void bl_multi_shutdown(void) {
  std::lock_guard<std::mutex> lk(g_multi.mu);
  if (g_multi.rccl.h) g_multi.drop_comms();
  for (int r = 0; r < BL_MAX_DEVICES; ++r) {
    if (g_multi.ctx[r]) bl_amd_ctx_destroy(g_multi.ctx[r]);
    g_multi.ctx[r] = nullptr;
  }
}

int bl_amd_analyze_corpus_multi(const int16_t *const *h_pcm, const int32_t *n_samples,
                                const int32_t *channels, const uint64_t *duration, int n_songs,
                                const int *devices, int n_devices, int flags,
                                bl_amd_song_result *h_results, float *h_matrix) {
  if (n_songs <= 0 || !h_pcm || !n_samples || !channels || !duration || !devices || n_devices <= 0 ||
      n_devices > BL_MAX_DEVICES || !h_results)
    return BL_UNEXPECTED;
  std::lock_guard<std::mutex> lk(g_multi.mu);
  const int W = n_devices;
  std::vector<ncclComm_t> comms;
  if (prepare_ranks(devices, W, flags, comms) != BL_OK) return BL_UNEXPECTED;
  /* shards: contiguous blocks for equal lengths, LPT by sample count otherwise */
  std::vector<std::vector<int>> shards(W);
  bool equal = true;
  for (int i = 1; i < n_songs && equal; ++i) equal = n_samples[i] == n_samples[0];
  if (equal) {
    const int base = n_songs / W, rem = n_songs % W;
    for (int r = 0, first = 0; r < W; ++r) {
      const int cnt = base + (r < rem ? 1 : 0);
      for (int i = 0; i < cnt; ++i) shards[r].push_back(first + i);
      first += cnt;
    }
  } else {
    std::vector<int> idx(n_songs);
    for (int i = 0; i < n_songs; ++i) idx[i] = i;
    std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return n_samples[a] > n_samples[b]; });
    typedef std::pair<long long, int> LR; /* (load, rank): smallest load first, ties by rank */
    std::priority_queue<LR, std::vector<LR>, std::greater<LR>> heap;
    for (int r = 0; r < W; ++r) heap.push(LR(0, r));
    for (int i : idx) {
      LR t = heap.top();
      heap.pop();
      shards[t.second].push_back(i);
      heap.push(LR(t.first + n_samples[i], t.second));
    }
  }
  int m = 1;
  for (int r = 0; r < W; ++r) m = std::max(m, (int)shards[r].size());
  std::vector<int32_t> order((size_t)W * m, -1);
  for (int r = 0; r < W; ++r)
    for (size_t i = 0; i < shards[r].size(); ++i) order[(size_t)r * m + i] = shards[r][i];
  /* the matrix is in the caller's order, so the row blocks are an even split of 0..N-1 */
  std::vector<int> row0(W), rows(W);
  for (int r = 0; r < W; ++r) {
    const int base = n_songs / W, rem = n_songs % W;
    rows[r] = h_matrix ? base + (r < rem ? 1 : 0) : 0;
    row0[r] = r * base + std::min(r, rem);
  }

  Barrier bar(W);
  std::atomic<int> failed{0};
  RankShared shared;
  memset(&shared, 0, sizeof shared);
  Call call{W, flags, n_songs, m, &order, devices, row0.data(), rows.data(), h_matrix, &bar, &failed, &shared};
  std::vector<std::thread> threads;
  for (int r = 0; r < W; ++r) {
    HostJob j{&shards[r], h_pcm, n_samples, channels, duration, h_results};
    threads.emplace_back(rank_main_host, &call, r, g_multi.ctx[r], comms[r], j);
  }
  for (auto &t : threads) t.join();
  return failed.load() ? BL_UNEXPECTED : BL_OK;
}

int bl_amd_analyze_corpus_multi_device(const bl_amd_shard *shards, int n_shards, int flags,
                                       bl_amd_song_result *h_results, float *h_matrix) {
  if (!shards || n_shards <= 0 || n_shards > BL_MAX_DEVICES) return BL_UNEXPECTED;
  long long total = 0;
  for (int r = 0; r < n_shards; ++r) {
    const bl_amd_shard &sh = shards[r];
    if (sh.n_songs < 0 || (sh.n_songs > 0 && (!sh.d_pcm || !sh.h_desc))) {
      fprintf(stderr, "bliss_amd: shard %d: n_songs = %d needs d_pcm and h_desc\n", r, sh.n_songs);
      return BL_UNEXPECTED;
    }
    total += sh.n_songs;
  }
  if (total <= 0 || total > INT32_MAX) return BL_UNEXPECTED;
  std::lock_guard<std::mutex> lk(g_multi.mu);
  const int W = n_shards, n_songs = (int)total;
  std::vector<int> devices(W);
  for (int r = 0; r < W; ++r) devices[r] = shards[r].device;
  std::vector<ncclComm_t> comms;
  if (prepare_ranks(devices.data(), W, flags, comms) != BL_OK) return BL_UNEXPECTED;
  /* output order = shard-major: shard r's songs are songs first[r] .. first[r] + n_songs - 1,
   * and those are the matrix rows rank r computes (its own songs against everybody's) */
  int m = 1;
  std::vector<int> first(W), rows(W);
  for (int r = 0, f = 0; r < W; ++r) {
    first[r] = f;
    f += shards[r].n_songs;
    m = std::max(m, shards[r].n_songs);
  }
  for (int r = 0; r < W; ++r) rows[r] = (h_matrix || shards[r].d_rows) ? shards[r].n_songs : 0;
  std::vector<int32_t> order((size_t)W * m, -1);
  for (int r = 0; r < W; ++r)
    for (int i = 0; i < shards[r].n_songs; ++i) order[(size_t)r * m + i] = first[r] + i;

  Barrier bar(W);
  std::atomic<int> failed{0};
  RankShared shared;
  memset(&shared, 0, sizeof shared);
  Call call{W, flags, n_songs, m, &order, devices.data(), first.data(), rows.data(), h_matrix, &bar, &failed, &shared};
  std::vector<std::thread> threads;
  for (int r = 0; r < W; ++r)
    threads.emplace_back(rank_main_device, &call, r, g_multi.ctx[r], comms[r], &shards[r],
                         h_results ? h_results + first[r] : nullptr);
  for (auto &t : threads) t.join();
  return failed.load() ? BL_UNEXPECTED : BL_OK;
}

} /* extern "C" */
