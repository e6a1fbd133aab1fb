/*
 * bl_multi.hip — the batch-of-songs mode across the GPUs of one node, from C
 * (bl_amd_analyze_corpus_multi of include/bliss_amd.h).
 *
 * Songs are independent (ref src/analyze.c:33-86 keeps no cross-song state), so the corpus is
 * sharded by song: equal-length corpora in contiguous blocks, mixed lengths by
 * longest-processing-time-first on the sample count.  One host thread and one context per
 * rank (a rank = one entry of `devices`); each rank runs the ordinary host-batch path on its
 * shard.  The only exchange is one all-gather of the 16-byte force vectors:
 *   BL_AMD_MULTI_GATHER_RCCL  ncclAllGather over xGMI, librccl loaded at first use (dlopen:
 *                             the single-device library does not depend on it);
 *   BL_AMD_MULTI_GATHER_PEER  every rank copies its block straight into every peer's buffer
 *                             (hipMemcpyPeerAsync) — 128 KiB per rank at 8 192 songs is
 *                             latency-bound, a 7-peer write is as good as a ring, and it also
 *                             allows two ranks on one device (how the tests run N = 2 on a
 *                             1-GPU box).
 * Then rank r computes rows [r N / W, (r + 1) N / W) of the N x N bl_distance matrix (caller
 * order) in its own HBM and, if asked, copies them to the host matrix.  No other collective.
 */
#include <dlfcn.h>
#include <rccl/rccl.h>
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

struct RankJob {
  int rank, world, device, flags;
  bl_amd_ctx *ctx;
  const std::vector<int> *mine; /* caller indices of this rank's songs */
  int m;                        /* padded songs per rank */
  int n_songs;
  const int16_t *const *h_pcm;
  const int32_t *n_samples, *channels;
  const uint64_t *duration;
  bl_amd_song_result *h_results;
  float *h_matrix;
  const std::vector<int32_t> *order; /* world * m entries: caller index or -1 */
  const int *devices;
  Barrier *bar;
  std::atomic<int> *failed;
  RankShared *shared;
  ncclComm_t comm;
};

void rank_main(RankJob j) {
  int ok = hipSetDevice(j.device) == hipSuccess;
  const int cnt = (int)j.mine->size();
  const int n = j.n_songs;
  std::vector<const void *> pcm(cnt);
  std::vector<int32_t> ns(cnt), ch(cnt);
  std::vector<uint64_t> du(cnt);
  std::vector<bl_amd_song_result> res(cnt);
  for (int i = 0; i < cnt; ++i) {
    const int s = (*j.mine)[i];
    pcm[i] = j.h_pcm[s]; ns[i] = j.n_samples[s]; ch[i] = j.channels[s]; du[i] = j.duration[s];
  }
  std::unique_lock<std::mutex> lk(j.ctx->mu);
  bl_amd_song_result *d_res = nullptr;
  if (ok && cnt > 0)
    ok = blr_analyze_host(j.ctx, pcm.data(), 0, ns.data(), ch.data(), du.data(), cnt, 0, res.data(), &d_res) == BL_OK;
  if (ok)
    for (int i = 0; i < cnt; ++i) j.h_results[(*j.mine)[i]] = res[i];

  hipStream_t s = j.ctx->streams[0];
  struct force_vector_s *d_my = nullptr, *d_gath = nullptr, *d_all = nullptr;
  int32_t *d_order = nullptr;
  float *d_rows = nullptr;
  const size_t blk = sizeof(struct force_vector_s) * (size_t)j.m;
  const int base = n / j.world, rem = n % j.world;
  const int my_rows = base + (j.rank < rem ? 1 : 0);
  const int row0 = j.rank * base + std::min(j.rank, rem);
  if (ok) {
    ok = s != nullptr || hipStreamCreateWithFlags(&j.ctx->streams[0], hipStreamNonBlocking) == hipSuccess;
    s = j.ctx->streams[0];
  }
  if (ok)
    ok = hipMalloc(&d_my, blk) == hipSuccess && hipMalloc(&d_gath, blk * j.world) == hipSuccess &&
         hipMalloc(&d_all, sizeof(struct force_vector_s) * (size_t)n) == hipSuccess &&
         hipMalloc(&d_order, sizeof(int32_t) * (size_t)j.m * j.world) == hipSuccess &&
         hipMemsetAsync(d_my, 0, blk, s) == hipSuccess &&
         (cnt == 0 || blk_extract_vecs(s, d_res, d_my, cnt) == BL_OK) &&
         hipMemcpyAsync(d_order, j.order->data(), sizeof(int32_t) * (size_t)j.m * j.world,
                        hipMemcpyHostToDevice, s) == hipSuccess &&
         hipStreamSynchronize(s) == hipSuccess;
  j.shared->d_gath[j.rank] = d_gath;
  if (!ok) j.failed->store(1);
  j.bar->wait(); /* every rank has its vectors and has published its gather buffer */
  const bool go = j.failed->load() == 0;
  if (go) {
    if (j.flags & BL_AMD_MULTI_GATHER_PEER) {
      for (int p = 0; p < j.world && ok; ++p)
        ok = hipMemcpyPeerAsync(j.shared->d_gath[p] + (size_t)j.rank * j.m, j.devices[p], d_my, j.device,
                                blk, s) == hipSuccess;
      ok = ok && hipStreamSynchronize(s) == hipSuccess;
    } else {
      const ncclResult_t r = g_multi.rccl.AllGather(d_my, d_gath, (size_t)j.m * 4, ncclFloat, j.comm, s);
      if (r != ncclSuccess) {
        fprintf(stderr, "bliss_amd: ncclAllGather failed on rank %d: %s\n", j.rank, g_multi.rccl.GetErrorString(r));
        ok = 0;
      }
      ok = ok && hipStreamSynchronize(s) == hipSuccess;
    }
    if (!ok) j.failed->store(1);
  }
  j.bar->wait(); /* all blocks have landed everywhere */
  if (go && j.failed->load() == 0 && my_rows > 0) {
    ok = blk_scatter_vecs(s, d_gath, d_order, d_all, j.m * j.world) == BL_OK;
    if (ok && j.h_matrix) {
      ok = hipMalloc(&d_rows, sizeof(float) * (size_t)my_rows * n) == hipSuccess &&
           blk_pairwise(s, d_all, n, row0, my_rows, d_rows, false, nullptr, nullptr) == BL_OK &&
           hipMemcpyAsync(j.h_matrix + (size_t)row0 * n, d_rows, sizeof(float) * (size_t)my_rows * n,
                          hipMemcpyDeviceToHost, s) == hipSuccess;
    }
    ok = ok && hipStreamSynchronize(s) == hipSuccess;
    if (!ok) j.failed->store(1);
  }
  (void)hipStreamSynchronize(s);
  j.bar->wait(); /* nobody frees a buffer a peer may still write to */
  void *bufs[] = {d_my, d_gath, d_all, d_order, d_rows};
  for (void *b : bufs)
    if (b) (void)hipFree(b);
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
  /* contexts: one per rank, kept between calls */
  for (int r = 0; r < W; ++r) {
    if (g_multi.ctx[r] && g_multi.ctx[r]->device != devices[r]) {
      bl_amd_ctx_destroy(g_multi.ctx[r]);
      g_multi.ctx[r] = nullptr;
    }
    if (!g_multi.ctx[r] && bl_amd_ctx_create(devices[r], &g_multi.ctx[r]) != BL_OK) return BL_UNEXPECTED;
  }
  std::vector<ncclComm_t> comms(W, nullptr);
  if (!(flags & BL_AMD_MULTI_GATHER_PEER)) {
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
      if (r != ncclSuccess) {
        fprintf(stderr, "bliss_amd: ncclCommInitAll failed: %s\n", g_multi.rccl.GetErrorString(r));
        g_multi.comms.clear();
        return BL_UNEXPECTED;
      }
      g_multi.comm_devices = devs;
    }
    comms = g_multi.comms;
  }
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

  Barrier bar(W);
  std::atomic<int> failed{0};
  RankShared shared;
  memset(&shared, 0, sizeof shared);
  std::vector<std::thread> threads;
  for (int r = 0; r < W; ++r) {
    RankJob j;
    j.rank = r; j.world = W; j.device = devices[r]; j.flags = flags; j.ctx = g_multi.ctx[r];
    j.mine = &shards[r]; j.m = m; j.n_songs = n_songs; j.h_pcm = h_pcm; j.n_samples = n_samples;
    j.channels = channels; j.duration = duration; j.h_results = h_results; j.h_matrix = h_matrix;
    j.order = &order; j.devices = devices; j.bar = &bar; j.failed = &failed; j.shared = &shared;
    j.comm = comms[r];
    threads.emplace_back(rank_main, j);
  }
  for (auto &t : threads) t.join();
  return failed.load() ? BL_UNEXPECTED : BL_OK;
}

} /* extern "C" */
