/*
 * bl_kernels.hip — gfx950 (MI355X, CDNA4) kernels and launch layer of the bliss
 * per-song analysis path.  Written for wave64 / 160 KiB LDS / 256 CUs; no other
 * target is supported.  Must be compiled with -ffp-contract=off: everything
 * outside bl_fft.h follows the reference's unfused x86-64 SSE2 arithmetic
 * (ref CMakeLists.txt:22, -std=c99) operation by operation.
 *
 * Kernels (reference code each one replaces):
 *   k_pcm_scan     sum, sum of squares, central histogram in one pass
 *                                               ref src/helpers.c:30-49,
 *                                               src/amplitude_sort.c:33-39
 *   k_trim         first / last non-zero sample  ref src/amplitude_sort.c:26-31
 *   k_song_prep    bl_mean / bl_variance values, start/end, reciprocal used by
 *                  the normalisation            ref src/tempo_atk_sort.c:101-107
 *   k_variance_wrap  exact int32-wrapping bl_variance for |mean| > 13571
 *   k_amp_finish   301-pass smoothing + integral ref src/amplitude_sort.c:41-79
 *   k_freq_frames  Hann + 512-pt f32 real DFT power, summed over the frames in the
 *                  reference's order            ref src/frequency_sort.c:67-94
 *   k_freq_finish  dB spectrum, 5 bands, score  ref src/frequency_sort.c:97-139
 *   k_env_windows3 normalise, 17-tap FIR, 512-pt f64 real DFT, f32-rounded
 *                  energy per window            ref src/tempo_atk_sort.c:109-153
 *   k_env_tail     IIR, box filters, peaks, tempo/attack
 *                                               ref src/tempo_atk_sort.c:184-284
 *   k_force        force, calm_or_loud          ref src/analyze.c:63-80
 *   k_pairwise     bl_distance / bl_cosine_similarity matrix
 *                                               ref src/analyze.c:96-100,135-140
 *   k_synth        integer synthetic PCM (benchmark corpus)
 */
#include <hip/hip_runtime.h>
#include <atomic>
#include <algorithm>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "bl_launch.h"
#include "bl_fft_lavc.h"
#include "bl_cos.h"
#include "bl_sqrt.h"
#include "bl_tail.h"

#define BL_HIP_CHECK(expr)                                                              \
  do {                                                                                  \
    hipError_t e_ = (expr);                                                             \
    if (e_ != hipSuccess) {                                                             \
      fprintf(stderr, "bliss_amd: %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(e_), \
              __FILE__, __LINE__);                                                      \
      return BL_UNEXPECTED;                                                             \
    }                                                                                   \
  } while (0)

/* FIR taps: literal digits of ref include/bandpass_coeffs.h:1-7 (symmetric) */
#define BL_C0 (-0.0023470)
#define BL_C1 0.0044613
#define BL_C2 (-0.0114627)
#define BL_C3 0.0226382
#define BL_C4 (-0.0405147)
#define BL_C5 0.0580037
#define BL_C6 (-0.0779167)
#define BL_C7 0.0882711
#define BL_C8 0.9065095

/* ordering point between LDS accesses of different lanes of ONE wave: a wave's LDS
 * instructions execute in order, so only the compiler has to be told */
__device__ __forceinline__ void bl_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

/* ------------------------------------------------------------------------- */
/* k_pcm_scan                                                                 */

/* One in-range count of the central histogram for each half of a packed word of two samples: bin = s + 2048 as
 * a 16-bit sum (v_pk_add_u16 for both halves), byte address = base + 4 * bin (v_mad_u32_u16 takes the half it is
 * told to), ds_add_u32.  NO range test: a sample outside [-2048, 2048) gives a bin in [4096, 65536) and an address
 * beyond the workgroup's LDS allocation — the histogram is the LAST thing in it — and the LDS discards
 * out-of-range writes (ISA: DS instructions, out-of-range addresses; checked on the device by
 * tests/test_gpu_parity.py::test_histogram_out_of_range_samples_are_dropped).  4 instructions per word instead of
 * 10 with extraction, compare and exec masks. */
typedef __attribute__((address_space(3))) unsigned bl_lds_u32;
/* What the range-test-free form rests on, checked where it can be: the histogram is the LAST object of the
 * workgroup's LDS (static_asserts at the two kernels that use it; k_pcm_scan also compares its static LDS size at
 * run time), so that 4 * bin >= 4 * BL_HIST_BINS lies behind the allocation or in the allocator's slack, where
 * nothing lives.  -DBL_AMD_CHECKED_HIST (make XDEFS=-DBL_AMD_CHECKED_HIST) builds the kernels with the range compare
 * instead: for debuggers and sanitizers that arm the LDS out-of-range trap. */
__device__ __forceinline__ void scan_hist_word(unsigned w, unsigned lds_base) {
#ifdef BL_AMD_CHECKED_HIST
  const unsigned b0 = (unsigned)((int)(short)(w & 0xFFFFu) + BL_HIST_BINS / 2);
  const unsigned b1 = (unsigned)((int)(short)(w >> 16) + BL_HIST_BINS / 2);
  bl_lds_u32 *h = (bl_lds_u32 *)(size_t)lds_base;
  if (b0 < BL_HIST_BINS) __hip_atomic_fetch_add(h + b0, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  if (b1 < BL_HIST_BINS) __hip_atomic_fetch_add(h + b1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#else
  typedef unsigned short us2 __attribute__((ext_vector_type(2)));
  us2 v;
  __builtin_memcpy(&v, &w, 4);
  v += (us2){BL_HIST_BINS / 2, BL_HIST_BINS / 2};
  unsigned b2;
  __builtin_memcpy(&b2, &v, 4);
  unsigned a0, a1;
  const unsigned one = 1u;
  /* one statement: between two of them hipcc pads with s_nop for hazards it cannot rule out; in this order every
   * address has an instruction between its computation and its use */
  asm volatile("v_mad_u32_u16 %0, %2, 4, %3 op_sel:[0,0,0,0]\n\t"
               "v_mad_u32_u16 %1, %2, 4, %3 op_sel:[1,0,0,0]\n\t"
               "ds_add_u32 %0, %4\n\t"
               "ds_add_u32 %1, %4"
               : "=&v"(a0), "=&v"(a1) : "v"(b2), "v"(lds_base), "v"(one) : "memory");
#endif
}

/* Everything the statistics take from one packed word of two samples: lo + hi into the 32-bit partial sum
 * (v_dot2_i32_i16 with ones), lo^2 + hi^2 (the same instruction; <= 2^31, read as unsigned) into the 64-bit sum of
 * squares by ONE v_mad_u64_u32 (r * 1 + sq; a 64-bit add is two instructions and every one of these issues in four
 * cycles: tools/gen_ubench_issue.py), and the histogram counts: 6 instructions per word. */
__device__ __forceinline__ void scan_word(unsigned w, int &s32, unsigned long long &sq, unsigned lds_hist, bool hist) {
  typedef short short2v __attribute__((ext_vector_type(2)));
  const short2v ones = {1, 1};
  short2v pr;
  __builtin_memcpy(&pr, &w, 4);
  s32 = __builtin_amdgcn_sdot2(pr, ones, s32, false);
  unsigned r; /* the builtin with a zero addend becomes v_mov 0 + v_dot2c: one instruction too many */
  asm("v_dot2_i32_i16 %0, %1, %1, 0" : "=v"(r) : "v"(w));
  asm("v_mad_u64_u32 %0, vcc, %1, 1, %0" : "+v"(sq) : "v"(r) : "vcc");
  if (hist) scan_hist_word(w, lds_hist);
}

/* sum, sum of squares and the central histogram of every song; the first / last non-zero sample is k_trim's.
 * Per 16-byte vector (8 samples): sums through v_dot2_i32_i16 (lo + hi and lo^2 + hi^2 per word; the latter read
 * as unsigned is exact up to 2^31), the histogram through scan_hist_word.  Two vectors per iteration keep two
 * loads in flight per lane. */
template <bool HIST>
__global__ __launch_bounds__(256) void k_pcm_scan(const int16_t *__restrict__ pcm,
                                                  const bl_dsong *__restrict__ songs,
                                                  bl_dstats *stats, unsigned *hist) {
  __shared__ unsigned lh[BL_HIST_BINS]; /* the only LDS of this kernel: nothing lies behind it */
  if (__builtin_amdgcn_groupstaticsize() != sizeof lh) __builtin_trap(); /* somebody added LDS: see scan_hist_word */
  const int tid = threadIdx.x;
  const bl_dsong sg = songs[blockIdx.y];
  const int16_t *p = pcm + sg.pcm_off;
  for (int i = tid; i < BL_HIST_BINS; i += 256) lh[i] = 0;
  __syncthreads();
  unsigned lds_base = (unsigned)(size_t)(bl_lds_u32 *)lh;
  asm volatile("" : "+v"(lds_base)); /* lives in a VGPR: as a scalar it is copied in front of every use */

  long long sum = 0;
  unsigned long long sq = 0;
  const unsigned nvec = (unsigned)sg.n >> 3;
  const uint4 *pv = reinterpret_cast<const uint4 *>(p);
  auto eat = [&](const uint4 q) {
    const unsigned w[4] = {q.x, q.y, q.z, q.w};
    int s32 = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) scan_word(w[k], s32, sq, lds_base, HIST);
    sum += s32;
  };
  const unsigned gstride = gridDim.x * 256u;
  unsigned v = blockIdx.x * 256u + tid;
  for (; v + gstride < nvec; v += 2 * gstride) {
    const uint4 q0 = pv[v], q1 = pv[v + gstride];
    eat(q0);
    eat(q1);
  }
  if (v < nvec) eat(pv[v]);
  if (blockIdx.x == 0 && tid < (sg.n & 7)) { /* the samples behind the last whole vector */
    const int sv = (int)p[8u * nvec + tid];
    sum += sv;
    sq += (unsigned)(sv * sv);
    const unsigned b = (unsigned)(sv + BL_HIST_BINS / 2);
    if (HIST && b < BL_HIST_BINS) atomicAdd(&lh[b], 1u);
  }
  /* wave reduction, then one pair of atomics per wave */
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    sum += __shfl_down(sum, off);
    sq += __shfl_down(sq, off);
  }
  bl_dstats *st = stats + blockIdx.y;
  if ((tid & 63) == 0) {
    atomicAdd(&st->sum, (unsigned long long)sum);
    atomicAdd(&st->sumsq, sq);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); /* the inline-asm adds are invisible to hipcc's counters */
  __syncthreads();
  unsigned *gh = hist + (size_t)blockIdx.y * BL_HIST_BINS;
  for (int i = tid; i < BL_HIST_BINS; i += 256) {
    const unsigned c = lh[i];
    if (c) atomicAdd(&gh[i], c);
  }
}

/* k_trim: the first and the last non-zero sample of every song (ref amplitude_sort.c:26-31, the two trim loops).
 * They sit within a few thousand samples of the ends of any real recording, so this is a search, not a pass: one
 * workgroup per song, wave 0 walks forward and wave 1 backward, 1 024 samples per step, until a vector with a
 * non-zero sample turns up.  (Tracked inside k_pcm_scan's loop it cost 14 instructions per 8 samples.)  An
 * all-zero song is the only one searched to the end; it is refused anyway (k_song_prep). */
__global__ __launch_bounds__(128) void k_trim(const int16_t *__restrict__ pcm, const bl_dsong *__restrict__ songs,
                                              bl_dstats *stats) {
  const bl_dsong sg = songs[blockIdx.x];
  const int16_t *p = pcm + sg.pcm_off;
  const uint4 *pv = reinterpret_cast<const uint4 *>(p);
  const int lane = threadIdx.x & 63, n = sg.n;
  const int nvec = n >> 3;
  const bool fwd = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) == 0;
  bl_dstats *st = stats + blockIdx.x;
  /* position of the first (fwd) / last non-zero 16-bit half of a non-zero vector */
  auto locate = [&](const uint4 q) -> int {
    const unsigned w[4] = {q.x, q.y, q.z, q.w};
    int at = fwd ? 8 : -1;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (fwd) {
        if (w[3 - k] >> 16) at = 2 * (3 - k) + 1;
        if (w[3 - k] & 0xFFFFu) at = 2 * (3 - k);
      } else {
        if (w[k] & 0xFFFFu) at = 2 * k;
        if (w[k] >> 16) at = 2 * k + 1;
      }
    }
    return at;
  };
  if (fwd) {
    unsigned first = 0xFFFFFFFFu;
    for (int v0 = 0; v0 < nvec; v0 += 128) {
      const int va = v0 + lane, vb = v0 + 64 + lane;
      const uint4 z = make_uint4(0, 0, 0, 0);
      const uint4 qa = va < nvec ? pv[va] : z, qb = vb < nvec ? pv[vb] : z;
      const unsigned long long ma = __ballot((qa.x | qa.y | qa.z | qa.w) != 0u);
      const unsigned long long mb = __ballot((qb.x | qb.y | qb.z | qb.w) != 0u);
      if (ma | mb) {
        const int src = ma ? __builtin_ctzll(ma) : __builtin_ctzll(mb);
        const unsigned mine = 8u * (unsigned)(ma ? va : vb) + (unsigned)locate(ma ? qa : qb);
        first = (unsigned)__shfl((int)mine, src);
        break;
      }
    }
    if (first == 0xFFFFFFFFu) /* nothing in the whole vectors: the up to seven samples behind them */
      for (int i = 8 * nvec; i < n; ++i)
        if (p[i] != 0) { first = (unsigned)i; break; }
    if (lane == 0) st->first = first;
  } else {
    int last = -1;
    for (int i = n - 1; i >= 8 * nvec; --i)
      if (p[i] != 0) { last = i; break; }
    if (last < 0)
      for (int v1 = nvec; v1 > 0; v1 -= 128) { /* vectors [v1 - 128, v1) */
        const int va = v1 - 1 - lane, vb = v1 - 65 - lane;
        const uint4 z = make_uint4(0, 0, 0, 0);
        const uint4 qa = va >= 0 ? pv[va] : z, qb = vb >= 0 ? pv[vb] : z;
        const unsigned long long ma = __ballot((qa.x | qa.y | qa.z | qa.w) != 0u);
        const unsigned long long mb = __ballot((qb.x | qb.y | qb.z | qb.w) != 0u);
        if (ma | mb) { /* lane 0 holds the highest vector of each half */
          const int src = ma ? __builtin_ctzll(ma) : __builtin_ctzll(mb);
          const int mine = 8 * (ma ? va : vb) + locate(ma ? qa : qb);
          last = __shfl(mine, src);
          break;
        }
      }
    if (lane == 0) st->last = last;
  }
}

__global__ void k_stats_init(bl_dstats *stats, int n_songs) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_songs) return;
  bl_dstats s;
  s.sum = 0; s.sumsq = 0; s.first = 0xFFFFFFFFu; s.last = -1;
  s.mean = 0; s.variance = 0; s.vprime = 0; s.rcp = 0; s.rcp_lo = 0; s.wrap_pass = 0; s.status = BL_OK;
  s.wrap_acc = 0;
  stats[i] = s;
}

/* ------------------------------------------------------------------------- */
/* k_song_prep: one thread per song                                           */

__device__ __forceinline__ void prep_finish(bl_dstats &s, int n) {
  if (s.variance == 0) s.status = BL_UNEXPECTED; /* reference divides by zero */
  /* ref tempo_atk_sort.c:105-113: x = (s/2^15 - mean/2^15) / (var/2^30)
   *   = RN((s - mean) / (var * 2^-15)) exactly (power-of-two scalings commute
   *   with rounding); vprime and its reciprocal feed bl_norm() below. */
  s.vprime = (double)s.variance / 32768.0;
  /* The envelope kernel works on x / 2 (an exact scaling: see bl_norm), so the reciprocal is
   * that of 2 * vprime, as an unevaluated sum rcp + rcp_lo accurate to ~2^-106. */
  const double v2 = 2.0 * s.vprime;
  s.rcp = 1.0 / v2;
  s.rcp_lo = __builtin_fma(-s.rcp, v2, 1.0) / v2;
  const double taps[9] = {BL_C0, BL_C1, BL_C2, BL_C3, BL_C4, BL_C5, BL_C6, BL_C7, BL_C8};
#pragma unroll
  for (int m = 0; m < 9; ++m) s.firc[m] = __builtin_fma(taps[m], s.rcp, taps[m] * s.rcp_lo);
  (void)n;
}

__global__ void k_song_prep(const bl_dsong *__restrict__ songs, bl_dstats *stats, int n_songs,
                            bl_amd_song_result *res) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_songs) return;
  bl_dstats s = stats[i];
  const bl_dsong sg = songs[i];
  const int n = sg.n;
  if (s.first == 0xFFFFFFFFu) { /* all-zero PCM: the reference's trim loops never end */
    s.status = BL_UNEXPECTED;
    s.first = 0; s.last = n - 1;
  }
  /* ref helpers.c:30-37: int32 accumulator (wraps), C truncating division */
  const int wrapped = (int)(unsigned)(s.sum & 0xFFFFFFFFull);
  s.mean = wrapped / n;
  /* ref helpers.c:39-49: sum of (int32)(v*v), v = sample - mean.  Without int32
   * overflow of v*v (|v| <= 46340, guaranteed when |mean| <= 13571) this is
   * sumsq - 2*mean*sum + n*mean^2 in exact integer arithmetic. */
  const long long m = s.mean;
  if (m > 13571 || m < -13571) {
    s.wrap_pass = 1;
  } else {
    const long long acc = (long long)s.sumsq - 2 * m * (long long)s.sum + (long long)n * m * m;
    s.variance = (int)(acc / n);
    prep_finish(s, n);
  }
  stats[i] = s;
  bl_amd_song_result *r = res + sg.out_idx;
  r->start = (int)s.first; r->end = s.last;
  r->mean = s.mean; r->variance = s.variance;
  r->n_frames = sg.n_frames; r->nb_frames = sg.nb_frames; r->n_windows = sg.n_windows;
  r->status = s.status;
}

/* exact restatement of ref helpers.c:39-49 including the int32 wrap of v*v;
 * only songs flagged by k_song_prep do any work */
__global__ __launch_bounds__(256) void k_variance_wrap(const int16_t *__restrict__ pcm,
                                                       const bl_dsong *__restrict__ songs,
                                                       bl_dstats *stats) {
  bl_dstats *st = stats + blockIdx.y;
  if (!st->wrap_pass) return;
  const bl_dsong sg = songs[blockIdx.y];
  const int16_t *p = pcm + sg.pcm_off;
  const int mean = st->mean;
  long long acc = 0;
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < (unsigned)sg.n; i += gridDim.x * 256u) {
    const int v = (int)p[i] - mean;
    acc += (int)((unsigned)v * (unsigned)v);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
  if ((threadIdx.x & 63) == 0)
    atomicAdd(reinterpret_cast<unsigned long long *>(&st->wrap_acc), (unsigned long long)acc);
}

__global__ void k_variance_wrap_finish(const bl_dsong *__restrict__ songs, bl_dstats *stats,
                                       int n_songs, bl_amd_song_result *res) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_songs) return;
  bl_dstats s = stats[i];
  if (!s.wrap_pass) return;
  s.variance = (int)(s.wrap_acc / songs[i].n);
  prep_finish(s, songs[i].n);
  stats[i] = s;
  res[songs[i].out_idx].variance = s.variance;
  res[songs[i].out_idx].status = s.status;
}

/* ------------------------------------------------------------------------- */
/* k_amp_finish: one block per song                                           */

#define BL_AMP_PASSES 301                         /* g = 0..300, ref amplitude_sort.c:41 */
#define BL_INT_LO ((32767 - 1000) - BL_HIST_LO)   /* local index of INTEGRAL_INF */
#define BL_INT_HI ((32767 + 1000) - BL_HIST_LO)   /* local index of INTEGRAL_SUP */

__global__ __launch_bounds__(256) void k_amp_finish(const bl_dsong *__restrict__ songs,
                                                    const bl_dstats *__restrict__ stats,
                                                    const unsigned *__restrict__ hist,
                                                    bl_amd_song_result *res) {
  __shared__ float buf[2][BL_HIST_BINS + 8];
  const int tid = threadIdx.x;
  const int song = blockIdx.x;
  const bl_dstats st = stats[song];
  const int n = songs[song].n;
  const unsigned *gh = hist + (size_t)song * BL_HIST_BINS;
  const int start = (int)st.first, end = st.last;
  if (tid < 8) { /* 3 zero cells left of bin 0, 5 right of the last bin */
    const int c = tid < 3 ? tid : BL_HIST_BINS + tid;
    buf[0][c] = 0.f; buf[1][c] = 0.f;
  }
  for (int i = tid; i < BL_HIST_BINS; i += 256) {
    unsigned c = gh[i];
    /* samples outside [start, end] are zeros and are not counted (ref :26-39) */
    if (i == BL_HIST_BINS / 2) c -= (unsigned)start + (unsigned)(n - 1 - end);
    /* float += 1 stops growing at 2^24 */
    buf[0][i + 3] = (float)min(c, 16777216u);
  }
  __syncthreads();
  int cur = 0;
  for (int g = 0; g < BL_AMP_PASSES; ++g) {
    const float *h = buf[cur] + 3;
    float *s = buf[cur ^ 1] + 3;
    /* only bins that can still reach the integral window [BL_INT_LO, BL_INT_HI] through the passes
     * that remain (3 bins per pass) are updated: from 3 807 of them in the first pass down to 2 001 */
    const int reach = 3 * (BL_AMP_PASSES - 1 - g);
    const int lo = max(BL_INT_LO - reach, 0), hi = min(BL_INT_HI + reach, BL_HIST_BINS - 1);
    for (int i = lo + tid; i <= hi; i += 256) {
      /* ref :49-55: f32 sum left to right, times (double)(1/27), stored as f32 */
      float acc = h[i - 3] + (3 * h[i - 2]);
      acc = acc + (6 * h[i - 1]);
      acc = acc + (7 * h[i]);
      acc = acc + (6 * h[i + 1]);
      acc = acc + (3 * h[i + 2]);
      acc = acc + h[i + 3];
      s[i] = (float)(1. / 27. * (double)acc);
    }
    __syncthreads();
    cur ^= 1;
  }
  /* ref :62-66 then :69-71 */
  float *s = buf[cur] + 3;
  float *v = buf[cur ^ 1] + 3;
  const float denom = (float)(start - end);
  for (int i = BL_INT_LO + tid; i <= BL_INT_HI; i += 256) {
    float t = s[i] / denom;
    t = (float)((double)t * 100.);
    v[i] = fabsf(t);
  }
  __syncthreads();
  if (tid == 0) {
    float integral = 0;
    for (int i = BL_INT_LO; i <= BL_INT_HI; ++i) integral += v[i];
    bl_amd_song_result *r = res + songs[song].out_idx;
    r->hist_integral = integral;
    r->v.amplitude = -0.2f * integral + 6.0f; /* ref :79 */
  }
}

/* ------------------------------------------------------------------------- */
/* k_freq_frames / k_freq_scan                                                */

/*
 * Hann window + 512-point f32 real DFT + per-bin power, summed over the frames in the reference's order
 * (ref src/frequency_sort.c:67-94).  k_freq_frames is the frequency analysis alone (four waves per workgroup);
 * k_freq_scan is the same body with eight waves and the statistics pass riding along (see freq_frames_lavc) —
 * what bl_analyze and the batch calls launch.
 *
 * One workgroup per song.  Every 16-lane group transforms TWO frames at once: all values are
 * 2-vectors (frame A in .x, frame B in .y), so the whole transform is v_pk_add / v_pk_mul_f32
 * on register pairs with no shuffling between the halves — twice the f32 rate of
 * the scalar VALU for the same instruction count (the one-frame-per-group kernel spent 40 % of
 * its VALU stream on v_mov's that re-paired (re, im) for the packed instructions hipcc formed).
 * A wave covers 8 consecutive frames per iteration, the WAVES waves 8 * WAVES.
 *
 * ref :88-93 adds every frame's power spectrum into one f32 accumulator per bin, frame after
 * frame; f32 addition does not associate, so the order is part of the result (15 000 frames
 * leave ~1e-5 of room in `frequency`).  The running spectrum goes round the waves like a baton:
 * wave w waits for the relay word to reach WAVES * it + w, adds its eight frames bin by bin in frame
 * order (its own power values re-laid out through its private exchange space: lane j owns bins
 * j, j + 64, j + 128, j + 192) and passes it on.  No workgroup barrier in the loop; the waves
 * stagger themselves.
 *
 * LDS: 4 WAVES exchange buffers of 272 (re, im) 2-vectors (16 bytes each: every exchange access is a
 * b128), twiddles, Hann, the running spectrum, the relay word: 76.9 KB for four waves -> two workgroups per
 * CU; 159.1 KB for eight waves with the histogram behind them -> one.  Either way two waves per SIMD (the
 * kernel wants ~200 VGPRs: 64 for the data, 64 for the frames in flight).
 */
typedef bl_c2<bl_f2> c2p; /* a complex number per frame of the pair */

/* LDS of a workgroup of W waves: 4 W exchange buffers, twiddles + Hann, the running spectrum + relay word, and —
 * k_freq_scan only — the central histogram, LAST (scan_hist_word relies on nothing lying behind it) */
#define BL_FREQ_XCH_BYTES(W) (4 * (W) * BL_FFT_XCH_ELEMS * 16)
#define BL_FREQ_ACC_OFF(W) (BL_FREQ_XCH_BYTES(W) + 2 * 256 * 8 + 512 * 4)
#define BL_FREQ_HIST_OFF(W) (BL_FREQ_ACC_OFF(W) + 256 * 4 + 64)
#define BL_FREQ_LDS_BYTES BL_FREQ_HIST_OFF(4)                                 /* k_freq_frames: 76.9 KB */
#define BL_FREQ_SCAN_WAVES 8
#define BL_FREQ_SCAN_LDS_BYTES (BL_FREQ_HIST_OFF(BL_FREQ_SCAN_WAVES) + 4 * BL_HIST_BINS) /* k_freq_scan: 159.1 KB */
/* row stride of the power staging: 2 rows = 16 banks (mod 32) apart, so the two 16-lane groups
 * that share a 32-lane store group land on disjoint banks */
/* every region of the layout, in order: [exchange buffers][twiddles 2 KB + pad 2 KB][Hann 2 KB][spectrum 1 KB][relay 64 B]
 * [histogram] — each ends where the next begins, the histogram ends where the allocation ends (what the
 * range-test-free ds_add of scan_hist_word relies on), and the kernel's launch passes exactly this size */
#define BL_FREQ_TW_OFF(W) BL_FREQ_XCH_BYTES(W)
#define BL_FREQ_HANN_OFF(W) (BL_FREQ_XCH_BYTES(W) + 2 * 256 * 8)
#define BL_FREQ_RELAY_OFF(W) (BL_FREQ_ACC_OFF(W) + 256 * 4)
static_assert(BL_FREQ_TW_OFF(BL_FREQ_SCAN_WAVES) + LV_TW_SLOTS * 16 * 8 <= BL_FREQ_HANN_OFF(BL_FREQ_SCAN_WAVES) &&
                  BL_FREQ_HANN_OFF(BL_FREQ_SCAN_WAVES) + 512 * 4 == BL_FREQ_ACC_OFF(BL_FREQ_SCAN_WAVES) &&
                  BL_FREQ_ACC_OFF(BL_FREQ_SCAN_WAVES) + 256 * 4 == BL_FREQ_RELAY_OFF(BL_FREQ_SCAN_WAVES) &&
                  BL_FREQ_RELAY_OFF(BL_FREQ_SCAN_WAVES) + 64 == BL_FREQ_HIST_OFF(BL_FREQ_SCAN_WAVES) &&
                  BL_FREQ_HIST_OFF(BL_FREQ_SCAN_WAVES) + 4 * BL_HIST_BINS == BL_FREQ_SCAN_LDS_BYTES &&
                  BL_FREQ_SCAN_LDS_BYTES <= 160 * 1024,
              "k_freq_scan: every LDS region ends where the next begins and the histogram is the LAST one (scan_hist_word)");
#define BL_FREQ_SROW 264

/* cross-lane move of a pair of floats through DPP (two 32-bit moves); CTRL 0x140 = row_mirror,
 * 0x120 + n = row_ror:n inside each 16-lane row */
template <int CTRL> __device__ __forceinline__ bl_f2 bl_dpp_f2(bl_f2 v) {
  const int x = __builtin_amdgcn_update_dpp(0, __float_as_int(v.x), CTRL, 0xF, 0xF, true);
  const int y = __builtin_amdgcn_update_dpp(0, __float_as_int(v.y), CTRL, 0xF, 0xF, true);
  return (bl_f2){__int_as_float(x), __int_as_float(y)};
}
/* the same with a value for the lanes that have no source (they keep `old`) */
template <int CTRL> __device__ __forceinline__ bl_f2 bl_dpp_f2_old(bl_f2 old, bl_f2 v) {
  const int x = __builtin_amdgcn_update_dpp(__float_as_int(old.x), __float_as_int(v.x), CTRL, 0xF, 0xF, false);
  const int y = __builtin_amdgcn_update_dpp(__float_as_int(old.y), __float_as_int(v.y), CTRL, 0xF, 0xF, false);
  return (bl_f2){__int_as_float(x), __int_as_float(y)};
}

/*
 * WAVES waves per workgroup (one workgroup per song).  SCAN: the statistics pass rides along — every PCM word the
 * transform loads also goes into the song's sum, sum of squares and central histogram (k_pcm_scan's arithmetic),
 * so the analysis reads the PCM twice instead of three times.
 *
 * The transform is libavcodec's, node for node (bl_fft_lavc.h; round 6): what the reference's av_rdft_calc computes
 * in the order it computes it, so that every frame's power values — and with them `frequency` — are the oracle's bit
 * for bit (the oracle under that order prints the reference's golden values to the last digit).
 * The input is gathered in split-radix order (lane L register r = element (lv_base(L) + K[r]) mod 256 of the frame:
 * immediate offsets from one per-lane base, every 8-byte element still loaded exactly once, 16 lanes per load inside
 * a 16-element neighbourhood), the leaves (fft16, or fft8 twice) run in that layout, ONE transpose through the
 * group's exchange buffer, then pass(32) with its products exchanged between lanes l and l ^ 8 by DPP, pass(64 .. 256)
 * in registers, rdft.c's post-pass with the partner by DPP (row mirror + shift) and re * re + im * im unfused.
 * Rounds 1-5 ran a fused radix-16 transform here (git 6a8cdc4: freq_frames_body; 558 instead of ~760 packed
 * instructions per wave-iteration, k_freq_scan 8.77 instead of 9.64 ms per 1 024 S180 songs) whose `frequency` agreed
 * with the oracle to a few 1e-6 absolute — inside the reference's own tolerance, not bit for bit.
 */
template <bool STEREO, int WAVES, bool SCAN>
__device__ __forceinline__ void freq_frames_lavc(const int16_t *__restrict__ pcm, const bl_dsong &sg,
                                                 const bl_tables &tb, float *spectrum, bl_dstats *st,
                                                 unsigned *gh) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int FPI = 8 * WAVES; /* frames per workgroup iteration */
  c2p *xch = reinterpret_cast<c2p *>(smem); /* 4 WAVES x 272 */
  c2f *lvtw = reinterpret_cast<c2f *>(smem + BL_FREQ_TW_OFF(WAVES)); /* [LV_TW_SLOTS][16 lanes] */
  float *hann = reinterpret_cast<float *>(smem + BL_FREQ_HANN_OFF(WAVES));
  float *accv = reinterpret_cast<float *>(smem + BL_FREQ_ACC_OFF(WAVES)); /* ps[0..255] so far */
  unsigned *lh = reinterpret_cast<unsigned *>(smem + BL_FREQ_HIST_OFF(WAVES)); /* SCAN: the histogram */
  typedef __attribute__((address_space(3))) volatile int lds_vint;
  lds_vint *relay = (lds_vint *)(smem + BL_FREQ_RELAY_OFF(WAVES));
  const int tid = threadIdx.x, g = tid >> 4, l = tid & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, gl = g & 3;
  const int16_t *p = pcm + sg.pcm_off;
  if (WAVES == 4 || tid < 256) {
    lvtw[tid] = tb.lv_tw[tid];
    hann[tid] = tb.hann[tid];
    hann[tid + 256] = tb.hann[tid + 256];
    accv[tid] = 0.f;
  }
  if (SCAN)
    for (int i = tid; i < BL_HIST_BINS; i += 64 * WAVES) lh[i] = 0;
  if (tid == 0) relay[0] = 0;
  __syncthreads();
  unsigned lds_hist = (unsigned)(size_t)(bl_lds_u32 *)lh;
  asm volatile("" : "+v"(lds_hist)); /* lives in a VGPR: as a scalar it is copied in front of every use */
  long long sum = 0;
  unsigned long long sq = 0;

  /* this lane's place in the split-radix order (bl_fft_lavc.h): T8 lanes 1, 5, 7, 9, 13; base element lv_base(l),
   * from which the lanes with base >= 251 wrap for every register but the first */
  const bool t16 = ((lv_t8_lane_mask() >> l) & 1u) == 0u;
  const unsigned long long bases = l < 8 ? lv_bases_packed(0) : lv_bases_packed(1);
  const int base0 = (int)((bases >> (8 * (l & 7))) & 0xFFu);
  const int basep = base0 >= 251 ? base0 - 256 : base0;
  const bool lo8 = l < 8;
  /* gather order of the registers, the same for every lane: lv_k_lo | lv_k_hi(true, .) = what lane 0 (base 0) loads.
   * Element of register r: (base0 + KG(r)) mod 256 = basep + KG(r) for r >= 1 (KG >= 16 > 5 >= -basep), base0 for r = 0 */
#define KG(r) lv_gather_index(0, (r))
  static_assert(lv_t8_lane_mask() == 0x22A2u && lv_bases_packed(0) == 0x06FE0A02FC040800ull &&
                    lv_bases_packed(1) == 0xFB0307FFFD050901ull && KG(0) == 0 && KG(1) == 128 && KG(15) == 176 &&
                    lv_gather_index(3, 1) == ((252 + KG(1)) & 255) && lv_gather_index(12, 0) == 255,
                "bl_fft_lavc.h: lane tables");

  c2p *gx = xch + g * BL_FFT_XCH_ELEMS; /* the transpose buffer of this 16-lane group */
  float *stage = reinterpret_cast<float *>(xch + (g - gl) * BL_FFT_XCH_ELEMS); /* wave-private [8][BL_FREQ_SROW] */
  /* one iteration ahead: 32 unconditional loads per lane (frame indices clamped into the song;
   * a frame past the end is transformed like any other and simply not added), so the HBM
   * latency of iteration it+1 hides behind the transforms of iteration it */
  uint2 pa[16], pb[16];
  constexpr bool stereo = STEREO; /* the channel handling is compiled in; k_freq_frames picks per workgroup */
  /* loads of registers [4 * part, 4 * part + 4) of both frames: the iteration issues its 32 loads in
   * four instalments between the phases of the transform (32 at once fill the vector-memory
   * queue and the wave sits in front of it: 1.6 k cycles per iteration) */
  auto fetch = [&](int f_, int part) {
    const int fa = min(f_, sg.n_frames - 1), fb = min(f_ + 1, sg.n_frames - 1);
    if (stereo) {
      const uint2 *qa = reinterpret_cast<const uint2 *>(p + (size_t)fa * 1024) + basep;
      const uint2 *qb = reinterpret_cast<const uint2 *>(p + (size_t)fb * 1024) + basep;
#pragma unroll
      for (int r = 4 * part; r < 4 * part + 4; ++r) {
        const int e = r == 0 ? base0 - basep : KG(r);
        pa[r] = qa[e]; pb[r] = qb[e];
      }
    } else {
      const unsigned *qa = reinterpret_cast<const unsigned *>(p + (size_t)fa * 512) + basep;
      const unsigned *qb = reinterpret_cast<const unsigned *>(p + (size_t)fb * 512) + basep;
#pragma unroll
      for (int r = 4 * part; r < 4 * part + 4; ++r) {
        const int e = r == 0 ? base0 - basep : KG(r);
        pa[r] = make_uint2(qa[e], 0u);
        pb[r] = make_uint2(qb[e], 0u);
      }
    }
  };
  /* the two mono samples (one complex DFT input) a lane takes from an 8-byte (stereo) or 4-byte
   * (mono) word, for both frames of the pair:
   * stereo, ref :69-75: (float)((L + R) / 2), the integer average truncates towards zero —
   * L + R converts exactly, half of it is exact, v_trunc does what the C division does;
   * mono, ref :76-80: (float)s */
  auto mono2 = [&](const uint2 wa, const uint2 wb, bl_f2 &s0, bl_f2 &s1) {
    const int a0 = (int)(short)(wa.x & 0xFFFFu), a1 = (int)(short)(wa.x >> 16);
    const int b0 = (int)(short)(wb.x & 0xFFFFu), b1 = (int)(short)(wb.x >> 16);
    if (stereo) {
      const int a2 = (int)(short)(wa.y & 0xFFFFu), a3 = (int)(short)(wa.y >> 16);
      const int b2 = (int)(short)(wb.y & 0xFFFFu), b3 = (int)(short)(wb.y >> 16);
      const bl_f2 h0 = (bl_f2){(float)(a0 + a1), (float)(b0 + b1)} * 0.5f;
      const bl_f2 h1 = (bl_f2){(float)(a2 + a3), (float)(b2 + b3)} * 0.5f;
      s0 = (bl_f2){__builtin_truncf(h0.x), __builtin_truncf(h0.y)};
      s1 = (bl_f2){__builtin_truncf(h1.x), __builtin_truncf(h1.y)};
    } else {
      s0 = (bl_f2){(float)a0, (float)b0};
      s1 = (bl_f2){(float)a1, (float)b1};
    }
  };
  auto bc = [](float w) { return (bl_f2){w, w}; };
  /* the lane's twiddles of the in-register passes stay in registers for the whole song; pass(32)'s carries the sign
   * of its half of the pair (lv_pass32_mul) */
  const c2f w32 = lvtw[LV_TW_P32 * 16 + l], w64 = lvtw[LV_TW_P64 * 16 + l];
  const float ws32 = lo8 ? -w32.im : w32.im;
  c2f w128[2], w256[4];
#pragma unroll
  for (int q = 0; q < 2; ++q) w128[q] = lvtw[(LV_TW_P128 + q) * 16 + l];
#pragma unroll
  for (int q = 0; q < 4; ++q) w256[q] = lvtw[(LV_TW_P256 + q) * 16 + l];
  const bl_f2 SH = bc(tb.lv_leafc[0]), C1 = bc(tb.lv_leafc[1]), C3 = bc(tb.lv_leafc[2]);
  const bl_f2 *hann2 = reinterpret_cast<const bl_f2 *>(hann) + basep;
  const int n_iter = (sg.n_frames + FPI - 1) / FPI;
#pragma unroll
  for (int part = 0; part < 4; ++part) fetch(8 * wave + 2 * gl, part);
  for (int it = 0; it < n_iter; ++it) {
    const int f = it * FPI + 8 * wave + 2 * gl;
    /* SCAN: the input stage with the statistics (a third of the iteration's instructions and all of its LDS
     * atomics) runs at priority 3, the leaves at 2, the rest at 0: of the two waves of a SIMD the one
     * that is feeding the LDS wins the VALU.  32.6 vs 33.8 ms per 4 096 songs (round 4); the other orders (later
     * phases first, as in k_env_windows3) made no difference, and k_freq_frames gains nothing from any. */
    if (SCAN) __builtin_amdgcn_s_setprio(3);
    bl_f2 re[16], im[16];
    /* SCAN: the statistics of every word as the transform's input stage consumes it (its registers die here).  The
     * frames of a song's last iteration that lie beyond its end (their loads were clamped onto the last frame) are
     * not counted. */
    int s32 = 0;
    auto word = [&](unsigned w) { scan_word(w, s32, sq, lds_hist, true); };
    const bool full = it + 1 < n_iter; /* wave-uniform */
    const bool va = f < sg.n_frames, vb = f + 1 < sg.n_frames;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      bl_f2 xr, xi;
      if (SCAN) {
        if (full) {
          word(pa[r].x); word(pb[r].x);
          if (stereo) { word(pa[r].y); word(pb[r].y); }
        } else {
          if (va) { word(pa[r].x); if (stereo) word(pa[r].y); }
          if (vb) { word(pb[r].x); if (stereo) word(pb[r].y); }
        }
      }
      mono2(pa[r], pb[r], xr, xi);
      const bl_f2 h = hann2[r == 0 ? base0 - basep : KG(r)]; /* hann[2 m], hann[2 m + 1] of this register's element m */
      re[r] = xr * (bl_f2){h.x, h.x};
      im[r] = xi * (bl_f2){h.y, h.y};
    }
    if (SCAN) sum += s32;
    fetch(f + FPI, 0);
    if (SCAN) __builtin_amdgcn_s_setprio(2);
    lv_leaves<bl_f2>(t16, re, im, SH, C1, C3);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      c2p v; v.re = re[r]; v.im = im[r];
      gx[r * 17 + l] = v;
    }
    bl_wave_sync();
    fetch(f + FPI, 1);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const c2p v = gx[l * 17 + j];
      re[j] = v.re; im[j] = v.im;
    }
    bl_wave_sync();
    fetch(f + FPI, 2);
    if (SCAN) __builtin_amdgcn_s_setprio(0);
    { /* pass(32) @ 0, 64, 96, 128, 192: registers (R, R + 1), lanes l and l ^ 8 */
      auto sel = [&](bl_f2 a, bl_f2 b) { return lo8 ? a : b; };
      constexpr int R32[5] = {0, 4, 6, 8, 12};
#pragma unroll
      for (int b = 0; b < 5; ++b) {
        const int R = R32[b];
        bl_f2 tA, tB;
        lv_pass32_mul<bl_f2>(re[R + 1], im[R + 1], bc(w32.re), bc(ws32), tA, tB);
        const bl_f2 pA = bl_dpp_f2<0x128>(tA), pB = bl_dpp_f2<0x128>(tB); /* row_ror:8 = lane ^ 8 */
        lv_pass32_fin<bl_f2>(re[R], im[R], re[R + 1], im[R + 1], tA, tB, pA, pB, sel);
      }
    }
    lv_pass_inlane<bl_f2, 0, 1>(re, im, bc(w64.re), bc(w64.im));
    lv_pass_inlane<bl_f2, 8, 1>(re, im, bc(w64.re), bc(w64.im));
    lv_pass_inlane<bl_f2, 12, 1>(re, im, bc(w64.re), bc(w64.im));
    lv_pass_inlane<bl_f2, 0, 2>(re, im, bc(w128[0].re), bc(w128[0].im));
    lv_pass_inlane<bl_f2, 1, 2>(re, im, bc(w128[1].re), bc(w128[1].im));
    lv_pass_inlane<bl_f2, 0, 4>(re, im, bc(w256[0].re), bc(w256[0].im));
    lv_pass_inlane<bl_f2, 1, 4>(re, im, bc(w256[1].re), bc(w256[1].im));
    lv_pass_inlane<bl_f2, 2, 4>(re, im, bc(w256[2].re), bc(w256[2].im));
    lv_pass_inlane<bl_f2, 3, 4>(re, im, bc(w256[3].re), bc(w256[3].im));
    fetch(f + FPI, 3);
    /* rdft.c's post-pass: the partner of i = l + 16 j is Z[256 - i], register 15 - j of lane 16 - l (row mirror +
     * shift by one); lane 0 is its own partner and takes its register 16 - j */
    bl_f2 own[8], mir[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const bl_f2 zr = j ? re[16 - j] : re[0];
      const bl_f2 zi = j ? im[16 - j] : im[0];
      const bl_f2 pr = bl_dpp_f2_old<0x111>(zr, bl_dpp_f2<0x140>(re[15 - j]));
      const bl_f2 pi = bl_dpp_f2_old<0x111>(zi, bl_dpp_f2<0x140>(im[15 - j]));
      const c2f w = lvtw[(LV_TW_POST + j) * 16 + l];
      lv_post_power<bl_f2>(re[j], im[j], pr, pi, bc(w.re), bc(w.im), bc(0.5f), own[j], mir[j]);
    }
    const bl_f2 mid = lv_mid_power<bl_f2>(re[8], im[8]);
    /* ref :88-93: re*re + im*im of bin d, for d = 1..255 (lane 0's own[0] / mir[0] are bins 0 / 256: never read) */
    float *sa = stage + (2 * gl) * BL_FREQ_SROW, *sb = sa + BL_FREQ_SROW;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      sa[l + 16 * k] = own[k].x; sb[l + 16 * k] = own[k].y;
      sa[256 - l - 16 * k] = mir[k].x; sb[256 - l - 16 * k] = mir[k].y;
    }
    if (l == 0) { sa[128] = mid.x; sb[128] = mid.y; }
    bl_wave_sync();
    /* the baton: frames 8 WAVES it + 8 w .. + 7 join the running spectrum after those of wave w - 1 */
    const int turn = WAVES * it + wave;
    /* frames beyond the song's last one (their loads were clamped onto it) are not added */
    const int n_live = sg.n_frames - (it * FPI + 8 * wave);
    /* this wave's 8 x 4 power values per lane are fetched BEFORE it asks for the baton (they are
     * its own), all 32 reads in flight at once; holding the baton then costs one read of the
     * running spectrum, eight dependent adds and a write.  (Reading them one by one behind the
     * frame-count test made the hold 3.2 k cycles: four waves x 3.2 k was the whole iteration.) */
    float sv[4][8];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int fr = 0; fr < 8; ++fr) sv[q][fr] = stage[fr * BL_FREQ_SROW + lane + 64 * q];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    while (__builtin_amdgcn_readfirstlane(relay[0]) < turn) __builtin_amdgcn_s_sleep(1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    float acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = accv[lane + 64 * q];
    if (n_live >= 8) { /* every iteration but a song's last: no per-frame test (hipcc makes selects of it) */
#pragma unroll
      for (int fr = 0; fr < 8; ++fr)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          acc[q] += sv[q][fr];
          /* one v_add_f32 each: paired into v_pk_add_f32 the operands need more moves than the
           * pairing saves, and this is the stretch during which the wave holds the baton */
          asm volatile("" : "+v"(acc[q]));
        }
    } else {
#pragma unroll
      for (int fr = 0; fr < 8; ++fr)
        if (fr < n_live) { /* wave-uniform */
#pragma unroll
          for (int q = 0; q < 4; ++q) acc[q] += sv[q][fr];
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) accv[lane + 64 * q] = acc[q];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    bl_wave_sync();
    if (lane == 0) relay[0] = turn + 1;
  }
  if (SCAN) {
    /* the samples behind the last whole frame (fewer than 512 per channel) */
    for (int i = sg.n_frames * 512 * sg.channels + tid; i < sg.n; i += 64 * WAVES) {
      const int sv = (int)p[i];
      sum += sv;
      sq += (unsigned)(sv * sv);
      const unsigned b = (unsigned)(sv + BL_HIST_BINS / 2);
      if (b < BL_HIST_BINS) atomicAdd(&lh[b], 1u);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      sum += __shfl_down(sum, off);
      sq += __shfl_down(sq, off);
    }
    if (lane == 0) {
      atomicAdd(&st->sum, (unsigned long long)sum);
      atomicAdd(&st->sumsq, sq);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); /* the inline-asm adds are invisible to hipcc's counters */
  }
  __syncthreads();
  if (WAVES == 4 || tid < 256) spectrum[(size_t)blockIdx.x * 256 + tid] = accv[tid];
  if (SCAN)
    for (int i = tid; i < BL_HIST_BINS; i += 64 * WAVES) gh[i] = lh[i]; /* the workgroup owns the song: plain stores */
}

#undef KG

/* one workgroup per song; the channel count is uniform per workgroup, so the branch costs one
 * scalar compare and each path keeps its compiled-in input side */
__global__ __launch_bounds__(256, 2) void k_freq_frames(const int16_t *__restrict__ pcm,
                                                        const bl_dsong *__restrict__ songs,
                                                        bl_tables tb, float *spectrum) {
  const bl_dsong sg = songs[blockIdx.x];
  if (sg.channels == 2) freq_frames_lavc<true, 4, false>(pcm, sg, tb, spectrum, nullptr, nullptr);
  else freq_frames_lavc<false, 4, false>(pcm, sg, tb, spectrum, nullptr, nullptr);
}

/* k_freq_scan: k_freq_frames and k_pcm_scan in one pass over the PCM — one 512-thread workgroup per song and CU
 * (the same eight waves per CU as two k_freq_frames workgroups, one histogram) */
__global__ __launch_bounds__(64 * BL_FREQ_SCAN_WAVES) void k_freq_scan(const int16_t *__restrict__ pcm,
                                                                       const bl_dsong *__restrict__ songs,
                                                                       bl_tables tb, float *spectrum,
                                                                       bl_dstats *stats, unsigned *hist) {
  const bl_dsong sg = songs[blockIdx.x];
  bl_dstats *st = stats + blockIdx.x;
  unsigned *gh = hist + (size_t)blockIdx.x * BL_HIST_BINS;
  if (sg.channels == 2) freq_frames_lavc<true, BL_FREQ_SCAN_WAVES, true>(pcm, sg, tb, spectrum, st, gh);
  else freq_frames_lavc<false, BL_FREQ_SCAN_WAVES, true>(pcm, sg, tb, spectrum, st, gh);
}

__global__ __launch_bounds__(256) void k_freq_finish(const float *__restrict__ spectrum,
                                                     const bl_dsong *__restrict__ songs,
                                                     bl_amd_song_result *res) {
  __shared__ float ps[256];
  __shared__ float wmax[4];
  const int d = threadIdx.x, song = blockIdx.x;
  const float acc = spectrum[(size_t)song * 256 + d];
  /* ref :97-102: sqrt(ps / 512), peak over d = 1..256 (ps[256] is 0) */
  float v = d == 0 ? 0.f : (float)sqrt((double)(acc / 512));
  float m = v;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_down(m, off));
  if ((d & 63) == 0) wmax[d >> 6] = m;
  __syncthreads();
  const float peak = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
  /* ref :105-107 */
  ps[d] = (float)(20 * log10((double)(v / peak)) - 3);
  __syncthreads();
  if (d == 0) { /* ref :110-139, f32 sequential sums, divisors 50 / 57 / 115 */
    float b0 = (ps[2] + ps[4]) / 2;
    float b1 = (ps[6] + ps[8]) / 2;
    float b2 = 0, b3 = 0, b4 = 0;
    for (int i = 10; i <= 60; ++i) b2 += ps[i];
    b2 /= 50;
    for (int i = 61; i <= 118; ++i) b3 += ps[i];
    b3 /= 57;
    for (int i = 119; i <= 234; ++i) b4 += ps[i];
    b4 /= 115;
    const float sum = b4 + b3 + b2 - b0 - b1;
    bl_amd_song_result *r = res + songs[song].out_idx;
    r->freq_peak = peak;
    r->v.frequency = (float)((1. / 3.) * (double)sum + 68. / 3.);
  }
}

/* ------------------------------------------------------------------------- */
/* envelope windows: shared arithmetic                                        */

/* ref tempo_atk_sort.c:109-114 for one sample, halved: x/2 with x = RN(((s/2^15) - (mean/2^15)) / vd)
 * = RN(k / V), k = s - mean (exact), V = variance * 2^-15 (power-of-two scalings commute with
 * rounding).  With 1 / (2V) = r + r_lo to ~2^-106, kd*r + RN(kd*r_lo) is k / (2V) with a relative
 * error below 2^-104 before the fma's single rounding; k / (2V) = k * 2^14 / variance with
 * |k| < 2^17, variance < 2^31 is either exactly representable or at least 2^-70 (relative) away
 * from the nearest rounding boundary, so the result is the correctly rounded quotient.
 * Why halved: every operation downstream (add, multiply by a constant, fma) scales exactly by a
 * power of two — nothing here comes near the subnormals — so the FIR outputs are y/2, the
 * spectrum X/2 and the power terms |X|^2 / 4 with bit-identical mantissas: the 1/4 that the
 * real-input split of the DFT owes (bl_fft512_power1) comes for free. */
__device__ __forceinline__ double bl_norm(int k, double rcp, double rcp_lo) {
  const double kd = (double)k;
  return __builtin_fma(kd, rcp, kd * rcp_lo);
}

/* ref :123-138.  The reference starts from y = 0 and adds nine products; 0 + c7 * p is c7 * p
 * bit for bit here (p = +0 gives +0: the inputs are never -0), so the first add is not issued. */
#define BL_FIR(X)                                                   \
  ({                                                                \
    double y_ = BL_C7 * (X(7) + X(9));                              \
    y_ += BL_C6 * (X(6) + X(10));                                   \
    y_ += BL_C5 * (X(5) + X(11));                                   \
    y_ += BL_C4 * (X(4) + X(12));                                   \
    y_ += BL_C3 * (X(3) + X(13));                                   \
    y_ += BL_C2 * (X(2) + X(14));                                   \
    y_ += BL_C1 * (X(1) + X(15));                                   \
    y_ += X(8) * BL_C8;                                             \
    y_ += BL_C0 * (X(0) + X(16));                                   \
    y_;                                                             \
  })

/* The same sum with each product folded into the running sum by an fma: eight roundings fewer per
 * output and eight instructions fewer (17 instead of 25).  NOT the reference's arithmetic: an
 * output differs by a few 1e-16 of its largest partial sum, which is the class of difference the
 * DFT behind it already has (ours, not FFTW's) and which the results see only through the f32
 * roundings of the ordered sum.  Selected by BL_AMD_FIR_FUSED=1; DESIGN.md §4.1 has the measured
 * flip rates that decide whether it is used. */
#define BL_FIR_FUSED(X)                                             \
  ({                                                                \
    double y_ = BL_C7 * (X(7) + X(9));                              \
    y_ = __builtin_fma(BL_C6, X(6) + X(10), y_);                    \
    y_ = __builtin_fma(BL_C5, X(5) + X(11), y_);                    \
    y_ = __builtin_fma(BL_C4, X(4) + X(12), y_);                    \
    y_ = __builtin_fma(BL_C3, X(3) + X(13), y_);                    \
    y_ = __builtin_fma(BL_C2, X(2) + X(14), y_);                    \
    y_ = __builtin_fma(BL_C1, X(1) + X(15), y_);                    \
    y_ = __builtin_fma(X(8), BL_C8, y_);                            \
    y_ = __builtin_fma(BL_C0, X(0) + X(16), y_);                    \
    y_;                                                             \
  })
/* Mode 2: the normalisation folded into the taps.  k = s - mean is an exact integer and so is every
 * pair sum; c'_m = RN(c_m / (2 vprime)) (k_song_prep) carries the division.  One rounding per tap
 * (the product inside the fma) where the reference has three (quotient, pair sum, product): the
 * output differs from the reference's by a few 1e-16 of its largest partial sum, as in mode 1, and
 * the 66 f64 instructions per round that normalise the samples are gone.  FC(m) names tap m. */
#define BL_FIR_FOLD(X, FC)                                          \
  ({                                                                \
    double y_ = FC(7) * (X(7) + X(9));                              \
    y_ = __builtin_fma(FC(6), X(6) + X(10), y_);                    \
    y_ = __builtin_fma(FC(5), X(5) + X(11), y_);                    \
    y_ = __builtin_fma(FC(4), X(4) + X(12), y_);                    \
    y_ = __builtin_fma(FC(3), X(3) + X(13), y_);                    \
    y_ = __builtin_fma(FC(2), X(2) + X(14), y_);                    \
    y_ = __builtin_fma(FC(1), X(1) + X(15), y_);                    \
    y_ = __builtin_fma(X(8), FC(8), y_);                            \
    y_ = __builtin_fma(FC(0), X(0) + X(16), y_);                    \
    y_;                                                             \
  })
#ifndef BL_FIR_FUSED_DEFAULT
#define BL_FIR_FUSED_DEFAULT 2
#endif
#define BL_FIR_SEL(MODE, X, FC) ((MODE) == 2 ? BL_FIR_FOLD(X, FC) : (MODE) == 1 ? BL_FIR_FUSED(X) : BL_FIR(X))

/* ------------------------------------------------------------------------- */
/* k_env_windows3: normalise + FIR + DFT + ordered sum, wave-autonomous        */
/*
 * One workgroup per CU: 7 compute waves + 1 summing wave (2 waves per SIMD, 213-220 VGPRs), no workgroup
 * barrier inside the loop.
 *
 * A compute wave walks a CONTIGUOUS run of rounds of four windows (one window per 16-lane group; the song's
 * rounds are split evenly over the compute waves of its workgroups).  Its private LDS slice holds five blocks
 * of 256 filtered samples as a ring: a round filters the 1 024 new samples (16 outputs per lane, from the 32
 * samples the lane loads itself: no cross-lane shift) into the four places the previous round has released and
 * finds the block it shares with that round where it was left.  Then the four 512-point f64 DFTs: inputs as
 * aligned ds_read_b128, two radix-16 passes over 16 lanes x 16 registers with the re and im transposes through
 * the place of the window's own block, partner values of the real-input split through DPP (row mirror + shift),
 * and the 4 x 257 power terms.  The first round of a run is preceded by a short pass that filters the one block
 * it cannot inherit.
 *
 * The f32-rounded, strictly ordered sum of ref tempo_atk_sort.c:142-149 is a dependent chain of three
 * instructions per term.  It runs on the eighth wave, IN TWO HALVES ON TWICE THE LANES: a compute wave hands over
 * terms 0..129 of the round it has just finished together with terms 130..256 of the round BEFORE (kept in 16
 * registers for one round); the summing wave adds the first halves on lanes 0-27 and, continuing from the partial
 * sums of its previous step, the second halves on lanes 32-59 — 390 dependent instructions per tile of 28 windows
 * instead of 771, the same additions in the same order.  The energies leave one step later.  Hand-over through
 * LDS sequence words (waves of one workgroup are always co-resident, so the bounded spins cannot deadlock).
 *
 * Who gets the VALU.  A SIMD gives its VALU to the wave with the highest s_setprio value and, among equals, to
 * the OLDEST wave — strictly: 96 % of the issue slots to the older of two busy waves (tools/gen_ubench_issue.py).
 * Two compute waves that share a SIMD and are held in step by the tile hand-over therefore do not share it: the
 * older one runs its round and waits, the younger one then runs alone with every LDS round trip of its own
 * exposed, and the tile waits for it.  PRIO gives every phase of a round a priority (4 bits per phase, phase 0 in
 * the lowest digit); the shipped table 0x222011 runs the second half of a round (transposes, second DFT pass,
 * hand-over, power terms) at 2, normalise + FIR and the FIR -> DFT exchange at 1 and the first DFT pass at 0:
 * whichever wave is further along — the one the tile is waiting for — wins, whatever its age.  278 vs 306 ms per
 * 8 192 S180 songs with identical results for 0x222111 (profiles/r04_env_variants.json);
 * the first pass at 0 another 1.0-1.3 % (three sweeps of five rounds; every table with that digit at 0 and the
 * second half at 2 or 3 did the same).
 */
#define EV_CWAVES 7
#define EV_TILE (4 * EV_CWAVES)             /* windows per tile */
#define EV_TROW 258                          /* terms row stride (doubles): even -> 16-byte rows */

/* cross-lane move of a double through DPP (two 32-bit moves).  CTRL 0x110+m = row_shr:m inside
 * each 16-lane row, 0x140 = row_mirror; lanes without a source read 0 (bound_ctrl) */
template <int CTRL> __device__ __forceinline__ double bl_dpp_f64(double v) {
  const unsigned long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(b & 0xFFFFFFFFull), CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(b >> 32), CTRL, 0xF, 0xF, true);
  return __longlong_as_double(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}

/* the same with a value for the lanes that have no source (they keep `old`) */
template <int CTRL> __device__ __forceinline__ double bl_dpp_f64_old(double old, double v) {
  const unsigned long long b = __double_as_longlong(v), o = __double_as_longlong(old);
  const int lo = __builtin_amdgcn_update_dpp((int)(unsigned)(o & 0xFFFFFFFFull), (int)(unsigned)(b & 0xFFFFFFFFull),
                                             CTRL, 0xF, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp((int)(unsigned)(o >> 32), (int)(unsigned)(b >> 32), CTRL, 0xF, 0xF, false);
  return __longlong_as_double(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}

/* Hand-over fences between waves of one workgroup: everything handed over lives in LDS, so
 * only the LDS counter has to drain.  A workgroup-scope fence also waits for vmcnt(0), i.e.
 * for the summing wave's global stores of the finished energies (and for prefetches in
 * flight) — ~1.5 k cycles of HBM latency on the critical path of every tile. */
__device__ __forceinline__ void ev_lds_release() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}
__device__ __forceinline__ void ev_lds_acquire() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

__device__ __forceinline__ void ev_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

/* A block of 256 filtered samples as 16 rows of eight 16-byte units (two samples each): unit c of row r at 16-byte
 * slot 9 r + 2 c, i.e. even rows on even slots and odd rows on the odd slots between them.  Both sides of the
 * FIR -> DFT exchange are then conflict-free: the eight lanes the LDS serves together write unit i of eight
 * consecutive rows (slots 9 r + 2 i: all different mod 8), and the sixteen lanes it serves together read units 0..7
 * of the two rows 2 m1 and 2 m1 + 1 (slots {0, 2, .. 14} and 9 + {0, 2, .. 14}: all different mod 16).  Rows 18
 * doubles apart (slot 9 r + c, rounds 2-3) served every DFT-input read in two turns: 64 of the 580 LDS cycles of a
 * round, the whole SQ_LDS_BANK_CONFLICT count of the kernel (tools/lds_model.py).  160 slots per block keep the
 * blocks of the four windows a multiple of 16 slots apart. */
#define EV3_BLK 320                          /* doubles per block */
#define EV3_ROW(r) (18 * (r))                /* first double of row r */
#define EV3_UNIT(c) (4 * (c))                /* first double of unit c within its row */
#define EV3_HEADS (5 * EV3_BLK)
#define EV3_SLOTS (EV3_HEADS + 64)           /* + 4 x 16 window heads */
#define EV3_TERMS_OFF (EV_CWAVES * EV3_SLOTS * 8)
#define EV3_TW_OFF (EV3_TERMS_OFF + EV_TILE * EV_TROW * 8)
#define EV3_FLAG_OFF (EV3_TW_OFF + 2 * 256 * 16)
#define EV3_ZERO_OFF (EV3_FLAG_OFF + 128)   /* 16 bytes of zeros: the 65th term pair of a second-half lane */
#define EV3_LDS_BYTES (EV3_ZERO_OFF + 16)

#define EV_PROBE_ROUNDS 16
#define EV_PROBE_SLOTS 12
#ifndef BL_ENV_PRIO
#define BL_ENV_PRIO 0x222011 /* the priority table the product launches */
#endif
/* the priority tables the measurement build instantiates beside it (tools/env_ab.py) */
#define EV_PRIO_TABS(X) X(0x000000) X(0x111111) X(0x322110) X(0x321000) X(0x222110) X(0x222111) X(0x232011) X(0x222112)
/* PROBE (measurement builds): s_memtime stamps of one workgroup's phases into `probe` */
template <int FIR_MODE, int PRIO, bool PROBE>
__global__ __launch_bounds__(64 * (EV_CWAVES + 1)) void k_env_windows3(
    const int16_t *__restrict__ pcm, const bl_dsong *__restrict__ songs,
    const bl_dstats *__restrict__ stats, bl_tables tb, float *energies, double *lc, long long *probe) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  double *terms = reinterpret_cast<double *>(smem + EV3_TERMS_OFF); /* [EV_TILE][257] */
  c2d *tw256 = reinterpret_cast<c2d *>(smem + EV3_TW_OFF);
  c2d *tw512 = tw256 + 256;
  typedef __attribute__((address_space(3))) volatile int lds_vint;
  lds_vint *flags = (lds_vint *)(smem + EV3_FLAG_OFF); /* [0..6] published by the compute waves, [8] by the summing wave */

  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), ln = tid & 63, g = ln >> 4, l = ln & 15;
  const bool probing = PROBE && blockIdx.x == 0 && blockIdx.y == 0 && probe != nullptr;
  auto stamp = [&](int round, int slot) {
    if (PROBE) {
      __builtin_amdgcn_sched_barrier(0); /* no arithmetic moves across a stamp */
      if (probing && round < EV_PROBE_ROUNDS && ln == 0)
        probe[(wave * EV_PROBE_ROUNDS + round) * EV_PROBE_SLOTS + slot] = (long long)__builtin_amdgcn_s_memtime();
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  const bl_dsong sg = songs[blockIdx.y];
  const bl_dstats st = stats[blockIdx.y];
  const int16_t *p = pcm + sg.pcm_off;
  if (tid < 256) {
    tw256[tid] = tb.tw256_d[((tid & 15) * (tid >> 4)) & 255];
    tw512[tid] = tb.tw512_d[tid];
  }
  if (tid < 36) flags[tid] = 0; /* the sequence words and the zero pair behind them */
  if (tid < EV_TILE) terms[tid * EV_TROW + 257] = 0.0; /* the pad behind term 256 is read as a term */
  __syncthreads();

  /* rounds of four windows, split evenly over the compute waves of the song's workgroups */
  const int n_rounds = (sg.n_windows + 3) / 4;
  const int n_units = EV_CWAVES * (int)gridDim.x;
  auto run_begin = [&](int u) -> int { return (int)((long long)n_rounds * u / n_units); };
  const int u0 = EV_CWAVES * (int)blockIdx.x;
  int steps = 0;
  for (int c = 0; c < EV_CWAVES; ++c) steps = max(steps, run_begin(u0 + c + 1) - run_begin(u0 + c));
  const int n_used = 256 * (sg.n_windows + 1);
  int seq = 0;

  if (wave == EV_CWAVES) {
    /* ---- summing wave ---- */
    __builtin_amdgcn_s_setprio(3);
    {
      /* Step st (1-based): every compute wave has published st.  Lane i < 28 (row i) adds terms 0..129 of round st
       * starting from 0 and keeps the partial sum; lane 32 + i takes the partial sum lane i made in step st - 1 and
       * continues round st - 1 over terms 130..256, then stores the energy.  The second-half lanes read 127 terms and
       * three zeros — (float)((double)sum + 0.0) is sum — so that every lane runs the same 130 additions.  Step
       * steps + 1 only has second halves (the compute waves publish them after their last round). */
      const int rowi = min(ln & 31, EV_TILE - 1);
      const bool own = ln < 32;
      const int c2 = min(rowi >> 2, EV_CWAVES - 1);
      const int q0 = run_begin(u0 + c2), q1 = run_begin(u0 + c2 + 1);
      const double2 *row = reinterpret_cast<const double2 *>(terms + rowi * EV_TROW);
      const double2 *zero2 = reinterpret_cast<const double2 *>(smem + EV3_ZERO_OFF);
      const double2 *tp = own ? row : row + 65;
      const double2 *tail = own ? row + 64 : zero2;
      float psum = 0.f;
      for (int st = 1; st <= steps + 1; ++st) {
        stamp(st - 1, 0);
        for (;;) {
          const int f = ln < EV_CWAVES ? flags[ln] : st;
          if (__all(f >= st)) break;
          __builtin_amdgcn_s_sleep(1);
        }
        ev_lds_acquire();
        stamp(st - 1, 1);
        const int rr = own ? st : st - 1; /* the round (1-based) this lane works on */
        const int rho = q0 + rr - 1, w = 4 * rho + (ln & 3);
        const bool live = (ln & 31) < EV_TILE && rr >= 1 && rr <= steps && rho < q1 && w < sg.n_windows;
        /* the partial sums of the previous step move from lane i to lane 32 + i */
        const float carried = __shfl(psum, ln & 31);
        float sum = own ? 0.f : carried;
        if (live) {
          /* 65 pairs of terms, fetched 8 pairs at a time, one block ahead of the chain that adds them: lgkmcnt
           * counts to 15, so with more than two blocks of 8 in flight the wait in front of a chain can only be
           * for everything.  The scheduling barriers keep hipcc from sinking the loads back down in front of
           * their uses, which would put one LDS latency per block on the step's critical path. */
          double2 ta[8], tb2[8];
#define EV_LOAD8(T, B) _Pragma("unroll") for (int k = 0; k < 8; ++k) T[k] = tp[8 * (B) + k];
#define EV_SUM8(T)                                                                                      \
  _Pragma("unroll") for (int k = 0; k < 8; ++k) {                                                       \
    sum = (float)((double)sum + T[k].x);                                                                \
    sum = (float)((double)sum + T[k].y);                                                                \
  }
#define EV_SB __builtin_amdgcn_sched_barrier(0);
          EV_LOAD8(ta, 0) EV_LOAD8(tb2, 1) EV_SB
          EV_SUM8(ta) EV_SB EV_LOAD8(ta, 2) EV_SB
          EV_SUM8(tb2) EV_SB EV_LOAD8(tb2, 3) EV_SB
          EV_SUM8(ta) EV_SB EV_LOAD8(ta, 4) EV_SB
          EV_SUM8(tb2) EV_SB EV_LOAD8(tb2, 5) EV_SB
          EV_SUM8(ta) EV_SB EV_LOAD8(ta, 6) EV_SB
          EV_SUM8(tb2) EV_SB EV_LOAD8(tb2, 7)
          const double2 tl = tail[0];
          EV_SB
          EV_SUM8(ta) EV_SB
          EV_SUM8(tb2)
          sum = (float)((double)sum + tl.x);
          sum = (float)((double)sum + tl.y);
#undef EV_LOAD8
#undef EV_SUM8
#undef EV_SB
        }
        /* the rows are read: hand them back before the energies are stored */
        ev_lds_release();
        if (ln == 0) flags[8] = st;
        psum = sum;
        if (live && !own) {
          energies[sg.env_off + w] = sum;
          lc[sg.env_off + w] = bl_tail_compress((double)sum, tb.log101);
        }
        stamp(st - 1, 2);
      }
    }
    return;
  }

  /* ---- compute waves ---- */
  /* phase boundary k (0..5) of a round: the priority of the phase that starts here.  k is a literal at every
   * call: one s_setprio (which is also a scheduling barrier: the phases stay apart in the instruction stream) */
  auto phase = [&](int k) {
    const int pr = (PRIO >> (4 * k)) & 3;
    if (pr == 0) __builtin_amdgcn_s_setprio(0);
    else if (pr == 1) __builtin_amdgcn_s_setprio(1);
    else if (pr == 2) __builtin_amdgcn_s_setprio(2);
    else __builtin_amdgcn_s_setprio(3);
  };
  double *buf = reinterpret_cast<double *>(smem) + wave * EV3_SLOTS;
  const int mean = st.mean;
  const double rcp = st.rcp, rcp_lo = st.rcp_lo;
#define FC(m) st.firc[m]
  /* mode 2 filters the integers k = s - mean themselves (the taps carry the division) */
  auto nrm = [&](int k) -> double { return FIR_MODE == 2 ? (double)k : bl_norm(k, rcp, rcp_lo); };
  const int r0 = run_begin(u0 + wave), r1 = run_begin(u0 + wave + 1);
  /* all 15 pass-1 twiddles of the lane live in registers for the whole run: 194 / 192 / 213 VGPRs in FIR modes
   * 0 / 1 / 2, no spill.  (Until round 6 modes 0 / 1 kept 12 and read three per round from LDS — a relic of a 220-VGPR
   * build; the 185-VGPR one had the room: 44.15 -> 41.99 ms per 1 024 S180 songs in mode 0, identical records.  The
   * eight split twiddles W512^(l + 16 k0) as well — 218 VGPRs — made mode 0 10 % SLOWER and mode 2 no faster:
   * profiles/EXPERIMENTS.md.) */
  constexpr int EV3_W1_REGS = 16;
  c2d w1r[EV3_W1_REGS];
#pragma unroll
  for (int k1 = 1; k1 < EV3_W1_REGS; ++k1) {
    w1r[k1] = tw256[k1 * 16 + l];
    asm volatile("" : "+v"(w1r[k1].re), "+v"(w1r[k1].im));
  }
  int base5 = (4 * r0) % 5; /* ring position of block 4 rho, the block shared with the previous round */

  if (r0 < r1) { /* the block the first round cannot inherit: samples [1024 r0, 1024 r0 + 256) */
    const int s0 = 1024 * r0 + 4 * ln; /* this lane's 4 outputs; inputs [s0 - 16, s0 + 4) */
    double r[20];
#pragma unroll
    for (int u = 0; u < 5; ++u) {
      const int i0 = s0 - 16 + 4 * u;
      const bool ok = i0 >= 0 && i0 + 4 <= n_used;
      const uint2 v = *reinterpret_cast<const uint2 *>(p + (ok ? i0 : 0));
      const unsigned w[2] = {ok ? v.x : 0u, ok ? v.y : 0u};
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int lo = (int)(short)(w[k] & 0xFFFFu), hi = (int)(short)(w[k] >> 16);
        r[4 * u + 2 * k] = ok ? nrm(lo - mean) : 0.0;
        r[4 * u + 2 * k + 1] = ok ? nrm(hi - mean) : 0.0;
      }
    }
    double *dst = buf + base5 * EV3_BLK + EV3_ROW(ln >> 2) + EV3_UNIT(2 * (ln & 3));
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#define XW(m) r[i + 16 - (m)]
      dst[EV3_UNIT(i >> 1) + (i & 1)] = BL_FIR_SEL(FIR_MODE, XW, FC);
#undef XW
    }
  }

  /* lane ln owns outputs 16 ln .. 16 ln + 15 of the round's 1 024 new samples and loads the 32
   * samples they read (four 16-byte loads), plus the sample that starts its zero-state output;
   * fetched one round ahead.  Buffer loads: the song is the buffer, the lane's byte offset one register
   * that moves on by 2 048 per round, and what lies beyond the song's last window reads as zero by the
   * hardware's range check — it only reaches windows that are never summed, so any sample will do there.
   * Two VALU instructions per round instead of the 21 that clamped 64-bit addresses took (round 5). */
  uint4 pre[4];
  short preh;
  /* descriptor word 3 = 0x00020000 (DATA_FORMAT 32) and "out of range reads as zero" are the gfx9 / CDNA raw-buffer
   * rules; num_records and the offsets are 32-bit byte counts: a song is at most INT_MAX samples (bl_dsong::n is an
   * int), so 2 * n_used < 2^32 */
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "k_env_windows3: the raw-buffer descriptor and its range check are written for gfx950"
#endif
  const __amdgpu_buffer_rsrc_t prs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<int16_t *>(p), 0, (int)(2u * (unsigned)n_used), 0x00020000);
  unsigned voff = 2u * (unsigned)(1024 * r0 + 240 + 16 * ln);  /* first input = first output - 16 */
  unsigned voffh = 2u * (unsigned)(1024 * r0 + 256 * g + l);
  typedef unsigned ev_v4u __attribute__((__vector_size__(16)));
  auto fetch = [&]() {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const ev_v4u v = __builtin_amdgcn_raw_buffer_load_b128(prs, (int)(voff + 16u * u), 0, 0);
      pre[u] = make_uint4(v[0], v[1], v[2], v[3]);
    }
    preh = (short)__builtin_amdgcn_raw_buffer_load_b16(prs, (int)voffh, 0, 0);
    voff += 2048u;
    voffh += 2048u;
  };
  fetch();
  double held[8]; /* terms 130..256 (mir[]) of the previous round */
#pragma unroll
  for (int k0 = 0; k0 < 8; ++k0) held[k0] = 0.0;
  /* a publication that carries nothing but the second halves of the round before */
  auto publish_held_only = [&]() {
    while (__builtin_amdgcn_readfirstlane(flags[8]) < seq - 1) __builtin_amdgcn_s_sleep(1);
    ev_lds_acquire();
    double *tg = terms + (4 * wave + g) * EV_TROW;
#pragma unroll
    for (int k0 = 0; k0 < 8; ++k0)
      if (k0 < 7 || l != 15) tg[256 - l - 16 * k0] = held[k0];
    ev_wave_sync(); /* no wait: see the publication at the end of a round */
  };
  for (int s = 0; s < steps; ++s) {
    ++seq;
    const int rho = r0 + s;
    if (rho >= r1) { /* this wave's run is one round shorter than its neighbours': nothing to hand over */
      publish_held_only();
      if (ln == 0) flags[wave] = seq;
      continue;
    }
    stamp(s, 0);
    phase(0);
    /* 1. normalise (ref :109-114) the 32 samples into registers */
    double yv[16], yh;
    {
      double r[32];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const unsigned w[4] = {pre[u].x, pre[u].y, pre[u].z, pre[u].w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int lo = (int)(short)(w[k] & 0xFFFFu), hi = (int)(short)(w[k] >> 16);
          r[8 * u + 2 * k] = nrm(lo - mean);
          r[8 * u + 2 * k + 1] = nrm(hi - mean);
        }
      }
      const int kh = (int)preh - mean;   /* this round's head sample: fetch() below overwrites preh */
      const double xh = nrm(kh);
      fetch(); /* next round's samples */
      /* 2. FIR (ref :123-138): outputs 16 ln .. 16 ln + 15 of the round's new samples */
#pragma unroll
      for (int i = 0; i < 16; ++i) {
#define XR(m) r[i + 16 - (m)]
        yv[i] = BL_FIR_SEL(FIR_MODE, XR, FC);
#undef XR
      }
      /* zero-state heads of the four windows: the first 16 outputs of a window start from a zeroed delay line
       * (ref :121); lane (g, l) filters sample l of window g with the taps that exist, tap m being the sample of
       * lane l - m of the same 16-lane row, zero when there is none (DPP row_shr:m).  Mode 2 gathers the taps as
       * integers: the pair sums k[l - m] + k[l - 16 + m] are exact either way, a 32-bit DPP move
       * costs half of a 64-bit one and folds into the add, and only the nine sums are converted —
       * 34 instead of 49 instructions, the same bits */
      if (FIR_MODE == 2) {
#define KH(m) __builtin_amdgcn_update_dpp(0, kh, 0x110 + (m), 0xF, 0xF, true) /* row_shr:m, 0 when there is no lane */
        const double p0 = (double)kh; /* tap 16 lies before the window: zero */
        const double p1 = (double)(KH(1) + KH(15)), p2 = (double)(KH(2) + KH(14)), p3 = (double)(KH(3) + KH(13));
        const double p4 = (double)(KH(4) + KH(12)), p5 = (double)(KH(5) + KH(11)), p6 = (double)(KH(6) + KH(10));
        const double p7 = (double)(KH(7) + KH(9)), p8 = (double)KH(8);
#undef KH
        double y_ = FC(7) * p7;
        y_ = __builtin_fma(FC(6), p6, y_);
        y_ = __builtin_fma(FC(5), p5, y_);
        y_ = __builtin_fma(FC(4), p4, y_);
        y_ = __builtin_fma(FC(3), p3, y_);
        y_ = __builtin_fma(FC(2), p2, y_);
        y_ = __builtin_fma(FC(1), p1, y_);
        y_ = __builtin_fma(p8, FC(8), y_);
        yh = __builtin_fma(FC(0), p0, y_);
      } else {
      double hx[17];
      hx[0] = xh;
      hx[1] = bl_dpp_f64<0x111>(xh);  hx[2] = bl_dpp_f64<0x112>(xh);  hx[3] = bl_dpp_f64<0x113>(xh);
      hx[4] = bl_dpp_f64<0x114>(xh);  hx[5] = bl_dpp_f64<0x115>(xh);  hx[6] = bl_dpp_f64<0x116>(xh);
      hx[7] = bl_dpp_f64<0x117>(xh);  hx[8] = bl_dpp_f64<0x118>(xh);  hx[9] = bl_dpp_f64<0x119>(xh);
      hx[10] = bl_dpp_f64<0x11A>(xh); hx[11] = bl_dpp_f64<0x11B>(xh); hx[12] = bl_dpp_f64<0x11C>(xh);
      hx[13] = bl_dpp_f64<0x11D>(xh); hx[14] = bl_dpp_f64<0x11E>(xh); hx[15] = bl_dpp_f64<0x11F>(xh);
      hx[16] = 0.0;
#define XH(m) hx[m]
      yh = BL_FIR_SEL(FIR_MODE, XH, FC);
#undef XH
      }
    }
    /* ring positions: window g reads block g (first half) and block g + 1 (second half); the
     * lanes of group g have just filtered block g + 1 */
    const int xa = base5 + g, xb = xa + 1;
    const int pa = xa >= 5 ? xa - 5 : xa, pb = xb >= 5 ? xb - 5 : xb;
    double *blk_a = buf + pa * EV3_BLK, *blk_b = buf + pb * EV3_BLK;
    stamp(s, 1);
    phase(1);
    ev_wave_sync(); /* previous round's LDS reads (DFT exchanges) are complete */
#pragma unroll
    for (int i = 0; i < 16; ++i) blk_b[EV3_ROW(l) + EV3_UNIT(i >> 1) + (i & 1)] = yv[i];
    buf[EV3_HEADS + ln] = yh;
    ev_wave_sync();
    /* 3. DFT input of window g: lane l holds y[32*m1 + 2*l], y[32*m1 + 2*l + 1] */
    double re[16], im[16];
    {
      const int off = EV3_ROW(l >> 3) + EV3_UNIT(l & 7); /* unit l & 7 of row 2 m1' or 2 m1' + 1 of the block */
      const double *ia = blk_a + off, *ib = blk_b + off;
      const double *i0 = l < 8 ? buf + EV3_HEADS + 16 * g + 2 * l : ia;
      re[0] = i0[0];
      im[0] = i0[1];
#pragma unroll
      for (int m1 = 1; m1 < 8; ++m1) { re[m1] = ia[EV3_ROW(2 * m1)]; im[m1] = ia[EV3_ROW(2 * m1) + 1]; }
#pragma unroll
      for (int m1 = 8; m1 < 16; ++m1) { re[m1] = ib[EV3_ROW(2 * (m1 - 8))]; im[m1] = ib[EV3_ROW(2 * (m1 - 8)) + 1]; }
    }
    ev_wave_sync(); /* window data is in registers; block g's place becomes exchange space */
    stamp(s, 2);
    phase(2);
    bl_fft16(re, im);
#pragma unroll
    for (int k1 = 1; k1 < 16; ++k1) {
      const c2d w = k1 < EV3_W1_REGS ? w1r[k1] : tw256[k1 * 16 + l];
      bl_cmul(re[bl_pos16(k1)], im[bl_pos16(k1)], w.re, w.im);
    }
    stamp(s, 3);
    phase(3);
    double *xg = blk_a; /* [16][18] doubles, re then im */
    const double2 *xrow = reinterpret_cast<const double2 *>(xg + l * 18);
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) xg[k1 * 18 + l] = re[bl_pos16(k1)];
    ev_wave_sync();
#pragma unroll
    for (int q = 0; q < 8; ++q) { const double2 v = xrow[q]; re[2 * q] = v.x; re[2 * q + 1] = v.y; }
    ev_wave_sync();
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) xg[k1 * 18 + l] = im[bl_pos16(k1)];
    ev_wave_sync();
#pragma unroll
    for (int q = 0; q < 8; ++q) { const double2 v = xrow[q]; im[2 * q] = v.x; im[2 * q + 1] = v.y; }
    ev_wave_sync();
    stamp(s, 4);
    phase(4);
    double *tg = terms + (4 * wave + g) * EV_TROW;
    /* The rows are free once the summing wave has taken tile seq - 1 out of them.  The wait stands here, in front
     * of the second DFT pass and the power terms, not behind them (36.3 ms per 1 024 songs against 37.8 with the pass in
     * front of it and 37.5 with the wait in front of the transposes): the second halves kept from the round before leave their registers first, this round's take
     * their place as they are computed (no copies, 16 registers fewer live), and the stores of the first halves
     * go out between the arithmetic instead of in one burst. */
    stamp(s, 5);
    phase(5);
    /* polled without s_sleep: the LDS round trip paces the loop, and a sleep quantum (64 cycles) behind the summing
     * wave's release is 0.5 % of the kernel (35.8 vs 36.0 ms per 1 024 songs, three rounds; the summing wave's own
     * poll and the other waits keep theirs: without it they measured the same or slower) */
    while (__builtin_amdgcn_readfirstlane(flags[8]) < seq - 1) {}
    ev_lds_acquire();
    stamp(s, 6);
#pragma unroll
    for (int k0 = 0; k0 < 8; ++k0)
      if (k0 < 7 || l != 15) tg[256 - l - 16 * k0] = held[k0]; /* terms 130..256 of the round before */
    bl_fft16(re, im);
    /* the partner of pair k = k1 + 16 k0 is Z[256 - k]: register 15 - k0 of lane (16 - k1) mod 16 —
     * a mirror of the 16-lane row followed by a shift by one, two DPP moves per dword and no LDS
     * round trip; lane 0 is its own partner and takes its register 16 - k0 (k0 = 0: Z[0] itself) */
    double mir7 = 0.0;
#pragma unroll
    for (int k0 = 0; k0 < 8; ++k0) {
      const double sr = re[bl_pos16(15 - k0)], si = im[bl_pos16(15 - k0)];
      const double zr = k0 ? re[bl_pos16(16 - k0)] : re[bl_pos16(0)];
      const double zi = k0 ? im[bl_pos16(16 - k0)] : im[bl_pos16(0)];
      /* row_mirror, then a shift by one inside the row: lane 0 has no source there and keeps
       * `old`, which is what it needs instead — its own register */
      const double pr = bl_dpp_f64_old<0x111>(zr, bl_dpp_f64<0x140>(sr));
      const double pi = bl_dpp_f64_old<0x111>(zi, bl_dpp_f64<0x140>(si));
      double own;
      bl_fft512_power1<double, false>(re[bl_pos16(k0)], im[bl_pos16(k0)], pr, pi, tw512[l + 16 * k0], own, held[k0]);
      tg[l + 16 * k0] = own; /* terms 0..127 of this round */
      if (k0 == 7) mir7 = held[7];
    }
    /* |X_128|^2 = |Z_128|^2 has no 1/4 of its own: give back the one the halved input took */
    const double mr = re[bl_pos16(8)], mi = im[bl_pos16(8)];
    const double mid = 4.0 * __builtin_fma(mr, mr, mi * mi);
    if (l == 0) tg[128] = mid;
    if (l == 15) tg[129] = mir7; /* term 129 belongs to the first half */
    /* The LDS executes one wave's instructions in order: the sequence word below lands after the terms above
     * whether this wave waits for them or not, and nothing in the next round needs them: no wait (~1 k cycles
     * of LDS queue per round with no arithmetic to cover them). */
    ev_wave_sync();
    if (ln == 0) flags[wave] = seq;
    stamp(s, 7);
    base5 = base5 == 0 ? 4 : base5 - 1; /* (4 (rho + 1)) mod 5 */
  }
  ++seq; /* the second halves of the last round: one more publication, nothing else in it */
  publish_held_only();
  if (ln == 0) flags[wave] = seq;
}
#undef FC

/* ------------------------------------------------------------------------- */
/* k_env_tail: one lane per song, three waves per 64 songs                    */
/*
 * Parts 2-3 of bl_envelope_sort are serial per song.  The 6th-order recurrence is a chain
 * of 8 dependent f64 operations per step, everything after y_j (onset difference, weighted
 * average, two box filters, peak test) another ~30; one wave issuing all of it in order needs
 * ~340 cycles per step.  Three waves of the workgroup share it as a pipeline over 38-step blocks
 * of 64 songs:
 *   wave 0  the recurrence (bl_tail_iir)                        -> y_j   (yblk, double-buffered)
 *   wave 1  onset weighting, atk, first box filter (bl_tail_ab)  -> o1    (oblk + per-lane counts)
 *   wave 2  second box filter, peak test (bl_tail_c)             -> beat
 * Every wave sits alone on a SIMD and is bound by its own dependent chain; a step costs what the
 * slowest stage costs — the recurrence, ~90 cycles.  The o1 stream is not one value per step at
 * the edges of a song (bl_box19): a block carries up to 48 values per lane and a count.
 */
#define BL_TAIL_OMAX 48 /* 38 + the 10 values box 1 flushes when a song ends */

__global__ __launch_bounds__(192) void k_env_tail(const bl_dsong *__restrict__ songs,
                                                  const double *__restrict__ lc, int n_songs,
                                                  bl_amd_song_result *res, int what) {
  __shared__ double yblk[2][38 * 64];            /* y_j of one block, [step][song] */
  __shared__ double oblk[2][BL_TAIL_OMAX * 64];  /* box-1 outputs of one block, [slot][song] */
  __shared__ int ocnt[2][64];                    /* how many of them per song */
  __shared__ double rings_ab[29 * 64];           /* wave 1: box-1 ring + its 10 `old` cells */
  __shared__ double rings_c[19 * 64];            /* wave 2: box-2 ring */
  __shared__ int flag_mem[4];
  typedef __attribute__((address_space(3))) volatile int lds_vint;
  /* [0]: y blocks produced, [1]: y blocks consumed, [2]: o1 blocks produced, [3]: o1 blocks consumed */
  lds_vint *flags = (lds_vint *)flag_mem;
  /* a handful of latency-bound waves that run beside the wide kernels: let them issue first */
  __builtin_amdgcn_s_setprio(3);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int song = blockIdx.x * 64 + lane;
  const bool valid = song < n_songs;
  bl_dsong sg;
  if (valid) sg = songs[song];
  else { sg.nb_frames = 0; sg.n_windows = 0; sg.env_off = 0; sg.n = 1; sg.duration = 1; }
  const int N = 2 * sg.nb_frames;
  int maxN = N;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) maxN = max(maxN, __shfl_xor(maxN, off));
  if (threadIdx.x < 4) flags[threadIdx.x] = 0;
  __syncthreads();
  const int n_blocks = (maxN + 37) / 38;

  if (wave == 0) {
    /* ---- the recurrence: input pairs (x_j, 0) -> (y_j, y_j+1) ---- */
    bl_tail_iir a;
    a.init();
    /* Every lane reads its own song's compressed envelope, 19 windows (one block) at a time and
     * two blocks ahead, straight into registers: three register sets rotate through "in use",
     * "arriving" and "being requested".  The loads are unconditional from clamped addresses
     * (under an exec mask hipcc waits vmcnt(0) after every few of them) and a window past the
     * song's end is zeroed where it is used.  Earlier forms: one coalesced load per song and a
     * transposition through LDS, fetched on demand (~18 k cycles of HBM latency in front of every
     * third block, more than the recurrence itself) or a tile ahead (the 64 x 3 v_readlane that
     * fetch song i's geometry still cost 9 k cycles per tile). */
    const double *mylc = lc + sg.env_off;
    const int nw = sg.n_windows;
    auto fetch = [&](int kb_, double (&dst)[19]) {
#pragma unroll
      for (int q = 0; q < 19; ++q) dst[q] = mylc[max(min(19 * kb_ + q, nw - 1), 0)];
    };
    auto block = [&](int kb, double (&cur)[19], double (&fut)[19]) {
      if (kb >= n_blocks) return;
      fetch(kb + 2, fut);
      double *yo = yblk[kb & 1] + lane;
      double ye[38];
#pragma unroll
      for (int q = 0; q < 19; ++q) a.pair(19 * kb + q < nw ? cur[q] : 0.0, ye[2 * q], ye[2 * q + 1]);
      /* the buffer is free once the block before the previous one has been consumed */
      while (__builtin_amdgcn_readfirstlane(flags[1]) < kb - 1) __builtin_amdgcn_s_sleep(1);
      ev_lds_acquire();
#pragma unroll
      for (int q = 0; q < 38; ++q) yo[q * 64] = ye[q];
      ev_lds_release();
      bl_wave_sync();
      if (lane == 0) flags[0] = kb + 1;
    };
    double pa[19], pb[19], pc[19];
    fetch(0, pa);
    fetch(1, pb);
    for (int kb = 0; kb < n_blocks; kb += 3) {
      block(kb, pa, pc);
      block(kb + 1, pb, pa);
      block(kb + 2, pc, pb);
    }
    return;
  }

  if (wave == 1) {
    /* ---- y_j -> weighting, atk, box 1 -> o1 ---- */
    bl_tail_ab t;
    t.init(sg.nb_frames, rings_ab + lane, 64);
    for (int kb = 0; kb < n_blocks; ++kb) {
      while (__builtin_amdgcn_readfirstlane(flags[0]) < kb + 1) __builtin_amdgcn_s_sleep(1);
      while (__builtin_amdgcn_readfirstlane(flags[3]) < kb - 1) __builtin_amdgcn_s_sleep(1);
      ev_lds_acquire();
      const double *yin = yblk[kb & 1] + lane;
      const int j = 38 * kb;
      bl_tail_fifo f;
      f.base = oblk[kb & 1] + lane;
      f.stride = 64;
      f.count = 0;
      /* A song in its steady state for the whole block takes the straight-line path; the others —
       * the first 40 steps (the same blocks for every song) and each song's own last dozen — take
       * the step-by-step one.  With equal lengths the branch is wave-uniform; with mixed lengths
       * both sides run (exec-masked) only for the block or two in which a song of the wave ends. */
      if (bl_tail_ab::chunk_ok(j, N)) {
        t.fast_chunk38(yin, 64, f.base, 64);
        f.count = 38;
      } else if (j < N) {
        for (int q = 0; q < 38; ++q) {
          const int jj = j + q;
          if (jj < N) {
            t.step(jj, yin[q * 64], f);
            if (jj == N - 1) t.finish(f);
          }
        }
      }
      ocnt[kb & 1][lane] = f.count;
      ev_lds_release();
      bl_wave_sync();
      if (lane == 0) { flags[1] = kb + 1; flags[2] = kb + 1; }
    }
    if (valid) {
      bl_amd_song_result *r = res + sg.out_idx;
      r->atk_sum = t.atk;
      r->v.attack = bl_tail_attack(t.atk, sg.n);
    }
    return;
  }

  /* ---- o1 -> box 2 -> peaks ---- */
  bl_tail_c c;
  c.init(sg.nb_frames, rings_c + lane, 64);
  for (int kb = 0; kb < n_blocks; ++kb) {
    while (__builtin_amdgcn_readfirstlane(flags[2]) < kb + 1) __builtin_amdgcn_s_sleep(1);
    ev_lds_acquire();
    const double *oin = oblk[kb & 1] + lane;
    const int cnt = ocnt[kb & 1][lane];
    if (cnt == 38 && c.chunk_ok()) {
      c.fast_chunk38(oin, 64);
    } else {
      for (int q = 0; q < BL_TAIL_OMAX; ++q)
        if (q < cnt) c.push(oin[q * 64]);
    }
    if (valid && c.taken == N && cnt > 0) c.finish(); /* the block that delivered the song's last output */
    ev_lds_release();
    bl_wave_sync();
    if (lane == 0) flags[3] = kb + 1;
  }
  if (!valid) return;
  bl_amd_song_result *r = res + sg.out_idx;
  r->beat = c.beat();
  r->v.tempo = bl_tail_tempo(c.beat(), sg.duration);
  (void)what;
}

/* ref analyze.c:63-80: force = fmax(tempo,0) + amplitude + frequency + fmax(attack,0)
 * (double sum, stored as float), then LOUD / CALM / UNKNOWN by its sign */
__global__ void k_force(bl_amd_song_result *res, int n_songs) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_songs) return;
  bl_amd_song_result *r = res + i;
  const float rating = (float)(fmax((double)r->v.tempo, 0.0) + (double)r->v.amplitude +
                               (double)r->v.frequency + fmax((double)r->v.attack, 0.0));
  r->force = rating;
  r->calm_or_loud = rating > 0 ? BL_LOUD : (rating < 0 ? BL_CALM : BL_UNKNOWN);
}

/* ------------------------------------------------------------------------- */
/* k_pairwise                                                                 */

/* ref analyze.c:96-100: f32 throughout, left-to-right; the sum whose root bl_distance returns */
__device__ __forceinline__ float bl_dist_sq(const float4 a, const float4 b) {
  const float d0 = a.x - b.x, d1 = a.y - b.y, d2 = a.z - b.z, d3 = a.w - b.w;
  return d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
}

__device__ __forceinline__ float bl_dist(const float4 a, const float4 b) {
  /* sqrt correctly rounded (-fhip-fp32-correctly-rounded-divide-sqrt); __fsqrt_rn is the 1-ulp native op */
  return sqrtf(bl_dist_sq(a, b));
}

/* ref analyze.c:135-140: the f32 dot product, left to right */
__device__ __forceinline__ float bl_dot(const float4 a, const float4 b) {
  return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
}

/* A workgroup owns BL_PW_ROWS rows x 1024 columns: every thread keeps its four column
 * vectors in registers and walks down the rows (the row vector is wave-uniform: scalar
 * loads), so a vector is fetched once per 16 outputs instead of once per output and the
 * index arithmetic is paid once.  Output: 16-byte stores, each row segment contiguous. */
#define BL_PW_ROWS 16
#ifndef BL_SQRT_VARIANT
#define BL_SQRT_VARIANT 1
#endif
/* SQ: 0 = the compiler's correctly rounded sqrtf everywhere, 1 / 2 = bl_sqrt_rn_fast<SQ> in its domain */
template <bool COSINE, int SQ = BL_SQRT_VARIANT>
__global__ __launch_bounds__(256) void k_pairwise(const float4 *__restrict__ vecs, int n,
                                                  int row_begin, int n_rows,
                                                  float *__restrict__ out) {
  const int j0 = (blockIdx.x * 256 + threadIdx.x) * 4;
  const int r0 = blockIdx.y * BL_PW_ROWS;
  const int r1 = min(r0 + BL_PW_ROWS, n_rows);
  /* cosine: what depends on one vector only — squared norm, its double root, the root's reciprocal (bl_cos.h) —
   * once per row of the workgroup (LDS) and once per column of the thread, not once per output */
  __shared__ double row_s[COSINE ? BL_PW_ROWS : 1], row_r[COSINE ? BL_PW_ROWS : 1];
  if (COSINE) {
    if ((int)threadIdx.x < r1 - r0) {
      const bl_cos_vec p = bl_cos_prep(vecs[row_begin + r0 + threadIdx.x]);
      row_s[threadIdx.x] = p.s;
      row_r[threadIdx.x] = p.r;
    }
    __syncthreads();
  }
  if (j0 >= n) return;
  float4 b[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) b[k] = vecs[min(j0 + k, n - 1)];
  bl_cos_vec cb[COSINE ? 4 : 1];
  if (COSINE) {
#pragma unroll
    for (int k = 0; k < 4; ++k) cb[k] = bl_cos_prep(b[k]);
  }
  const bool vec_ok = j0 + 4 <= n && (n & 3) == 0 && ((reinterpret_cast<size_t>(out) & 15) == 0);
  for (int row = r0; row < r1; ++row) {
    const float4 a = vecs[row_begin + row];
    float *orow = out + (size_t)row * n;
    float r[4];
    if (SQ == 3) { /* measurement builds only (BL_AMD_MEASURE): the store stream alone, no arithmetic */
#pragma unroll
      for (int k = 0; k < 4; ++k) r[k] = a.x;
    } else if (COSINE) {
      /* q' = dot * (ra * rb) where its float is provably the reference's (bl_cos.h); a wave with any output
       * near a float rounding boundary, a zero dot product or a degenerate norm takes the plain expression */
      const double ra = row_r[row - r0];
      float dot[4];
      bool fast = true;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        dot[k] = bl_dot(a, b[k]);
        fast = bl_cos_fast(dot[k], ra * cb[k].r, r[k]) && fast;
      }
      if (!__all(fast)) {
        bl_cos_vec ca;
        ca.s = row_s[row - r0];
#pragma unroll
        for (int k = 0; k < 4; ++k) r[k] = bl_cos_plain(dot[k], ca, cb[k]);
      }
    } else {
      /* the five-instruction root where every sum of the wave is in its domain (bl_sqrt.h), the
       * compiler's sqrtf otherwise: a zero (the diagonal, duplicate songs), a tiny or a non-finite
       * sum — about one wave-row in forty at N = 10 000.  Both are the correctly rounded root. */
      float q[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) q[k] = bl_dist_sq(a, b[k]);
      const unsigned worst = max(max(bl_sqrt_fast_key(q[0]), bl_sqrt_fast_key(q[1])),
                                 max(bl_sqrt_fast_key(q[2]), bl_sqrt_fast_key(q[3])));
      if (SQ != 0 && __all(worst <= BL_SQRT_FAST_SPAN)) {
#pragma unroll
        for (int k = 0; k < 4; ++k) r[k] = bl_sqrt_rn_fast<SQ == 2 ? 2 : 1>(q[k]);
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) r[k] = sqrtf(q[k]);
      }
    }
    if (vec_ok) { /* plain stores: with the non-temporal hint the same stream is 6 % slower (70.8 vs 66.6 us) */
      *reinterpret_cast<float4 *>(orow + j0) = make_float4(r[0], r[1], r[2], r[3]);
    } else {
      for (int k = 0; k < 4 && j0 + k < n; ++k) orow[j0 + k] = r[k];
    }
  }
}

/* Exhaustive check of bl_sqrt_rn_fast: every f32 bit pattern in [first, first + count) that lies
 * in the fast domain against (float)sqrt((double)s); counts[0] += values checked, counts[1] +=
 * mismatches, counts[2] += mismatches of the compiler's sqrtf over ALL patterns of the range
 * (zero, denormals, infinities included; NaN results compare equal to NaN). */
template <int V>
__global__ __launch_bounds__(256) void k_sqrt_sweep(unsigned long long first, unsigned long long count,
                                                    unsigned long long *counts) {
  unsigned long long checked = 0, bad_fast = 0, bad_slow = 0;
  for (unsigned long long i = blockIdx.x * 256ull + threadIdx.x; i < count; i += gridDim.x * 256ull) {
    const float s = __uint_as_float((unsigned)(first + i));
    const float want = (float)sqrt((double)s);
    const float slow = sqrtf(s);
    if (!(slow == want || (slow != slow && want != want))) ++bad_slow;
    if (bl_sqrt_fast_ok(s)) {
      ++checked;
      if (__float_as_uint(bl_sqrt_rn_fast<V>(s)) != __float_as_uint(want)) ++bad_fast;
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    checked += __shfl_down(checked, off);
    bad_fast += __shfl_down(bad_fast, off);
    bad_slow += __shfl_down(bad_slow, off);
  }
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(&counts[0], checked);
    atomicAdd(&counts[1], bad_fast);
    atomicAdd(&counts[2], bad_slow);
  }
}

/* ------------------------------------------------------------------------- */
/* seeded playlist: ref python/examples/make_m3u_playlist.py:62-72                */
/* distances from one seed vector to every song (bl_distance arithmetic), then the songs
 * in order of increasing distance.  The order is the stable argsort: rank(i) = number of
 * songs that are closer, or equally close with a smaller index — an exact, deterministic
 * O(n^2) count (4.3e9 comparisons at n = 65 536, a few ms) instead of a comparison sort. */
__global__ __launch_bounds__(256) void k_seed_dist(const float4 *__restrict__ vecs, int n, int seed,
                                                   float *__restrict__ dist) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j < n) dist[j] = bl_dist(vecs[seed], vecs[j]);
}

__global__ __launch_bounds__(256) void k_rank_order(const float *__restrict__ dist, int n,
                                                    int *__restrict__ order) {
  __shared__ float tile[256];
  const int i = blockIdx.x * 256 + threadIdx.x;
  const float di = i < n ? dist[i] : 0.f;
  int rank = 0;
  for (int j0 = 0; j0 < n; j0 += 256) {
    const int j = j0 + threadIdx.x;
    tile[threadIdx.x] = j < n ? dist[j] : 0.f;
    __syncthreads();
    const int lim = min(256, n - j0);
    for (int k = 0; k < lim; ++k) {
      const float dj = tile[k];
      rank += (dj < di || (dj == di && j0 + k < i)) ? 1 : 0;
    }
    __syncthreads();
  }
  if (i < n) order[rank] = i;
}

/* ------------------------------------------------------------------------- */
/* k_synth: integer-only synthetic PCM, same bytes as oracle/orc_synth.c       */

__device__ __forceinline__ unsigned syn_mix32(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352dU;
  x ^= x >> 15; x *= 0x846ca68bU;
  x ^= x >> 16;
  return x;
}
__device__ __forceinline__ int syn_psin(unsigned ph) {
  const int x = (int)(ph & 65535u) - 32768;
  const int ax = x < 0 ? -x : x;
  return -((x * (32768 - ax)) / 8192);
}
__device__ __forceinline__ short syn_sample(unsigned seed, unsigned rate, unsigned channels,
                                            unsigned i) {
  const unsigned f = i / channels, c = i - f * channels;
  const unsigned h = syn_mix32(seed * 0x9E3779B9u + 1u);
  const unsigned f1 = 110u + (h & 255u);
  const unsigned f2 = 2000u + ((h >> 8) & 2047u);
  const unsigned bpm = 90u + ((h >> 20) & 63u);
  const unsigned a1 = 3000u + ((h >> 26) & 31u) * 100u;
  const unsigned period = rate * 60u / bpm;
  const unsigned pos = f % period;
  const int env = 32768 - (int)(((unsigned long long)pos * 29491u) / period);
  const unsigned ph1 = (unsigned)((((unsigned long long)f * f1) << 16) / rate);
  const unsigned ph2 = (unsigned)((((unsigned long long)f * f2) << 16) / rate) + c * 16384u;
  const int tone = (syn_psin(ph1) * (int)a1 + syn_psin(ph2) * 2500) / 32768;
  const int sig = (tone * env) / 32768;
  const int noise = (int)(syn_mix32(seed ^ syn_mix32(i + 0x1234567u)) % 1601u) - 800;
  return (short)(sig + noise);
}

__global__ __launch_bounds__(256) void k_synth(int16_t *pcm, const bl_dsong *__restrict__ songs,
                                               unsigned seed_base, unsigned rate) {
  const bl_dsong sg = songs[blockIdx.y];
  int16_t *p = pcm + sg.pcm_off;
  const unsigned seed = seed_base + (unsigned)sg.out_idx; /* records may be length-sorted */
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < (unsigned)sg.n; i += gridDim.x * 256u)
    p[i] = syn_sample(seed, rate, (unsigned)sg.channels, i);
}


/* ------------------------------------------------------------------------- */
/* small data-movement kernels of the batch / multi-device paths                */

/* same-rate S32 -> S16 narrowing of a 32-bit source (what the reference's resampler does
 * for an S32 input at the target rate, ref src/decode.c:388-392 -> swr_convert): arithmetic
 * shift by 16.  Parity unpinned (libswresample is absent). */
__global__ __launch_bounds__(256) void k_narrow_s32(const int4 *__restrict__ in, uint2 *__restrict__ out,
                                                    size_t nvec, const int32_t *__restrict__ in_s,
                                                    int16_t *__restrict__ out_s, size_t n) {
  const size_t stride = (size_t)gridDim.x * 256u;
  const size_t t = (size_t)blockIdx.x * 256u + threadIdx.x;
  for (size_t v = t; v < nvec; v += stride) { /* 16 bytes in, 8 bytes out per lane */
    const int4 q = in[v];
    uint2 o;
    o.x = ((unsigned)q.x >> 16) | ((unsigned)q.y & 0xFFFF0000u);
    o.y = ((unsigned)q.z >> 16) | ((unsigned)q.w & 0xFFFF0000u);
    out[v] = o;
  }
  for (size_t i = 4 * nvec + t; i < n; i += stride) out_s[i] = (int16_t)(in_s[i] >> 16);
}

__global__ __launch_bounds__(256) void k_scatter_vecs(const float4 *__restrict__ in,
                                                      const int32_t *__restrict__ order,
                                                      float4 *__restrict__ out, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n && order[i] >= 0) out[order[i]] = in[i]; /* -1: padding slot of a short shard */
}

__global__ __launch_bounds__(256) void k_extract_vecs(const bl_amd_song_result *__restrict__ res,
                                                      float4 *__restrict__ out, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) {
    const struct force_vector_s v = res[i].v;
    out[i] = make_float4(v.tempo, v.amplitude, v.frequency, v.attack);
  }
}

/* ========================================================================= */
/* launchers (declared in bl_launch.h)                                        */

size_t blk_tables_bytes(void) { return 256 * 16 * 2 + 512 * 4 + LV_TW_SLOTS * 16 * 8; }

void blk_tables_fill_host(unsigned char *h) {
  /* twiddle / window tables, computed in double on the host */
  const double pi = 3.14159265358979323846;
  c2d *t256 = reinterpret_cast<c2d *>(h);
  c2d *t512 = t256 + 256;
  float *hann = reinterpret_cast<float *>(t512 + 256);
  for (int k = 0; k < 256; ++k) {
    t256[k].re = cos(2 * pi * k / 256); t256[k].im = -sin(2 * pi * k / 256);
    t512[k].re = cos(2 * pi * k / 512); t512[k].im = -sin(2 * pi * k / 512);
  }
  /* ref frequency_sort.c:40-42 */
  for (int i = 0; i < 512; ++i) hann[i] = (float)(.5f * (1.0f - cos(2 * M_PI * i / (512 - 1))));
  /* libavcodec's cosine tables, per lane and pass (bl_fft_lavc.h) */
  float leafc[4];
  lv_fill_tables(reinterpret_cast<float(*)[2]>(hann + 512), leafc);
}

bl_tables blk_tables_bind(const void *d_mem) {
  bl_tables tb;
  const unsigned char *d = static_cast<const unsigned char *>(d_mem);
  tb.tw256_d = reinterpret_cast<const c2d *>(d);
  tb.tw512_d = tb.tw256_d + 256;
  tb.hann = reinterpret_cast<const float *>(tb.tw512_d + 256);
  tb.lv_tw = reinterpret_cast<const c2f *>(tb.hann + 512);
  {
    float tw[LV_TW_SLOTS * 16][2];
    lv_fill_tables(tw, tb.lv_leafc);
  }
  tb.log101 = log((double)(1 + 100.0f)); /* ref tempo_atk_sort.c:188, log(1 + mu) */
  return tb;
}

int blk_configure_device(void) {
  for (const void *fn : {reinterpret_cast<const void *>(k_env_windows3<0, BL_ENV_PRIO, false>),
                         reinterpret_cast<const void *>(k_env_windows3<1, BL_ENV_PRIO, false>),
                         reinterpret_cast<const void *>(k_env_windows3<2, BL_ENV_PRIO, false>)})
    BL_HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, EV3_LDS_BYTES));
#ifdef BL_AMD_MEASURE
  BL_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_env_windows3<2, BL_ENV_PRIO, true>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, EV3_LDS_BYTES));
#define X(T)                                                                                              \
  BL_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_env_windows3<2, T, false>),           \
                                   hipFuncAttributeMaxDynamicSharedMemorySize, EV3_LDS_BYTES));           \
  BL_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_env_windows3<2, T, true>),            \
                                   hipFuncAttributeMaxDynamicSharedMemorySize, EV3_LDS_BYTES));
  EV_PRIO_TABS(X)
#undef X
#endif
  BL_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_freq_frames),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, BL_FREQ_LDS_BYTES));
  BL_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_freq_scan),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, BL_FREQ_SCAN_LDS_BYTES));
  return BL_OK;
}

namespace {

struct Mark {
  blk_mark_fn fn;
  void *user;
  int k;
  hipStream_t s;
  Mark(blk_mark_fn f, void *u, int kk, hipStream_t ss) : fn(f), user(u), k(kk), s(ss) {
    if (fn) fn(user, k, s, 1);
  }
  ~Mark() {
    if (fn) fn(user, k, s, 0);
  }
};

} // namespace

static long long *g_env_probe = nullptr;
#ifdef BL_AMD_MEASURE

/* measurement builds only: pick a priority table of EV_PRIO_TABS at run time (-1: the compiled default; bits 24..:
 * the PROBE instantiation) and give the stamps a device buffer of 8 x EV_PROBE_ROUNDS x EV_PROBE_SLOTS int64 */
static int g_env_variant = -1;
extern "C" __attribute__((visibility("default"))) int bl_amd_measure_env(int variant, void *d_probe) {
  g_env_variant = variant;
  g_env_probe = static_cast<long long *>(d_probe);
  return BL_OK;
}
#endif

/* Which form of the 17-tap FIR k_env_windows3 runs:
 *   0  the reference's unfused order (BL_FIR) — bit-identical window energies;
 *   1  each product folded into the sum by an fma (BL_FIR_FUSED);
 *   2  as 1, with the normalisation folded into the taps (BL_FIR_FOLD) — the default.
 * bl_amd_set_fir_mode() wins over the environment variable BL_AMD_FIR_FUSED, which wins over the
 * compiled default.  Read on every launch, so one process can run all of them (the A/B tools do). */
static std::atomic<int> g_fir_mode{-1};
int blk_fir_mode() {
  int m = g_fir_mode.load(std::memory_order_relaxed);
  if (m < 0) {
    const char *e = getenv("BL_AMD_FIR_FUSED");
    m = e && *e ? atoi(e) : BL_FIR_FUSED_DEFAULT;
  }
  return m < 0 || m > 2 ? BL_FIR_FUSED_DEFAULT : m;
}
extern "C" int bl_amd_fir_mode(void) { return blk_fir_mode(); }
extern "C" int bl_amd_set_fir_mode(int mode) {
  if (mode < -1 || mode > 2) return BL_UNEXPECTED;
  g_fir_mode.store(mode, std::memory_order_relaxed);
  return BL_OK;
}

namespace {

/* Which root the distance kernel uses: the compiled default BL_SQRT_VARIANT (bl_sqrt.h).  Only a
 * measurement build (make measure: -DBL_AMD_MEASURE, tools/dist_bench.py) also reads
 * BL_AMD_SQRT_VARIANT=0|1|2|3 at run time — 0 = the compiler's sqrtf only, 3 = no arithmetic at
 * all, the store stream alone (results invalid).  The product build ignores the variable.
 * rocprofv3 at N = 10 000, us per launch: 78.9 / 71.2 / 70.7 / 66.6 (profiles/r03_distance.json). */
int blk_sqrt_variant() {
#ifdef BL_AMD_MEASURE
  const char *e = getenv("BL_AMD_SQRT_VARIANT");
  const int v = e && *e ? atoi(e) : BL_SQRT_VARIANT;
  return v < 0 || v > 3 ? BL_SQRT_VARIANT : v;
#else
  return BL_SQRT_VARIANT;
#endif
}

int grid_x_for(long long units_max, int n_songs, int blocks_per_cu, int n_cu) {
  /* enough blocks to fill the chip several times over, never more than the
   * longest song has work for */
  long long want = ((long long)n_cu * blocks_per_cu + n_songs - 1) / n_songs;
  if (want < 1) want = 1;
  if (want > units_max) want = units_max;
  if (want < 1) want = 1;
  if (want > 65535) want = 65535;
  return (int)want;
}

} // namespace

/* analysis of one launch group (n_songs <= 32768: gridDim.y) */
int blk_analyze(const blk_analyze_args &a) {
  const int n_songs = a.n_songs, what = a.what;
  hipStream_t stream = a.stream;
  const int gx_scan = grid_x_for(((long long)a.max_n / 8 + 255) / 256, n_songs, 8, a.n_cu);
  const int tb64 = (n_songs + 63) / 64;
  hipLaunchKernelGGL(k_stats_init, dim3(tb64), dim3(64), 0, stream, a.stats, n_songs);
  /* With all three analyzers asked for, the statistics ride along with the frequency pass (k_freq_scan): two
   * passes over the PCM instead of three.  A measurement build can take them apart again (BL_AMD_FUSED_SCAN=0). */
  bool fused = what == 7;
#ifdef BL_AMD_MEASURE
  if (const char *e = getenv("BL_AMD_FUSED_SCAN")) fused = fused && atoi(e) != 0;
#endif
  if (fused) {
    Mark m(a.mark, a.mark_user, PK_FREQ_SCAN, stream);
    hipLaunchKernelGGL(k_freq_scan, dim3(n_songs), dim3(64 * BL_FREQ_SCAN_WAVES), BL_FREQ_SCAN_LDS_BYTES, stream,
                       a.pcm, a.songs, a.tb, a.spectrum, a.stats, a.hist);
  } else {
    BL_HIP_CHECK(hipMemsetAsync(a.hist, 0, sizeof(unsigned) * BL_HIST_BINS * (size_t)n_songs, stream));
    Mark m(a.mark, a.mark_user, PK_SCAN, stream);
    hipLaunchKernelGGL(k_pcm_scan<true>, dim3(gx_scan, n_songs), dim3(256), 0, stream, a.pcm, a.songs,
                       a.stats, a.hist);
  }
  hipLaunchKernelGGL(k_trim, dim3(n_songs), dim3(128), 0, stream, a.pcm, a.songs, a.stats);
  hipLaunchKernelGGL(k_song_prep, dim3(tb64), dim3(64), 0, stream, a.songs, a.stats, n_songs,
                     a.results);
  hipLaunchKernelGGL(k_variance_wrap, dim3(gx_scan, n_songs), dim3(256), 0, stream, a.pcm, a.songs,
                     a.stats);
  hipLaunchKernelGGL(k_variance_wrap_finish, dim3(tb64), dim3(64), 0, stream, a.songs, a.stats,
                     n_songs, a.results);
  /* Order: the envelope windows first, then the serial envelope tail (three latency-bound waves per 64 songs: it
   * leaves the chip free) beside what is left — the amplitude kernel and, when the statistics were not fused into
   * it, the frequency pass; k_force joins the two.  The tail's 143 KB workgroups only reach a CU when the dispatcher
   * has nothing else pending for it, so they have to be resident BEFORE the kernels they run beside begin:
   *   fused    the tail follows the window kernel on the main stream; amplitude and frequency finish go to the side
   *            stream behind an event (353.9 -> 349.9 ms per 8 192 songs: launched the other way round the
   *            amplitude kernel took the CUs first and the tail ran after it, not beside it);
   *   separate the tail goes to the side stream, the main stream runs the short amplitude kernel and then the wide
   *            frequency pass (launched after that pass, the tail started ~60 ms late). */
  bool tail_async = false, head_async = false;
  hipStream_t rest_stream = stream; /* where the amplitude kernel and the frequency finish go */
  if (what & 4) {
    const int fir_mode = blk_fir_mode();
    /* one 512-thread workgroup per CU; the blocks of a song split its rounds of four windows
     * into contiguous runs, one per compute wave: at least four rounds per run, so that the
     * block a run filters before its first round stays a small part of it */
    auto launch_env = [&](int first, int count, int maxn) -> int {
      Mark m(a.mark, a.mark_user, PK_ENV, stream);
      const int gx2 = grid_x_for(std::max(1, (2 * (maxn / 512)) / (4 * 4 * EV_CWAVES)), count, 2, a.n_cu);
      const dim3 grid(gx2, count), block(64 * (EV_CWAVES + 1));
#define EV_LAUNCH(M, T, P)                                                                           \
  hipLaunchKernelGGL((k_env_windows3<M, T, P>), grid, block, EV3_LDS_BYTES, stream, a.pcm, a.songs + first, \
                     a.stats + first, a.tb, a.energies, a.lc, g_env_probe)
#ifdef BL_AMD_MEASURE
      /* bl_amd_measure_env(): A/B of the priority tables (FIR mode 2 only), with or without the phase stamps */
      const int tab = g_env_variant & 0xFFFFFF;
      const bool stamps = g_env_variant >= 0 && (g_env_variant >> 24) != 0;
      if (fir_mode == 2 && g_env_variant >= 0 && (tab != BL_ENV_PRIO || stamps)) {
        if (tab == BL_ENV_PRIO) EV_LAUNCH(2, BL_ENV_PRIO, true);
#define X(T) else if (tab == (T)) { if (stamps) EV_LAUNCH(2, T, true); else EV_LAUNCH(2, T, false); }
        EV_PRIO_TABS(X)
#undef X
        else return BL_UNEXPECTED;
      } else
#endif
      if (fir_mode == 2) EV_LAUNCH(2, BL_ENV_PRIO, false);
      else if (fir_mode == 1) EV_LAUNCH(1, BL_ENV_PRIO, false);
      else EV_LAUNCH(0, BL_ENV_PRIO, false);
      return BL_OK;
#undef EV_LAUNCH
    };
    /* the serial tail of the songs [first, first + count), on the side stream when there is something to
     * overlap it with */
    const bool side = (what & 3) && a.side;
    auto launch_tail = [&](int first, int count, bool last) -> int {
      hipStream_t ts = stream;
      if (side && !last) {
        /* the long songs of a mixed batch: their tail goes to the second side stream, behind its own event, and runs
         * under the window kernel of the rest (on the first side stream it would stand in front of the amplitude
         * kernel and the frequency finish, which only wait for the window kernel behind this one) */
        BL_HIP_CHECK(hipEventRecord(a.ev_head, stream));
        BL_HIP_CHECK(hipStreamWaitEvent(a.side2, a.ev_head, 0));
        ts = a.side2;
        head_async = true;
      } else if (side && fused) {
        /* the tail stays here; the rest waits on the side stream for the window kernel in front of it */
        BL_HIP_CHECK(hipEventRecord(a.ev_env, stream));
        BL_HIP_CHECK(hipStreamWaitEvent(a.side, a.ev_env, 0));
        rest_stream = a.side;
        tail_async = true;
      } else if (side) {
        BL_HIP_CHECK(hipEventRecord(a.ev_env, stream));
        BL_HIP_CHECK(hipStreamWaitEvent(a.side, a.ev_env, 0));
        ts = a.side;
        tail_async = true;
      }
      Mark m(a.mark, a.mark_user, PK_TAIL, ts);
      hipLaunchKernelGGL(k_env_tail, dim3((count + 63) / 64), dim3(192), 0, ts, a.songs + first, a.lc, count,
                         a.results, what);
      return BL_OK;
    };
    /* Mixed lengths (records sorted longest first): the tail of a ten-minute song is a ~15 ms serial chain, and
     * launched behind the window kernel of the whole batch it outlasts the frequency pass it is meant to hide
     * behind.  The long songs [0, n_head) get their own window launch, and their tail runs on the side stream
     * under the window kernel of the rest. */
    const int n_head = (a.n_head > 0 && a.n_head < n_songs && side && a.side2) ? a.n_head : 0;
    if (n_head) {
      if (launch_env(0, n_head, a.max_n) != BL_OK || launch_tail(0, n_head, false) != BL_OK) return BL_UNEXPECTED;
      if (launch_env(n_head, n_songs - n_head, a.max_n_rest) != BL_OK ||
          launch_tail(n_head, n_songs - n_head, true) != BL_OK)
        return BL_UNEXPECTED;
    } else {
      if (launch_env(0, n_songs, a.max_n) != BL_OK || launch_tail(0, n_songs, true) != BL_OK) return BL_UNEXPECTED;
    }
  }
  if (what & 1) {
    Mark m(a.mark, a.mark_user, PK_AMP, rest_stream);
    hipLaunchKernelGGL(k_amp_finish, dim3(n_songs), dim3(256), 0, rest_stream, a.songs, a.stats, a.hist,
                       a.results);
  }
  if (what & 2) {
    if (!fused) {
      Mark m(a.mark, a.mark_user, PK_FREQ, rest_stream);
      hipLaunchKernelGGL(k_freq_frames, dim3(n_songs), dim3(256), BL_FREQ_LDS_BYTES, rest_stream, a.pcm,
                         a.songs, a.tb, a.spectrum);
    }
    Mark m(a.mark, a.mark_user, PK_FREQ_FIN, rest_stream);
    hipLaunchKernelGGL(k_freq_finish, dim3(n_songs), dim3(256), 0, rest_stream, a.spectrum, a.songs,
                       a.results);
  }
  /* whatever ran on the side stream (tails, or the amplitude / frequency finish) joins the main stream here */
  if (tail_async) {
    BL_HIP_CHECK(hipEventRecord(a.ev_tail, a.side));
    BL_HIP_CHECK(hipStreamWaitEvent(stream, a.ev_tail, 0));
  }
  if (head_async) {
    BL_HIP_CHECK(hipEventRecord(a.ev_tail2, a.side2));
    BL_HIP_CHECK(hipStreamWaitEvent(stream, a.ev_tail2, 0));
  }
  if (what == 7) hipLaunchKernelGGL(k_force, dim3(tb64), dim3(64), 0, stream, a.results, n_songs);
  BL_HIP_CHECK(hipGetLastError());
  return BL_OK;
}

int blk_synth(hipStream_t s, int16_t *pcm, const bl_dsong *d_songs, int n_songs, int max_n,
              int n_cu, unsigned seed_base, unsigned rate) {
  const int gx = grid_x_for(((long long)max_n + 255) / 256, n_songs, 8, n_cu);
  hipLaunchKernelGGL(k_synth, dim3(gx, n_songs), dim3(256), 0, s, pcm, d_songs, seed_base, rate);
  BL_HIP_CHECK(hipGetLastError());
  return BL_OK;
}

int blk_pairwise(hipStream_t s, const struct force_vector_s *d_vecs, int n, int row_begin,
                 int n_rows, float *d_out, bool cosine, blk_mark_fn mark, void *mark_user) {
  const float4 *v = reinterpret_cast<const float4 *>(d_vecs);
  const int gx = (n + 1023) / 1024;
  const int chunk = 65535 * BL_PW_ROWS; /* rows per launch (gridDim.y limit) */
  for (int r0 = 0; r0 < n_rows; r0 += chunk) {
    const int cnt = n_rows - r0 < chunk ? n_rows - r0 : chunk;
    const int gy = (cnt + BL_PW_ROWS - 1) / BL_PW_ROWS;
    Mark m(mark, mark_user, PK_DIST, s);
    if (cosine)
      hipLaunchKernelGGL((k_pairwise<true, 0>), dim3(gx, gy), dim3(256), 0, s, v, n, row_begin + r0, cnt,
                         d_out + (size_t)r0 * n);
#ifdef BL_AMD_MEASURE
    else if (blk_sqrt_variant() == 0)
      hipLaunchKernelGGL((k_pairwise<false, 0>), dim3(gx, gy), dim3(256), 0, s, v, n, row_begin + r0, cnt,
                         d_out + (size_t)r0 * n);
    else if (blk_sqrt_variant() == 3)
      hipLaunchKernelGGL((k_pairwise<false, 3>), dim3(gx, gy), dim3(256), 0, s, v, n, row_begin + r0, cnt,
                         d_out + (size_t)r0 * n);
    else if (blk_sqrt_variant() == 2)
      hipLaunchKernelGGL((k_pairwise<false, 2>), dim3(gx, gy), dim3(256), 0, s, v, n, row_begin + r0, cnt,
                         d_out + (size_t)r0 * n);
    else
      hipLaunchKernelGGL((k_pairwise<false, 1>), dim3(gx, gy), dim3(256), 0, s, v, n, row_begin + r0, cnt,
                         d_out + (size_t)r0 * n);
#else
    else
      hipLaunchKernelGGL((k_pairwise<false, BL_SQRT_VARIANT>), dim3(gx, gy), dim3(256), 0, s, v, n,
                         row_begin + r0, cnt, d_out + (size_t)r0 * n);
#endif
  }
  BL_HIP_CHECK(hipGetLastError());
  return BL_OK;
}

/* Sweep of bl_cos_fast against the plain expression over pseudo-random (dot, na, nb): norms over 2^-40..2^40
 * (a quarter of them within 2^-4..2^16, where force vectors live), dot = u * sqrt(na nb) with u in [-1, 1], and for
 * each such triple the 8 neighbouring floats of dot.  counts: [0] triples, [1] triples the fast path accepts,
 * [2] accepted triples whose float differs from the plain expression's (must be 0), [3] largest |q' - q| seen,
 * in ulp of the double quotient (provable bound: < 6), [4] triples whose q lies within 64 ulp of a float rounding
 * boundary, [5] of those, how many the unguarded (float)q' would get wrong. */
__global__ __launch_bounds__(256) void k_cos_sweep(unsigned long long seed, int per_thread, unsigned long long *counts) {
  unsigned long long st = seed + 0x9E3779B97F4A7C15ull * (blockIdx.x * 256ull + threadIdx.x + 1);
  auto next = [&]() -> unsigned {
    st ^= st << 13; st ^= st >> 7; st ^= st << 17;
    return (unsigned)(st >> 32) ^ (unsigned)st;
  };
  auto rnd_norm = [&]() -> float {
    const unsigned r = next();
    const int span = (r & 3u) ? 80 : 20, base = (r & 3u) ? -40 : -4;
    const int e = base + (int)((r >> 2) % (unsigned)span);
    return ldexpf(1.0f + (float)(next() >> 9) * (1.0f / 8388608.0f), e);
  };
  unsigned long long n = 0, n_fast = 0, bad = 0, max_ulp = 0, near = 0, near_bad = 0;
  for (int it = 0; it < per_thread; ++it) {
    bl_cos_vec a, b;
    a.n = rnd_norm(); b.n = rnd_norm();
    a.s = sqrt((double)a.n); a.r = 1.0 / a.s;
    b.s = sqrt((double)b.n); b.r = 1.0 / b.s;
    const float u = (float)((int)next()) * (1.0f / 2147483648.0f);
    const float d0 = (float)((double)u * (a.s * b.s));
    for (int j = -4; j < 4; ++j) {
      const float dot = __uint_as_float(__float_as_uint(d0) + (unsigned)j);
      const float want = bl_cos_plain(dot, a, b);
      float got;
      const bool ok = bl_cos_fast(dot, a.r * b.r, got);
      const double q = (double)dot / (a.s * b.s), qf = (double)dot * (a.r * b.r);
      const long long bq = __double_as_longlong(q), bf = __double_as_longlong(qf);
      ++n;
      if (ok) {
        ++n_fast;
        if (__float_as_uint(got) != __float_as_uint(want)) ++bad;
      }
      if (q == q && qf == qf && q != 0.0 && (bq >> 63) == (bf >> 63)) {
        const unsigned long long d = (unsigned long long)(bq > bf ? bq - bf : bf - bq);
        if (d < (1ull << 40)) max_ulp = max(max_ulp, d);
        const unsigned lo = (unsigned)bq & 0x1FFFFFFFu;
        if (lo - (0x10000000u - 64u) <= 128u) {
          ++near;
          if (__float_as_uint(got) != __float_as_uint(want)) ++near_bad;
        }
      }
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    n += __shfl_down(n, off); n_fast += __shfl_down(n_fast, off); bad += __shfl_down(bad, off);
    near += __shfl_down(near, off); near_bad += __shfl_down(near_bad, off);
    max_ulp = max(max_ulp, (unsigned long long)__shfl_down(max_ulp, off));
  }
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(&counts[0], n); atomicAdd(&counts[1], n_fast); atomicAdd(&counts[2], bad);
    atomicMax(&counts[3], max_ulp); atomicAdd(&counts[4], near); atomicAdd(&counts[5], near_bad);
  }
}

int blk_cos_sweep(hipStream_t s, unsigned long long seed, int per_thread, unsigned long long *d_counts, int n_cu) {
  hipLaunchKernelGGL(k_cos_sweep, dim3(n_cu * 8), dim3(256), 0, s, seed, per_thread, d_counts);
  BL_HIP_CHECK(hipGetLastError());
  return BL_OK;
}

int blk_sqrt_sweep(hipStream_t s, unsigned long long first, unsigned long long count,
                   unsigned long long *d_counts, int n_cu) {
  if (blk_sqrt_variant() == 2)
    hipLaunchKernelGGL(k_sqrt_sweep<2>, dim3(n_cu * 8), dim3(256), 0, s, first, count, d_counts);
  else /* variant 0 ships no fast form; the sweep then checks form 1 and the fallback */
    hipLaunchKernelGGL(k_sqrt_sweep<1>, dim3(n_cu * 8), dim3(256), 0, s, first, count, d_counts);
  BL_HIP_CHECK(hipGetLastError());
  return BL_OK;
}

int blk_playlist(hipStream_t s, const struct force_vector_s *d_vecs, int n, int seed_index,
                 int32_t *d_order, float *d_dist) {
  const int gx = (n + 255) / 256;
  hipLaunchKernelGGL(k_seed_dist, dim3(gx), dim3(256), 0, s, reinterpret_cast<const float4 *>(d_vecs),
                     n, seed_index, d_dist);
  hipLaunchKernelGGL(k_rank_order, dim3(gx), dim3(256), 0, s, d_dist, n, d_order);
  BL_HIP_CHECK(hipGetLastError());
  return BL_OK;
}

int blk_narrow_s32(hipStream_t s, const int32_t *d_in, int16_t *d_out, size_t n, int n_cu) {
  /* the vector body needs a 16-byte aligned input and an 8-byte aligned output; anything
   * else (and the last n % 4 elements) goes element by element */
  const bool aligned = ((reinterpret_cast<size_t>(d_in) & 15) == 0) && ((reinterpret_cast<size_t>(d_out) & 7) == 0);
  const size_t nvec = aligned ? n / 4 : 0;
  const size_t work = aligned ? nvec : n;
  size_t blocks = (work + 255) / 256;
  const size_t cap = (size_t)n_cu * 16;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(k_narrow_s32, dim3((unsigned)blocks), dim3(256), 0, s,
                     reinterpret_cast<const int4 *>(d_in), reinterpret_cast<uint2 *>(d_out), nvec, d_in,
                     d_out, n);
  BL_HIP_CHECK(hipGetLastError());
  return BL_OK;
}

int blk_scatter_vecs(hipStream_t s, const struct force_vector_s *d_in, const int32_t *d_order,
                     struct force_vector_s *d_out, int n) {
  hipLaunchKernelGGL(k_scatter_vecs, dim3((n + 255) / 256), dim3(256), 0, s,
                     reinterpret_cast<const float4 *>(d_in), d_order, reinterpret_cast<float4 *>(d_out), n);
  BL_HIP_CHECK(hipGetLastError());
  return BL_OK;
}

int blk_extract_vecs(hipStream_t s, const bl_amd_song_result *d_res, struct force_vector_s *d_out,
                     int n) {
  hipLaunchKernelGGL(k_extract_vecs, dim3((n + 255) / 256), dim3(256), 0, s, d_res,
                     reinterpret_cast<float4 *>(d_out), n);
  BL_HIP_CHECK(hipGetLastError());
  return BL_OK;
}

int blk_scan_one(hipStream_t s, const int16_t *pcm, const bl_dsong *d_songs, bl_dstats *d_stats,
                 unsigned *d_hist, int n, int n_cu) {
  const int gx = grid_x_for(((long long)n / 8 + 255) / 256, 1, 8, n_cu);
  BL_HIP_CHECK(hipMemsetAsync(d_hist, 0, sizeof(unsigned) * BL_HIST_BINS, s));
  hipLaunchKernelGGL(k_stats_init, dim3(1), dim3(64), 0, s, d_stats, 1);
  hipLaunchKernelGGL(k_pcm_scan<true>, dim3(gx, 1), dim3(256), 0, s, pcm, d_songs, d_stats, d_hist);
  hipLaunchKernelGGL(k_trim, dim3(1), dim3(128), 0, s, pcm, d_songs, d_stats);
  BL_HIP_CHECK(hipGetLastError());
  return BL_OK;
}

int blk_variance_wrap_one(hipStream_t s, const int16_t *pcm, const bl_dsong *d_songs,
                          bl_dstats *d_stats, int n, int n_cu) {
  const int gx = grid_x_for(((long long)n / 8 + 255) / 256, 1, 8, n_cu);
  hipLaunchKernelGGL(k_variance_wrap, dim3(gx, 1), dim3(256), 0, s, pcm, d_songs, d_stats);
  BL_HIP_CHECK(hipGetLastError());
  return BL_OK;
}
