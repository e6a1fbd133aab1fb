/*
 * bl_rs_kernels.hip — device form of the rate converter (bl_resample.c), for batches of songs
 * that are resident in HBM at their native rate: in -> 22 050 Hz stereo s16, the format of
 * bl_amd_analyze_batch_device.  Same arithmetic as the host form, sample for sample (the host
 * form is the one pinned on the reference's digests, ref tests/test_decode.c:35-36,55-56; the
 * GPU tests hold this one against it bit for bit): the filter bank is the host's, uploaded;
 * the position of output n is computed directly, w_n = w0 + floor(n * dst_incr / (src_incr *
 * phases)), phase = floor(n * dst_incr / src_incr) mod phases; sources of at most 16 bits use
 * the Q15 bank and a wrapping 32-bit accumulator; wider sources use float with the eight
 * strided partial sums of fused multiply-adds combined pairwise, then rint(v * 32768) clipped.
 *
 * Three kernels, chosen by the plan (blk_resample):
 *   k_resample_1p  one phase and a whole-number step — 44.1 kHz (and 88.2 kHz): the taps sit in
 *                  scalar registers, a lane computes four consecutive outputs from one pass over
 *                  the frames they share;
 *   k_resample_pm  many phases whose cycle advances a whole number of input frames — 48 kHz: a
 *                  wave takes one phase at a time (taps wave-uniform), its lanes outputs one
 *                  cycle apart, the next tile's input is fetched while one is computed;
 *   k_resample     everything else (other rates, up-sampling): one workgroup converts RS_TILE
 *                  consecutive output frames; it stages the input span those outputs read as
 *                  (L, R) pairs and, when it fits, the bank into LDS (rows 16-byte aligned, an
 *                  odd number of 16-byte units apart, so that lanes on different phases spread
 *                  over the banks), one output frame per lane at a time — LDS-read bound: taps *
 *                  12 bytes per output frame.
 * All three reflect the input at the song's edges exactly as the host form does.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>

#include "bliss.h"
#include "bl_runtime.h"

#define RS_TILE 1024
#define RS_THREADS 256
#define RSG_THREADS 1024 /* generic kernel: its LDS footprint allows one workgroup per CU */
#define RS_LDS_LIMIT (160 * 1024)

namespace {

template <bool F32> struct rs_elem { typedef int type; };
template <> struct rs_elem<true> { typedef float type; };

__device__ __forceinline__ int rs_clip16(int v) { return v > 32767 ? 32767 : v < -32768 ? -32768 : v; }

typedef float rs_f2 __attribute__((ext_vector_type(2)));
typedef int rs_i2 __attribute__((ext_vector_type(2)));
template <bool F32> struct rs_pair { typedef rs_i2 type; };
template <> struct rs_pair<true> { typedef rs_f2 type; };

/* one output frame: x = the window's frames as (L, R) pairs, c = its row of taps (16-byte aligned,
 * zero padded to taps8) */
template <bool F32>
__device__ __forceinline__ unsigned rs_output(const typename rs_pair<F32>::type *__restrict__ x,
                                              const typename rs_elem<F32>::type *__restrict__ c, int taps8) {
  if constexpr (F32) {
    rs_f2 a[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) a[q] = rs_f2{0.0f, 0.0f};
    for (int i = 0; i < taps8; i += 8) {
      const float4 c0 = *reinterpret_cast<const float4 *>(c + i), c1 = *reinterpret_cast<const float4 *>(c + i + 4);
      const float cf[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
#pragma unroll
      for (int q = 0; q < 8; ++q) a[q] = __builtin_elementwise_fma(x[i + q], rs_f2{cf[q], cf[q]}, a[q]);
    }
    const rs_f2 v = ((a[0] + a[4]) + (a[2] + a[6])) + ((a[1] + a[5]) + (a[3] + a[7]));
    const float ra = fminf(fmaxf(rintf(v.x * 32768.0f), -32768.0f), 32767.0f);
    const float rb = fminf(fmaxf(rintf(v.y * 32768.0f), -32768.0f), 32767.0f);
    return ((unsigned)(int)ra & 0xFFFFu) | ((unsigned)(int)rb << 16);
  } else {
    int a = 1 << 14, b = 1 << 14; /* wraps like the host's accumulator */
    for (int i = 0; i < taps8; i += 8) {
      const int4 c0 = *reinterpret_cast<const int4 *>(c + i), c1 = *reinterpret_cast<const int4 *>(c + i + 4);
      const int cf[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
#pragma unroll
      for (int q = 0; q < 8; ++q) { /* 16-bit operands: the 24-bit multiplier is exact */
        const rs_i2 v = x[i + q];
        a = (int)((unsigned)a + (unsigned)__mul24(v.x, cf[q]));
        b = (int)((unsigned)b + (unsigned)__mul24(v.y, cf[q]));
      }
    }
    return ((unsigned)rs_clip16(a >> 15) & 0xFFFFu) | ((unsigned)rs_clip16(b >> 15) << 16);
  }
}

template <bool F32, bool BANK_LDS>
__global__ __launch_bounds__(RSG_THREADS) void k_resample(const void *__restrict__ in,
                                                         const bl_rs_dsong *__restrict__ songs,
                                                         const void *__restrict__ bank_g, bl_rs_geom G,
                                                         int16_t *__restrict__ out) {
  typedef typename rs_elem<F32>::type T;
  typedef typename rs_pair<F32>::type T2;
  extern __shared__ __align__(16) unsigned char rs_smem[];
  const bl_rs_dsong sg = songs[blockIdx.y];
  const long long n_begin = (long long)blockIdx.x * G.tiles_per_wg * RS_TILE;
  if (n_begin >= sg.out_frames) return;
  const long long n_end = min((long long)sg.out_frames, n_begin + (long long)G.tiles_per_wg * RS_TILE);
  const int tid = threadIdx.x;
  const unsigned pc = (unsigned)G.phase_count;
  const int L = G.taps, taps8 = G.taps8;
  const bool stereo = sg.channels == 2;

  auto position = [&](long long n, int &index) -> long long {
    const unsigned long long t = (unsigned long long)n * G.dst_incr / G.src_incr;
    index = (int)(t % pc);
    return (long long)G.w0 + (long long)(t / pc);
  };

  T2 *xs = reinterpret_cast<T2 *>(rs_smem);   /* frames as (L, R); a mono source in both */
  T *lb = reinterpret_cast<T *>(xs + G.span); /* G.span is even: 16-byte aligned */
  const int lb_stride = taps8 + 4;            /* rows 16-byte aligned, an odd number of 16-byte units apart */
  if (BANK_LDS) { /* once per workgroup: it walks a run of tiles */
    const T *bg = static_cast<const T *>(bank_g);
    const int total = (int)pc * taps8;
    for (int i = tid; i < total; i += RSG_THREADS) {
      const int r = i / taps8, q = i - r * taps8;
      lb[r * lb_stride + q] = bg[(size_t)r * G.alloc + q];
    }
  }

  for (long long n0 = n_begin; n0 < n_end; n0 += RS_TILE) {
  const int cnt = (int)min((long long)RS_TILE, n_end - n0);
  int idx_unused;
  const long long w_first = position(n0, idx_unused);
  const long long w_last = position(n0 + cnt - 1, idx_unused);
  const int span = (int)(w_last - w_first) + taps8;
  __syncthreads(); /* the previous tile's windows have been read */

  /* input span: ext position e = w_first + k; ext[0..L) mirrors the first samples about
   * sample 0, ext[L + N ...] mirrors the last `refl` about the end, nothing beyond */
  const long long N = sg.frames;
  for (int k0 = tid; k0 < span; k0 += 4 * RSG_THREADS) {
    T2 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { /* four loads in flight; clamped, not masked (a masked load is waited for at once) */
      const long long e = w_first + (k0 + u * RSG_THREADS) - L;
      const long long j = e - N;
      long long xi = e < 0 ? -e : (e < N ? e : N - 1 - j);
      const bool ok = j < sg.refl;
      xi = min(max(xi, 0LL), N - 1);
      if constexpr (F32) {
        const int32_t *p = static_cast<const int32_t *>(in) + sg.in_off;
        if (stereo) {
          const int2 q = reinterpret_cast<const int2 *>(p)[xi];
          v[u].x = (float)q.x * (1.0f / 2147483648.0f);
          v[u].y = (float)q.y * (1.0f / 2147483648.0f);
        } else {
          v[u].x = (float)p[xi] * (1.0f / 2147483648.0f) * (float)0.70710678118654752440;
          v[u].y = v[u].x;
        }
      } else {
        const int16_t *p = static_cast<const int16_t *>(in) + sg.in_off;
        if (stereo) {
          const unsigned q = reinterpret_cast<const unsigned *>(p)[xi];
          v[u].x = (int)(short)(q & 0xFFFFu);
          v[u].y = (int)(short)(q >> 16);
        } else {
          v[u].x = ((int)p[xi] * 23170 + 16384) >> 15; /* Q15 1/sqrt(2) */
          v[u].y = v[u].x;
        }
      }
      if (!ok) v[u] = T2{0, 0};
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (k0 + u * RSG_THREADS < span) xs[k0 + u * RSG_THREADS] = v[u];
  }
  __syncthreads();

  unsigned *o = reinterpret_cast<unsigned *>(out + sg.out_off) + n0;
  for (int j = tid; j < cnt; j += RSG_THREADS) {
    int index;
    const long long w = position(n0 + j, index);
    const int base = (int)(w - w_first);
    const T *c = BANK_LDS ? lb + index * lb_stride : static_cast<const T *>(bank_g) + (size_t)index * G.alloc;
    o[j] = rs_output<F32>(xs + base, c, taps8);
  }
  } /* tiles */
}

/* ---- one phase, whole-number step: 44.1 kHz (step 2, 66 taps) and 88.2 kHz (step 4, 132) ----
 *
 * Every output uses the same coefficient row, so the row lives in scalar registers, and
 * consecutive outputs read overlapping input: a lane computes RS1_OUT consecutive output frames
 * from one pass over the L + (RS1_OUT - 1) * D input frames they cover, everything unrolled.
 * s16: the channels are staged planar, two consecutive frames per dword, and a v_dot2 takes two
 * taps at a time (the step is even, so frame pairs and tap pairs stay aligned for every output).
 * float: frames are staged as (L, R) pairs and one packed fma serves both channels; each
 * output keeps the eight strided partial sums of the host form, and because the frames are
 * walked in order each partial sum sees its taps in the host's order.  A lane's frames are 8 * D
 * / 2 apart from its neighbour's: 16 bytes of padding after every 8 frames keep its ds_read_b128
 * off its neighbours' banks.  A mono source is staged into both channels. */
#define RS1_OUT 4
typedef short rs_s2 __attribute__((ext_vector_type(2)));

template <bool F32, int D, int L> struct rs1_geom {
  static constexpr int F = L + (RS1_OUT - 1) * D;                    /* frames one lane reads */
  static constexpr int SPAN = ((RS_TILE / RS1_OUT - 1) * RS1_OUT * D + F + 7) & ~7;
  static constexpr int W0 = L - (L - 1) / 2;
  static constexpr size_t LDS = F32 ? (size_t)(SPAN + SPAN / 8 * 2) * 8 : (size_t)SPAN * 2 * 2;
  static_assert(F % 8 == 0 && L % 2 == 0 && D % 2 == 0, "pairing");
};

template <bool F32, int D, int L>
__global__ __launch_bounds__(RS_THREADS) void k_resample_1p(const void *__restrict__ in,
                                                            const bl_rs_dsong *__restrict__ songs,
                                                            const void *__restrict__ bank_g,
                                                            int16_t *__restrict__ out) {
  typedef rs1_geom<F32, D, L> GE;
  extern __shared__ __align__(16) unsigned char rs_smem[];
  const bl_rs_dsong sg = songs[blockIdx.y];
  const long long n0 = (long long)blockIdx.x * RS_TILE;
  if (n0 >= sg.out_frames) return;
  const int cnt = (int)min((long long)RS_TILE, (long long)sg.out_frames - n0);
  const int tid = threadIdx.x;
  const bool stereo = sg.channels == 2;
  const long long N = sg.frames;
  const long long x_first = (long long)GE::W0 + n0 * D - L; /* input index of staged frame 0 */

  /* input index of staged frame k, or -1 for "nothing there" (beyond the flush reflection) */
  auto source = [&](int k) -> long long {
    const long long e = x_first + k;
    if (e < 0) return -e;
    if (e < N) return e;
    const long long j = e - N;
    return j < sg.refl ? N - 1 - j : -1;
  };

  /* A tile away from the song's edges (all but the first and the last of a stereo song) is
   * staged with 16-byte loads that are all in flight before the first LDS write. */
  const bool interior = x_first >= 0 && x_first + GE::SPAN <= N;
  if constexpr (F32) {
    rs_f2 *xs = reinterpret_cast<rs_f2 *>(rs_smem);
    const int32_t *p = static_cast<const int32_t *>(in) + sg.in_off;
    if (stereo && interior && ((reinterpret_cast<size_t>(p + 2 * x_first) & 15) == 0)) {
      constexpr int NV = GE::SPAN / 2, PER = (NV + RS_THREADS - 1) / RS_THREADS; /* 2 frames per load */
      const int4 *src = reinterpret_cast<const int4 *>(p + 2 * x_first);
      int4 v[PER];
#pragma unroll
      for (int i = 0; i < PER; ++i) v[i] = src[min(tid + RS_THREADS * i, NV - 1)]; /* clamped, not masked */
#pragma unroll
      for (int i = 0; i < PER; ++i) {
        const int idx = tid + RS_THREADS * i, k = 2 * idx;
        if (idx < NV) {
          float4 w;
          w.x = (float)v[i].x * (1.0f / 2147483648.0f);
          w.y = (float)v[i].y * (1.0f / 2147483648.0f);
          w.z = (float)v[i].z * (1.0f / 2147483648.0f);
          w.w = (float)v[i].w * (1.0f / 2147483648.0f);
          *reinterpret_cast<float4 *>(xs + k + (k >> 3) * 2) = w;
        }
      }
    } else if (stereo && interior) { /* 8-byte aligned only (an odd first frame): one frame per load, all in flight */
      constexpr int PER = (GE::SPAN + RS_THREADS - 1) / RS_THREADS;
      const int2 *src = reinterpret_cast<const int2 *>(p + 2 * x_first);
      int2 v[PER];
#pragma unroll
      for (int i = 0; i < PER; ++i) v[i] = src[min(tid + RS_THREADS * i, GE::SPAN - 1)];
#pragma unroll
      for (int i = 0; i < PER; ++i) {
        const int k = tid + RS_THREADS * i;
        if (k < GE::SPAN)
          xs[k + (k >> 3) * 2] = rs_f2{(float)v[i].x * (1.0f / 2147483648.0f), (float)v[i].y * (1.0f / 2147483648.0f)};
      }
    } else {
      for (int k = tid; k < GE::SPAN; k += RS_THREADS) {
        const long long xi = source(k);
        rs_f2 v = {0.0f, 0.0f};
        if (xi >= 0) {
          if (stereo) {
            const int2 q = reinterpret_cast<const int2 *>(p)[xi];
            v.x = (float)q.x * (1.0f / 2147483648.0f);
            v.y = (float)q.y * (1.0f / 2147483648.0f);
          } else {
            v.x = (float)p[xi] * (1.0f / 2147483648.0f) * (float)0.70710678118654752440;
            v.y = v.x;
          }
        }
        xs[k + (k >> 3) * 2] = v;
      }
    }
  } else {
    unsigned *c0 = reinterpret_cast<unsigned *>(rs_smem);
    unsigned *c1 = c0 + GE::SPAN / 2;
    const int16_t *p = static_cast<const int16_t *>(in) + sg.in_off;
    if (stereo && interior && ((reinterpret_cast<size_t>(p + 2 * x_first) & 15) == 0)) {
      constexpr int NV = GE::SPAN / 4, PER = (NV + RS_THREADS - 1) / RS_THREADS; /* 4 frames per load */
      const uint4 *src = reinterpret_cast<const uint4 *>(p + 2 * x_first);
      uint4 v[PER];
#pragma unroll
      for (int i = 0; i < PER; ++i) v[i] = src[min(tid + RS_THREADS * i, NV - 1)]; /* clamped, not masked */
#pragma unroll
      for (int i = 0; i < PER; ++i) {
        const int idx = tid + RS_THREADS * i;
        if (idx < NV) {
          uint2 a, b;
          a.x = (v[i].x & 0xFFFFu) | (v[i].y << 16);
          a.y = (v[i].z & 0xFFFFu) | (v[i].w << 16);
          b.x = (v[i].x >> 16) | (v[i].y & 0xFFFF0000u);
          b.y = (v[i].z >> 16) | (v[i].w & 0xFFFF0000u);
          reinterpret_cast<uint2 *>(c0)[idx] = a;
          reinterpret_cast<uint2 *>(c1)[idx] = b;
        }
      }
    } else if (!stereo && interior && ((reinterpret_cast<size_t>(p + x_first) & 15) == 0)) {
      /* mono: 8 frames per load, up-mixed (Q15 1/sqrt(2)) into both channels */
      constexpr int NV = GE::SPAN / 8, PER = (NV + RS_THREADS - 1) / RS_THREADS;
      const uint4 *src = reinterpret_cast<const uint4 *>(p + x_first);
      uint4 v[PER];
#pragma unroll
      for (int i = 0; i < PER; ++i) v[i] = src[min(tid + RS_THREADS * i, NV - 1)];
#pragma unroll
      for (int i = 0; i < PER; ++i) {
        const int idx = tid + RS_THREADS * i;
        if (idx < NV) {
          const unsigned w[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
          unsigned o[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int lo = (((int)(short)(w[k] & 0xFFFFu)) * 23170 + 16384) >> 15;
            const int hi = (((int)(short)(w[k] >> 16)) * 23170 + 16384) >> 15;
            o[k] = ((unsigned)lo & 0xFFFFu) | ((unsigned)hi << 16);
          }
          reinterpret_cast<uint4 *>(c0)[idx] = make_uint4(o[0], o[1], o[2], o[3]);
          reinterpret_cast<uint4 *>(c1)[idx] = make_uint4(o[0], o[1], o[2], o[3]);
        }
      }
    } else if (stereo && interior) { /* 4-byte aligned only (an odd first frame): one frame per load, all in flight */
      constexpr int NP = GE::SPAN / 2, PER = (NP + RS_THREADS - 1) / RS_THREADS;
      const unsigned *src = reinterpret_cast<const unsigned *>(p + 2 * x_first);
      unsigned f0[PER], f1[PER];
#pragma unroll
      for (int i = 0; i < PER; ++i) {
        const int m = min(tid + RS_THREADS * i, NP - 1);
        f0[i] = src[2 * m];
        f1[i] = src[2 * m + 1];
      }
#pragma unroll
      for (int i = 0; i < PER; ++i) {
        const int m = tid + RS_THREADS * i;
        if (m < NP) {
          c0[m] = (f0[i] & 0xFFFFu) | (f1[i] << 16);
          c1[m] = (f0[i] >> 16) | (f1[i] & 0xFFFF0000u);
        }
      }
    } else {
      auto frame = [&](int k) -> unsigned { /* (L, R) of staged frame k as two int16 */
        const long long xi = source(k);
        if (xi < 0) return 0u;
        if (stereo) return reinterpret_cast<const unsigned *>(p)[xi];
        const unsigned m = (unsigned)((((int)p[xi] * 23170 + 16384) >> 15) & 0xFFFF); /* Q15 1/sqrt(2) */
        return m | (m << 16);
      };
      for (int m = tid; m < GE::SPAN / 2; m += RS_THREADS) {
        const unsigned f0 = frame(2 * m), f1 = frame(2 * m + 1);
        c0[m] = (f0 & 0xFFFFu) | (f1 << 16);
        c1[m] = (f0 >> 16) | (f1 & 0xFFFF0000u);
      }
    }
  }
  __syncthreads();

  unsigned res[RS1_OUT];
  if constexpr (F32) {
    const rs_f2 *xs = reinterpret_cast<const rs_f2 *>(rs_smem);
    const float *cg = static_cast<const float *>(bank_g);
    rs_f2 acc[RS1_OUT][8];
#pragma unroll
    for (int k = 0; k < RS1_OUT; ++k)
#pragma unroll
      for (int q = 0; q < 8; ++q) acc[k][q] = rs_f2{0.0f, 0.0f};
    const int chunk0 = tid * (RS1_OUT * D / 8 > 0 ? RS1_OUT * D / 8 : 1);
    static_assert((RS1_OUT * D) % 8 == 0, "a lane starts on a chunk of 8 frames");
#pragma unroll
    for (int cch = 0; cch < GE::F / 8; ++cch) {
      const float4 *src = reinterpret_cast<const float4 *>(xs + (size_t)(chunk0 + cch) * 10);
      float4 v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = src[i];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int f = cch * 8 + i;
        const rs_f2 x = (i & 1) ? rs_f2{v[i >> 1].z, v[i >> 1].w} : rs_f2{v[i >> 1].x, v[i >> 1].y};
#pragma unroll
        for (int k = 0; k < RS1_OUT; ++k) {
          const int t = f - D * k;
          if (t >= 0 && t < L) {
            const float cf = cg[t];
            acc[k][t & 7] = __builtin_elementwise_fma(x, rs_f2{cf, cf}, acc[k][t & 7]);
          }
        }
      }
    }
#pragma unroll
    for (int k = 0; k < RS1_OUT; ++k) {
      const rs_f2 *a = acc[k];
      const rs_f2 v = ((a[0] + a[4]) + (a[2] + a[6])) + ((a[1] + a[5]) + (a[3] + a[7]));
      const float ra = fminf(fmaxf(rintf(v.x * 32768.0f), -32768.0f), 32767.0f);
      const float rb = fminf(fmaxf(rintf(v.y * 32768.0f), -32768.0f), 32767.0f);
      res[k] = ((unsigned)(int)ra & 0xFFFFu) | ((unsigned)(int)rb << 16);
    }
  } else {
    const unsigned *c0 = reinterpret_cast<const unsigned *>(rs_smem);
    const unsigned *c1 = c0 + GE::SPAN / 2;
    const int *cg = static_cast<const int *>(bank_g);
    int acc0[RS1_OUT], acc1[RS1_OUT];
#pragma unroll
    for (int k = 0; k < RS1_OUT; ++k) acc0[k] = acc1[k] = 1 << 14;
    const int d0 = tid * (RS1_OUT * D / 2); /* first dword (frame pair) of this lane */
#pragma unroll
    for (int cch = 0; cch < GE::F / 8; ++cch) {
      const uint4 va = *reinterpret_cast<const uint4 *>(c0 + d0 + cch * 4);
      const uint4 vb = *reinterpret_cast<const uint4 *>(c1 + d0 + cch * 4);
      const unsigned xa[4] = {va.x, va.y, va.z, va.w}, xb[4] = {vb.x, vb.y, vb.z, vb.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int f = cch * 8 + 2 * i;
#pragma unroll
        for (int k = 0; k < RS1_OUT; ++k) {
          const int t = f - D * k;
          if (t >= 0 && t + 1 < L) {
            const unsigned cp = ((unsigned)cg[t] & 0xFFFFu) | ((unsigned)cg[t + 1] << 16);
            const rs_s2 cv = __builtin_bit_cast(rs_s2, cp);
            acc0[k] = __builtin_amdgcn_sdot2(__builtin_bit_cast(rs_s2, xa[i]), cv, acc0[k], false);
            acc1[k] = __builtin_amdgcn_sdot2(__builtin_bit_cast(rs_s2, xb[i]), cv, acc1[k], false);
          }
        }
      }
    }
#pragma unroll
    for (int k = 0; k < RS1_OUT; ++k)
      res[k] = ((unsigned)rs_clip16(acc0[k] >> 15) & 0xFFFFu) | ((unsigned)rs_clip16(acc1[k] >> 15) << 16);
  }

  unsigned *o = reinterpret_cast<unsigned *>(out + sg.out_off) + n0 + RS1_OUT * tid;
  const int left = cnt - RS1_OUT * tid;
  if (left >= RS1_OUT && ((reinterpret_cast<size_t>(o) & 15) == 0)) {
    *reinterpret_cast<uint4 *>(o) = make_uint4(res[0], res[1], res[2], res[3]);
  } else {
#pragma unroll
    for (int k = 0; k < RS1_OUT; ++k)
      if (k < left) o[k] = res[k];
  }
}

template <bool F32, int D, int L>
int rs_launch_1p(hipStream_t s, const void *d_in, const bl_rs_dsong *d_songs, int n_songs, int max_out_frames,
                 const void *d_bank, int16_t *d_out) {
  const unsigned tiles = (unsigned)((max_out_frames + RS_TILE - 1) / RS_TILE);
  typedef rs1_geom<F32, D, L> GE;
  const size_t lds = GE::LDS;
  hipLaunchKernelGGL((k_resample_1p<F32, D, L>), dim3(tiles, (unsigned)n_songs), dim3(RS_THREADS), lds, s,
                     d_in, d_songs, d_bank, d_out);
  BL_HIP_CHECK(hipGetLastError());
  return BL_OK;
}

/* ---- many phases, whole-number advance per phase cycle: 48 kHz (147 phases, 320 frames) ----
 *
 * Outputs n and n + phases use the same coefficient row, `adv` input frames apart.  A wave
 * therefore takes one phase r at a time and lets lane m compute output n0 + r + phases * m: the
 * row is wave-uniform (scalar loads, no LDS traffic for coefficients) and lane m reads a window
 * that starts adv * m frames after lane 0's.  adv is a multiple of 64, so the lanes' windows are
 * staged as 64 separate regions of adv + L frames (the L frames two neighbours share are stored
 * twice) at an odd dword stride: every ds_read of the wave is then conflict-free and a window
 * never straddles a gap.  s16: channels planar, two frames per dword, v_dot2 with the row packed
 * into int16 pairs on the scalar unit — shifted by one tap when the window starts on an odd
 * frame.  float: one channel per pass (a stereo tile does not fit the LDS), eight partial sums in
 * the host's order.  Results go through an LDS tile so that the stores to HBM are contiguous. */
#define RSP_WAVES 8
#define RSP_THREADS (64 * RSP_WAVES)

struct bl_rs_pm {
  int phase_count, adv, w0, rstride, row_stride; /* rstride: dwords between the lanes' regions, odd */
  int tiles_per_wg;
};

#define RSP_REG (64 / RSP_WAVES) /* regions a wave stages */

template <bool F32, int L>
__global__ __launch_bounds__(RSP_THREADS) void k_resample_pm(const void *__restrict__ in,
                                                             const bl_rs_dsong *__restrict__ songs,
                                                             const void *__restrict__ bank_g, bl_rs_pm P,
                                                             int16_t *__restrict__ out) {
  extern __shared__ __align__(16) unsigned char rs_smem[];
  const bl_rs_dsong sg = songs[blockIdx.y];
  const int pc = P.phase_count, adv = P.adv, R = P.rstride;
  const int T = pc * 64;
  const int tiles_total = (int)(((long long)sg.out_frames + T - 1) / T);
  const int tile_begin = blockIdx.x * P.tiles_per_wg;
  if (tile_begin >= tiles_total) return;
  const int tile_end = min(tile_begin + P.tiles_per_wg, tiles_total);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool stereo = sg.channels == 2;
  const long long N = sg.frames;
  /* the staged frames start `delta` (< 4) before the first window so that they start on a
   * 16-byte boundary of the input; every window is `delta` further into its region */
  const int delta = (((P.w0 - L) % 4) + 4) % 4;
  const int rfr = adv + L + delta; /* frames a region holds */

  unsigned *ob = reinterpret_cast<unsigned *>(rs_smem); /* the tile's output frames, (L, R) int16 */

  /* phase r = wave, wave + RSP_WAVES, ...: row (r * adv) mod pc, first frame (r * adv) div pc */
  const int idx0 = (wave * adv) % pc, base0 = (wave * adv) / pc + delta;
  const int idx_step = (RSP_WAVES * adv) % pc, base_step = (RSP_WAVES * adv) / pc;

  auto x_first_of = [&](int tile) -> long long { /* input index of the tile's staged frame 0 */
    return (long long)P.w0 + (long long)tile * 64 * adv - L - delta;
  };
  /* input index of frame x_first + f, -1: nothing there (beyond the flush reflection) */
  auto source = [&](long long x_first, int f) -> long long {
    const long long e = x_first + f;
    if (e < 0) return -e;
    if (e < N) return e;
    const long long j = e - N;
    return j < sg.refl ? N - 1 - j : -1;
  };

  if constexpr (!F32) {
    unsigned *c0 = ob + T, *c1 = c0 + 64 * R;
    const int16_t *p = static_cast<const int16_t *>(in) + sg.in_off;
    const int nq = (rfr + 2 + 3) / 4; /* 16-byte units (4 frames) per region */
    constexpr int PFU = 2;            /* units per lane and region: adv + L + 2 <= 512 frames */
    const bool can_vector = ((reinterpret_cast<size_t>(p) & 15) == 0) && (adv % 4 == 0) && nq <= 64 * PFU;
    /* a tile away from the song's edges is fetched with 16-byte loads into registers while the
     * previous tile is being computed */
    auto interior = [&](int tile) -> bool {
      const long long xf = x_first_of(tile);
      return can_vector && xf >= 0 && xf + 63LL * adv + 4LL * nq <= N;
    };
    uint4 pf[RSP_REG][PFU];
    auto prefetch = [&](int tile) {
      if (stereo) {
        const uint4 *src = reinterpret_cast<const uint4 *>(p + 2 * x_first_of(tile));
#pragma unroll
        for (int k = 0; k < RSP_REG; ++k) {
          const int m = wave + RSP_WAVES * k;
#pragma unroll
          for (int u = 0; u < PFU; ++u) {
            /* unconditional (clamped): a load under a lane mask makes the compiler wait for it at once */
            pf[k][u] = src[(m * adv) / 4 + min(lane + 64 * u, nq - 1)];
          }
        }
      } else { /* mono: the same four frames are 8 bytes */
        const uint2 *src = reinterpret_cast<const uint2 *>(p + x_first_of(tile));
#pragma unroll
        for (int k = 0; k < RSP_REG; ++k) {
          const int m = wave + RSP_WAVES * k;
#pragma unroll
          for (int u = 0; u < PFU; ++u) {
            const uint2 v = src[(m * adv) / 4 + min(lane + 64 * u, nq - 1)];
            pf[k][u] = make_uint4(v.x, v.y, 0u, 0u);
          }
        }
      }
    };
    auto commit = [&]() {
#pragma unroll
      for (int k = 0; k < RSP_REG; ++k) {
        const int m = wave + RSP_WAVES * k;
#pragma unroll
        for (int u = 0; u < PFU; ++u) {
          const int q = lane + 64 * u;
          if (q < nq) {
            const uint4 v = pf[k][u];
            unsigned *d0 = c0 + m * R + 2 * q, *d1 = c1 + m * R + 2 * q;
            if (stereo) {
              d0[0] = (v.x & 0xFFFFu) | (v.y << 16);
              d0[1] = (v.z & 0xFFFFu) | (v.w << 16);
              d1[0] = (v.x >> 16) | (v.y & 0xFFFF0000u);
              d1[1] = (v.z >> 16) | (v.w & 0xFFFF0000u);
            } else { /* up-mix (Q15 1/sqrt(2)) into both channels */
              const unsigned w[2] = {v.x, v.y};
#pragma unroll
              for (int h = 0; h < 2; ++h) {
                const int lo = (((int)(short)(w[h] & 0xFFFFu)) * 23170 + 16384) >> 15;
                const int hi = (((int)(short)(w[h] >> 16)) * 23170 + 16384) >> 15;
                d0[h] = d1[h] = ((unsigned)lo & 0xFFFFu) | ((unsigned)hi << 16);
              }
            }
          }
        }
      }
    };
    auto stage_slow = [&](int tile) { /* edges, mono, odd alignments: frame by frame */
      const long long xf = x_first_of(tile);
      auto frame = [&](int f) -> unsigned { /* (L, R) of staged frame f as two int16 */
        const long long xi = source(xf, f);
        if (xi < 0) return 0u;
        if (stereo) return reinterpret_cast<const unsigned *>(p)[xi];
        const unsigned mm = (unsigned)((((int)p[xi] * 23170 + 16384) >> 15) & 0xFFFF); /* Q15 1/sqrt(2) */
        return mm | (mm << 16);
      };
      const int rd = rfr / 2 + 1;
      for (int m = wave; m < 64; m += RSP_WAVES)
        for (int j = lane; j < rd; j += 64) {
          const unsigned f0 = frame(m * adv + 2 * j), f1 = frame(m * adv + 2 * j + 1);
          c0[m * R + j] = (f0 & 0xFFFFu) | (f1 << 16);
          c1[m * R + j] = (f0 >> 16) | (f1 & 0xFFFF0000u);
        }
    };

    /* Per phase, the row as L / 2 + 1 pairs of Q15 taps in the pairing its windows need: a
     * window that starts on the odd half of a dword takes (-1, 0), (1, 2), ..., (L-1, L) with zeros
     * outside the row, an even one (0, 1), ..., and a zero pair.  Built once per workgroup (the
     * phases do not depend on the tile).  Scalar loads of the row inside the loop would share
     * the LDS reads' wait counter and serialise with them. */
    const int *bank = static_cast<const int *>(bank_g);
    unsigned *tp = c1 + 64 * R;
    for (int e = tid; e < pc * (L / 2 + 1); e += RSP_THREADS) {
      const int r = e / (L / 2 + 1), j = e - r * (L / 2 + 1);
      const int *row = bank + (size_t)((r * adv) % pc) * P.row_stride;
      const int t0 = 2 * j - (((r * adv) / pc + delta) & 1), t1 = t0 + 1;
      const unsigned lo = (t0 >= 0 && t0 < L) ? ((unsigned)row[t0] & 0xFFFFu) : 0u;
      const unsigned hi = (t1 >= 0 && t1 < L) ? ((unsigned)row[t1] << 16) : 0u;
      tp[e] = lo | hi;
    }
    bool fetched = interior(tile_begin);
    if (fetched) prefetch(tile_begin);
    for (int tile = tile_begin; tile < tile_end; ++tile) {
      if (fetched) commit();
      else stage_slow(tile);
      __syncthreads();
      fetched = tile + 1 < tile_end && interior(tile + 1);
      if (fetched) prefetch(tile + 1);

      /* Two consecutive phases per pass: their windows start 2 or 3 frames apart (adv / phases), so
       * one read of L / 2 + 3 dwords per channel serves both — the second phase uses the same
       * registers one or two dwords further on (its tap pairs in the table already have the parity
       * of its own start).  Lane j takes a phase's j-th pair of taps from the table; v_readlane
       * hands it to the whole wave. */
      constexpr int NP = L / 2 + 1;
      for (int pr = wave; 2 * pr < pc; pr += RSP_WAVES) {
        const int ra = 2 * pr, rb = min(ra + 1, pc - 1); /* rb == ra: the odd phase out at the end */
        const int da = ((ra * adv) / pc + delta) >> 1, diff = (((rb * adv) / pc + delta) >> 1) - da;
        const unsigned vpa = tp[ra * NP + min(lane, L / 2)], vpb = tp[rb * NP + min(lane, L / 2)];
        const unsigned *a0 = c0 + lane * R + da, *a1 = c1 + lane * R + da;
        /* all of the reads go out before the first multiply (left alone, the compiler issues them
         * four at a time and waits for each group) */
        unsigned xa[NP + 2], xb[NP + 2];
#pragma unroll
        for (int j = 0; j < NP + 2; ++j) {
          xa[j] = a0[j];
          xb[j] = a1[j];
        }
#pragma unroll
        for (int j = 0; j < NP + 2; ++j) asm volatile("" : "+v"(xa[j]), "+v"(xb[j]));
        auto phase = [&](int r, unsigned vp, const unsigned *wa, const unsigned *wb) {
          int acc0 = 1 << 14, acc1 = 1 << 14;
#pragma unroll
          for (int j = 0; j < NP; ++j) {
            const rs_s2 cv = __builtin_bit_cast(rs_s2, (unsigned)__builtin_amdgcn_readlane((int)vp, j));
            acc0 = __builtin_amdgcn_sdot2(__builtin_bit_cast(rs_s2, wa[j]), cv, acc0, false);
            acc1 = __builtin_amdgcn_sdot2(__builtin_bit_cast(rs_s2, wb[j]), cv, acc1, false);
          }
          ob[r + pc * lane] =
              ((unsigned)rs_clip16(acc0 >> 15) & 0xFFFFu) | ((unsigned)rs_clip16(acc1 >> 15) << 16);
        };
        phase(ra, vpa, xa, xb);
        if (diff == 1) phase(rb, vpb, xa + 1, xb + 1);      /* wave-uniform */
        else if (diff == 2) phase(rb, vpb, xa + 2, xb + 2);
      }
      __syncthreads();
      const long long n0 = (long long)tile * T;
      const int cnt = (int)min((long long)T, (long long)sg.out_frames - n0);
      unsigned *o = reinterpret_cast<unsigned *>(out + sg.out_off) + n0;
      for (int i = tid; i < cnt; i += RSP_THREADS) o[i] = ob[i];
    }
  } else {
    float *xs = reinterpret_cast<float *>(ob + T);
    const int32_t *p = static_cast<const int32_t *>(in) + sg.in_off;
    const float *bank = static_cast<const float *>(bank_g);
    for (int tile = tile_begin; tile < tile_end; ++tile) {
      const long long xf = x_first_of(tile);
      for (int ch = 0; ch < (stereo ? 2 : 1); ++ch) {
        __syncthreads(); /* every wave is done with the previous samples and output tile */
        const int nu = (rfr + 1) / 2; /* 16-byte units (2 frames, both channels) per region */
        const bool fast = stereo && ((reinterpret_cast<size_t>(p) & 15) == 0) && xf >= 0 &&
                          xf + 63LL * adv + 2LL * nu <= N && nu <= 256;
        if (fast) { /* away from the song's edges: 16-byte loads, 16 in flight per lane */
          const int4 *src = reinterpret_cast<const int4 *>(p + 2 * xf);
#pragma unroll
          for (int half = 0; half < 2; ++half) {
            int4 v[RSP_REG / 2][4];
#pragma unroll
            for (int k = 0; k < RSP_REG / 2; ++k) {
              const int m = wave + RSP_WAVES * (half * (RSP_REG / 2) + k);
#pragma unroll
              for (int u = 0; u < 4; ++u) v[k][u] = src[(m * adv) / 2 + min(lane + 64 * u, nu - 1)];
            }
#pragma unroll
            for (int k = 0; k < RSP_REG / 2; ++k) {
              const int m = wave + RSP_WAVES * (half * (RSP_REG / 2) + k);
#pragma unroll
              for (int u = 0; u < 4; ++u) {
                const int q = lane + 64 * u;
                if (q < nu) {
                  xs[m * R + 2 * q] = (float)(ch ? v[k][u].y : v[k][u].x) * (1.0f / 2147483648.0f);
                  xs[m * R + 2 * q + 1] = (float)(ch ? v[k][u].w : v[k][u].z) * (1.0f / 2147483648.0f);
                }
              }
            }
          }
        } else
        for (int m = wave; m < 64; m += RSP_WAVES) {
          for (int j0 = 0; j0 < rfr; j0 += 256) {
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const int j = j0 + lane + 64 * u;
              v[u] = 0.0f;
              if (j < rfr) {
                const long long xi = source(xf, m * adv + j);
                if (xi >= 0)
                  v[u] = stereo ? (float)p[2 * xi + ch] * (1.0f / 2147483648.0f)
                                : (float)p[xi] * (1.0f / 2147483648.0f) * (float)0.70710678118654752440;
              }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const int j = j0 + lane + 64 * u;
              if (j < rfr) xs[m * R + j] = v[u];
            }
          }
        }
        __syncthreads();
        /* lane i holds tap i (a second register the taps from 64 on), handed out by v_readlane */
        static_assert(L <= 128, "two registers of taps");
        auto load_taps = [&](int idx, float &lo, float &hi) {
          const float *row = bank + (size_t)idx * P.row_stride;
          lo = row[min(lane, L - 1)];
          hi = row[min(64 + lane, L - 1)];
        };
        int idx = idx0, base = base0;
        float t_lo, t_hi;
        load_taps(idx, t_lo, t_hi);
        for (int r = wave; r < pc; r += RSP_WAVES) {
          int idx_n = idx + idx_step, base_n = base + base_step;
          if (idx_n >= pc) { idx_n -= pc; ++base_n; }
          float n_lo, n_hi;
          load_taps(idx_n, n_lo, n_hi);
          const float *x = xs + lane * R + base;
          float xv[L];
#pragma unroll
          for (int i = 0; i < L; ++i) xv[i] = x[i];
#pragma unroll
          for (int i = 0; i < L; ++i) asm volatile("" : "+v"(xv[i]));
          float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
          for (int i = 0; i < L; ++i) {
            const float cf = __builtin_bit_cast(
                float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, i < 64 ? t_lo : t_hi), i & 63));
            a[i & 7] = __builtin_fmaf(xv[i], cf, a[i & 7]);
          }
          const float v = ((a[0] + a[4]) + (a[2] + a[6])) + ((a[1] + a[5]) + (a[3] + a[7]));
          const int q = (int)fminf(fmaxf(rintf(v * 32768.0f), -32768.0f), 32767.0f);
          short *o16 = reinterpret_cast<short *>(ob + r + pc * lane);
          if (stereo) o16[ch] = (short)q;
          else ob[r + pc * lane] = ((unsigned)q & 0xFFFFu) | ((unsigned)q << 16);
          t_lo = n_lo;
          t_hi = n_hi;
          idx = idx_n;
          base = base_n;
        }
      }
      __syncthreads();
      const long long n0 = (long long)tile * T;
      const int cnt = (int)min((long long)T, (long long)sg.out_frames - n0);
      unsigned *o = reinterpret_cast<unsigned *>(out + sg.out_off) + n0;
      for (int i = tid; i < cnt; i += RSP_THREADS) o[i] = ob[i];
    }
  }
}

template <bool F32, int L>
int rs_launch_pm(hipStream_t s, const void *d_in, const bl_rs_dsong *d_songs, int n_songs, int max_out_frames,
                 const void *d_bank, const bl_rs_pm &P, size_t lds, int16_t *d_out) {
  static bool configured[BL_RS_MAX_DEVICES] = {false};
  int dev = 0;
  BL_HIP_CHECK(hipGetDevice(&dev));
  if (dev >= 0 && dev < BL_RS_MAX_DEVICES && !configured[dev]) {
    BL_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_resample_pm<F32, L>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, RS_LDS_LIMIT));
    configured[dev] = true;
  }
  const long long T = (long long)P.phase_count * 64;
  const unsigned tiles = (unsigned)((max_out_frames + T - 1) / T);
  const unsigned groups = (tiles + (unsigned)P.tiles_per_wg - 1) / (unsigned)P.tiles_per_wg;
  hipLaunchKernelGGL((k_resample_pm<F32, L>), dim3(groups, (unsigned)n_songs), dim3(RSP_THREADS), lds, s, d_in,
                     d_songs, d_bank, P, d_out);
  BL_HIP_CHECK(hipGetLastError());
  return BL_OK;
}

template <bool F32, bool BANK_LDS>
int rs_launch(hipStream_t s, const void *d_in, const bl_rs_dsong *d_songs, int n_songs, int max_out_frames,
              const void *d_bank, const bl_rs_geom &g, int16_t *d_out, size_t lds) {
  static bool configured[BL_RS_MAX_DEVICES] = {false};
  int dev = 0;
  BL_HIP_CHECK(hipGetDevice(&dev));
  if (dev >= 0 && dev < BL_RS_MAX_DEVICES && !configured[dev]) {
    BL_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_resample<F32, BANK_LDS>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, RS_LDS_LIMIT));
    configured[dev] = true;
  }
  /* a workgroup walks a run of tiles so that the bank is staged once for all of them; short
   * runs when the batch is small */
  const long long tiles = (max_out_frames + RS_TILE - 1) / RS_TILE;
  bl_rs_geom gg = g;
  gg.tiles_per_wg = (int)std::min<long long>(16, std::max<long long>(1, tiles * n_songs / 2048));
  const unsigned groups = (unsigned)((tiles + gg.tiles_per_wg - 1) / gg.tiles_per_wg);
  hipLaunchKernelGGL((k_resample<F32, BANK_LDS>), dim3(groups, (unsigned)n_songs), dim3(RSG_THREADS), lds, s,
                     d_in, d_songs, d_bank, gg, d_out);
  BL_HIP_CHECK(hipGetLastError());
  return BL_OK;
}

} // namespace

int blk_resample_geom(int phase_count, int taps, int alloc, int src_incr, int dst_incr, bl_rs_geom *g,
                      size_t *lds_bytes, int *bank_in_lds) {
  g->phase_count = phase_count;
  g->taps = taps;
  g->taps8 = (taps + 7) & ~7;
  g->alloc = alloc;
  g->w0 = taps - (taps - 1) / 2;
  g->src_incr = (unsigned long long)src_incr;
  g->dst_incr = (unsigned long long)dst_incr;
  /* the widest input span a tile can read: first to last window start, plus one window */
  const unsigned long long adv =
      (unsigned long long)(RS_TILE - 1) * g->dst_incr / (g->src_incr * (unsigned long long)phase_count);
  g->span = ((int)adv + 2 + g->taps8 + 1) & ~1;
  const size_t samples = (size_t)g->span * 8; /* (L, R) pairs of 32-bit elements */
  const size_t bank = (size_t)phase_count * (size_t)(g->taps8 + 4) * 4;
  if (samples > RS_LDS_LIMIT) return BL_UNEXPECTED;
  *bank_in_lds = samples + bank <= RS_LDS_LIMIT;
  *lds_bytes = samples + (*bank_in_lds ? bank : 0);
  return BL_OK;
}

int blk_resample(hipStream_t s, const void *d_in, int in_is_s32, const bl_rs_dsong *d_songs, int n_songs,
                 int max_out_frames, const void *d_bank, const bl_rs_geom &g, size_t lds_bytes,
                 int bank_in_lds, int16_t *d_out) {
  if (n_songs <= 0 || max_out_frames <= 0) return BL_UNEXPECTED;
  if (g.phase_count == 1 && g.dst_incr % g.src_incr == 0 && !getenv("BL_AMD_RS_GENERIC")) {
    const unsigned long long step = g.dst_incr / g.src_incr;
    if (step == 2 && g.taps == 66)
      return in_is_s32 ? rs_launch_1p<true, 2, 66>(s, d_in, d_songs, n_songs, max_out_frames, d_bank, d_out)
                       : rs_launch_1p<false, 2, 66>(s, d_in, d_songs, n_songs, max_out_frames, d_bank, d_out);
    if (step == 4 && g.taps == 132)
      return in_is_s32 ? rs_launch_1p<true, 4, 132>(s, d_in, d_songs, n_songs, max_out_frames, d_bank, d_out)
                       : rs_launch_1p<false, 4, 132>(s, d_in, d_songs, n_songs, max_out_frames, d_bank, d_out);
  }
  if (g.phase_count > 1 && g.dst_incr % g.src_incr == 0 && g.taps == 72 && !getenv("BL_AMD_RS_GENERIC")) {
    const unsigned long long adv = g.dst_incr / g.src_incr; /* input frames per cycle of the phases */
    bl_rs_pm P;
    P.phase_count = g.phase_count;
    P.adv = (int)adv;
    P.w0 = g.w0;
    P.row_stride = g.alloc;
    const int rfr = (int)adv + g.taps + 4; /* + the alignment lead, see the kernel */
    P.rstride = in_is_s32 ? ((rfr + 2) | 1) : ((rfr / 2 + 1) | 1);
    const size_t lds = ((size_t)g.phase_count * 64 + (size_t)(in_is_s32 ? 1 : 2) * 64 * (size_t)P.rstride +
                        (in_is_s32 ? 0 : (size_t)g.phase_count * (g.taps / 2 + 1))) * 4;
    /* a workgroup walks a run of tiles (the next one is fetched while one is computed); short
     * runs when the batch is small so that every CU still gets work */
    const long long T = (long long)g.phase_count * 64;
    const long long all_tiles = (long long)n_songs * ((max_out_frames + T - 1) / T);
    P.tiles_per_wg = (int)std::min<long long>(8, std::max<long long>(1, all_tiles / 1024));
    if (adv % 2 == 0 && adv <= 4096 && lds <= RS_LDS_LIMIT)
      return in_is_s32 ? rs_launch_pm<true, 72>(s, d_in, d_songs, n_songs, max_out_frames, d_bank, P, lds, d_out)
                       : rs_launch_pm<false, 72>(s, d_in, d_songs, n_songs, max_out_frames, d_bank, P, lds, d_out);
  }
  if (in_is_s32)
    return bank_in_lds ? rs_launch<true, true>(s, d_in, d_songs, n_songs, max_out_frames, d_bank, g, d_out, lds_bytes)
                       : rs_launch<true, false>(s, d_in, d_songs, n_songs, max_out_frames, d_bank, g, d_out, lds_bytes);
  return bank_in_lds ? rs_launch<false, true>(s, d_in, d_songs, n_songs, max_out_frames, d_bank, g, d_out, lds_bytes)
                     : rs_launch<false, false>(s, d_in, d_songs, n_songs, max_out_frames, d_bank, g, d_out, lds_bytes);
}
