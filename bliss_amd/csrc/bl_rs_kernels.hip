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
 * One workgroup converts RS_TILE consecutive output frames of one song.  It stages the input
 * span those outputs read (reflected at the song's edges exactly as the host form does) and,
 * when it fits, the bank into LDS; then each lane computes one output frame at a time.  The
 * work is LDS-read bound (taps * (1 coefficient + channels samples) reads per output frame);
 * bank rows are padded to an odd stride so that lanes on different phases spread over the banks.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "bliss.h"
#include "bl_runtime.h"

#define RS_TILE 1024
#define RS_THREADS 256
#define RS_LDS_LIMIT (150 * 1024)

namespace {

template <bool F32> struct rs_elem { typedef int type; };
template <> struct rs_elem<true> { typedef float type; };

__device__ __forceinline__ int rs_clip16(int v) { return v > 32767 ? 32767 : v < -32768 ? -32768 : v; }

template <bool F32, bool STEREO>
__device__ __forceinline__ unsigned rs_output(const typename rs_elem<F32>::type *__restrict__ x0,
                                              const typename rs_elem<F32>::type *__restrict__ x1,
                                              const typename rs_elem<F32>::type *__restrict__ c, int taps8) {
  if constexpr (F32) {
    float a[8] = {0, 0, 0, 0, 0, 0, 0, 0}, b[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < taps8; i += 8) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const float cf = c[i + q];
        a[q] = __builtin_fmaf(x0[i + q], cf, a[q]);
        if (STEREO) b[q] = __builtin_fmaf(x1[i + q], cf, b[q]);
      }
    }
    const float va = ((a[0] + a[4]) + (a[2] + a[6])) + ((a[1] + a[5]) + (a[3] + a[7]));
    float ra = rintf(va * 32768.0f);
    ra = fminf(fmaxf(ra, -32768.0f), 32767.0f);
    const unsigned l = (unsigned)(int)ra & 0xFFFFu;
    if (!STEREO) return l | (l << 16);
    const float vb = ((b[0] + b[4]) + (b[2] + b[6])) + ((b[1] + b[5]) + (b[3] + b[7]));
    float rb = rintf(vb * 32768.0f);
    rb = fminf(fmaxf(rb, -32768.0f), 32767.0f);
    return l | ((unsigned)(int)rb << 16);
  } else {
    unsigned a = 1u << 14, b = 1u << 14;
    for (int i = 0; i < taps8; i += 8) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int cf = c[i + q];
        a += (unsigned)(x0[i + q] * cf);
        if (STEREO) b += (unsigned)(x1[i + q] * cf);
      }
    }
    const unsigned l = (unsigned)rs_clip16((int)a >> 15) & 0xFFFFu;
    if (!STEREO) return l | (l << 16);
    return l | ((unsigned)rs_clip16((int)b >> 15) << 16);
  }
}

template <bool F32, bool BANK_LDS>
__global__ __launch_bounds__(RS_THREADS) void k_resample(const void *__restrict__ in,
                                                         const bl_rs_dsong *__restrict__ songs,
                                                         const void *__restrict__ bank_g, bl_rs_geom G,
                                                         int16_t *__restrict__ out) {
  typedef typename rs_elem<F32>::type T;
  extern __shared__ __align__(16) unsigned char rs_smem[];
  const bl_rs_dsong sg = songs[blockIdx.y];
  const long long n0 = (long long)blockIdx.x * RS_TILE;
  if (n0 >= sg.out_frames) return;
  const int cnt = (int)min((long long)RS_TILE, (long long)sg.out_frames - n0);
  const int tid = threadIdx.x;
  const unsigned pc = (unsigned)G.phase_count;
  const int L = G.taps, taps8 = G.taps8;
  const bool stereo = sg.channels == 2;

  auto position = [&](long long n, int &index) -> long long {
    const unsigned long long t = (unsigned long long)n * G.dst_incr / G.src_incr;
    index = (int)(t % pc);
    return (long long)G.w0 + (long long)(t / pc);
  };
  int idx_unused;
  const long long w_first = position(n0, idx_unused);
  const long long w_last = position(n0 + cnt - 1, idx_unused);
  const int span = (int)(w_last - w_first) + taps8;

  T *x0 = reinterpret_cast<T *>(rs_smem);
  T *x1 = x0 + G.span;
  T *lb = x1 + G.span;
  const int lb_stride = taps8 + 1;

  /* input span: ext position e = w_first + k; ext[0..L) mirrors the first samples about
   * sample 0, ext[L + N ...] mirrors the last `refl` about the end, nothing beyond */
  const long long N = sg.frames;
  for (int k = tid; k < span; k += RS_THREADS) {
    const long long e = w_first + k - L;
    long long xi = e;
    bool ok = true;
    if (e < 0) xi = -e;
    else if (e >= N) {
      const long long j = e - N;
      ok = j < sg.refl;
      xi = N - 1 - j;
    }
    T a = 0, b = 0;
    if (ok) {
      if constexpr (F32) {
        const int32_t *p = static_cast<const int32_t *>(in) + sg.in_off;
        if (stereo) {
          const int2 v = reinterpret_cast<const int2 *>(p)[xi];
          a = (float)v.x * (1.0f / 2147483648.0f);
          b = (float)v.y * (1.0f / 2147483648.0f);
        } else {
          a = (float)p[xi] * (1.0f / 2147483648.0f) * (float)0.70710678118654752440;
        }
      } else {
        const int16_t *p = static_cast<const int16_t *>(in) + sg.in_off;
        if (stereo) {
          const unsigned v = reinterpret_cast<const unsigned *>(p)[xi];
          a = (int)(short)(v & 0xFFFFu);
          b = (int)(short)(v >> 16);
        } else {
          a = ((int)p[xi] * 23170 + 16384) >> 15; /* Q15 1/sqrt(2) */
        }
      }
    }
    x0[k] = a;
    if (stereo) x1[k] = b;
  }
  if (BANK_LDS) {
    const T *bg = static_cast<const T *>(bank_g);
    const int total = (int)pc * taps8;
    for (int i = tid; i < total; i += RS_THREADS) {
      const int r = i / taps8, q = i - r * taps8;
      lb[r * lb_stride + q] = bg[(size_t)r * G.alloc + q];
    }
  }
  __syncthreads();

  unsigned *o = reinterpret_cast<unsigned *>(out + sg.out_off) + n0;
  for (int j = tid; j < cnt; j += RS_THREADS) {
    int index;
    const long long w = position(n0 + j, index);
    const int base = (int)(w - w_first);
    const T *c = BANK_LDS ? lb + index * lb_stride : static_cast<const T *>(bank_g) + (size_t)index * G.alloc;
    o[j] = stereo ? rs_output<F32, true>(x0 + base, x1 + base, c, taps8)
                  : rs_output<F32, false>(x0 + base, x0 + base, c, taps8);
  }
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
  const unsigned tiles = (unsigned)((max_out_frames + RS_TILE - 1) / RS_TILE);
  hipLaunchKernelGGL((k_resample<F32, BANK_LDS>), dim3(tiles, (unsigned)n_songs), dim3(RS_THREADS), lds, s,
                     d_in, d_songs, d_bank, g, d_out);
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
  g->span = (int)adv + 2 + g->taps8;
  const size_t samples = 2 * (size_t)g->span * 4;
  const size_t bank = (size_t)phase_count * (size_t)(g->taps8 + 1) * 4;
  if (samples > RS_LDS_LIMIT) return BL_UNEXPECTED;
  *bank_in_lds = samples + bank <= RS_LDS_LIMIT;
  *lds_bytes = samples + (*bank_in_lds ? bank : 0);
  return BL_OK;
}

int blk_resample(hipStream_t s, const void *d_in, int in_is_s32, const bl_rs_dsong *d_songs, int n_songs,
                 int max_out_frames, const void *d_bank, const bl_rs_geom &g, size_t lds_bytes,
                 int bank_in_lds, int16_t *d_out) {
  if (n_songs <= 0 || max_out_frames <= 0) return BL_UNEXPECTED;
  if (in_is_s32)
    return bank_in_lds ? rs_launch<true, true>(s, d_in, d_songs, n_songs, max_out_frames, d_bank, g, d_out, lds_bytes)
                       : rs_launch<true, false>(s, d_in, d_songs, n_songs, max_out_frames, d_bank, g, d_out, lds_bytes);
  return bank_in_lds ? rs_launch<false, true>(s, d_in, d_songs, n_songs, max_out_frames, d_bank, g, d_out, lds_bytes)
                     : rs_launch<false, false>(s, d_in, d_songs, n_songs, max_out_frames, d_bank, g, d_out, lds_bytes);
}
