/*
 * bl_fft.h — 512-point real forward DFT, spread over 16 lanes x 16 registers.
 *
 * Replaces the two third-party FFTs the reference calls on the hot path:
 *   libavcodec av_rdft_calc (f32)  — ref src/frequency_sort.c:83
 *   FFTW3 fftw_execute r2c (f64)   — ref src/tempo_atk_sort.c:141
 * Only |X_k|^2 (k = 0..256) is consumed by either caller
 * (ref src/frequency_sort.c:88-93, src/tempo_atk_sort.c:142-149).
 *
 * Mapping (CDNA4 wave64 = 4 independent 16-lane groups, one transform each):
 *   real x[0..511] is packed as 256 complex z[m] = x[2m] + i x[2m+1];
 *   m = 16*m1 + n0: lane n0 holds z for m1 = 0..15 in registers.
 *   pass 1: 16-point DFT over m1 in registers, twiddle W256^(n0*k1),
 *   one 16x16 transpose through LDS (row stride 17 -> conflict free),
 *   pass 2: 16-point DFT over n0 -> lane k1 holds Z[k1 + 16*k0];
 *   real split: lane k1 forms the 8 pairs (k, 256-k), k = k1+16*k0, k0 < 8,
 *   fetching the partner half-row from lane (16-k1)%16 through LDS.
 * FMA is used inside the transform (it is our FFT; its rounding is as
 * implementation-defined as FFTW's), never outside it.
 *
 * Everything here is __host__ __device__ so tests can run the exact same
 * arithmetic lane-by-lane on the CPU (tests/host/test_fft_host.cpp).
 */
#ifndef BL_FFT_H_
#define BL_FFT_H_

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define BL_HD __host__ __device__ __forceinline__
#else
#include <math.h>
#define BL_HD static inline
#endif

template <typename T> struct bl_c2 { T re, im; };

BL_HD double bl_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
BL_HD float bl_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
#if defined(__HIPCC__)
/* two f32 transforms side by side (frame A in .x, frame B in .y): every operation of the
 * templates below becomes one v_pk_*_f32 on a register pair (k_freq_frames) */
typedef float bl_f2 __attribute__((ext_vector_type(2)));
BL_HD bl_f2 bl_fma(bl_f2 a, bl_f2 b, bl_f2 c) { return __builtin_elementwise_fma(a, b, c); }
#endif

/* forward 4-point DFT in place: (a,b,c,d) = x0..x3 -> X0..X3, W4 = -i */
template <typename T>
BL_HD void bl_r4(T &ar, T &ai, T &br, T &bi, T &cr, T &ci, T &dr, T &di) {
  T t0r = ar + cr, t0i = ai + ci, t1r = ar - cr, t1i = ai - ci;
  T t2r = br + dr, t2i = bi + di, t3r = br - dr, t3i = bi - di;
  ar = t0r + t2r; ai = t0i + t2i;
  cr = t0r - t2r; ci = t0i - t2i;
  br = t1r + t3i; bi = t1i - t3r;
  dr = t1r - t3i; di = t1i + t3r;
}

/* (r,i) *= (wr + i wi) */
template <typename T> BL_HD void bl_cmul(T &r, T &i, T wr, T wi) {
  T p = i * wi, q = i * wr;
  T nr = bl_fma(r, wr, -p);
  T ni = bl_fma(r, wi, q);
  r = nr; i = ni;
}

/*
 * Forward 16-point DFT in place.  Input index n = 0..15 natural order.
 * Output X[k] is left at position bl_pos16(k) = 4*(k%4) + k/4.
 */
BL_HD constexpr int bl_pos16(int k) { return 4 * (k & 3) + (k >> 2); }

template <typename T> BL_HD void bl_fft16(T (&re)[16], T (&im)[16]) {
  const T C1 = (T)0.92387953251128673848, S1 = (T)0.38268343236508978178;
  const T R = (T)0.70710678118654752440;
#pragma unroll
  for (int n0 = 0; n0 < 4; ++n0)
    bl_r4(re[n0], im[n0], re[4 + n0], im[4 + n0], re[8 + n0], im[8 + n0], re[12 + n0], im[12 + n0]);
  /* element 4*k1 + n0 now holds A[n0][k1]; it wants W16^(n0*k1) before the second pass.  The
   * general twiddles are multiplied here; the four that are R (1 -+ i) or R (-1 - i) are left as
   * their unscaled sums and the factor R goes into the second pass's additions as an fma (one
   * rounding instead of two, eight multiplications fewer) */
  bl_cmul(re[5], im[5], C1, -S1);   /* exponent 1: (n0,k1) = (1,1) */
  bl_cmul(re[13], im[13], S1, -C1); /* exponent 3: (1,3) */
  bl_cmul(re[7], im[7], S1, -C1);   /* exponent 3: (3,1) */
  bl_cmul(re[15], im[15], -C1, S1); /* exponent 9: (3,3) */
  /* exponent 2, (1,2) idx 9 and (2,1) idx 6: R * ((a + b) + i (b - a)) */
  { T a = re[9], b = im[9]; re[9] = a + b; im[9] = b - a; }
  { T a = re[6], b = im[6]; re[6] = a + b; im[6] = b - a; }
  /* exponent 6, (2,3) idx 14 and (3,2) idx 11: R * ((b - a) - i (a + b)) */
  { T a = re[14], b = im[14]; re[14] = b - a; im[14] = -(a + b); }
  { T a = re[11], b = im[11]; re[11] = b - a; im[11] = -(a + b); }
  /* k1 = 0: no twiddles */
  bl_r4(re[0], im[0], re[1], im[1], re[2], im[2], re[3], im[3]);
  { /* k1 = 1: c = R * (re[6], im[6]) */
    const T ar = re[4], ai = im[4], br = re[5], bi = im[5], cr = re[6], ci = im[6], dr = re[7], di = im[7];
    const T t0r = bl_fma(R, cr, ar), t0i = bl_fma(R, ci, ai), t1r = bl_fma(-R, cr, ar), t1i = bl_fma(-R, ci, ai);
    const T t2r = br + dr, t2i = bi + di, t3r = br - dr, t3i = bi - di;
    re[4] = t0r + t2r; im[4] = t0i + t2i;
    re[6] = t0r - t2r; im[6] = t0i - t2i;
    re[5] = t1r + t3i; im[5] = t1i - t3r;
    re[7] = t1r - t3i; im[7] = t1i + t3r;
  }
  { /* k1 = 2: b = R * (re[9], im[9]), c = -i * (re[10], im[10]), d = R * (re[11], im[11]) */
    const T ar = re[8], ai = im[8], cr = im[10], ci = -re[10];
    const T t0r = ar + cr, t0i = ai + ci, t1r = ar - cr, t1i = ai - ci;
    const T u2r = re[9] + re[11], u2i = im[9] + im[11], u3r = re[9] - re[11], u3i = im[9] - im[11];
    re[8] = bl_fma(R, u2r, t0r);   im[8] = bl_fma(R, u2i, t0i);
    re[10] = bl_fma(-R, u2r, t0r); im[10] = bl_fma(-R, u2i, t0i);
    re[9] = bl_fma(R, u3i, t1r);   im[9] = bl_fma(-R, u3r, t1i);
    re[11] = bl_fma(-R, u3i, t1r); im[11] = bl_fma(R, u3r, t1i);
  }
  { /* k1 = 3: c = R * (re[14], im[14]) */
    const T ar = re[12], ai = im[12], br = re[13], bi = im[13], cr = re[14], ci = im[14], dr = re[15], di = im[15];
    const T t0r = bl_fma(R, cr, ar), t0i = bl_fma(R, ci, ai), t1r = bl_fma(-R, cr, ar), t1i = bl_fma(-R, ci, ai);
    const T t2r = br + dr, t2i = bi + di, t3r = br - dr, t3i = bi - di;
    re[12] = t0r + t2r; im[12] = t0i + t2i;
    re[14] = t0r - t2r; im[14] = t0i - t2i;
    re[13] = t1r + t3i; im[13] = t1i - t3r;
    re[15] = t1r - t3i; im[15] = t1i + t3r;
  }
  /* element 4*k1 + k0 holds X[k1 + 4*k0] */
}

/* LDS footprint of one 16-lane transform, in complex elements */
#define BL_FFT_XCH_ELEMS (16 * 17) /* transpose buffer, row stride 17 */
#define BL_FFT_PAR_ELEMS (16 * 8)  /* partner half rows */

/*
 * Phase A (per lane n0): registers hold z[16*m1 + n0], m1 = 0..15.
 * Does pass 1, applies W256^(n0*k1) and writes row-major [k1][n0] (stride 17).
 * tw256: twiddles laid out per use, tw256[k1 * 16 + n0] = exp(-2 pi i n0 k1 / 256): the 16
 * lanes of a group read consecutive 16-byte entries (indexing by the exponent n0*k1 put
 * lanes l and l + 64/k1 on the same banks: up to 8-way LDS conflicts).
 */
template <typename T>
BL_HD void bl_fft512_phaseA(int n0, T (&re)[16], T (&im)[16], const bl_c2<T> *tw256,
                            bl_c2<T> *xch) {
  bl_fft16(re, im);
#pragma unroll
  for (int k1 = 0; k1 < 16; ++k1) {
    const int p = bl_pos16(k1);
    T r = re[p], i = im[p];
    if (k1 != 0) {
      bl_c2<T> w = tw256[k1 * 16 + n0];
      if (n0 != 0) bl_cmul(r, i, w.re, w.im);
    }
    bl_c2<T> v; v.re = r; v.im = i;
    xch[k1 * 17 + n0] = v;
  }
}

/*
 * Phase B (per lane k1, after a barrier): reads column -> pass 2 ->
 * registers hold Z[k1 + 16*k0] at position bl_pos16(k0); publishes the upper
 * half row (k0 = 8..15) for the partner lane.
 */
template <typename T>
BL_HD void bl_fft512_phaseB_load(int k1, T (&re)[16], T (&im)[16], const bl_c2<T> *xch) {
#pragma unroll
  for (int n0 = 0; n0 < 16; ++n0) {
    bl_c2<T> v = xch[k1 * 17 + n0];
    re[n0] = v.re; im[n0] = v.im;
  }
}
/* second half of phase B; with a barrier after the load half, `par` may alias `xch` */
template <typename T>
BL_HD void bl_fft512_phaseB_publish(int k1, T (&re)[16], T (&im)[16], bl_c2<T> *par) {
  bl_fft16(re, im);
#pragma unroll
  for (int k0 = 8; k0 < 16; ++k0) {
    bl_c2<T> v; v.re = re[bl_pos16(k0)]; v.im = im[bl_pos16(k0)];
    par[k1 * 8 + (k0 - 8)] = v;
  }
}
template <typename T>
BL_HD void bl_fft512_phaseB(int k1, T (&re)[16], T (&im)[16], const bl_c2<T> *xch,
                            bl_c2<T> *par) {
  bl_fft512_phaseB_load<T>(k1, re, im, xch);
  bl_fft512_phaseB_publish<T>(k1, re, im, par);
}

/*
 * Power of the real-input spectrum from a lane's own half row and its partner's.
 * re/im: Z[k1 + 16*k0] at position bl_pos16(k0); pr/pi[k0] (k0 = 0..7): Z[256 - k],
 * k = k1 + 16*k0, i.e. the partner lane's register 15-k0 (lane 0: its own 16-k0,
 * itself for k0 = 0).
 *   own[k0]  = |X_k|^2      , k = k1 + 16*k0      (k0 = 0..7  -> k in 0..127)
 *   mir[k0]  = |X_(256-k)|^2                       (-> 129..256)
 *   mid      = |X_128|^2 (meaningful in lane 0 only)
 * tw512: W512^k = exp(-2 pi i k/512), k = 0..255.
 */
/* one pair: Z_k = (zr, zi), partner Z_(256-k) = (pr, pi), w = W512^k.  QUARTER = false: the
 * caller fed the transform x / 2, which already carries the 1/4 (an exact scaling). */
template <typename T, bool QUARTER = true>
BL_HD void bl_fft512_power1(T zr, T zi, T pr, T pi, bl_c2<T> w, T &own, T &mir) {
  const T er = zr + pr, ei = zi - pi;
  const T orr = zi + pi, oi = pr - zr;
  const T q = oi * w.im, s = oi * w.re;
  const T tr = bl_fma(orr, w.re, -q);
  const T ti = bl_fma(orr, w.im, s);
  const T ar = er + tr, ai = ei + ti, br = er - tr, bi = ei - ti;
  own = bl_fma(ar, ar, ai * ai);
  mir = bl_fma(br, br, bi * bi);
  if (QUARTER) { own = (T)0.25 * own; mir = (T)0.25 * mir; }
}

template <typename T>
BL_HD void bl_fft512_power(int k1, const T (&re)[16], const T (&im)[16], const T (&pr)[8],
                           const T (&pi)[8], const bl_c2<T> *tw512, T (&own)[8], T (&mir)[8],
                           T &mid) {
#pragma unroll
  for (int k0 = 0; k0 < 8; ++k0)
    bl_fft512_power1<T>(re[bl_pos16(k0)], im[bl_pos16(k0)], pr[k0], pi[k0], tw512[k1 + 16 * k0],
                        own[k0], mir[k0]);
  const T mr = re[bl_pos16(8)], mi = im[bl_pos16(8)];
  mid = bl_fma(mr, mr, mi * mi);
}

/* index into a partner half-row buffer [lane][8] for lane k1, pair k0 (k0 = 0..7);
 * returns -1 when the partner is the lane's own Z[0] (k1 = 0, k0 = 0) */
BL_HD int bl_partner_slot(int k1, int k0) {
  if (k1 != 0) return ((16 - k1) & 15) * 8 + (7 - k0);
  return k0 == 0 ? -1 : (8 - k0);
}

/*
 * Phase C (per lane k1, after a barrier): partner fetch from a complex buffer + power.
 */
template <typename T>
BL_HD void bl_fft512_phaseC(int k1, const T (&re)[16], const T (&im)[16], const bl_c2<T> *tw512,
                            const bl_c2<T> *par, T (&own)[8], T (&mir)[8], T &mid) {
  T pr[8], pi[8];
#pragma unroll
  for (int k0 = 0; k0 < 8; ++k0) {
    const int sl = bl_partner_slot(k1, k0);
    if (sl < 0) { pr[k0] = re[bl_pos16(0)]; pi[k0] = im[bl_pos16(0)]; }
    else { const bl_c2<T> v = par[sl]; pr[k0] = v.re; pi[k0] = v.im; }
  }
  bl_fft512_power<T>(k1, re, im, pr, pi, tw512, own, mir, mid);
}

/* pass 1 in registers: 16-point DFT over m1 and the W256^(n0*k1) twiddles; the value
 * for k1 stays at position bl_pos16(k1) */
template <typename T>
BL_HD void bl_fft512_pass1(int n0, T (&re)[16], T (&im)[16], const bl_c2<T> *tw256) {
  bl_fft16(re, im);
#pragma unroll
  for (int k1 = 1; k1 < 16; ++k1) {
    const int p = bl_pos16(k1);
    const bl_c2<T> w = tw256[k1 * 16 + n0];
    T r = re[p], i = im[p];
    bl_cmul(r, i, w.re, w.im);
    if (n0 != 0) { re[p] = r; im[p] = i; }
  }
}

#endif /* BL_FFT_H_ */
