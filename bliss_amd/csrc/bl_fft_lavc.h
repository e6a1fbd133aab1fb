/*
 * bl_fft_lavc.h — the 512-point f32 real DFT of the frequency analysis in libavcodec's operation order,
 * spread over 16 lanes x 16 registers.
 *
 * ref src/frequency_sort.c:65,83 calls av_rdft_init(9, DFT_R2C) / av_rdft_calc.  libavcodec is absent from
 * /root/reference; its generic C path (fft_template.c + rdft.c, FFmpeg 0.7 .. 4.4) is restated in
 * oracle/orc_fft_lavc.c, and under that operation order — and under none of the other f32 DFTs tried — the oracle
 * prints every golden value of ref tests/test_analyze.c:30-35,62-68 to the last digit.  An f32 DFT is visible in
 * `frequency` at the 1e-6 level, so the kernel evaluates the SAME expression DAG: same operands, same operation
 * per node, every product and sum rounded to float, no fused operation anywhere (compile with -ffp-contract=off).
 * Rewrites used, all bit-exact in IEEE arithmetic: a - (-b) = a + b, a + (-b) = a - b, (-a) * b = -(a * b),
 * a + b = b + a.  tests/host/test_fft_lavc_host.cpp runs this lane code on the CPU against orc_fft_lavc.c bit for
 * bit.
 *
 * The transform.  z[m] = x[2m] + i x[2m+1] (m < 256), permuted to split-radix order (position p holds
 * z[lv_index(p)]), then the in-place conjugate-pair split-radix recursion
 *     fft(n) @ b  =  fft(n/2) @ b,  fft(n/4) @ b + n/2,  fft(n/4) @ b + 3n/4,  pass(n) @ b
 * with fft4 / fft8 / fft16 as written-out leaves; pass(n) @ b is n/4 butterflies on positions
 * (b + k, b + k + n/4, b + k + n/2, b + k + 3n/4) with twiddle cos_n[k], cos_n[n/4 - k].  Then rdft.c's post-pass
 * on the pairs (Z_i, Z_(256-i)).
 *
 * Mapping onto a 16-lane group (one transform per group; T = a pair of frames in k_freq_scan):
 *   layout B  lane L, register r  = position 16 L + r.  Every 16-block is a leaf: an fft16 (lanes of type T16) or
 *             two fft8 (type T8: L = 1, 5, 7, 9, 13) — all in registers.  The input is gathered in this order
 *             straight from the frame (lv_index: 16 lanes read one 8-byte element each out of a 16-element
 *             neighbourhood).
 *   one 16 x 16 transpose through LDS
 *   layout A  lane l, register j  = position 16 j + l.  pass(64) @ 0, 128, 192, pass(128) @ 0, pass(256) @ 0 are
 *             in-lane (the four operands of a butterfly are 16, 32, 64 positions apart: registers j + {1, 2, 4} m).
 *             pass(32) @ 0, 64, 96, 128, 192 pairs lane k (a0, a2 in registers R, R + 1) with lane k + 8 (a1, a3):
 *             each multiplies its own operand by the twiddle, the products are exchanged (DPP row_ror:8 on the
 *             device) and each finishes the two outputs it owns.  After the passes position = frequency index, and
 *             the post-pass partner Z_(256-i) of i = 16 j + l is register 15 - j of lane 16 - l (lane 0: its own
 *             register 16 - j), as in bl_fft.h.
 * Lane 0's butterflies are the recursion's TRANSFORM_ZERO (k = 0, no multiplication); here they run through the
 * general code with the twiddle (1, 0): x * 1 - y * 0 = x exactly, except that an exact zero may come out with the
 * other sign — which no later operation can turn into a different magnitude and the power re^2 + im^2 cannot see.
 */
#ifndef BL_FFT_LAVC_H_
#define BL_FFT_LAVC_H_

#include "bl_fft.h"

/* ---- structure ------------------------------------------------------------ */

/* lanes whose 16-block is two fft8 (positions 16, 80, 112, 144, 208 of the recursion) */
BL_HD constexpr bool lv_lane_is_t16(int L) { return !(L == 1 || L == 5 || L == 7 || L == 9 || L == 13); }

/* lv_index(16 L + r) = (lv_base(L) + K[r]) mod 256: which z[] a position holds */
BL_HD constexpr int lv_base(int L) {
  constexpr int B[16] = {0, 8, 4, 252, 2, 10, 254, 6, 1, 9, 5, 253, 255, 7, 3, 251};
  return B[L & 15];
}
BL_HD constexpr int lv_k_lo(int r) { /* registers 0..7, every lane */
  constexpr int K[8] = {0, 128, 64, 192, 32, 160, 224, 96};
  return K[r & 7];
}
BL_HD constexpr int lv_k_hi(bool t16, int r) { /* registers 8..15 */
  constexpr int K16[8] = {16, 144, 80, 208, 240, 112, 48, 176};
  constexpr int K8[8] = {240, 112, 48, 176, 16, 144, 208, 80};
  return t16 ? K16[r & 7] : K8[r & 7];
}
BL_HD constexpr int lv_index(int L, int r) {
  return (lv_base(L) + (r < 8 ? lv_k_lo(r) : lv_k_hi(lv_lane_is_t16(L), r - 8))) & 255;
}

/* the same for the kernel, which cannot index a table by its lane: the T8 lanes as a bit mask, lv_base() as a byte per
 * lane in two 64-bit words (lanes 0..7, 8..15) */
BL_HD constexpr unsigned lv_t8_lane_mask() {
  unsigned m = 0;
  for (int L = 0; L < 16; ++L) m |= lv_lane_is_t16(L) ? 0u : 1u << L;
  return m;
}
BL_HD constexpr unsigned long long lv_bases_packed(int half) {
  unsigned long long v = 0;
  for (int i = 0; i < 8; ++i) v |= (unsigned long long)lv_base(8 * half + i) << (8 * i);
  return v;
}

/* per-lane twiddle slots of layout A (table lv_tw[slot * 16 + lane], filled by lv_fill_tables) */
enum {
  LV_TW_P32 = 0,  /* pass(32):  k = lane & 7 */
  LV_TW_P64 = 1,  /* pass(64):  k = lane */
  LV_TW_P128 = 2, /* pass(128): k = lane + 16 q, q = 0, 1 */
  LV_TW_P256 = 4, /* pass(256): k = lane + 16 q, q = 0..3 */
  LV_TW_POST = 8, /* post-pass: i = lane + 16 j, j = 0..7: (tcos[i], tsin[i]) */
  LV_TW_SLOTS = 16
};
/* leaf constants: lv_leafc[0] = sqrthalf, [1] = cos_16[1], [2] = cos_16[3] */

#include <math.h>
/* Host only.  The tables, exactly as libavcodec builds them: cos_m[i] = (float)cos(i * 2 pi / m) for i <= m / 4, mirrored
 * cos_m[m / 2 - i] = cos_m[i]; a pass reads (cos_n[k], cos_n[n / 4 - k]), the post-pass (cos_512[i],
 * cos_512[128 + i]).  k = 0 is TRANSFORM_ZERO: (1, 0). */
static inline float lv_cos_tab(int m, int i) {
  const double freq = 2 * 3.14159265358979323846 / m;
  if (i > m / 4) i = m / 2 - i;
  return (float)cos(i * freq);
}
static inline void lv_fill_tables(float (*tw)[2] /* [LV_TW_SLOTS * 16] */, float leafc[4]) {
  for (int l = 0; l < 16; ++l) {
    const int n_of[4] = {32, 64, 128, 256}, slot_of[4] = {LV_TW_P32, LV_TW_P64, LV_TW_P128, LV_TW_P256};
    for (int p = 0; p < 4; ++p) {
      const int n = n_of[p], reps = p == 0 ? 1 : n / 64;
      for (int q = 0; q < reps; ++q) {
        const int k = p == 0 ? (l & 7) : l + 16 * q;
        float *w = tw[(slot_of[p] + q) * 16 + l];
        if (k == 0) { w[0] = 1.0f; w[1] = 0.0f; }
        else { w[0] = lv_cos_tab(n, k); w[1] = lv_cos_tab(n, n / 4 - k); }
      }
    }
    for (int j = 0; j < 8; ++j) {
      const int i = l + 16 * j;
      tw[(LV_TW_POST + j) * 16 + l][0] = lv_cos_tab(512, i);
      tw[(LV_TW_POST + j) * 16 + l][1] = lv_cos_tab(512, 128 + i);
    }
  }
  leafc[0] = (float)0.70710678118654752440; /* sqrthalf = (float)M_SQRT1_2 */
  leafc[1] = lv_cos_tab(16, 1);
  leafc[2] = lv_cos_tab(16, 3);
  leafc[3] = 0.0f;
}

/* ---- butterflies (fft_template.c: BF, BUTTERFLIES, TRANSFORM) --------------- */

/* BUTTERFLIES(a0, a1, a2, a3) on (t1, t2, t5, t6) */
template <typename T>
BL_HD void lv_butterflies(T &a0r, T &a0i, T &a1r, T &a1i, T &a2r, T &a2i, T &a3r, T &a3i, T t1, T t2, T t5, T t6) {
  const T t3 = t5 - t1, s5 = t5 + t1;
  a2r = a0r - s5; a0r = a0r + s5;
  a3i = a1i - t3; a1i = a1i + t3;
  const T t4 = t2 - t6, s6 = t2 + t6;
  a3r = a1r - t4; a1r = a1r + t4;
  a2i = a0i - s6; a0i = a0i + s6;
}

/* TRANSFORM(a0, a1, a2, a3, wre, wim): CMUL(t1, t2, a2, (wre, -wim)); CMUL(t5, t6, a3, (wre, wim)) */
template <typename T>
BL_HD void lv_transform(T &a0r, T &a0i, T &a1r, T &a1i, T &a2r, T &a2i, T &a3r, T &a3i, T wre, T wim) {
  const T t1 = a2r * wre + a2i * wim; /* a2r * wre - a2i * (-wim) */
  const T t2 = a2i * wre - a2r * wim; /* a2r * (-wim) + a2i * wre */
  const T t5 = a3r * wre - a3i * wim;
  const T t6 = a3r * wim + a3i * wre;
  lv_butterflies(a0r, a0i, a1r, a1i, a2r, a2i, a3r, a3i, t1, t2, t5, t6);
}

template <typename T>
BL_HD void lv_transform_zero(T &a0r, T &a0i, T &a1r, T &a1i, T &a2r, T &a2i, T &a3r, T &a3i) {
  const T t1 = a2r, t2 = a2i, t5 = a3r, t6 = a3i;
  lv_butterflies(a0r, a0i, a1r, a1i, a2r, a2i, a3r, a3i, t1, t2, t5, t6);
}

/* fft4 on four consecutive positions */
template <typename T> BL_HD void lv_fft4(T *zr, T *zi) {
  const T t3 = zr[0] - zr[1], t1 = zr[0] + zr[1];
  const T t8 = zr[3] - zr[2], t6 = zr[3] + zr[2];
  zr[2] = t1 - t6; zr[0] = t1 + t6;
  const T t4 = zi[0] - zi[1], t2 = zi[0] + zi[1];
  const T t7 = zi[2] - zi[3], t5 = zi[2] + zi[3];
  zi[3] = t4 - t8; zi[1] = t4 + t8;
  zr[3] = t3 - t7; zr[1] = t3 + t7;
  zi[2] = t2 - t5; zi[0] = t2 + t5;
}

/* fft8 on eight consecutive positions */
template <typename T> BL_HD void lv_fft8(T *zr, T *zi, T sqrthalf) {
  lv_fft4(zr, zi);
  const T t1 = zr[4] + zr[5]; zr[5] = zr[4] - zr[5]; /* BF(t1, z[5].re, z[4].re, -z[5].re) */
  const T t2 = zi[4] + zi[5]; zi[5] = zi[4] - zi[5];
  const T t5 = zr[6] + zr[7]; zr[7] = zr[6] - zr[7];
  const T t6 = zi[6] + zi[7]; zi[7] = zi[6] - zi[7];
  lv_butterflies(zr[0], zi[0], zr[2], zi[2], zr[4], zi[4], zr[6], zi[6], t1, t2, t5, t6);
  lv_transform(zr[1], zi[1], zr[3], zi[3], zr[5], zi[5], zr[7], zi[7], sqrthalf, sqrthalf);
}

/* ---- layout B: the leaves of a lane ------------------------------------------ */

/* The kernel gathers every lane's 16 elements in ONE order — immediate load offsets, no per-lane table —, the order
 * of the T16 lanes: register r holds z[(lv_base(L) + K[r]) mod 256], K = lv_k_lo | lv_k_hi(true, .).  For a T16 lane
 * that is position 16 L + r.  A T8 lane's second fft8 wants its inputs in another order (lv_k_hi(false, .)): there
 * position 16 L + 8 + t sits in register 8 + LV_S8[t]. */
BL_HD constexpr int lv_s8(int t) {
  constexpr int S[8] = {4, 5, 6, 7, 0, 1, 3, 2};
  return S[t & 7];
}
BL_HD constexpr int lv_gather_index(int L, int r) { /* what register r of lane L is loaded with */
  return (lv_base(L) + (r < 8 ? lv_k_lo(r) : lv_k_hi(true, r - 8))) & 255;
}

/* in: registers in gather order (above); out: register r = position 16 L + r, transformed.  t16 lanes: fft16; the
 * others: fft8, fft8.  Both kinds start with fft8 on positions 0..7 and an fft4 on registers 12..15 (positions
 * 12..15 of a T16 lane, 8..11 of a T8 lane). */
template <typename T> BL_HD void lv_leaves(bool t16, T (&re)[16], T (&im)[16], T sqrthalf, T c1, T c3) {
  lv_fft8(re, im, sqrthalf);
  lv_fft4(re + 12, im + 12);
  if (t16) {
    lv_fft4(re + 8, im + 8);
    lv_transform_zero(re[0], im[0], re[4], im[4], re[8], im[8], re[12], im[12]);
    lv_transform(re[2], im[2], re[6], im[6], re[10], im[10], re[14], im[14], sqrthalf, sqrthalf);
    lv_transform(re[1], im[1], re[5], im[5], re[9], im[9], re[13], im[13], c1, c3);
    lv_transform(re[3], im[3], re[7], im[7], re[11], im[11], re[15], im[15], c3, c1);
  } else { /* the rest of fft8 on positions 8..15 = registers 12, 13, 14, 15, 8, 9, 11, 10 */
    T z0r = re[12], z0i = im[12], z1r = re[13], z1i = im[13], z2r = re[14], z2i = im[14], z3r = re[15], z3i = im[15];
    T z4r = re[8], z4i = im[8], z5r = re[9], z5i = im[9], z6r = re[11], z6i = im[11], z7r = re[10], z7i = im[10];
    const T t1 = z4r + z5r; z5r = z4r - z5r;
    const T t2 = z4i + z5i; z5i = z4i - z5i;
    const T t5 = z6r + z7r; z7r = z6r - z7r;
    const T t6 = z6i + z7i; z7i = z6i - z7i;
    lv_butterflies(z0r, z0i, z2r, z2i, z4r, z4i, z6r, z6i, t1, t2, t5, t6);
    lv_transform(z1r, z1i, z3r, z3i, z5r, z5i, z7r, z7i, sqrthalf, sqrthalf);
    re[8] = z0r; im[8] = z0i; re[9] = z1r; im[9] = z1i; re[10] = z2r; im[10] = z2i; re[11] = z3r; im[11] = z3i;
    re[12] = z4r; im[12] = z4i; re[13] = z5r; im[13] = z5i; re[14] = z6r; im[14] = z6i; re[15] = z7r; im[15] = z7i;
  }
}

/* ---- layout A: the passes ------------------------------------------------------ */

/* pass(32) @ 16 R, first half: this lane's product.  lo lanes (l < 8) hold a0 (register R) and a2 (R + 1), hi lanes
 * a1 and a3; (wre, ws) = (cos_32[k], lo ? -cos_32[8 - k] : cos_32[8 - k]), k = l & 7:
 *   lo: (tA, tB) = (t1, t2) = a2 * (wre, -wim)       hi: (tA, tB) = (t5, t6) = a3 * (wre, wim) */
template <typename T> BL_HD void lv_pass32_mul(T xr, T xi, T wre, T ws, T &tA, T &tB) {
  tA = xr * wre - xi * ws;
  tB = xr * ws + xi * wre;
}
/* second half, with the partner lane's products (pA, pB):
 *   lo: s5 = t5 + t1 = pA + tA, s6 = t2 + t6 = tB + pB;  a2 = a0 - (s5, s6), a0 = a0 + (s5, s6)
 *   hi: t3 = t5 - t1 = tA - pA, t4 = t2 - t6 = pB - tB;  a3 = (a1.re - t4, a1.im - t3), a1 = (a1.re + t4, a1.im + t3)
 * `lo` selects per lane (a v_cndmask on the device): ur goes to the real parts, ui to the imaginary ones. */
template <typename T, typename SEL>
BL_HD void lv_pass32_fin(T &r0, T &i0, T &r1, T &i1, T tA, T tB, T pA, T pB, SEL lo_sel) {
  const T sumA = pA + tA, difA = tA - pA, sumB = tB + pB, difB = pB - tB;
  const T ur = lo_sel(sumA, difB), ui = lo_sel(sumB, difA);
  r1 = r0 - ur; r0 = r0 + ur;
  i1 = i0 - ui; i0 = i0 + ui;
}

/* one in-lane butterfly of pass(64 / 128 / 256): registers j0, j0 + s, j0 + 2 s, j0 + 3 s */
template <typename T, int J0, int S> BL_HD void lv_pass_inlane(T (&re)[16], T (&im)[16], T wre, T wim) {
  lv_transform(re[J0], im[J0], re[J0 + S], im[J0 + S], re[J0 + 2 * S], im[J0 + 2 * S], re[J0 + 3 * S], im[J0 + 3 * S],
               wre, wim);
}

/* ---- rdft.c post-pass + the power of ref src/frequency_sort.c:88-93 ------------- */

/* one pair: Z_i = (zr, zi), Z_(256-i) = (pr, pi), (tcos, tsin) of i.  own = |X_i|^2, mir = |X_(256-i)|^2 with
 * re * re + im * im as the reference writes it (two products, one sum, all rounded). */
template <typename T> BL_HD void lv_post_power(T zr, T zi, T pr, T pi, T tcos, T tsin, T half, T &own, T &mir) {
  const T evr = half * (zr + pr);
  const T odi = half * (pr - zr);
  const T evi = half * (zi - pi);
  const T odr = half * (zi + pi);
  const T osr = odr * tcos + odi * tsin; /* DFT_R2C: the "negative sin" form */
  const T osi = odi * tcos - odr * tsin;
  const T xr = evr + osr, xi = evi + osi;   /* data[2 i], data[2 i + 1] */
  const T yr = evr - osr, yi = osi - evi;   /* data[512 - 2 i], data[512 - 2 i + 1] */
  own = xr * xr + xi * xi;
  mir = yr * yr + yi * yi;
}
/* i = 128: data[256] = Z_128.re, data[257] = -Z_128.im */
template <typename T> BL_HD T lv_mid_power(T zr, T zi) { return zr * zr + zi * zi; }

#endif /* BL_FFT_LAVC_H_ */
