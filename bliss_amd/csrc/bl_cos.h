/*
 * bl_cos.h — bl_cosine_similarity of ONE pair at the cost of a handful of instructions.
 *
 * The reference (ref src/analyze.c:135-140):
 *     dot, na, nb   f32 sums of four f32 products each, left to right
 *     return (float)((double)dot / (sqrt((double)na) * sqrt((double)nb)));
 * i.e. three double roundings after the f32 part — the product of the roots, the quotient, the narrowing to
 * float.  hipcc expands the f64 divide into ~25 instructions and the two f64 roots into ~30 each, per output.
 *
 * What depends on one vector only is computed once per vector (bl_cos_prep): n, s = sqrt((double)n) and
 * r = 1 / s, both correctly rounded.  Per pair the fast path evaluates
 *     q' = (double)dot * (ra * rb)
 * which differs from the reference's double quotient q = RN(dot / RN(sa * sb)) by less than 6 ulp of q: six
 * roundings of relative size u = 2^-53 separate the two (ra, rb, ra * rb, dot * R, the reference's sa * sb, and
 * its quotient), i.e. 6 u relative, which is between 3 and 6 ulp of q depending on where q sits in its binade
 * (plus the case of q' and q on either side of a power of two, where the ulp halves: still below 12 of the smaller
 * ulp).  (float)q' equals (float)q whenever no float rounding boundary — a double whose low 29 mantissa bits are
 * 0x10000000 — lies within that distance of q', so the fast result is taken only if the low 29 bits of q' are further
 * than BL_COS_GUARD = 16 ulp from 0x10000000 and q' is a normal number of a magnitude where a float has its full 24 bits;
 * everything else (one pair in ~10^7, zero or non-finite norms, zero dot products) takes the plain expression.
 * Not taken on trust: bl_amd_selftest_cos() sweeps random and boundary-seeking (dot, na, nb) triples on the
 * device against the plain expression and reports the largest |q' - q| it saw, in ulp.
 */
#ifndef BL_COS_H_
#define BL_COS_H_

#include <hip/hip_runtime.h>

#define BL_COS_GUARD 16 /* ulp of the double quotient; the provable distance is < 6 (12 across a binade edge) */

struct bl_cos_vec { double s, r; float n; };

/* squared norm in the reference's order; its double root and the root's reciprocal */
__device__ __forceinline__ bl_cos_vec bl_cos_prep(const float4 v) {
  bl_cos_vec p;
  p.n = v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  p.s = sqrt((double)p.n);
  p.r = 1.0 / p.s;
  return p;
}

/* the plain expression, from the prepared roots (same three roundings as the reference) */
__device__ __forceinline__ float bl_cos_plain(float dot, const bl_cos_vec &a, const bl_cos_vec &b) {
  return (float)((double)dot / (a.s * b.s));
}

/* q' and whether (float)q' is guaranteed to be the reference's result */
__device__ __forceinline__ bool bl_cos_fast(float dot, double rab, float &out) {
  const double q = (double)dot * rab;
  const unsigned long long bits = (unsigned long long)__double_as_longlong(q);
  const unsigned lo = (unsigned)bits & 0x1FFFFFFFu;
  const unsigned e = (unsigned)(bits >> 52) & 0x7FFu;
  out = (float)q;
  /* not within BL_COS_GUARD of a float rounding boundary; 2^-100 <= |q'| <= 2^100 */
  return (lo - (0x10000000u - BL_COS_GUARD)) > 2u * BL_COS_GUARD && (e - 923u) <= 200u;
}

#endif /* BL_COS_H_ */
