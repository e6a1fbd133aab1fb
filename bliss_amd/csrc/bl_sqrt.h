/*
 * bl_sqrt.h — correctly rounded f32 square root for bl_distance (ref src/analyze.c:96-100: the
 * reference's sqrt of an f32 sum, which gcc compiles to sqrtss — IEEE correctly rounded).
 *
 * hipcc's own correctly rounded sqrtf (-fhip-fp32-correctly-rounded-divide-sqrt) costs ~20 VALU
 * instructions per value (input scaling for denormals, v_sqrt_f32, two next-up / next-down
 * candidates with compares and selects) and made k_pairwise VALU-bound.  For a normal, not tiny,
 * finite argument the same result takes five:
 *     y = v_sqrt_f32(s)               within 1 ulp
 *     h = 0.5 * v_rsq_f32(s)          ~ 1 / (2 y)
 *     r = fma(-y, y, s)               the residual s - y^2, exact in f32 when y is within 1 ulp
 *     y' = fma(r, h, y)               one rounding: the correctly rounded root (Markstein)
 * Not taken on trust: bl_amd_selftest_sqrt() runs bl_sqrt_rn_fast over every f32 bit pattern of
 * its domain on the GPU and compares with (float)sqrt((double)s), which is the correctly rounded
 * f32 root (53 >= 2 * 24 + 2 bits make the double rounding harmless); tests/test_gpu_parity.py
 * asserts zero mismatches.  Outside the domain (zero, denormal or tiny, infinite, NaN) the caller
 * takes the compiler's sqrtf.
 */
#ifndef BL_SQRT_H_
#define BL_SQRT_H_

#include <hip/hip_runtime.h>

#define BL_SQRT_FAST_LO 0x1p-100f /* below: the residual is no longer exact (and v_sqrt flushes denormals) */
#define BL_SQRT_FAST_HI 0x1p+126f

/* s in [BL_SQRT_FAST_LO, BL_SQRT_FAST_HI].  V = 1: v_sqrt + v_rsq + Markstein's final step;
 * V = 2: v_rsq only, one coupled Newton step on (y, h) and the final step (Goldschmidt / IA-64
 * form) — one transcendental instead of two, three more fmas. */
template <int V> __device__ __forceinline__ float bl_sqrt_rn_fast(float s) {
  if (V == 1) {
    const float y = __builtin_amdgcn_sqrtf(s);
    const float h = 0.5f * __builtin_amdgcn_rsqf(s);
    const float r = __builtin_fmaf(-y, y, s);
    return __builtin_fmaf(r, h, y);
  }
  const float q = __builtin_amdgcn_rsqf(s);
  const float y0 = s * q, h0 = 0.5f * q;
  const float e = __builtin_fmaf(-y0, h0, 0.5f);
  const float y1 = __builtin_fmaf(y0, e, y0), h1 = __builtin_fmaf(h0, e, h0);
  const float d = __builtin_fmaf(-y1, y1, s);
  return __builtin_fmaf(d, h1, y1);
}

/* in the domain <=> (bits - lo) <= (hi - lo) as unsigned integers: non-negative floats order like
 * their bit patterns; zero, negative, infinite and NaN patterns all fall outside */
__device__ __forceinline__ unsigned bl_sqrt_fast_key(float s) {
  return __float_as_uint(s) - __float_as_uint(BL_SQRT_FAST_LO);
}
#define BL_SQRT_FAST_SPAN (0x7E800000u - 0x0D800000u) /* bits(2^126) - bits(2^-100) */
__device__ __forceinline__ bool bl_sqrt_fast_ok(float s) { return bl_sqrt_fast_key(s) <= BL_SQRT_FAST_SPAN; }

#endif /* BL_SQRT_H_ */
