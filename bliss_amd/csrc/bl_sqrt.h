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

/* s in [BL_SQRT_FAST_LO, BL_SQRT_FAST_HI] */
__device__ __forceinline__ float bl_sqrt_rn_fast(float s) {
  const float y = __builtin_amdgcn_sqrtf(s);
  const float h = 0.5f * __builtin_amdgcn_rsqf(s);
  const float r = __builtin_fmaf(-y, y, s);
  return __builtin_fmaf(r, h, y);
}

__device__ __forceinline__ bool bl_sqrt_fast_ok(float s) {
  return s >= BL_SQRT_FAST_LO && s <= BL_SQRT_FAST_HI; /* false for NaN */
}

#endif /* BL_SQRT_H_ */
