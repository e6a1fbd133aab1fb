/*
 * bl_resample.h — internal: the rate converter behind bl_audio_decode() (bl_resample.c) and the
 * plan (filter bank + stepping) that the device form (k_resample, bl_kernels.hip) shares with it.
 */
#ifndef BL_RESAMPLE_H_
#define BL_RESAMPLE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
  int phase_count, taps, alloc; /* rows of `alloc` (taps rounded up to 16, zero padded) */
  int src_incr, dst_incr;       /* one output advances dst_incr / src_incr phases of the input */
  int dst_incr_div, dst_incr_mod;
  int is_float;
  float *fbank;   /* (phase_count + 1) rows: wider-than-16-bit sources */
  int16_t *ibank; /* Q15: sources of at most 16 bits */
} bl_rs_plan;

/* stepping and filter length only (no bank): enough for bl_rs_out_frames().  0 / -1. */
int bl_rs_plan_geometry(bl_rs_plan *f, int out_rate, int in_rate);
/* 0 / -1.  want_float selects which bank is built. */
int bl_rs_plan_build(bl_rs_plan *f, int out_rate, int in_rate, int want_float);
void bl_rs_plan_free(bl_rs_plan *f);
/* frames of output for `frames` of input (0: too short); *refl = samples reflected at the end */
size_t bl_rs_out_frames(const bl_rs_plan *f, size_t frames, size_t *refl);

/* `in`: `frames` interleaved frames of `channels` (1 or 2) channels at `in_rate` Hz, int16, or
 * (in_is_s32 = 1) int32 left-justified, or (in_is_s32 = 2) IEEE float with full scale +-1.
 * *out: malloc'd interleaved stereo s16 at `out_rate` Hz.
 * BL_OK / BL_UNEXPECTED. */
int bl_resample_to_stereo_s16(const void *in, int in_is_s32, size_t frames, int channels, int in_rate,
                              int out_rate, int16_t **out, size_t *out_frames);

#ifdef __cplusplus
}
#endif
#endif
