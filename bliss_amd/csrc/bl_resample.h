/*
 * bl_resample.h — internal: the rate converter behind bl_audio_decode() (bl_resample.c).
 */
#ifndef BL_RESAMPLE_H_
#define BL_RESAMPLE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* `in`: `frames` interleaved frames of `channels` (1 or 2) channels at `in_rate` Hz, int16, or
 * (in_is_s32) int32 left-justified.  *out: malloc'd interleaved stereo s16 at `out_rate` Hz.
 * BL_OK / BL_UNEXPECTED. */
int bl_resample_to_stereo_s16(const void *in, int in_is_s32, size_t frames, int channels, int in_rate,
                              int out_rate, int16_t **out, size_t *out_frames);

#ifdef __cplusplus
}
#endif
#endif
