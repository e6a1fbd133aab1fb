/*
 * bl_device.h — internal interface between the C host layer (bl_api.c) and the
 * HIP translation unit (bl_kernels.hip).  Not installed; the public C-ABI is
 * include/bliss.h + include/bliss_amd.h.
 */
#ifndef BL_DEVICE_H_
#define BL_DEVICE_H_

#include <stddef.h>
#include <stdint.h>
#include "bliss_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* central amplitude histogram kept on the device: 16-bit values s with
 * |s| <= 2048 are the only ones that can reach the integral window
 * [INT16_MAX-1000, INT16_MAX+1000] through 301 passes of a +-3 stencil
 * (ref src/amplitude_sort.c:5-10,41-59: 301*3 = 903 bins of reach). */
#define BL_HIST_BINS 4096
#define BL_HIST_LO (32768 - 2048) /* reference bin index of local bin 0 */

/* device-side helpers used by the reference-API shims in bl_api.c */
int bld_ready(void); /* BL_OK when a device is initialised, else tries device 0 */
int bld_mean_variance_host(const int16_t *h_pcm, int n, int have_mean, int mean_in,
                           int *mean_out, int *variance_out);
/* single analyzers on one host-resident song (what = 1 amplitude, 2 frequency,
 * 4 envelope, 7 all); fills *res */
int bld_analyze_one_host(const int16_t *h_pcm, int n, int channels, uint64_t duration, int what,
                         bl_amd_song_result *res);

#ifdef __cplusplus
}
#endif
#endif
