/*
 * orc_synth.c — TEST INFRASTRUCTURE (see bliss_oracle.h).
 *
 * Integer-only, random-access synthetic PCM: sample i of song `seed` is a pure
 * function of (seed, rate, channels, i), so the container, the GPU box's host
 * and the device generator in bliss_amd/csrc (bl_synth_kernel) all produce the
 * same bytes without libm.  The signal is a beat-gated mix of two parabolic
 * "sines" plus +-800 LSB of hashed noise; |sample| < 9400 (no clipping), mean
 * ~ 0 (bl_mean's int32 sum cannot wrap), first/last sample non-silent with
 * overwhelming probability.  This is the synthetic distribution of
 * SURVEY.md §8(d); the reference has no generator of its own.
 */
#include "bliss_oracle.h"

static inline uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU;
  x ^= x >> 15; x *= 0x846ca68bU;
  x ^= x >> 16;
  return x;
}

/* parabolic sine of a 16-bit phase, output in [-32768, 32768] */
static inline int32_t psin(uint32_t phase16) {
  int32_t x = (int32_t)(phase16 & 65535u) - 32768;
  int32_t ax = x < 0 ? -x : x;
  return -((x * (32768 - ax)) / 8192);
}

int16_t orc_synth_sample(uint32_t seed, uint32_t rate, uint32_t channels, uint32_t i) {
  uint32_t f = i / channels, c = i - f * channels;
  uint32_t h = mix32(seed * 0x9E3779B9u + 1u);
  uint32_t f1 = 110u + (h & 255u);
  uint32_t f2 = 2000u + ((h >> 8) & 2047u);
  uint32_t bpm = 90u + ((h >> 20) & 63u);
  uint32_t a1 = 3000u + ((h >> 26) & 31u) * 100u; /* 3000..6100 */
  uint32_t period = rate * 60u / bpm;
  uint32_t pos = f % period;
  int32_t env = 32768 - (int32_t)(((uint64_t)pos * 29491u) / period);
  uint32_t ph1 = (uint32_t)((((uint64_t)f * f1) << 16) / rate);
  uint32_t ph2 = (uint32_t)((((uint64_t)f * f2) << 16) / rate) + c * 16384u;
  int32_t tone = (psin(ph1) * (int32_t)a1 + psin(ph2) * 2500) / 32768;
  int32_t sig = (tone * env) / 32768;
  int32_t noise = (int32_t)(mix32(seed ^ mix32(i + 0x1234567u)) % 1601u) - 800;
  return (int16_t)(sig + noise);
}

void orc_synth_fill(int16_t *pcm, uint32_t n, uint32_t seed, uint32_t rate,
                    uint32_t channels) {
  for (uint32_t i = 0; i < n; ++i) pcm[i] = orc_synth_sample(seed, rate, channels, i);
}
