/* A caller in the style of the reference's examples: it includes <bliss.h> only and uses printf
 * format macros, abs and fabs, which the reference header provides through libavformat's includes.
 * Compiled with -fsyntax-only by tests/test_abi.py: include/bliss.h must keep such sources building. */
#include <bliss.h>

int report(const char *path) {
  struct bl_song song;
  if (bl_analyze(path, &song) == BL_UNEXPECTED) return EXIT_FAILURE;
  printf("duration %" PRIu64 " s, last sample %" PRId16 "\n", song.duration,
         ((int16_t *)song.sample_array)[song.nSamples - 1]);
  const float gap = (float)fabs((double)((int16_t *)song.sample_array)[0] / (double)INT16_MAX);
  printf("first sample %f of full scale, |tempo| %d\n", gap, abs((int)song.force_vector.tempo));
  bl_free_song(&song);
  return EXIT_SUCCESS;
}
