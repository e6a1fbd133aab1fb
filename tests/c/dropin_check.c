/*
 * Drop-in check at the C level: a plain C99 caller that knows nothing but include/bliss.h,
 * linked against libbliss_amd.so the way a caller of the reference links libbliss.so.
 * Expected values: the goldens the reference pins for audio/song.flac
 * (ref tests/test_analyze.c:30-55, tests/test_decode.c:16-17), at the reference's own 1e-5.
 * Exercises the by-value struct ABI of bl_distance / bl_cosine_similarity, the caller-owned
 * uninitialised struct bl_song, the analyzers on a decoded song and the release path.
 * With a second argument, the reference's other fixture (audio/song_s32.flac: 48 kHz, 24 bit) is
 * analysed as ref tests/test_analyze.c:59-89 does: it goes through the rate converter.
 * usage: dropin_check <path to song.flac> [<path to song_s32.flac>]; exit status 0 = all passed.
 */
#include <math.h>
#include <stdio.h>
#include <string.h>

#include "bliss.h"

static int failures;

static void near(const char *what, double got, double want, double tol) {
  if (!(fabs(got - want) <= tol)) {
    printf("FAIL %-24s got %.9g want %.9g\n", what, got, want);
    ++failures;
  }
}
static void same_int(const char *what, long got, long want) {
  if (got != want) {
    printf("FAIL %-24s got %ld want %ld\n", what, got, want);
    ++failures;
  }
}
static void same_str(const char *what, const char *got, const char *want) {
  if (!got || strcmp(got, want) != 0) {
    printf("FAIL %-24s got \"%s\" want \"%s\"\n", what, got ? got : "(null)", want);
    ++failures;
  }
}

int main(int argc, char **argv) {
  if (argc < 2) return 2;
  struct bl_song song; /* deliberately not initialised, as the reference's callers do */
  const int rc = bl_analyze(argv[1], &song);
  same_int("bl_analyze return", rc, BL_CALM);
  same_int("calm_or_loud", song.calm_or_loud, BL_CALM);

  const struct { const char *name; double got, want; } f[] = {
      {"force", song.force, -20.777929},
      {"tempo", song.force_vector.tempo, -8.945454},
      {"amplitude", song.force_vector.amplitude, -10.641844},
      {"frequency", song.force_vector.frequency, -10.136086},
      {"attack", song.force_vector.attack, -15.560563},
  };
  for (unsigned i = 0; i < sizeof f / sizeof f[0]; ++i) near(f[i].name, f[i].got, f[i].want, 1e-5);

  same_int("channels", song.channels, 2);
  same_int("nSamples", song.nSamples, 488138);
  same_int("sample_rate", song.sample_rate, 22050);
  same_int("bitrate", song.bitrate, 233864);
  same_int("nb_bytes_per_sample", song.nb_bytes_per_sample, 2);
  same_int("duration", (long)song.duration, 11);
  same_str("artist", song.artist, "David TMX");
  same_str("title", song.title, "Renaissance");
  same_str("album", song.album, "Renaissance");
  same_str("tracknumber", song.tracknumber, "02");
  same_str("genre", song.genre, "Pop");

  /* the analyzers one by one on the decoded song give the fields bl_analyze assembled */
  struct envelope_result_s env;
  bl_envelope_sort(&song, &env);
  near("bl_envelope_sort tempo", env.tempo, song.force_vector.tempo, 0.0);
  near("bl_envelope_sort attack", env.attack, song.force_vector.attack, 0.0);
  near("bl_amplitude_sort", bl_amplitude_sort(&song), song.force_vector.amplitude, 0.0);
  near("bl_frequency_sort", bl_frequency_sort(&song), song.force_vector.frequency, 0.0);

  /* helpers on the caller-visible sample_array */
  int16_t *pcm = (int16_t *)song.sample_array;
  const int mean = bl_mean(pcm, song.nSamples);
  long long acc = 0;
  for (int i = 0; i < song.nSamples; ++i) acc += pcm[i];
  same_int("bl_mean", mean, (long)(acc / song.nSamples)); /* no int32 wrap on this file */
  if (bl_variance(pcm, song.nSamples, mean) <= 0) { puts("FAIL bl_variance <= 0"); ++failures; }

  /* structs by value */
  struct force_vector_s a = song.force_vector, b = song.force_vector;
  near("bl_distance(v, v)", bl_distance(a, b), 0.0, 0.0);
  near("bl_cosine(v, v)", bl_cosine_similarity(a, b), 1.0, 1e-6);
  b.tempo += 3.0f;
  b.attack -= 4.0f;
  near("bl_distance 3-4-5", bl_distance(a, b), 5.0, 1e-6);
  {
    const double dot = (double)a.tempo * b.tempo + (double)a.amplitude * b.amplitude +
                       (double)a.frequency * b.frequency + (double)a.attack * b.attack;
    const double na = sqrt((double)a.tempo * a.tempo + (double)a.amplitude * a.amplitude +
                           (double)a.frequency * a.frequency + (double)a.attack * a.attack);
    const double nb = sqrt((double)b.tempo * b.tempo + (double)b.amplitude * b.amplitude +
                           (double)b.frequency * b.frequency + (double)b.attack * b.attack);
    near("bl_cosine_similarity", bl_cosine_similarity(a, b), dot / (na * nb), 1e-6);
  }

  /* two-file forms, caller-owned structs filled as a side effect */
  struct bl_song s1, s2;
  near("bl_distance_file(f, f)", bl_distance_file(argv[1], argv[1], &s1, &s2), 0.0, 0.0);
  near("s1.force", s1.force, song.force, 0.0);
  bl_free_song(&s1);
  bl_free_song(&s2);
  same_int("bl_analyze(missing)", bl_analyze("/nonexistent/file.flac", &s1), BL_UNEXPECTED);

  if (argc > 2) { /* ref tests/test_analyze.c:59-89 */
    struct bl_song w;
    same_int("s32: bl_analyze return", bl_analyze(argv[2], &w), BL_CALM);
    near("s32: force", w.force, -20.821571, 1e-5);
    near("s32: tempo", w.force_vector.tempo, -8.218182, 1e-5);
    near("s32: amplitude", w.force_vector.amplitude, -10.641695, 1e-5);
    near("s32: frequency", w.force_vector.frequency, -10.179875, 1e-5);
    near("s32: attack", w.force_vector.attack, -15.561186, 1e-5);
    same_int("s32: channels", w.channels, 2);
    same_int("s32: nSamples", w.nSamples, 488140);
    same_int("s32: sample_rate", w.sample_rate, 22050);
    same_int("s32: nb_bytes_per_sample", w.nb_bytes_per_sample, 2);
    same_int("s32: duration", (long)w.duration, 11);
    same_str("s32: artist", w.artist, "David TMX");
    same_str("s32: genre", w.genre, "Pop");
    bl_free_song(&w);
  }

  if (bl_version() <= 0) { puts("FAIL bl_version"); ++failures; }
  bl_free_song(&song);
  if (song.sample_array != NULL || song.artist != NULL) { puts("FAIL bl_free_song leaves pointers"); ++failures; }
  printf("%s (%d failures)\n", failures ? "FAILED" : "OK", failures);
  return failures ? 1 : 0;
}
