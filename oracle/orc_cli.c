/*
 * orc_cli.c — TEST INFRASTRUCTURE.  Command-line driver of the CPU oracle.
 *   orc_cli raw  <file.s16le> <channels> <duration>     analyse raw PCM
 *   orc_cli synth <seed> <rate> <channels> <seconds>     analyse a synthetic song
 *   orc_cli synthn <seed> <rate> <channels> <n_samples> <duration>   the same, any length (n interleaved samples)
 *   orc_cli time <seed> <rate> <channels> <seconds> <count>   time `count` songs
 * Prints one JSON object per song.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "bliss_oracle.h"

static void print_result(const orc_result *r) {
  printf("{\"tempo\": %.9g, \"amplitude\": %.9g, \"frequency\": %.9g, \"attack\": %.9g, "
         "\"force\": %.9g, \"calm_or_loud\": %d, \"start\": %d, \"end\": %d, \"mean\": %d, "
         "\"variance\": %d, \"n_frames\": %d, \"nb_frames\": %d, \"n_windows\": %d, "
         "\"beat\": %d, \"atk_sum\": %.17g, \"min_peak_margin\": %.6g}\n",
         r->tempo, r->amplitude, r->frequency, r->attack, r->force, r->calm_or_loud,
         r->start, r->end, r->mean, r->variance, r->n_frames, r->nb_frames,
         r->n_windows, r->beat, r->atk_sum, r->min_peak_margin);
}

static double now(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

int main(int argc, char **argv) {
  if (argc >= 5 && !strcmp(argv[1], "raw")) {
    FILE *f = fopen(argv[2], "rb");
    if (!f) { perror(argv[2]); return 1; }
    fseek(f, 0, SEEK_END);
    long bytes = ftell(f);
    fseek(f, 0, SEEK_SET);
    int16_t *pcm = (int16_t *)malloc(bytes);
    if (fread(pcm, 1, bytes, f) != (size_t)bytes) return 1;
    fclose(f);
    orc_result r;
    orc_analyze_pcm(pcm, (int)(bytes / 2), atoi(argv[3]), strtoull(argv[4], 0, 10), &r);
    print_result(&r);
    free(pcm);
    return 0;
  }
  if (argc >= 7 && !strcmp(argv[1], "synthn")) {
    uint32_t seed = strtoul(argv[2], 0, 10), rate = strtoul(argv[3], 0, 10), ch = strtoul(argv[4], 0, 10);
    uint32_t n = strtoul(argv[5], 0, 10);
    int16_t *pcm = (int16_t *)malloc((size_t)n * 2 + 2);
    orc_synth_fill(pcm, n, seed, rate, ch);
    orc_result r;
    orc_analyze_pcm(pcm, (int)n, (int)ch, strtoull(argv[6], 0, 10), &r);
    print_result(&r);
    free(pcm);
    return 0;
  }
  if (argc >= 6 && (!strcmp(argv[1], "synth") || !strcmp(argv[1], "time"))) {
    uint32_t seed = strtoul(argv[2], 0, 10), rate = strtoul(argv[3], 0, 10);
    uint32_t ch = strtoul(argv[4], 0, 10), secs = strtoul(argv[5], 0, 10);
    uint32_t n = rate * ch * secs;
    int count = (argc >= 7) ? atoi(argv[6]) : 1;
    int16_t *pcm = (int16_t *)malloc((size_t)n * 2);
    double total = 0;
    for (int s = 0; s < count; ++s) {
      orc_synth_fill(pcm, n, seed + s, rate, ch);
      orc_result r;
      double t0 = now();
      orc_analyze_pcm(pcm, (int)n, (int)ch, secs, &r);
      total += now() - t0;
      if (!strcmp(argv[1], "synth")) print_result(&r);
    }
    if (!strcmp(argv[1], "time"))
      printf("{\"songs\": %d, \"seconds\": %.6f, \"songs_per_s\": %.6f}\n", count, total,
             count / total);
    free(pcm);
    return 0;
  }
  fprintf(stderr, "usage: see orc_cli.c header\n");
  return 2;
}
