// How sensitive is the integer `beat` (ref src/tempo_atk_sort.c:277-280) to the last bits of the
// window energies?  300 random songs; the oracle's energies are perturbed by up to +-U ulp(f32) each and
// the tail (bl_tail.h, host build) is re-run.  Backs the "why the DFT stays" paragraph of DESIGN.md: an
// energy computed by Parseval differs from the reference's f32-rounded running sum by a few ulp.
// Test infrastructure (links oracle/liboracle.so).
// Build: g++ -O2 -ffp-contract=off tools/beat_sensitivity.cpp -Loracle -loracle -Wl,-rpath,$PWD/oracle -lm
// Measured (round 1, one trial per song): +-1 ulp: beat changes in 1 of 300 songs; +-8 ulp: 4; +-64 ulp: 18.
// Round 3: many trials per song give q = P(beat changes | one energy moves one ulp), which multiplied by the
// measured number of energies an f64-level change of the FIR moves per song (tools/fir_ab.py) is the expected
// beat flip rate of that change.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
#include <random>
#include "../bliss_amd/csrc/bl_tail.h"
#include "../oracle/bliss_oracle.h"
static int beat_of(const std::vector<float> &en, int nb_frames) {
  std::vector<double> scratch(48);
  bl_tail t; t.init(nb_frames, scratch.data(), 1);
  const double log101 = log((double)(1 + 100.0f));
  const int N = 2 * nb_frames;
  for (int j = 0; j < N; ++j) t.step(j, (j & 1) == 0 ? bl_tail_compress((double)en[j / 2], log101) : 0.0);
  t.finish();
  return t.beat();
}
int main(int argc, char **argv) {
  // usage: beat_sensitivity [ulps = 1] [trials per song = 1] [songs = 300] [seconds: 0 = 3..33 | fixed]
  // Every trial perturbs each energy by a uniform integer in [-ulps, +ulps] f32 ulps.  With ulps = 1 the
  // ratio (trials with a changed beat) / (energies moved) estimates q = P(beat changes | ONE energy moves
  // by one ulp): the regime is linear (a trial changes beat with probability ~ 1e-3).
  const int ulps = argc > 1 ? atoi(argv[1]) : 1;
  const int trials = argc > 2 ? atoi(argv[2]) : 1;
  const unsigned n_songs = argc > 3 ? (unsigned)atoi(argv[3]) : 300;
  const double fixed_secs = argc > 4 ? atof(argv[4]) : 0;
  std::mt19937 rng(7);
  std::uniform_real_distribution<double> U(0, 1);
  std::normal_distribution<double> G(0, 1);
  int songs = 0, changed = 0, total_abs = 0; double minm = 1e9;
  long long n_trials = 0, moved = 0, windows = 0;
  for (unsigned seed = 1; seed <= n_songs; ++seed) {
    const unsigned rates[5] = {8000, 11025, 22050, 44100, 48000};
    unsigned rate = rates[rng() % 5], ch = 1 + rng() % 2; double secs = 3 + 30 * U(rng);
    if (fixed_secs > 0) { secs = fixed_secs; rate = 44100; ch = 2; }
    unsigned frames = (unsigned)(rate * secs), n = frames * ch;
    std::vector<int16_t> pcm(n);
    double level = pow(10, 1.5 + 2.9 * U(rng)); int nt = 1 + rng() % 4; double f[4], a[4], ph[4];
    for (int k = 0; k < nt; ++k) { f[k] = pow(10, 1.5 + U(rng) * (log10(rate / 2.2) - 1.5)); a[k] = 0.2 + 0.8 * U(rng); ph[k] = 6.28 * U(rng); }
    double bpm = 50 + 150 * U(rng), nl = level * pow(10, -3 + 2.5 * U(rng)), dc = (rng() % 4 == 0) ? (U(rng) - 0.5) * 32000 : 0;
    for (unsigned i = 0; i < frames; ++i) {
      double t = (double)i / rate, x = 0;
      for (int k = 0; k < nt; ++k) x += a[k] * sin(2 * M_PI * f[k] * t + ph[k]);
      x *= 0.55 + 0.45 * (sin(2 * M_PI * bpm / 60 * t) > 0.6);
      x = x / nt * level + nl * G(rng) + dc;
      for (unsigned c = 0; c < ch; ++c) { double v = x * (0.6 + 0.4 * (c == 0)) + 2 * G(rng); v = v > 32767 ? 32767 : (v < -32768 ? -32768 : v); pcm[i * ch + c] = (int16_t)lrint(v); }
    }
    if (n < 5120) continue;
    orc_result r; memset(&r, 0, sizeof r);
    std::vector<float> en(2 * (n / 512) + 4, 0.f);
    orc_envelope(pcm.data(), (int)n, (unsigned long long)secs + 1, &r, en.data());
    if (r.min_peak_margin < minm) minm = r.min_peak_margin;
    const int b0 = beat_of(en, r.nb_frames);
    ++songs; windows += r.n_windows;
    bool song_changed = false;
    for (int t = 0; t < trials; ++t) {
      std::vector<float> e2 = en;
      for (int w = 0; w < r.n_windows; ++w) {
        float &v = e2[w];
        int k = (int)(rng() % (2 * ulps + 1)) - ulps;
        if (k) ++moved;
        for (int q = 0; q < abs(k); ++q) v = nextafterf(v, k > 0 ? INFINITY : -INFINITY);
      }
      const int b1 = beat_of(e2, r.nb_frames);
      ++n_trials;
      if (b1 != b0) { ++changed; total_abs += abs(b1 - b0); song_changed = true; }
    }
    (void)song_changed;
  }
  printf("{\"ulps\": %d, \"songs\": %d, \"trials\": %lld, \"windows_per_song_avg\": %.0f, \"trials_with_beat_changed\": %d, "
         "\"sum_abs_beat_delta\": %d, \"energies_moved\": %lld, \"beat_changes_per_energy_moved\": %.3g, "
         "\"min_peak_margin\": %.3g}\n",
         ulps, songs, n_trials, (double)windows / songs, changed, total_abs, moved,
         moved ? (double)changed / (double)moved : 0.0, minm);
}
