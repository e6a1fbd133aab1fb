// Host-side unit test of bliss_amd/csrc/bl_tail.h against the oracle: the streaming
// tail must reproduce beat (exactly) and atk_sum (bit-exactly on the CPU, same libm)
// from the oracle's window energies.  Build with -ffp-contract=off, link liboracle.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../bliss_amd/csrc/bl_tail.h"
#include "../../oracle/bliss_oracle.h"

static int check(unsigned seed, unsigned rate, unsigned ch, unsigned secs, unsigned extra) {
  unsigned n = rate * ch * secs + extra;
  std::vector<int16_t> pcm(n);
  orc_synth_fill(pcm.data(), n, seed, rate, ch);
  orc_result r;
  memset(&r, 0, sizeof r);
  std::vector<float> en(2 * (n / 512) + 4, 0.f);
  orc_envelope(pcm.data(), (int)n, secs, &r, en.data());
  std::vector<double> scratch(48);
  bl_tail t;
  t.init(r.nb_frames, scratch.data(), 1);
  const double log101 = log((double)(1 + 100.0f)); // C semantics: log() of a double
  for (int j = 0; j < 2 * r.nb_frames; ++j) {
    double x = 0;
    if ((j & 1) == 0) x = bl_tail_compress((double)en[j / 2], log101);
    t.step(j, x);
  }
  t.finish();
  float tempo = bl_tail_tempo(t.beat(), secs), attack = bl_tail_attack(t.atk, (int)n);
  int ok = t.beat() == r.beat && t.atk == r.atk_sum && tempo == r.tempo && attack == r.attack;
  printf("seed %u n %u: beat %d/%d atk %.17g/%.17g tempo %g attack %g %s\n", seed, n, t.beat(),
         r.beat, t.atk, r.atk_sum, tempo, attack, ok ? "ok" : "MISMATCH");
  return ok;
}

int main() {
  int ok = 1;
  ok &= check(1, 22050, 2, 11, 0);
  ok &= check(2, 44100, 2, 30, 0);
  ok &= check(3, 44100, 1, 20, 777);
  ok &= check(4, 8000, 1, 1, 0);      // 8000 samples -> N = 30: short-array edges
  ok &= check(5, 5120, 1, 1, 0);      // N = 20: the minimum the reference supports
  ok &= check(6, 5632, 1, 1, 0);      // N = 22
  puts(ok ? "OK" : "FAIL");
  return ok ? 0 : 1;
}
