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

static int check(unsigned seed, unsigned rate, unsigned ch, unsigned secs, unsigned extra, bool use_fast = true) {
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
  const int N = 2 * r.nb_frames;
  for (int j = 0; j < N; ++j)
    t.step(j, (j & 1) == 0 ? bl_tail_compress((double)en[j / 2], log101) : 0.0);
  t.finish();
  float tempo = bl_tail_tempo(t.beat(), secs), attack = bl_tail_attack(t.atk, (int)n);
  int ok = t.beat() == r.beat && t.atk == r.atk_sum && tempo == r.tempo && attack == r.attack;
  // the three-stage form used by k_env_tail (recurrence | weighting + box 1 | box 2 + peaks):
  // 38-step blocks from j = 0 with a FIFO of up to 48 box-1 outputs per block between the last
  // two, register-ring chunks wherever a stage is in its steady state for the whole block
  {
    std::vector<double> s_ab(29), s_c(19), yv(N + 38, 0.0), fifo(48);
    bl_tail_iir a;
    bl_tail_ab ab;
    bl_tail_c c;
    a.init();
    ab.init(r.nb_frames, s_ab.data(), 1);
    c.init(r.nb_frames, s_c.data(), 1);
    for (int j = 0; j < N; j += 2) a.pair(bl_tail_compress((double)en[j / 2], log101), yv[j], yv[j + 1]);
    for (int j = 0; j < N; j += 38) {
      bl_tail_fifo f;
      f.base = fifo.data(); f.stride = 1; f.count = 0;
      if (use_fast && bl_tail_ab::chunk_ok(j, N)) {
        ab.fast_chunk38(&yv[j], 1, fifo.data(), 1);
        f.count = 38;
      } else {
        for (int q = j; q < j + 38 && q < N; ++q) {
          ab.step(q, yv[q], f);
          if (q == N - 1) ab.finish(f);
        }
      }
      if (f.count > 48) { printf("  fifo overflow %d\n", f.count); ok = 0; }
      if (use_fast && f.count == 38 && c.chunk_ok()) c.fast_chunk38(fifo.data(), 1);
      else
        for (int q = 0; q < f.count; ++q) c.push(fifo[q]);
      if (c.taken == N) c.finish();
    }
    const int ok2 = c.beat() == r.beat && ab.atk == r.atk_sum && c.taken == N;
    if (!ok2) printf("  three-stage: beat %d atk %.17g taken %d MISMATCH\n", c.beat(), ab.atk, c.taken);
    ok &= ok2;
  }
  printf("seed %u n %u: beat %d/%d atk %.17g/%.17g tempo %g attack %g %s\n", seed, n, t.beat(),
         r.beat, t.atk, r.atk_sum, tempo, attack, ok ? "ok" : "MISMATCH");
  return ok;
}

// bl_div_const must agree with IEEE division bit for bit
static int check_div() {
  unsigned long long st = 88172645463325252ull;
  int bad = 0;
  for (int i = 0; i < 20000000; ++i) {
    st ^= st << 13; st ^= st >> 7; st ^= st << 17;
    double x = (double)(st >> 11) * (1.0 / 9007199254740992.0);   // [0,1)
    int e = (int)((st >> 3) % 80) - 60;
    x = ldexp(x, e);
    if (st & 1) x = -x;
    volatile double a = x / 19.0, b = x / 10.0;
    if (BL_DIV19(x) != a || BL_DIV10(x) != b) ++bad;
  }
  printf("bl_div_const vs IEEE '/': %d mismatches in 2e7 samples\n", bad);
  return bad == 0;
}

int main() {
  int ok = check_div();
  ok &= check(1, 22050, 2, 11, 0);
  ok &= check(2, 44100, 2, 30, 0);
  ok &= check(3, 44100, 1, 20, 777);
  ok &= check(4, 8000, 1, 1, 0);      // 8000 samples -> N = 30: short-array edges
  ok &= check(5, 5120, 1, 1, 0);      // N = 20: the minimum the reference supports
  ok &= check(6, 5632, 1, 1, 0);      // N = 22
  ok &= check(7, 22050, 2, 7, 0, false);  // generic path only
  ok &= check(8, 13312, 1, 1, 0);     // N = 52
  ok &= check(9, 14000, 1, 1, 0);     // N = 54
  puts(ok ? "OK" : "FAIL");
  return ok ? 0 : 1;
}
