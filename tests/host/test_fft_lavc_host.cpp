// Host-side unit test of bliss_amd/csrc/bl_fft_lavc.h: the device's lane code of the frequency analysis' f32 DFT
// (16 lanes emulated phase by phase, T = float) against the oracle's restatement of libavcodec's operation order
// (oracle/orc_fft_lavc.c) — every power value re*re + im*im of bins 1..255 BIT FOR BIT, on Hann-windowed integer
// frames like the kernel's and on random floats.  Also: the gather index decomposition against the recursive
// split-radix permutation.  Build: g++ -O2 -std=c++17 -ffp-contract=off test_fft_lavc_host.cpp -x c orc_fft_lavc.c ...
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../bliss_amd/csrc/bl_fft_lavc.h"

extern "C" void orc_lavc_rdft512_f32(float *x);

static int srp(int i, int n) {
  if (n <= 2) return i & 1;
  int m = n >> 1;
  if (!(i & m)) return srp(i, m) * 2;
  m >>= 1;
  if (i & m) return srp(i, m) * 4 + 1; /* inverse == !(i & m) with inverse = 0: the forward transform */
  return srp(i, m) * 4 - 1;
}

static float g_tw[LV_TW_SLOTS * 16][2], g_leafc[4];

static int run(const std::vector<float> &x) {
  std::vector<float> ref(x);
  orc_lavc_rdft512_f32(ref.data());
  float want[256];
  for (int d = 1; d < 256; ++d) want[d] = (ref[2 * d] * ref[2 * d]) + (ref[2 * d + 1] * ref[2 * d + 1]);

  float re[16][16], im[16][16];
  for (int L = 0; L < 16; ++L)
    for (int r = 0; r < 16; ++r) {
      const int m = lv_gather_index(L, r); /* one load order for every lane, as in the kernel */
      re[L][r] = x[2 * m]; im[L][r] = x[2 * m + 1];
    }
  for (int L = 0; L < 16; ++L) lv_leaves<float>(lv_lane_is_t16(L), re[L], im[L], g_leafc[0], g_leafc[1], g_leafc[2]);
  float ar[16][16], ai[16][16]; /* layout A: [lane l][register j] = position 16 j + l */
  for (int l = 0; l < 16; ++l)
    for (int j = 0; j < 16; ++j) { ar[l][j] = re[j][l]; ai[l][j] = im[j][l]; }
  const int blocks32[5] = {0, 4, 6, 8, 12};
  for (int b = 0; b < 5; ++b) {
    const int R = blocks32[b];
    float tA[16], tB[16];
    for (int l = 0; l < 16; ++l) {
      const float *w = g_tw[LV_TW_P32 * 16 + l];
      lv_pass32_mul<float>(ar[l][R + 1], ai[l][R + 1], w[0], l < 8 ? -w[1] : w[1], tA[l], tB[l]);
    }
    for (int l = 0; l < 16; ++l) {
      const bool lo = l < 8;
      auto sel = [lo](float a, float b) { return lo ? a : b; };
      lv_pass32_fin<float>(ar[l][R], ai[l][R], ar[l][R + 1], ai[l][R + 1], tA[l], tB[l], tA[l ^ 8], tB[l ^ 8], sel);
    }
  }
  for (int l = 0; l < 16; ++l) {
    auto W = [&](int slot, int c) { return g_tw[slot * 16 + l][c]; };
    lv_pass_inlane<float, 0, 1>(ar[l], ai[l], W(LV_TW_P64, 0), W(LV_TW_P64, 1));
    lv_pass_inlane<float, 8, 1>(ar[l], ai[l], W(LV_TW_P64, 0), W(LV_TW_P64, 1));
    lv_pass_inlane<float, 12, 1>(ar[l], ai[l], W(LV_TW_P64, 0), W(LV_TW_P64, 1));
    lv_pass_inlane<float, 0, 2>(ar[l], ai[l], W(LV_TW_P128, 0), W(LV_TW_P128, 1));
    lv_pass_inlane<float, 1, 2>(ar[l], ai[l], W(LV_TW_P128 + 1, 0), W(LV_TW_P128 + 1, 1));
    lv_pass_inlane<float, 0, 4>(ar[l], ai[l], W(LV_TW_P256, 0), W(LV_TW_P256, 1));
    lv_pass_inlane<float, 1, 4>(ar[l], ai[l], W(LV_TW_P256 + 1, 0), W(LV_TW_P256 + 1, 1));
    lv_pass_inlane<float, 2, 4>(ar[l], ai[l], W(LV_TW_P256 + 2, 0), W(LV_TW_P256 + 2, 1));
    lv_pass_inlane<float, 3, 4>(ar[l], ai[l], W(LV_TW_P256 + 3, 0), W(LV_TW_P256 + 3, 1));
  }
  float got[257];
  for (int d = 0; d <= 256; ++d) got[d] = -1.f;
  for (int l = 0; l < 16; ++l)
    for (int j = 0; j < 8; ++j) {
      const int i = 16 * j + l;
      if (i == 0) continue;
      const int lp = (16 - l) & 15, jp = l ? 15 - j : 16 - j;
      float own, mir;
      lv_post_power<float>(ar[l][j], ai[l][j], ar[lp][jp], ai[lp][jp], g_tw[(LV_TW_POST + j) * 16 + l][0],
                           g_tw[(LV_TW_POST + j) * 16 + l][1], 0.5f, own, mir);
      got[i] = own; got[256 - i] = mir;
    }
  got[128] = lv_mid_power<float>(ar[0][8], ai[0][8]);
  int bad = 0;
  for (int d = 1; d < 256; ++d)
    if (memcmp(&got[d], &want[d], 4) != 0) {
      if (bad < 5) printf("bin %d: got %.9g want %.9g\n", d, got[d], want[d]);
      ++bad;
    }
  return bad;
}

int main() {
  /* the gather decomposition is the split-radix permutation of a 256-point forward transform */
  int revtab[256];
  for (int i = 0; i < 256; ++i) revtab[-srp(i, 256) & 255] = i;
  for (int L = 0; L < 16; ++L)
    for (int r = 0; r < 16; ++r)
      if (revtab[lv_index(L, r)] != 16 * L + r) { printf("lv_index(%d, %d) wrong\nFAIL\n", L, r); return 1; }
  for (int L = 0; L < 16; ++L) /* ... and what the T8 lanes find in their registers 8..15 */
    for (int t = 0; t < 8; ++t) {
      const int reg = lv_lane_is_t16(L) ? 8 + t : 8 + lv_s8(t);
      if (lv_gather_index(L, reg) != lv_index(L, 8 + t)) { printf("gather order (%d, %d) wrong\nFAIL\n", L, t); return 1; }
    }
  lv_fill_tables(g_tw, g_leafc);
  int bad = 0, frames = 0;
  std::vector<float> hann(512);
  for (int i = 0; i < 512; ++i) hann[i] = (float)(.5f * (1.0f - cos(2 * M_PI * i / (512 - 1))));
  srand(12345);
  for (int t = 0; t < 400; ++t, ++frames) { /* Hann-windowed integers: what the kernel transforms */
    std::vector<float> x(512);
    const int amp = t < 100 ? 32767 : t < 200 ? 3000 : t < 300 ? 40 : 1;
    for (int i = 0; i < 512; ++i) {
      int s = (rand() % (2 * amp + 1)) - amp;
      if (t % 7 == 3) s = (int)(amp * sin(0.01 * i * (t + 1)));          /* tonal */
      if (t % 50 == 49) s = 0;                                            /* digital silence */
      x[i] = (float)s * hann[i];
    }
    bad += run(x);
  }
  for (int t = 0; t < 100; ++t, ++frames) { /* random floats */
    std::vector<float> x(512);
    for (auto &v : x) v = (float)((rand() / (double)RAND_MAX - 0.5) * 2e4);
    bad += run(x);
  }
  printf("%d frames, %d power values differ\n", frames, bad);
  if (bad) { printf("FAIL\n"); return 1; }
  printf("OK\n");
  return 0;
}
