/*
 * bl_decode.c — host ingest behind bl_audio_decode().
 *
 * Replaces ref src/decode.c:27-213 (libavformat/libavcodec/libswresample) for
 * the container formats that need no third-party code: RIFF/WAVE PCM (8 / 16 / 24 / 32 bit
 * integer, 32-bit float) and native FLAC (8 to 32 bit).  It fills struct bl_song as
 * fill_song_properties()/bl_audio_decode() do (ref src/decode.c:187-193,
 * 215-349): malloc'd interleaved s16 `sample_array`, nSamples = interleaved
 * count, nb_bytes_per_sample = 2, duration = whole seconds, strdup'd tags
 * (the reference's "<no title>"-style defaults when absent, ref
 * src/decode.c:263-308), filename.
 *
 * Sources wider than 16 bits are narrowed like a same-rate S32 -> S16 sample
 * format conversion (ref src/decode.c:323-346,388-392: libswresample): the
 * sample left-justified in 32 bits, arithmetic >> 16 — for 24-bit audio the
 * top 16 bits.  The FLAC decoder itself is pinned for 24-bit input by the
 * STREAMINFO MD5 of the reference's audio/song_s32*.flac (tests/test_ingest.py);
 * the same-rate narrowing step is libswresample's and stays parity-unpinned.
 *
 * Integrity.  Every FLAC frame's CRC-8 / CRC-16 is checked, frame numbers must follow each
 * other and the decoded length must equal STREAMINFO's: a damaged or truncated file fails
 * (BL_UNEXPECTED) instead of being analysed with altered or missing audio.  The stored MD5 of the
 * whole stream is checked on request (bl_amd_flac_verify), not on every decode.
 *
 * Sample rate.  The reference always hands 22 050 Hz PCM to the analyzers
 * (ref src/decode.c:7-9,317-346), and every analyzer constant assumes it: a file at any other
 * rate is converted (bl_resample.c, a restatement of libswresample's default resampler pinned
 * by the digests of ref tests/test_decode.c:35-36,55-56) to 22 050 Hz stereo s16, resampled = 1.
 * bl_amd_decode_allow_native_rate(1) / BL_AMD_ALLOW_NATIVE_RATE=1 switches the conversion off.
 * Sources that are already 22 050 Hz s16 (the reference's own audio/song.flac) decode to the
 * byte-identical sample_array (MD5 pinned by ref tests/test_decode.c:16-17); a 22 050 Hz mono
 * file stays mono.
 */
#include <ctype.h>
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "bliss.h"
#include "bliss_amd.h"
#include "bl_resample.h"

#define BL_DECODE_RATE 22050 /* ref src/decode.c:7 SAMPLE_RATE */

/* -1: not set by the caller, the environment decides.  Read and written with relaxed atomics: the
 * setter may run on one thread while decodes run on others, and either value is a valid answer. */
static int g_allow_native_rate = -1;

void bl_amd_decode_allow_native_rate(int allow) {
  __atomic_store_n(&g_allow_native_rate, allow != 0, __ATOMIC_RELAXED);
}

static int native_rate_allowed(void) {
  const int v = __atomic_load_n(&g_allow_native_rate, __ATOMIC_RELAXED);
  if (v >= 0) return v;
  const char *e = getenv("BL_AMD_ALLOW_NATIVE_RATE"); /* not cached: getenv is cheap next to a decode */
  return e && *e && strcmp(e, "0") != 0;
}

/* a sample of `bps` significant bits as the s16 the analyzers read: left-justify in 32 bits,
 * arithmetic >> 16 */
static inline int16_t narrow_sample(int32_t v, uint32_t bps) {
  if (bps > 16) return (int16_t)(v >> (bps - 16));
  return (int16_t)((uint32_t)v << (16 - bps));
}

/* Where the decoders put the interleaved samples: narrowed to s16 (the analyzers' format), or —
 * for a source wider than 16 bits that still has to go through the rate converter —
 * left-justified in 32 bits, which is what FFmpeg's decoders hand to libswresample. */
typedef struct {
  int16_t *p16;
  int32_t *p32;
  size_t n, cap;
  int wide;
  int is_float; /* wide only: p32 holds IEEE float bit patterns (RIFF format tag 3) */
  int non_s16;  /* the source's sample format is not S16 for FFmpeg (u8, 24 / 32 bit, float): the
                 * reference sends such a file through libswresample even at 22 050 Hz */
  uint32_t bps; /* significant bits of the source */
  int native;   /* native_rate_allowed() as this decode saw it when it started */
} pcm_sink;

static int sink_reserve(pcm_sink *s, size_t more) {
  if (s->n + more <= s->cap) return 0;
  size_t cap = (s->cap + more) * 2;
  if (s->wide) {
    int32_t *np = (int32_t *)realloc(s->p32, cap * sizeof(int32_t) + 16);
    if (!np) return -1;
    s->p32 = np;
  } else {
    int16_t *np = (int16_t *)realloc(s->p16, cap * sizeof(int16_t) + 16);
    if (!np) return -1;
    s->p16 = np;
  }
  s->cap = cap;
  return 0;
}

static inline void sink_put(pcm_sink *s, int32_t v) {
  if (s->wide) s->p32[s->n++] = (int32_t)((uint32_t)v << (32 - s->bps));
  else s->p16[s->n++] = narrow_sample(v, s->bps);
}

static void sink_free(pcm_sink *s) {
  free(s->p16);
  free(s->p32);
  s->p16 = NULL;
  s->p32 = NULL;
}

static int wants_rate_conversion(uint32_t rate, int native);
/* keep all 32 bits: the source is wider than 16 bits and a float stage follows — the rate
 * converter, or the mono up-mix of a same-rate file (bl_audio_decode) */
static int wants_wide(uint32_t bits, uint32_t rate, uint32_t channels, int native) {
  return bits > 16 && (wants_rate_conversion(rate, native) || (channels == 1 && !native));
}

/* ----------------------------------------------------------------------- */
typedef struct {
  const uint8_t *p;
  size_t len, pos; /* byte position */
  uint64_t acc;    /* bit accumulator, MSB first */
  int nacc;        /* valid bits in acc */
  int err;
} bitrd;

static void br_init(bitrd *b, const uint8_t *p, size_t len, size_t pos) {
  b->p = p; b->len = len; b->pos = pos; b->acc = 0; b->nacc = 0; b->err = 0;
}

static inline void br_fill(bitrd *b) {
  while (b->nacc <= 56) {
    uint64_t byte = 0;
    if (b->pos < b->len) byte = b->p[b->pos];
    else if (b->pos > b->len + 16) { b->err = 1; }
    b->pos++;
    b->acc |= byte << (56 - b->nacc);
    b->nacc += 8;
  }
}

static inline uint32_t br_bits(bitrd *b, int n) { /* n in 0..32 */
  if (n == 0) return 0;
  if (b->nacc < n) br_fill(b);
  uint32_t v = (uint32_t)(b->acc >> (64 - n));
  b->acc <<= n;
  b->nacc -= n;
  return v;
}

static inline int32_t br_sbits(bitrd *b, int n) {
  if (n == 0) return 0;
  uint32_t v = br_bits(b, n);
  uint32_t m = 1u << (n - 1);
  return (int32_t)((v ^ m) - m);
}

static inline uint32_t br_unary(bitrd *b) { /* count zeros before the next 1 */
  uint32_t q = 0;
  for (;;) {
    if (b->nacc == 0) br_fill(b);
    if (b->acc == 0) { /* all buffered bits are zero */
      q += (uint32_t)b->nacc;
      b->nacc = 0;
      if (b->err || q > (1u << 24)) { b->err = 1; return q; }
      continue;
    }
    int lz = __builtin_clzll(b->acc);
    if (lz >= b->nacc) { q += (uint32_t)b->nacc; b->acc = 0; b->nacc = 0; continue; }
    q += (uint32_t)lz;
    b->acc <<= (lz + 1);
    b->nacc -= lz + 1;
    return q;
  }
}

static inline void br_align(bitrd *b) {
  int drop = b->nacc & 7;
  b->acc <<= drop;
  b->nacc -= drop;
}

/* byte offset of the next unread bit (must be aligned) */
static inline size_t br_bytepos(const bitrd *b) { return b->pos - (size_t)(b->nacc / 8); }

/* ----------------------------------------------------------------------- */
static int read_file(const char *path, uint8_t **data, size_t *len) {
  FILE *f = fopen(path, "rb");
  if (!f) return -1;
  if (fseek(f, 0, SEEK_END) != 0) { fclose(f); return -1; }
  long sz = ftell(f);
  if (sz < 0) { fclose(f); return -1; }
  rewind(f);
  uint8_t *buf = (uint8_t *)malloc((size_t)sz + 1);
  if (!buf) { fclose(f); return -1; }
  if (fread(buf, 1, (size_t)sz, f) != (size_t)sz) { free(buf); fclose(f); return -1; }
  fclose(f);
  *data = buf;
  *len = (size_t)sz;
  return 0;
}

static uint32_t le32(const uint8_t *p) {
  return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}
static uint32_t le16(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }

/* ------------------------------- WAV ----------------------------------- */
static int decode_wav(const uint8_t *d, size_t len, struct bl_song *song, pcm_sink *sink) {
  if (len < 12 || memcmp(d, "RIFF", 4) || memcmp(d + 8, "WAVE", 4)) return BL_UNEXPECTED;
  size_t pos = 12;
  int have_fmt = 0;
  uint32_t fmt_tag = 0, channels = 0, rate = 0, bits = 0;
  while (pos + 8 <= len) {
    uint32_t sz = le32(d + pos + 4);
    const uint8_t *body = d + pos + 8;
    if (pos + 8 + (size_t)sz > len) sz = (uint32_t)(len - pos - 8);
    if (!memcmp(d + pos, "fmt ", 4) && sz >= 16) {
      fmt_tag = le16(body); channels = le16(body + 2); rate = le32(body + 4);
      bits = le16(body + 14);
      if (fmt_tag == 0xFFFE && sz >= 26) fmt_tag = le16(body + 24); /* extensible */
      have_fmt = 1;
    } else if (!memcmp(d + pos, "data", 4)) {
      const int is_float = fmt_tag == 3 && bits == 32; /* IEEE float, full scale +-1 */
      const int is_pcm = fmt_tag == 1 && (bits == 8 || bits == 16 || bits == 24 || bits == 32);
      if (!have_fmt || (!is_float && !is_pcm) || channels < 1 || channels > 2 || rate == 0)
        return BL_UNEXPECTED;
      const uint32_t bytes = bits / 8;
      uint32_t n = sz / bytes;
      n -= n % channels;
      if (n == 0) return BL_UNEXPECTED;
      /* 8-bit PCM is unsigned; as s16 it is (v - 128) << 8, the conversion every 16-bit path starts from */
      sink->bps = bits == 8 ? 16 : bits;
      sink->wide = wants_wide(bits, rate, channels, sink->native);
      sink->is_float = is_float && sink->wide;
      sink->non_s16 = bits != 16;
      if (sink_reserve(sink, n)) return BL_UNEXPECTED;
      for (uint32_t i = 0; i < n; ++i) {
        const uint8_t *q = body + (size_t)bytes * i;
        int32_t v;
        if (is_float) {
          const uint32_t u = le32(q);
          if (sink->wide) { /* on its way to the rate converter: the float as it is */
            sink->p32[sink->n++] = (int32_t)u;
            continue;
          }
          /* same-rate FLT -> S16: lrintf(x * 2^15), clipped; not-a-numbers count as silence */
          float x;
          memcpy(&x, &u, 4);
          if (!(x == x) || x > 4.0f || x < -4.0f) x = x > 0 ? 4.0f : (x < 0 ? -4.0f : 0.0f);
          const long r = lrintf(x * 32768.0f);
          sink->p16[sink->n++] = (int16_t)(r > 32767 ? 32767 : r < -32768 ? -32768 : r);
          continue;
        }
        if (bits == 8) v = ((int32_t)q[0] - 128) * 256;
        else if (bits == 16) v = (int16_t)le16(q);
        else if (bits == 24) v = (int32_t)((le16(q + 1) << 8 | q[0]) << 8) >> 8;
        else v = (int32_t)le32(q);
        sink_put(sink, v);
      }
      song->nSamples = (int)n;
      song->channels = (int)channels;
      song->sample_rate = (int)rate;
      song->nb_bytes_per_sample = 2;
      song->duration = (uint64_t)(n / channels) / rate;
      song->bitrate = (int)(rate * channels * bits);
      return BL_OK;
    }
    pos += 8 + (size_t)sz + (sz & 1);
  }
  return BL_UNEXPECTED;
}

/* ------------------------------- MD5 ----------------------------------- */
/* RFC 1321, used only to check a FLAC stream against the signature of the unencoded audio
 * that its STREAMINFO block carries (bl_amd_flac_verify). */
typedef struct {
  uint32_t h[4];
  uint64_t len;
  uint8_t buf[64];
  size_t fill;
} md5_state;

static void md5_block(uint32_t h[4], const uint8_t *p) {
  static const uint8_t rot[64] = {7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22,
                                  5, 9,  14, 20, 5, 9,  14, 20, 5, 9,  14, 20, 5, 9,  14, 20,
                                  4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23,
                                  6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21};
  /* K[i] = floor(2^32 * |sin(i + 1)|) */
  static const uint32_t K[64] = {
      0xd76aa478, 0xe8c7b756, 0x242070db, 0xc1bdceee, 0xf57c0faf, 0x4787c62a, 0xa8304613, 0xfd469501,
      0x698098d8, 0x8b44f7af, 0xffff5bb1, 0x895cd7be, 0x6b901122, 0xfd987193, 0xa679438e, 0x49b40821,
      0xf61e2562, 0xc040b340, 0x265e5a51, 0xe9b6c7aa, 0xd62f105d, 0x02441453, 0xd8a1e681, 0xe7d3fbc8,
      0x21e1cde6, 0xc33707d6, 0xf4d50d87, 0x455a14ed, 0xa9e3e905, 0xfcefa3f8, 0x676f02d9, 0x8d2a4c8a,
      0xfffa3942, 0x8771f681, 0x6d9d6122, 0xfde5380c, 0xa4beea44, 0x4bdecfa9, 0xf6bb4b60, 0xbebfbc70,
      0x289b7ec6, 0xeaa127fa, 0xd4ef3085, 0x04881d05, 0xd9d4d039, 0xe6db99e5, 0x1fa27cf8, 0xc4ac5665,
      0xf4292244, 0x432aff97, 0xab9423a7, 0xfc93a039, 0x655b59c3, 0x8f0ccc92, 0xffeff47d, 0x85845dd1,
      0x6fa87e4f, 0xfe2ce6e0, 0xa3014314, 0x4e0811a1, 0xf7537e82, 0xbd3af235, 0x2ad7d2bb, 0xeb86d391};
  uint32_t w[16];
  for (int i = 0; i < 16; ++i) w[i] = le32(p + 4 * i);
  uint32_t a = h[0], b = h[1], c = h[2], d = h[3];
  for (int i = 0; i < 64; ++i) {
    uint32_t f;
    int g;
    if (i < 16) { f = (b & c) | (~b & d); g = i; }
    else if (i < 32) { f = (d & b) | (~d & c); g = (5 * i + 1) & 15; }
    else if (i < 48) { f = b ^ c ^ d; g = (3 * i + 5) & 15; }
    else { f = c ^ (b | ~d); g = (7 * i) & 15; }
    const uint32_t t = a + f + K[i] + w[g];
    a = d; d = c; c = b;
    b = b + ((t << rot[i]) | (t >> (32 - rot[i])));
  }
  h[0] += a; h[1] += b; h[2] += c; h[3] += d;
}

static void md5_init(md5_state *m) {
  m->h[0] = 0x67452301; m->h[1] = 0xefcdab89; m->h[2] = 0x98badcfe; m->h[3] = 0x10325476;
  m->len = 0; m->fill = 0;
}

static void md5_update(md5_state *m, const uint8_t *p, size_t n) {
  m->len += n;
  while (n) {
    const size_t take = 64 - m->fill < n ? 64 - m->fill : n;
    memcpy(m->buf + m->fill, p, take);
    m->fill += take; p += take; n -= take;
    if (m->fill == 64) { md5_block(m->h, m->buf); m->fill = 0; }
  }
}

static void md5_final(md5_state *m, uint8_t out[16]) {
  const uint64_t bits = m->len * 8;
  const uint8_t one = 0x80, zero = 0;
  md5_update(m, &one, 1);
  while (m->fill != 56) md5_update(m, &zero, 1);
  uint8_t lenb[8];
  for (int i = 0; i < 8; ++i) lenb[i] = (uint8_t)(bits >> (8 * i));
  md5_update(m, lenb, 8);
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) out[4 * i + j] = (uint8_t)(m->h[i] >> (8 * j));
}

/* ------------------------------- FLAC ---------------------------------- */
typedef struct {
  uint32_t rate, channels, bps, max_block;
  uint64_t total; /* samples per channel */
} flac_info;

static int flac_residual(bitrd *b, int32_t *out, uint32_t blocksize, uint32_t order) {
  uint32_t method = br_bits(b, 2);
  if (method > 1) return -1;
  int pbits = method ? 5 : 4;
  uint32_t esc = method ? 31 : 15;
  uint32_t porder = br_bits(b, 4);
  uint32_t parts = 1u << porder;
  if ((blocksize >> porder) << porder != blocksize && porder) return -1;
  uint32_t idx = order;
  for (uint32_t p = 0; p < parts; ++p) {
    uint32_t cnt = blocksize >> porder;
    if (p == 0) {
      if (cnt < order) return -1;
      cnt -= order;
    }
    uint32_t k = br_bits(b, pbits);
    if (idx + cnt > blocksize) return -1;
    if (k == esc) {
      uint32_t raw = br_bits(b, 5);
      for (uint32_t i = 0; i < cnt; ++i) out[idx++] = br_sbits(b, (int)raw);
    } else {
      for (uint32_t i = 0; i < cnt; ++i) {
        uint32_t q = br_unary(b);
        uint32_t u = (q << k) | br_bits(b, (int)k);
        out[idx++] = (int32_t)(u >> 1) ^ -(int32_t)(u & 1);
      }
    }
    if (b->err) return -1;
  }
  return 0;
}

static int flac_subframe(bitrd *b, int32_t *out, uint32_t blocksize, uint32_t bps) {
  if (br_bits(b, 1)) return -1; /* padding bit */
  uint32_t type = br_bits(b, 6);
  uint32_t wasted = 0;
  if (br_bits(b, 1)) wasted = br_unary(b) + 1;
  if (wasted >= bps) return -1;
  bps -= wasted;
  if (type == 0) { /* CONSTANT */
    int32_t v = br_sbits(b, (int)bps);
    for (uint32_t i = 0; i < blocksize; ++i) out[i] = v;
  } else if (type == 1) { /* VERBATIM */
    for (uint32_t i = 0; i < blocksize; ++i) out[i] = br_sbits(b, (int)bps);
  } else if (type >= 8 && type <= 12) { /* FIXED, order type-8 */
    uint32_t order = type - 8;
    if (order > blocksize) return -1;
    for (uint32_t i = 0; i < order; ++i) out[i] = br_sbits(b, (int)bps);
    if (flac_residual(b, out, blocksize, order)) return -1;
    for (uint32_t i = order; i < blocksize; ++i) {
      int64_t pred = 0;
      switch (order) {
        case 1: pred = out[i - 1]; break;
        case 2: pred = 2 * (int64_t)out[i - 1] - out[i - 2]; break;
        case 3: pred = 3 * (int64_t)out[i - 1] - 3 * (int64_t)out[i - 2] + out[i - 3]; break;
        case 4:
          pred = 4 * (int64_t)out[i - 1] - 6 * (int64_t)out[i - 2] + 4 * (int64_t)out[i - 3] -
                 out[i - 4];
          break;
        default: break;
      }
      out[i] = (int32_t)(out[i] + pred);
    }
  } else if (type >= 32) { /* LPC, order type-31 */
    uint32_t order = type - 31;
    if (order > blocksize) return -1;
    int32_t coef[32];
    for (uint32_t i = 0; i < order; ++i) out[i] = br_sbits(b, (int)bps);
    uint32_t prec = br_bits(b, 4) + 1;
    if (prec == 16) return -1;
    int32_t shift = br_sbits(b, 5);
    if (shift < 0) return -1;
    for (uint32_t i = 0; i < order; ++i) coef[i] = br_sbits(b, (int)prec);
    if (flac_residual(b, out, blocksize, order)) return -1;
    for (uint32_t i = order; i < blocksize; ++i) {
      int64_t acc = 0;
      for (uint32_t j = 0; j < order; ++j) acc += (int64_t)coef[j] * out[i - 1 - j];
      out[i] = (int32_t)(out[i] + (acc >> shift));
    }
  } else {
    return -1;
  }
  if (wasted)
    for (uint32_t i = 0; i < blocksize; ++i) out[i] = (int32_t)((uint32_t)out[i] << wasted);
  return b->err ? -1 : 0;
}

static void flac_tags(const uint8_t *body, uint32_t sz32, struct bl_song *song) {
  /* all bounds in size_t and by subtraction: the lengths are attacker-controlled 32-bit
   * fields and `pos + l` must not wrap */
  const size_t sz = sz32;
  if (sz < 8) return;
  const size_t vlen = le32(body);
  if (vlen > sz - 8) return;
  size_t pos = 4 + vlen;
  uint32_t count = le32(body + pos);
  pos += 4;
  for (uint32_t c = 0; c < count && sz - pos >= 4; ++c) {
    const size_t l = le32(body + pos);
    pos += 4;
    if (l > sz - pos) return;
    const char *kv = (const char *)body + pos;
    const char *eq = (const char *)memchr(kv, '=', l);
    if (eq) {
      size_t kl = (size_t)(eq - kv), vl = l - kl - 1;
      char key[32];
      if (kl < sizeof(key)) {
        for (size_t i = 0; i < kl; ++i) key[i] = (char)toupper((unsigned char)kv[i]);
        key[kl] = 0;
        char **dst = NULL;
        if (!strcmp(key, "ARTIST")) dst = &song->artist;
        else if (!strcmp(key, "TITLE")) dst = &song->title;
        else if (!strcmp(key, "ALBUM")) dst = &song->album;
        else if (!strcmp(key, "TRACKNUMBER")) dst = &song->tracknumber;
        else if (!strcmp(key, "GENRE")) dst = &song->genre;
        if (dst && !*dst) {
          *dst = (char *)malloc(vl + 1);
          if (*dst) { memcpy(*dst, eq + 1, vl); (*dst)[vl] = 0; }
        }
      }
    }
    pos += l;
  }
}

/* sig (optional): receives the MD5 of the decoded samples at their native width (what FLAC
 * calls the signature of the unencoded audio) and the signature stored in STREAMINFO */
/* FLAC frame checksums: CRC-8 (x^8 + x^2 + x + 1) over the frame header, CRC-16 (x^16 + x^15 +
 * x^2 + 1) over the whole frame.  A header whose CRC-8 does not match is a false sync code
 * inside audio data and is skipped; a frame whose CRC-16 does not match is damaged audio, and
 * the decode fails instead of handing altered samples to the analyzers. */
static uint8_t flac_crc8(const uint8_t *p, size_t n) {
  uint8_t c = 0;
  for (size_t i = 0; i < n; ++i) {
    c ^= p[i];
    for (int k = 0; k < 8; ++k) c = (uint8_t)((c & 0x80) ? (c << 1) ^ 0x07 : c << 1);
  }
  return c;
}

/* bl_audio_decode is the call callers run on many threads at once: the table is built exactly
 * once, with the stores ordered before any reader (pthread_once) */
static uint16_t g_crc16_tab[256];
static pthread_once_t g_crc16_once = PTHREAD_ONCE_INIT;
static void flac_crc16_init(void) {
  for (int v = 0; v < 256; ++v) {
    uint16_t c = (uint16_t)(v << 8);
    for (int k = 0; k < 8; ++k) c = (uint16_t)((c & 0x8000) ? (c << 1) ^ 0x8005 : c << 1);
    g_crc16_tab[v] = c;
  }
}

static uint16_t flac_crc16(const uint8_t *p, size_t n) {
  pthread_once(&g_crc16_once, flac_crc16_init);
  uint16_t c = 0;
  for (size_t i = 0; i < n; ++i) c = (uint16_t)((c << 8) ^ g_crc16_tab[(c >> 8) ^ p[i]]);
  return c;
}

typedef struct {
  md5_state md;
  uint8_t stored[16];
} flac_sig;

static int decode_flac(const uint8_t *d, size_t len, struct bl_song *song, pcm_sink *sink,
                       flac_sig *sig) {
  if (len < 42 || memcmp(d, "fLaC", 4)) return BL_UNEXPECTED;
  size_t pos = 4;
  flac_info fi;
  memset(&fi, 0, sizeof(fi));
  int last = 0, have_info = 0;
  while (!last && pos + 4 <= len) {
    last = d[pos] >> 7;
    uint32_t type = d[pos] & 0x7f;
    uint32_t sz = ((uint32_t)d[pos + 1] << 16) | ((uint32_t)d[pos + 2] << 8) | d[pos + 3];
    const uint8_t *body = d + pos + 4;
    if (pos + 4 + sz > len) return BL_UNEXPECTED;
    if (type == 0 && sz >= 34) {
      fi.max_block = ((uint32_t)body[2] << 8) | body[3];
      fi.rate = ((uint32_t)body[10] << 12) | ((uint32_t)body[11] << 4) | (body[12] >> 4);
      fi.channels = ((body[12] >> 1) & 7) + 1;
      fi.bps = (((uint32_t)body[12] & 1) << 4 | (body[13] >> 4)) + 1;
      fi.total = ((uint64_t)(body[13] & 15) << 32) | ((uint64_t)body[14] << 24) |
                 ((uint64_t)body[15] << 16) | ((uint64_t)body[16] << 8) | body[17];
      if (sig) memcpy(sig->stored, body + 18, 16);
      have_info = 1;
    } else if (type == 4) {
      flac_tags(body, sz, song);
    }
    pos += 4 + sz;
  }
  if (!have_info || fi.bps < 8 || fi.bps > 32 || fi.channels < 1 || fi.channels > 2 || fi.rate == 0)
    return BL_UNEXPECTED;
  size_t audio_start = pos;

  sink->bps = fi.bps;
  sink->wide = wants_wide(fi.bps, fi.rate, fi.channels, sink->native);
  sink->non_s16 = fi.bps > 16; /* FFmpeg's FLAC decoder delivers S16 up to 16 bits, S32 above */
  if (sink_reserve(sink, fi.total ? (size_t)fi.total * fi.channels / 2 + 8 : (size_t)1 << 19))
    return BL_UNEXPECTED;
  uint32_t maxb = fi.max_block ? fi.max_block : 65535;
  int32_t *ch[2];
  ch[0] = (int32_t *)malloc(sizeof(int32_t) * (size_t)(maxb + 16) * 2);
  if (!ch[0]) return BL_UNEXPECTED;
  ch[1] = ch[0] + maxb + 16;

  static const uint32_t bs_tab[16] = {0,    192,  576,  1152, 2304, 4608, 0,     0,
                                      256,  512,  1024, 2048, 4096, 8192, 16384, 32768};
  int rc = BL_OK;
  uint64_t frames_seen = 0, frames_done = 0; /* FLAC frames / inter-channel sample frames decoded */
  while (pos + 6 < len) {
    if (d[pos] != 0xFF || (d[pos + 1] & 0xFE) != 0xF8) { ++pos; continue; } /* resync */
    bitrd b;
    br_init(&b, d, len, pos + 2);
    uint32_t bs_code = br_bits(&b, 4), sr_code = br_bits(&b, 4);
    uint32_t chan = br_bits(&b, 4), ss_code = br_bits(&b, 3);
    if (br_bits(&b, 1) || sr_code == 15 || bs_code == 0 || chan > 10) { ++pos; continue; }
    /* UTF-8 style coded frame number (fixed block size) or first-sample number (variable) */
    uint32_t first = br_bits(&b, 8);
    int extra = 0;
    if (first >= 0xFE) extra = 6; else if (first >= 0xFC) extra = 5; else if (first >= 0xF8) extra = 4;
    else if (first >= 0xF0) extra = 3; else if (first >= 0xE0) extra = 2; else if (first >= 0xC0) extra = 1;
    else if (first >= 0x80) { ++pos; continue; }
    uint64_t coded = extra ? (first & (0x3Fu >> extra)) : first;
    for (int i = 0; i < extra; ++i) coded = (coded << 6) | (br_bits(&b, 8) & 0x3F);
    const int variable_blocks = d[pos + 1] & 1;
    uint32_t blocksize = bs_tab[bs_code];
    if (bs_code == 6) blocksize = br_bits(&b, 8) + 1;
    else if (bs_code == 7) blocksize = br_bits(&b, 16) + 1;
    if (sr_code == 12) br_bits(&b, 8);
    else if (sr_code == 13 || sr_code == 14) br_bits(&b, 16);
    {
      const size_t hdr_end = br_bytepos(&b);
      const uint32_t crc8 = br_bits(&b, 8);
      if (b.err || hdr_end > len || flac_crc8(d + pos, hdr_end - pos) != crc8) { ++pos; continue; }
    }
    if (blocksize == 0 || blocksize > maxb + 16) { ++pos; continue; }
    static const uint32_t ss_tab[8] = {0, 8, 12, 0, 16, 20, 24, 32};
    uint32_t bps = ss_code == 0 ? fi.bps : ss_tab[ss_code];
    if (bps != fi.bps) { rc = BL_UNEXPECTED; break; } /* reserved code or a mid-stream change */
    /* frames follow each other without gaps: a frame lost to damage (its header no longer
     * passes the CRC-8 and was skipped as a false sync) shows up here */
    if (coded != (variable_blocks ? frames_done : frames_seen)) { rc = BL_UNEXPECTED; break; }
    uint32_t nch = chan < 8 ? chan + 1 : 2;
    if (nch != fi.channels) { rc = BL_UNEXPECTED; break; }
    if (bps == 32 && chan >= 8) {
      /* a side channel of a 32-bit stream has 33 bits: beyond the 32-bit reader and sample
       * buffers of this decoder.  Refused, not decoded to wrong samples (the frame CRCs would
       * still pass). */
      fprintf(stderr, "bliss_amd: 32-bit FLAC with inter-channel decorrelation is not supported\n");
      rc = BL_UNEXPECTED;
      break;
    }
    int bad = 0;
    for (uint32_t c = 0; c < nch && !bad; ++c) {
      uint32_t cb = bps;
      if ((chan == 8 && c == 1) || (chan == 9 && c == 0) || (chan == 10 && c == 1)) cb += 1;
      bad = flac_subframe(&b, ch[c], blocksize, cb);
    }
    if (bad) { rc = BL_UNEXPECTED; break; }
    br_align(&b);
    {
      const size_t body_end = br_bytepos(&b);
      const uint32_t crc16 = br_bits(&b, 16);
      if (b.err || body_end > len || flac_crc16(d + pos, body_end - pos) != crc16) { rc = BL_UNEXPECTED; break; }
    }
    pos = br_bytepos(&b);
    ++frames_seen;
    frames_done += blocksize;
    if (sink_reserve(sink, (size_t)blocksize * nch)) { rc = BL_UNEXPECTED; break; }
    for (uint32_t i = 0; i < blocksize; ++i) {
      int32_t l = ch[0][i], r = nch == 2 ? ch[1][i] : 0;
      if (chan == 8) r = l - r;                 /* left/side  */
      else if (chan == 9) l = l + r;            /* side/right */
      else if (chan == 10) {                    /* mid/side   */
        int32_t side = r, mid = (int32_t)(((uint32_t)l << 1) | ((uint32_t)side & 1));
        l = (mid + side) >> 1;
        r = (mid - side) >> 1;
      }
      sink_put(sink, l);
      if (nch == 2) sink_put(sink, r);
      if (sig) { /* little-endian, (bps + 7) / 8 bytes per sample, interleaved */
        uint8_t raw[8];
        const uint32_t nb = (bps + 7) / 8;
        for (uint32_t k = 0; k < nb; ++k) {
          raw[k] = (uint8_t)((uint32_t)l >> (8 * k));
          raw[nb + k] = (uint8_t)((uint32_t)r >> (8 * k));
        }
        md5_update(&sig->md, raw, nb * nch);
      }
    }
  }
  free(ch[0]);
  const size_t n = sink->n;
  if (rc != BL_OK || n == 0) return BL_UNEXPECTED;
  if (fi.total && frames_done != fi.total) return BL_UNEXPECTED; /* truncated, or frames lost at the end */
  song->nSamples = (int)n;
  song->channels = (int)fi.channels;
  song->sample_rate = (int)fi.rate;
  song->nb_bytes_per_sample = 2;
  song->duration = (uint64_t)(n / fi.channels) / fi.rate;
  {
    /* libavformat reports bit_rate = file bits / stream duration for FLAC
     * (ref src/decode.c:229 copies it; tests/test_analyze.c:40 pins 233864). */
    double secs = (double)(n / fi.channels) / (double)fi.rate;
    (void)audio_start;
    song->bitrate = secs > 0 ? (int)((double)len * 8.0 / secs) : 0;
  }
  return BL_OK;
}

static int wants_rate_conversion(uint32_t rate, int native) {
  return rate != BL_DECODE_RATE && !native;
}

/* ref include/bliss.h:234-235 / src/decode.c:27-213 */
int bl_audio_decode(char const *const filename, struct bl_song *const song) {
  uint8_t *data = NULL;
  size_t len = 0;
  /* the reference NULLs / fills every pointer field itself: callers pass
   * uninitialised structs (ref tests/test_analyze.c:27-28) */
  song->sample_array = NULL;
  song->artist = song->title = song->album = song->tracknumber = song->genre = NULL;
  song->filename = NULL;
  song->resampled = 0;
  song->calm_or_loud = 0;
  song->force = 0;
  memset(&song->force_vector, 0, sizeof(song->force_vector));
  if (!filename || read_file(filename, &data, &len) != 0) {
    fprintf(stderr, "Couldn't open file: %s\n", filename ? filename : "(null)");
    return BL_UNEXPECTED;
  }
  int rc = BL_UNEXPECTED;
  pcm_sink sink;
  memset(&sink, 0, sizeof sink);
  sink.native = native_rate_allowed();
  if (len >= 4 && !memcmp(data, "fLaC", 4)) rc = decode_flac(data, len, song, &sink, NULL);
  else if (len >= 12 && !memcmp(data, "RIFF", 4)) rc = decode_wav(data, len, song, &sink);
  else fprintf(stderr, "Unsupported container (WAV integer PCM / FLAC only): %s\n", filename);
  free(data);
  if (rc == BL_OK && wants_rate_conversion((uint32_t)song->sample_rate, sink.native)) {
    /* ref src/decode.c:317-346: anything that is not 22 050 Hz s16 goes through the rate
     * converter and comes out as 22 050 Hz stereo s16 */
    int16_t *out = NULL;
    size_t out_frames = 0;
    const size_t frames = sink.n / (size_t)song->channels;
    rc = bl_resample_to_stereo_s16(sink.wide ? (const void *)sink.p32 : (const void *)sink.p16,
                                   sink.wide ? (sink.is_float ? 2 : 1) : 0, frames, song->channels, song->sample_rate,
                                   BL_DECODE_RATE, &out, &out_frames);
    if (rc == BL_OK && (out_frames == 0 || out_frames * 2 > (size_t)INT32_MAX)) {
      free(out);
      rc = BL_UNEXPECTED;
    }
    if (rc != BL_OK) {
      fprintf(stderr, "bliss_amd: could not convert %s from %d Hz to %d Hz\n", filename,
              song->sample_rate, BL_DECODE_RATE);
    } else {
      sink_free(&sink);
      song->sample_array = (int8_t *)out;
      song->nSamples = (int)(out_frames * 2);
      song->channels = 2;
      song->sample_rate = BL_DECODE_RATE;
      song->resampled = 1;
    }
  } else if (rc == BL_OK && sink.native) { /* the caller's own business: as decoded */
    song->sample_array = (int8_t *)sink.p16;
    sink.p16 = NULL;
  } else if (rc == BL_OK) {
    /* 22 050 Hz already.  ref src/decode.c:312-346: a source whose sample format is not S16 still
     * goes through libswresample (resampled = 1): no filter at equal rates, but the format
     * conversion — S32 -> S16 is >> 16, FLT -> S16 lrintf(x * 2^15) clipped, both done while the
     * samples were stored — and, for a mono source, the up-mix to the stereo output layout with
     * gain 1/sqrt(2): in float for wide sources (S32 -> FLT * 2^-31, times the float coefficient,
     * FLT -> S16), in Q15 for an 8-bit one.  Third-party arithmetic (libswresample), restated
     * from its published conversion functions: parity unpinned — the reference holds no vector
     * of a same-rate non-S16 file. */
    if (sink.non_s16) song->resampled = 1;
    if (sink.non_s16 && song->channels == 1) {
      const size_t n = sink.n;
      int16_t *o = n <= (size_t)INT32_MAX / 2 ? (int16_t *)malloc(2 * n * sizeof(int16_t) + 16) : NULL;
      if (!o) {
        rc = BL_UNEXPECTED;
      } else {
        const float g = (float)M_SQRT1_2;
        for (size_t i = 0; i < n; ++i) {
          int16_t v;
          if (sink.wide) {
            float x;
            if (sink.is_float) {
              memcpy(&x, &sink.p32[i], sizeof x);
              if (!(x == x) || x > 4.0f || x < -4.0f) x = x > 0 ? 4.0f : (x < 0 ? -4.0f : 0.0f);
            } else {
              x = (float)sink.p32[i] * (1.0f / 2147483648.0f);
            }
            const long r = lrintf(x * g * 32768.0f);
            v = (int16_t)(r > 32767 ? 32767 : r < -32768 ? -32768 : r);
          } else {
            v = (int16_t)(((int32_t)sink.p16[i] * 23170 + 16384) >> 15);
          }
          o[2 * i] = o[2 * i + 1] = v;
        }
        song->sample_array = (int8_t *)o;
        song->nSamples = (int)(2 * n);
      }
    } else {
      song->sample_array = (int8_t *)sink.p16;
      sink.p16 = NULL;
    }
    /* ref src/decode.c:191-193 reports two channels whatever the file has — true of everything that
     * went through the converter (stereo output layout).  The one case it is not true of is a MONO
     * S16 file at 22 050 Hz: no converter runs, and the reference's append_buffer_to_song() copies
     * 2 x nb_samples x 2 bytes out of a frame that holds half of that (ref :353-354,368-370) — it reads
     * past the decoder's buffer, so there is no defined reference result to reproduce.  Such a file
     * keeps channels = 1 here and goes through the analyzers' mono branch. */
    if (sink.non_s16 || song->channels == 2) song->channels = 2;
  }
  sink_free(&sink);
  if (rc != BL_OK) {
    free(song->artist); free(song->title); free(song->album);
    free(song->tracknumber); free(song->genre);
    song->artist = song->title = song->album = song->tracknumber = song->genre = NULL;
    return BL_UNEXPECTED;
  }
  song->filename = strdup(filename);
  /* defaults of ref src/decode.c:263-308 */
  if (!song->artist) song->artist = strdup("<no artist>");
  if (!song->title) song->title = strdup("<no title>");
  if (!song->album) song->album = strdup("<no album>");
  if (!song->tracknumber) song->tracknumber = strdup("");
  if (!song->genre) song->genre = strdup("<no genre>");
  song->tracknumber[strcspn(song->tracknumber, "/")] = '\0'; /* ref src/decode.c:267 */
  return BL_OK;
}

/* include/bliss_amd.h: decode a FLAC file and compare the MD5 of the decoded samples (native
 * width, before the narrowing to s16) with the signature of the unencoded audio stored in its
 * STREAMINFO block.  Returns 1 when they match, 0 when they differ, BL_UNEXPECTED when the file
 * cannot be decoded; both digests are returned when the pointers are non-NULL. */
int bl_amd_flac_verify(const char *filename, uint8_t computed[16], uint8_t stored[16]) {
  uint8_t *data = NULL;
  size_t len = 0;
  if (!filename || read_file(filename, &data, &len) != 0) return BL_UNEXPECTED;
  struct bl_song song;
  memset(&song, 0, sizeof song);
  flac_sig sig;
  md5_init(&sig.md);
  memset(sig.stored, 0, sizeof sig.stored);
  pcm_sink sink;
  memset(&sink, 0, sizeof sink);
  const int rc = decode_flac(data, len, &song, &sink, &sig);
  free(data);
  sink_free(&sink);
  free(song.artist); free(song.title); free(song.album); free(song.tracknumber); free(song.genre);
  if (rc != BL_OK) return BL_UNEXPECTED;
  uint8_t got[16];
  md5_final(&sig.md, got);
  if (computed) memcpy(computed, got, 16);
  if (stored) memcpy(stored, sig.stored, 16);
  return memcmp(got, sig.stored, 16) == 0;
}
