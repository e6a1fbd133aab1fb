/* The C caller of INTEGRATION.md section 4 for a corpus resident in the GPUs' memory (BASELINE
 * configs[2] from C), compiled against include/bliss_amd.h alone: C99, no HIP headers.  Built with
 * -fsyntax-only on the CPU (tests/test_abi.py) — it needs device pointers to run. */
#include <stdlib.h>

#include <bliss_amd.h>

int analyze_resident_corpus(int n_gpus, int songs_per_gpu, const int16_t *const *d_pcm,
                            const bl_amd_song_desc *const *desc, float *const *d_rows,
                            bl_amd_song_result *results /* n_gpus * songs_per_gpu, shard-major */) {
  bl_amd_shard sh[16];
  if (n_gpus < 1 || n_gpus > 16) return BL_UNEXPECTED;
  for (int r = 0; r < n_gpus; ++r) {
    sh[r] = (bl_amd_shard){.device = r, .n_songs = songs_per_gpu, .d_pcm = d_pcm[r], .h_desc = desc[r],
                           .d_results = NULL, .d_rows = d_rows ? d_rows[r] : NULL};
  }
  /* every rank analyses its arena in place, one all-gather of the 16-byte force vectors, rank r
   * leaves the rows of its own songs against all N in d_rows[r] */
  return bl_amd_analyze_corpus_multi_device(sh, n_gpus, BL_AMD_MULTI_GATHER_RCCL, results, NULL);
}
