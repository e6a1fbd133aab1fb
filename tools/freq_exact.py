#!/usr/bin/env python3
"""`frequency` (and freq_peak, force) of the HIP path against the CPU oracle BIT FOR BIT: since round 6 the kernel's f32 DFT
evaluates libavcodec's operation order node for node (bl_fft_lavc.h), as the oracle does (orc_fft_lavc.c).  Random synthetic
songs of mixed length, mono and stereo, plus the reference's recording.  Prints one JSON object.
usage: python tools/freq_exact.py [--songs 48] [--seed 1] [--max-seconds 40]"""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--songs", type=int, default=48)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--max-seconds", type=int, default=40)
    a = ap.parse_args()
    import bliss_amd
    from bliss_amd import _lib
    from tests.oracle_py import Oracle
    lib = bliss_amd.load()
    assert lib.bl_amd_init(0) == 0
    orc = Oracle()
    rng = np.random.default_rng(a.seed)
    rate = 22050
    secs = rng.integers(3, a.max_seconds + 1, a.songs)
    ch = rng.integers(1, 3, a.songs)
    extra = rng.integers(0, 1024, a.songs)
    lengths = [int(rate * c * s + e) for c, s, e in zip(ch, secs, extra)]
    corpus = bliss_amd.DeviceCorpus(lengths, ch.tolist(), secs.tolist())
    corpus.synth(seed_base=900000 + 1000 * a.seed, sample_rate=rate)
    corpus.analyze()
    got = corpus.fetch()
    pcm = corpus.pcm.cpu().numpy()
    out = {"songs": a.songs, "differing": {"frequency": 0, "freq_peak": 0, "force": 0}, "worst_abs_diff_frequency": 0.0, "examples": []}
    for i in range(a.songs):
        o = int(corpus.desc[i].pcm_offset)
        ref = orc.analyze(pcm[o:o + lengths[i]], int(ch[i]), int(secs[i]))
        for k in ("frequency", "freq_peak", "force"):
            g, r = np.float32(got[i][k]), np.float32(ref[k])
            if g.view(np.int32) != r.view(np.int32):
                out["differing"][k] += 1
                if len(out["examples"]) < 6:
                    out["examples"].append({"song": i, "field": k, "gpu": float(g), "oracle": float(r), "channels": int(ch[i]), "n": lengths[i]})
        out["worst_abs_diff_frequency"] = max(out["worst_abs_diff_frequency"], abs(float(got[i]["frequency"]) - float(ref["frequency"])))
    # the reference's recording through bl_analyze
    song = _lib.BlSong()
    path = os.path.join(ROOT, "tests", "golden", "song.flac")
    assert lib.bl_analyze(path.encode(), C.byref(song)) in (0, 1)
    out["song_flac"] = {"frequency": "%.6f" % song.force_vector.frequency, "force": "%.6f" % song.force,
                        "golden": {"frequency": "-10.136086", "force": "-20.777929"}}
    lib.bl_free_song(C.byref(song))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
