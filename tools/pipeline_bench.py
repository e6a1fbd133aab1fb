#!/usr/bin/env python
"""A collection at its native rate, resident in HBM: device rate conversion + analysis, songs/s.
`--songs` synthetic stereo s16 songs of `--seconds` s at `--rate` Hz (the synth of bench.py run at
that rate), converted by bl_amd_resample_batch_device into the arena bl_amd_analyze_batch_device
reads; `--verify K` songs are converted on the host and analysed by the CPU oracle."""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--songs", type=int, default=2048)
    ap.add_argument("--seconds", type=int, default=180)
    ap.add_argument("--rate", type=int, default=44100)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--verify", type=int, default=2)
    args = ap.parse_args()
    import numpy as np
    import torch
    import bliss_amd
    from bliss_amd import _lib
    lib = bliss_amd.load()
    assert lib.bl_amd_init(0) == 0
    n, rate = args.songs, args.rate
    frames = rate * args.seconds
    per_in = (2 * frames + 7) & ~7
    of = lib.bl_amd_resample_out_frames(frames, rate)
    per_out = (2 * of + 7) & ~7
    d_in = torch.zeros(per_in * n + 64, dtype=torch.int16, device="cuda")
    d_out = torch.zeros(per_out * n + 64, dtype=torch.int16, device="cuda")
    res = torch.zeros(n * C.sizeof(_lib.SongResult), dtype=torch.uint8, device="cuda")
    syn = (_lib.SongDesc * n)()
    rd = (_lib.ResampleDesc * n)()
    sd = (_lib.SongDesc * n)()
    for i in range(n):
        syn[i].pcm_offset, syn[i].n_samples, syn[i].channels, syn[i].duration = i * per_in, 2 * frames, 2, args.seconds
        rd[i].in_offset, rd[i].out_offset, rd[i].frames, rd[i].channels = i * per_in, i * per_out, frames, 2
        sd[i].pcm_offset, sd[i].n_samples, sd[i].channels, sd[i].duration = i * per_out, 2 * of, 2, args.seconds
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert lib.bl_amd_synth_pcm_device(d_in.data_ptr(), syn, n, 1000, rate, s) == 0
    torch.cuda.synchronize()
    t_conv = t_all = None
    for _ in range(args.reps + 1):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record()
        assert lib.bl_amd_resample_batch_device(d_in.data_ptr(), 0, rd, n, rate, d_out.data_ptr(), s) == 0
        e[1].record()
        assert lib.bl_amd_analyze_batch_device(d_out.data_ptr(), sd, n, res.data_ptr(), s) == 0
        e[2].record()
        torch.cuda.synchronize()
        a, b = e[0].elapsed_time(e[1]), e[0].elapsed_time(e[2])
        if t_all is None or b < t_all:
            t_conv, t_all = a, b
    r = bliss_amd.results_to_numpy(res.cpu().numpy().tobytes())
    ok = bool((r["status"] == 0).all())
    verified = 0
    if args.verify:
        from tests.oracle_py import Oracle
        orc = Oracle()
        for i in range(min(args.verify, n)):
            src = orc.synth(1000 + i, rate, 2, 2 * frames)
            assert np.array_equal(src, d_in[i * per_in:i * per_in + 2 * frames].cpu().numpy())
            pcm = bliss_amd.resample_host(src, 2, rate)
            assert np.array_equal(pcm, d_out[i * per_out:i * per_out + 2 * of].cpu().numpy()), "converter mismatch"
            o = orc.analyze(pcm, 2, args.seconds)
            for k in ("start", "end", "mean", "variance", "n_frames", "nb_frames", "beat"):
                assert int(r[k][i]) == int(o[k]), (i, k)
            for k in ("tempo", "amplitude", "frequency", "attack"):
                assert abs(float(r[k][i]) - o[k]) <= 1e-4 * max(1.0, abs(o[k])), (i, k)
            verified += 1
    print(json.dumps(dict(tool="pipeline_bench", songs=n, seconds=args.seconds, in_rate=rate, kind="s16 stereo",
                          parity="parity unpinned (s16 path): the converter's Q15 arithmetic has no reference vector",
                          convert_ms=round(t_conv, 2), analyze_ms=round(t_all - t_conv, 2), total_ms=round(t_all, 2),
                          songs_per_s=round(n / t_all * 1e3, 1), results_ok=ok, verified_songs=verified,
                          verification="host converter + CPU oracle; converter output bit-exact, integers exact, floats 1e-4 rel")))


if __name__ == "__main__":
    main()
