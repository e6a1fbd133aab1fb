#!/usr/bin/env python
"""Throughput of the device rate converter (bl_amd_resample_batch_device): `--songs` songs of
`--seconds` s stereo at each input rate / sample kind, resident in HBM, one call per batch.
Prints one JSON object; algorithmic bytes = input bytes read once + output bytes written once."""
import argparse
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--songs", type=int, default=1024)
    ap.add_argument("--seconds", type=int, default=180)
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    import torch
    import bliss_amd
    from bliss_amd import _lib
    lib = bliss_amd.load()
    assert lib.bl_amd_init(0) == 0
    rows = []
    for rate, kind in ((44100, "s16"), (48000, "s16"), (44100, "s32"), (48000, "s32"), (88200, "s16"), (88200, "s32"), (96000, "s32")):
        frames = rate * args.seconds
        dt = torch.int16 if kind == "s16" else torch.int32
        per_song = (2 * frames + 7) & ~7
        d_in = torch.randint(-20000, 20000, (per_song * args.songs,), dtype=torch.int32, device="cuda")
        d_in = (d_in.to(dt) if kind == "s16" else d_in * 65536)
        of = lib.bl_amd_resample_out_frames(frames, rate)
        out_per_song = (2 * of + 7) & ~7
        d_out = torch.zeros(out_per_song * args.songs + 64, dtype=torch.int16, device="cuda")
        desc = (_lib.ResampleDesc * args.songs)()
        for i in range(args.songs):
            desc[i].in_offset, desc[i].out_offset = i * per_song, i * out_per_song
            desc[i].frames, desc[i].channels = frames, 2
        s = torch.cuda.current_stream().cuda_stream
        best = None
        for _ in range(args.reps + 1):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            assert lib.bl_amd_resample_batch_device(d_in.data_ptr(), int(kind == "s32"), desc, args.songs, rate,
                                                    d_out.data_ptr(), C.c_void_p(s)) == 0
            b.record()
            torch.cuda.synchronize()
            ms = a.elapsed_time(b)
            best = ms if best is None else min(best, ms)
        in_bytes = args.songs * frames * 2 * (2 if kind == "s16" else 4)
        out_bytes = args.songs * of * 4
        rows.append(dict(in_rate=rate, kind=kind, ms=round(best, 2), songs_per_s=round(args.songs / best * 1e3, 1),
                         out_frames_per_s=round(args.songs * of / best * 1e3 / 1e9, 2),
                         algorithmic_GBps=round((in_bytes + out_bytes) / best / 1e6, 1)))
        del d_in, d_out
        torch.cuda.empty_cache()
    print(json.dumps(dict(tool="resample_bench", songs=args.songs, seconds=args.seconds, channels=2,
                          unit_out_frames="G frames/s", rows=rows)))


if __name__ == "__main__":
    main()
