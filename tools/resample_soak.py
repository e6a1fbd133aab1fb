#!/usr/bin/env python
"""Randomised parity soak of the device rate converter against the host form (the one pinned on
the reference's digests): random input rates (standard and odd), sample kinds, channel counts,
lengths around filter / tile boundaries, random arena offsets, several songs per call.
Prints one JSON object; any mismatch is listed (and the exit code is 1)."""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

RATES = [44100, 48000, 88200, 96000, 32000, 24000, 16000, 11025, 8000, 12000, 64000, 176400, 192000,
         37800, 44056, 47952, 22254, 50000, 44099, 22051, 30000, 21000]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--calls", type=int, default=120)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    import torch
    import bliss_amd
    lib = bliss_amd.load()
    assert lib.bl_amd_init(0) == 0
    rng = np.random.default_rng(args.seed)
    songs = frames_total = 0
    bad = []
    per_rate = {}
    for call in range(args.calls):
        rate = int(RATES[call % len(RATES)]) if call < 3 * len(RATES) else int(rng.integers(6000, 200000))
        kind = np.int32 if rng.random() < 0.5 else np.int16
        n = int(rng.integers(1, 7))
        taps = int(np.ceil(32 / min(22050 * 0.97 / rate, 1.0))) + 2
        per_out = rate / 22050
        lens, chans, pcms = [], [], []
        for _ in range(n):
            mode = rng.integers(0, 4)
            if mode == 0:
                fr = taps + int(rng.integers(0, 40))                       # barely longer than the filter
            elif mode == 1:
                fr = int(per_out * 1024 * rng.integers(1, 4)) + int(rng.integers(-3, 4))   # tile edges
            elif mode == 2:
                fr = int(per_out * 147 * 64 * rng.integers(1, 3)) + int(rng.integers(-5, 6))   # phase-cycle tiles
            else:
                fr = int(rng.integers(taps, 60000))
            fr = max(fr, taps)
            ch = int(rng.integers(1, 3))
            if kind == np.int16:
                x = rng.integers(-32768, 32768, fr * ch).astype(np.int16)
            else:
                x = rng.integers(-(1 << 31), 1 << 31, fr * ch).astype(np.int32)
            lens.append(fr); chans.append(ch); pcms.append(x)
        total = sum((p.size + 7) & ~7 for p in pcms)
        arena = np.zeros(total, dtype=kind)
        off = 0
        for p in pcms:
            arena[off:off + p.size] = p
            off += (p.size + 7) & ~7
        out, placed = bliss_amd.resample_batch_device(torch.from_numpy(arena).cuda(), lens, chans, rate)
        torch.cuda.synchronize()
        host = out.cpu().numpy()
        for i, (p, ch) in enumerate(zip(pcms, chans)):
            want = bliss_amd.resample_host(p, ch, rate)
            got = host[placed[i][0]:placed[i][0] + placed[i][1]]
            songs += 1
            frames_total += want.size // 2
            if want.size != got.size or not np.array_equal(want, got):
                bad.append(dict(call=call, rate=rate, kind=str(np.dtype(kind)), frames=lens[i], channels=ch))
        per_rate[rate] = per_rate.get(rate, 0) + n
    print(json.dumps(dict(tool="resample_soak", seed=args.seed, calls=args.calls, songs=songs,
                          output_frames=frames_total, distinct_rates=len(per_rate), mismatches=len(bad),
                          mismatch_list=bad[:20])))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
