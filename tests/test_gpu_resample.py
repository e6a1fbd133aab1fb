"""Device rate converter (bl_amd_resample_batch_device, bl_rs_kernels.hip) against the host form
(bl_amd_resample_host), which is the one pinned on the reference's digests (tests/test_ingest.py,
ref tests/test_decode.c:35-36,55-56): bit for bit, both sample kinds, every code path of the kernel
(bank in LDS / in global memory, one phase / many, down- and up-sampling, mono up-mix, song edges,
tile boundaries), then end to end into the analysis."""
import ctypes as C

import numpy as np
import pytest

import bliss_amd
from bliss_amd import _lib

pytestmark = pytest.mark.gpu


def _songs(rng, rate, kind, lengths, channels):
    out = []
    for n, ch in zip(lengths, channels):
        t = np.arange(n)
        base = 9000 * np.sin(2 * np.pi * 330 * t / rate) + 2500 * np.sin(2 * np.pi * 4100 * t / rate)
        sig = np.stack([base + rng.integers(-700, 700, n) for _ in range(ch)], axis=1)
        sig[:3] = [[32767] * ch, [-32768] * ch, [12345] * ch]     # full scale at the reflected edge
        sig[-2:] = [[-32768] * ch, [32767] * ch]
        pcm = np.clip(sig, -32768, 32767).astype(np.int64).reshape(-1)
        if kind == "s32":
            pcm = (pcm << 16) + rng.integers(0, 1 << 16, pcm.size)
            out.append(np.clip(pcm, -(1 << 31), (1 << 31) - 1).astype(np.int32))
        else:
            out.append(pcm.astype(np.int16))
    return out


def _device_convert(songs, channels, rate):
    import torch
    total = sum((s.size + 7) & ~7 for s in songs)
    arena = np.zeros(total, dtype=songs[0].dtype)
    off = 0
    for s in songs:
        arena[off:off + s.size] = s
        off += (s.size + 7) & ~7
    d_in = torch.from_numpy(arena).cuda()
    out, placed = bliss_amd.resample_batch_device(d_in, [s.size // c for s, c in zip(songs, channels)], channels, rate)
    torch.cuda.synchronize()
    host = out.cpu().numpy()
    return [host[o:o + n] for o, n in placed], out, placed


@pytest.mark.parametrize("rate,kind", [
    (44100, "s16"), (44100, "s32"),       # one phase, 66 taps: the common case
    (48000, "s16"), (48000, "s32"),       # 147 phases, bank in LDS
    (32000, "s32"), (96000, "s16"),       # 441 / 147 phases, longer filters
    (8000, "s16"), (11025, "s32"),        # up-sampling (factor 1, 32 taps)
    (88200, "s16"), (88200, "s32"),       # one phase, step 4, 132 taps
    (192000, "s32"), (176400, "s16"),     # bank too large for the LDS: read through L1/L2
    (44099, "s16"), (22051, "s32"), (12345, "s16"),   # no small rational: 1 024 phases, fractional stepping
])
def test_device_equals_host_bit_for_bit(gpu_lib, rate, kind):
    rng = np.random.default_rng(rate + (kind == "s32"))
    per_out = rate / 22050
    lengths = [int(1024 * per_out * 3 + 11), int(1024 * per_out) + 1, int(1023 * per_out), 700, int(5000 * per_out)]
    channels = [2, 1, 2, 2, 1]
    songs = _songs(rng, rate, kind, lengths, channels)
    got, _, placed = _device_convert(songs, channels, rate)
    for i, (s, ch) in enumerate(zip(songs, channels)):
        want = bliss_amd.resample_host(s, ch, rate)
        assert want.size == placed[i][1] == 2 * gpu_lib.bl_amd_resample_out_frames(s.size // ch, rate)
        assert np.array_equal(got[i], want), (rate, kind, i, int(np.argmax(got[i] != want)))
        if ch == 1:
            assert np.array_equal(got[i][0::2], got[i][1::2])


def test_converted_batch_feeds_the_analysis(gpu_lib, oracle):
    """48 kHz s16 songs in HBM -> device converter -> bl_amd_analyze_batch_device on the converter's
    output arena; the oracle analyses the host-converted PCM of the same songs."""
    import torch
    rng = np.random.default_rng(5)
    rate, secs = 48000, 12
    songs = []
    for seed in (3, 4, 5):
        s = oracle.synth(seed, rate, 2, rate * 2 * secs)
        songs.append(s)
    channels = [2, 2, 2]
    got, d_out, placed = _device_convert(songs, channels, rate)
    n = len(songs)
    desc = (_lib.SongDesc * n)()
    for i, (off, ns) in enumerate(placed):
        desc[i].pcm_offset, desc[i].n_samples, desc[i].channels, desc[i].duration = off, ns, 2, secs
    res = torch.zeros(n * C.sizeof(_lib.SongResult), dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    assert gpu_lib.bl_amd_analyze_batch_device(d_out.data_ptr(), desc, n, res.data_ptr(), C.c_void_p(stream)) == 0
    torch.cuda.synchronize()
    r = bliss_amd.results_to_numpy(res.cpu().numpy().tobytes())
    for i, s in enumerate(songs):
        want_pcm = bliss_amd.resample_host(s, 2, rate)
        assert np.array_equal(got[i], want_pcm)
        o = oracle.analyze(want_pcm, 2, secs)
        assert r["status"][i] == 0
        for k in ("start", "end", "mean", "variance", "n_frames", "nb_frames", "beat"):
            assert int(r[k][i]) == int(o[k]), (i, k)
        for k in ("tempo", "amplitude", "frequency", "attack"):
            assert abs(float(r[k][i]) - o[k]) <= 1e-4 * max(1.0, abs(o[k])), (i, k)


@pytest.mark.parametrize("rate,kind", [(44100, "s16"), (48000, "s32")])
def test_host_batch_at_native_rate(gpu_lib, oracle, rate, kind):
    """bl_amd_analyze_batch_host_rate: host PCM at 44.1 / 48 kHz -> transfer, device conversion,
    analysis, in waves; against host converter + CPU oracle.  One mono song in the batch."""
    secs = 9
    songs, chans = [], [2, 1, 2]
    for i, ch in enumerate(chans):
        s = oracle.synth(20 + i, rate, ch, rate * ch * secs + 2 * i * ch).astype(np.int64)
        if kind == "s32":
            s = (s << 16) + (np.arange(s.size) * 2654435761 % 65536)
            songs.append(s.astype(np.int32))
        else:
            songs.append(s.astype(np.int16))
    r = bliss_amd.analyze_batch_host_rate(songs, chans, secs, rate)
    for i, (s, ch) in enumerate(zip(songs, chans)):
        pcm = bliss_amd.resample_host(s, ch, rate)
        o = oracle.analyze(pcm, 2, secs)
        assert r["status"][i] == 0
        for k in ("start", "end", "mean", "variance", "n_frames", "nb_frames", "beat"):
            assert int(r[k][i]) == int(o[k]), (rate, kind, i, k)
        for k in ("tempo", "amplitude", "frequency", "attack"):
            assert abs(float(r[k][i]) - o[k]) <= 1e-4 * max(1.0, abs(o[k])), (rate, kind, i, k)
    # 22 050 Hz through the same entry point is the plain host batch
    plain = [oracle.synth(30, 22050, 2, 22050 * 2 * 6)]
    a = bliss_amd.analyze_batch_host_rate(plain, 2, 6, 22050)
    b = bliss_amd.analyze_batch_host(plain, 2, 6)
    assert all(np.array_equal(a[k], b[k]) for k in a.dtype.names)
    # too short to convert / odd sample count for stereo: rejected before anything is staged
    ptr_bad = [np.zeros(40, np.int16)]
    with pytest.raises(RuntimeError):
        bliss_amd.analyze_batch_host_rate(ptr_bad, 2, 1, 44100)


def test_converter_rejects_bad_descriptors(gpu_lib):
    import torch
    d_in = torch.zeros(1 << 16, dtype=torch.int16, device="cuda")
    d_out = torch.zeros(1 << 16, dtype=torch.int16, device="cuda")
    ok = dict(in_offset=0, out_offset=0, frames=20000, channels=2)
    for bad in (dict(frames=10), dict(channels=3), dict(out_offset=1), dict(in_offset=3), dict(frames=0)):
        d = (_lib.ResampleDesc * 1)()
        for k, v in {**ok, **bad}.items():
            setattr(d[0], k, v)
        assert gpu_lib.bl_amd_resample_batch_device(d_in.data_ptr(), 0, d, 1, 44100, d_out.data_ptr(), None) \
            == _lib.BL_UNEXPECTED
    d = (_lib.ResampleDesc * 1)()
    for k, v in ok.items():
        setattr(d[0], k, v)
    assert gpu_lib.bl_amd_resample_batch_device(d_in.data_ptr(), 0, d, 1, 0, d_out.data_ptr(), None) == _lib.BL_UNEXPECTED
    assert gpu_lib.bl_amd_resample_batch_device(d_in.data_ptr(), 0, d, 1, 44100, d_out.data_ptr(), None) == _lib.BL_OK
    torch.cuda.synchronize()
