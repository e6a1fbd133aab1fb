"""Host ingest (bl_audio_decode) beyond the 16-bit fixture: 24-bit FLAC pinned by the STREAMINFO
signature of the reference's own audio/song_s32*.flac, the rate converter pinned by the digests the
reference's tests hold for the same two files, the S32 -> S16 narrowing, WAV 24/32-bit, and
malformed metadata.  No GPU involved."""
import ctypes as C
import hashlib
import os
import struct

import numpy as np
import pytest

from bliss_amd import _lib

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")


def _crc8(data):
    c = 0
    for b in data:
        c ^= b
        for _ in range(8):
            c = ((c << 1) ^ 0x07) & 0xFF if c & 0x80 else (c << 1) & 0xFF
    return c


def _crc16(data):
    c = 0
    for b in data:
        c ^= b << 8
        for _ in range(8):
            c = ((c << 1) ^ 0x8005) & 0xFFFF if c & 0x8000 else (c << 1) & 0xFFFF
    return c


def write_flac_verbatim(path, samples, channels, rate, bps, comment_block=None, stereo_mode=None):
    """Minimal FLAC writer (VERBATIM subframes, one frame per 4096 samples): test input for the
    decoder, bit depth 16, 24 or 32; independent channels, or (stereo_mode="left_side") the left
    channel and a side channel left - right of bps + 1 bits."""
    samples = np.asarray(samples, dtype=np.int64).reshape(-1, channels)
    total = samples.shape[0]
    nb = bps // 8
    raw = b"".join(int(v).to_bytes(nb, "little", signed=True) for v in samples.reshape(-1))
    md5 = hashlib.md5(raw).digest()
    info = struct.pack(">HH", 4096, 4096) + b"\0\0\0" + b"\0\0\0"
    packed = (rate << 44) | ((channels - 1) << 41) | ((bps - 1) << 36) | total
    info += packed.to_bytes(8, "big") + md5
    blocks = [(0, info)]
    if comment_block is not None:
        blocks.append((4, comment_block))
    out = bytearray(b"fLaC")
    for i, (t, body) in enumerate(blocks):
        last = 0x80 if i == len(blocks) - 1 else 0
        out += bytes([last | t]) + len(body).to_bytes(3, "big") + body
    ss_code = {16: 4, 24: 6, 32: 7}[bps]
    for fno, start in enumerate(range(0, total, 4096)):
        blk = samples[start:start + 4096]
        bs = blk.shape[0]
        hdr = bytearray([0xFF, 0xF8])
        hdr.append((7 << 4) | 0)                       # blocksize: 16-bit field follows; rate: from STREAMINFO
        hdr.append(((8 if stereo_mode == "left_side" else channels - 1) << 4) | (ss_code << 1))
        assert fno < 128
        hdr.append(fno)                                # UTF-8 coded frame number (1 byte)
        hdr += (bs - 1).to_bytes(2, "big")
        hdr.append(_crc8(hdr))
        bits = []
        for c in range(channels):
            bits.append("0" + "000001" + "0")          # padding, VERBATIM, no wasted bits
            side = stereo_mode == "left_side" and c == 1
            w = bps + 1 if side else bps
            for v in (blk[:, 0] - blk[:, 1] if side else blk[:, c]):
                bits.append(format(int(v) & ((1 << w) - 1), "0%db" % w))
        s = "".join(bits)
        s += "0" * (-len(s) % 8)
        frame = bytes(hdr) + int(s, 2).to_bytes(len(s) // 8, "big")
        out += frame + _crc16(frame).to_bytes(2, "big")
    open(path, "wb").write(bytes(out))
    return md5


def _decode(lib, path):
    song = _lib.BlSong()
    rc = lib.bl_audio_decode(str(path).encode(), C.byref(song))
    if rc != _lib.BL_OK:
        return rc, None, None
    pcm = np.ctypeslib.as_array(C.cast(song.sample_array, C.POINTER(C.c_int16)), shape=(song.nSamples,)).copy()
    meta = dict(channels=song.channels, rate=song.sample_rate, duration=song.duration,
                nb=song.nb_bytes_per_sample, resampled=song.resampled, title=song.title)
    lib.bl_free_song(C.byref(song))
    return rc, pcm, meta


@pytest.mark.parametrize("name,stored", [
    ("song.flac", "8a1bd824951c0433cc47fec5bf41d0a9"),      # == ref tests/test_decode.c:16-17
    ("song_s32.flac", None), ("song_s32_mono.flac", None)])
def test_flac_decoder_matches_streaminfo_signature(lib, name, stored):
    """The MD5 of the decoded samples at native width equals the signature the encoder stored:
    pins the 24-bit path of the decoder on the reference's own fixtures."""
    got = (C.c_uint8 * 16)()
    want = (C.c_uint8 * 16)()
    assert lib.bl_amd_flac_verify(os.path.join(GOLD, name).encode(), got, want) == 1
    assert bytes(got) == bytes(want) and any(bytes(want))
    if stored:
        assert bytes(got).hex() == stored


# ref tests/test_decode.c:35-36,55-56: MD5 of the 22 050 Hz stereo s16 audio the reference's decoder
# (libswresample behind it) produces for its two 48 kHz / 24-bit fixtures
RESAMPLED_MD5 = {"song_s32.flac": "eb9f31a7b9ed022d66ff82b76e7c3c18",
                 "song_s32_mono.flac": "747dbfcd75bebc23ebe2024935aede36"}


@pytest.mark.parametrize("name", sorted(RESAMPLED_MD5))
def test_rate_conversion_reproduces_reference_digest(lib, name):
    """48 kHz / 24 bit -> 22 050 Hz stereo s16 (ref src/decode.c:317-346, 379-401): byte-identical
    to what the reference's tests pin, metadata as ref tests/test_analyze.c:69-88."""
    lib.bl_amd_decode_allow_native_rate(0)
    rc, pcm, meta = _decode(lib, os.path.join(GOLD, name))
    assert rc == _lib.BL_OK
    assert hashlib.md5(pcm.tobytes()).hexdigest() == RESAMPLED_MD5[name]
    assert pcm.size == 488140 and meta["channels"] == 2 and meta["rate"] == 22050
    assert meta["nb"] == 2 and meta["resampled"] == 1 and meta["duration"] == 11
    assert meta["title"] == b"Renaissance"
    if "mono" in name:
        assert np.array_equal(pcm[0::2], pcm[1::2])


def test_oracle_reproduces_reference_goldens_of_the_resampled_fixture(lib, oracle):
    """ref tests/test_analyze.c:59-68 (EPSILON 1e-5): second pin of the oracle, on the output of the
    rate converter."""
    rc, pcm, meta = _decode(lib, os.path.join(GOLD, "song_s32.flac"))
    assert rc == _lib.BL_OK
    r = oracle.analyze(pcm, 2, meta["duration"])
    gold = dict(force=-20.821571, tempo=-8.218182, amplitude=-10.641695, frequency=-10.179875,
                attack=-15.561186)
    for k, want in gold.items():
        assert abs(r[k] - want) <= 1e-5, (k, r[k], want)
        assert "%.6f" % float(np.float32(r[k])) == "%.6f" % want, (k, float(np.float32(r[k])), want)   # last printed digit


def test_native_rate_opt_in(lib):
    """With the opt-in the file is handed over at its own rate, narrowed, not converted."""
    p = os.path.join(GOLD, "song_s32.flac")
    lib.bl_amd_decode_allow_native_rate(1)
    try:
        rc, pcm, meta = _decode(lib, p)
        assert rc == _lib.BL_OK and meta["rate"] == 48000 and meta["channels"] == 2
        assert meta["nb"] == 2 and meta["resampled"] == 0 and meta["duration"] == 11
        assert pcm.size == 2 * 531307 and meta["title"] == b"Renaissance"
        rc, mono, meta = _decode(lib, os.path.join(GOLD, "song_s32_mono.flac"))
        assert rc == _lib.BL_OK and meta["channels"] == 1 and mono.size == 531307
    finally:
        lib.bl_amd_decode_allow_native_rate(0)


def _write_wav16(path, pcm, channels, rate):
    import wave
    with wave.open(str(path), "wb") as w:
        w.setnchannels(channels); w.setsampwidth(2); w.setframerate(rate)
        w.writeframes(np.asarray(pcm, dtype=np.int16).tobytes())


def _kaiser_bank(in_rate, out_rate=22050):
    """The converter's filter design restated in numpy (double), for the s16 checks below:
    Kaiser(9)-windowed sinc, cutoff 0.97, ceil(32 / factor) taps made even, one row per phase of
    out/in in lowest terms, normalised by the DC gain of phase 0."""
    from math import gcd
    factor = min(out_rate * 0.97 / in_rate, 1.0)
    taps = int(np.ceil(32 / factor)); taps += taps & 1
    phases = out_rate // gcd(out_rate, in_rate)
    assert phases <= 1024
    center = (taps - 1) // 2
    i = np.arange(taps)[None, :]
    ph = np.arange(phases)[:, None]
    x = np.pi * ((i - center) - ph / phases) * factor
    y = np.where(x == 0, 1.0, np.sin(x) / np.where(x == 0, 1.0, x))
    w = 2.0 * x / (factor * taps * np.pi)
    y = y * np.i0(9.0 * np.sqrt(np.maximum(1 - w * w, 0)))
    return y / y[0].sum(), center, phases


@pytest.mark.parametrize("in_rate", [44100, 48000, 32000, 96000])
def test_s16_rate_conversion_against_numpy_restatement(lib, tmp_path, in_rate):
    """16-bit sources take the integer path (Q15 coefficients, int32 accumulator, rounding
    (v + 2^14) >> 15): compared sample for sample with a numpy restatement away from the edges."""
    from math import gcd
    rng = np.random.default_rng(in_rate)
    frames = 3 * in_rate + 17
    t = np.arange(frames)
    sig = (9000 * np.sin(2 * np.pi * 440 * t / in_rate) + 3000 * np.sin(2 * np.pi * 5000 * t / in_rate)
           + rng.integers(-500, 500, frames)).astype(np.int16)
    stereo = np.stack([sig, (sig // 2 + 100).astype(np.int16)], axis=1)
    p = tmp_path / "s.wav"
    _write_wav16(p, stereo.reshape(-1), 2, in_rate)
    rc, pcm, meta = _decode(lib, p)
    assert rc == _lib.BL_OK and meta["rate"] == 22050 and meta["resampled"] == 1 and meta["channels"] == 2
    out = pcm.reshape(-1, 2)
    assert abs(out.shape[0] - frames * 22050 / in_rate) <= 1.0
    bank, center, phases = _kaiser_bank(in_rate)
    q = np.clip(np.rint((bank * 32768).astype(np.float32)), -32768, 32767).astype(np.int64)
    g = gcd(22050, in_rate)
    step = in_rate // g                       # input advance per output, in units of 1/phases sample
    taps = bank.shape[1]
    for n in (40, 41, 1000, 12345, out.shape[0] - 60):
        pos = n * step
        s0, phs = pos // phases, pos % phases
        for c in range(2):
            win = stereo[s0 - center:s0 - center + taps, c].astype(np.int64)
            want = int(np.clip(((win * q[phs]).sum() + (1 << 14)) >> 15, -32768, 32767))
            assert out[n, c] == want, (in_rate, n, c)


def test_rate_conversion_properties(lib, tmp_path):
    """DC in -> the same DC out (coefficients sum to one), a mono source comes out as two equal
    channels at gain 1/sqrt(2) (ref src/decode.c:338: out layout stereo), length follows the ratio."""
    frames = 44100 * 2
    p = tmp_path / "dc.wav"
    _write_wav16(p, np.full(2 * frames, 12000, np.int16), 2, 44100)
    rc, pcm, meta = _decode(lib, p)
    assert rc == _lib.BL_OK and pcm.size == 2 * (frames // 2)
    assert np.all(np.abs(pcm.astype(int) - 12000) <= 1)
    _write_wav16(p, np.full(frames, 12000, np.int16), 1, 44100)
    rc, pcm, meta = _decode(lib, p)
    assert rc == _lib.BL_OK and meta["channels"] == 2 and pcm.size == 2 * (frames // 2)
    assert np.array_equal(pcm[0::2], pcm[1::2])
    assert np.all(np.abs(pcm.astype(int) - round(12000 / np.sqrt(2))) <= 1)
    # too short to fill the filter once: fails, does not crash
    _write_wav16(p, np.zeros(2 * 20, np.int16), 2, 44100)
    rc, _, _ = _decode(lib, p)
    assert rc == _lib.BL_UNEXPECTED


def test_24bit_flac_is_narrowed_with_shift(lib, tmp_path):
    rng = np.random.default_rng(3)
    s24 = rng.integers(-(1 << 23), 1 << 23, 2 * 9000)
    s24[:4] = [(1 << 23) - 1, -(1 << 23), -1, 255]
    p = tmp_path / "v24.flac"
    md5 = write_flac_verbatim(p, s24, 2, 22050, 24)
    rc, pcm, meta = _decode(lib, p)
    assert rc == _lib.BL_OK and meta["nb"] == 2 and meta["rate"] == 22050
    # left-justified in 32 bits, arithmetic >> 16  ==  24-bit value >> 8
    assert np.array_equal(pcm, ((s24 << 8) >> 16).astype(np.int16))
    got = (C.c_uint8 * 16)()
    assert lib.bl_amd_flac_verify(str(p).encode(), got, None) == 1 and bytes(got) == md5
    s16 = rng.integers(-32768, 32768, 2 * 5000)
    p16 = tmp_path / "v16.flac"
    write_flac_verbatim(p16, s16, 2, 22050, 16)
    rc, pcm, _ = _decode(lib, p16)
    assert rc == _lib.BL_OK and np.array_equal(pcm, s16.astype(np.int16))


def test_wav_24_and_32_bit(lib, tmp_path):
    rng = np.random.default_rng(4)
    n = 2 * 7000
    s32 = rng.integers(-(1 << 31), 1 << 31, n, dtype=np.int64)
    for bits in (24, 32):
        vals = s32 >> (32 - bits)
        raw = b"".join(int(v).to_bytes(bits // 8, "little", signed=True) for v in vals)
        fmt = struct.pack("<HHIIHH", 1, 2, 22050, 22050 * 2 * bits // 8, 2 * bits // 8, bits)
        body = b"WAVE" + b"fmt " + struct.pack("<I", len(fmt)) + fmt + b"data" + struct.pack("<I", len(raw)) + raw
        p = tmp_path / f"w{bits}.wav"
        p.write_bytes(b"RIFF" + struct.pack("<I", len(body)) + body)
        rc, pcm, meta = _decode(lib, p)
        assert rc == _lib.BL_OK and meta["channels"] == 2
        assert np.array_equal(pcm, (vals >> (bits - 16)).astype(np.int16))


def _wav(path, fmt_tag, channels, rate, bits, raw):
    fmt = struct.pack("<HHIIHH", fmt_tag, channels, rate, rate * channels * bits // 8, channels * bits // 8, bits)
    body = b"WAVE" + b"fmt " + struct.pack("<I", len(fmt)) + fmt + b"data" + struct.pack("<I", len(raw)) + raw
    path.write_bytes(b"RIFF" + struct.pack("<I", len(body)) + body)


def test_wav_float_and_8_bit(lib, tmp_path):
    """RIFF format tag 3 (32-bit float, full scale +-1) and unsigned 8-bit PCM.  Same rate: the
    sample-format conversions lrintf(x * 2^15) clipped / (v - 128) << 8; another rate: through the
    converter's float / s16 path like any other source of that width."""
    rng = np.random.default_rng(12)
    x = np.concatenate([rng.uniform(-1.2, 1.2, 2 * 9000), [1.0, -1.0, 0.999999, np.nan, np.inf, -np.inf, 0.0, 3e-5]]).astype(np.float32)
    p = tmp_path / "f.wav"
    _wav(p, 3, 2, 22050, 32, x.tobytes())
    rc, pcm, meta = _decode(lib, p)
    # not an S16 source: the reference routes it through libswresample even at 22 050 Hz (ref
    # src/decode.c:312-321), so resampled = 1 although no rate changes
    assert rc == _lib.BL_OK and meta["rate"] == 22050 and meta["resampled"] == 1 and meta["channels"] == 2
    clean = np.where(np.isnan(x), 0.0, np.clip(x, -4.0, 4.0)).astype(np.float32)
    want = np.clip(np.rint(clean * np.float32(32768.0)), -32768, 32767).astype(np.int16)
    assert np.array_equal(pcm, want)
    u8 = rng.integers(0, 256, 2 * 8000).astype(np.uint8)
    _wav(p, 1, 2, 22050, 8, u8.tobytes())
    rc, pcm, _ = _decode(lib, p)
    assert rc == _lib.BL_OK and np.array_equal(pcm, ((u8.astype(np.int32) - 128) * 256).astype(np.int16))
    # another rate: float WAV == the same samples as left-justified int32 through the float path, when
    # the floats are exactly k * 2^-31 multiples (then both conversions start from identical floats)
    k = (rng.integers(-(1 << 23), 1 << 23, 2 * 30000).astype(np.int64) << 8)
    f = (k.astype(np.float64) / 2147483648.0).astype(np.float32)
    _wav(p, 3, 2, 44100, 32, f.tobytes())
    rc, a, meta = _decode(lib, p)
    assert rc == _lib.BL_OK and meta["rate"] == 22050 and meta["resampled"] == 1
    _wav(p, 1, 2, 44100, 32, k.astype(np.int32).tobytes())
    rc, b, _ = _decode(lib, p)
    assert rc == _lib.BL_OK and np.array_equal(a, b)
    _wav(p, 1, 2, 44100, 8, u8.tobytes())
    rc, c, _ = _decode(lib, p)
    import bliss_amd
    assert rc == _lib.BL_OK and np.array_equal(c, bliss_amd.resample_host(((u8.astype(np.int32) - 128) * 256).astype(np.int16), 2, 44100))


def test_same_rate_sources_follow_the_reference_layout(lib, tmp_path):
    """ref src/decode.c:191-193,312-346 at 22 050 Hz: a source that is not S16 goes through the
    converter — it counts as resampled, reports two channels, and a MONO one is up-mixed to stereo
    with gain 1/sqrt(2) (float for wide sources, Q15 for 8 bit) — libswresample's arithmetic
    restated, parity unpinned; a mono S16 file, for which the reference's copy loop is undefined,
    stays mono."""
    rng = np.random.default_rng(21)
    n = 9000
    g = np.float32(np.sqrt(0.5))
    # mono 24-bit FLAC
    s24 = rng.integers(-(1 << 23), 1 << 23, n)
    p = tmp_path / "m24.flac"
    write_flac_verbatim(p, s24, 1, 22050, 24)
    rc, pcm, meta = _decode(lib, p)
    assert rc == _lib.BL_OK and meta["channels"] == 2 and meta["resampled"] == 1 and pcm.size == 2 * n
    x = (s24.astype(np.int64) << 8).astype(np.float32) * np.float32(1.0 / 2147483648.0)
    want = np.clip(np.rint((x * g) * np.float32(32768.0)), -32768, 32767).astype(np.int16)
    assert np.array_equal(pcm[0::2], want) and np.array_equal(pcm[1::2], want)
    # mono float WAV
    f = rng.uniform(-1.1, 1.1, n).astype(np.float32)
    w = tmp_path / "m.wav"
    _wav(w, 3, 1, 22050, 32, f.tobytes())
    rc, pcm, meta = _decode(lib, w)
    want = np.clip(np.rint((f * g) * np.float32(32768.0)), -32768, 32767).astype(np.int16)
    assert rc == _lib.BL_OK and meta["resampled"] == 1 and np.array_equal(pcm[0::2], want) and np.array_equal(pcm[1::2], want)
    # mono 8-bit WAV: Q15 up-mix of (v - 128) << 8
    u8 = rng.integers(0, 256, n).astype(np.uint8)
    _wav(w, 1, 1, 22050, 8, u8.tobytes())
    rc, pcm, meta = _decode(lib, w)
    v = (u8.astype(np.int32) - 128) * 256
    want = ((v * 23170 + 16384) >> 15).astype(np.int16)
    assert rc == _lib.BL_OK and meta["resampled"] == 1 and np.array_equal(pcm[0::2], want) and np.array_equal(pcm[1::2], want)
    # mono S16: the reference over-reads its decode buffer for such a file (ref src/decode.c:353-370), so
    # there is nothing defined to mirror: untouched samples, one channel, not resampled
    s16 = rng.integers(-32768, 32768, n)
    write_flac_verbatim(p, s16, 1, 22050, 16)
    rc, pcm, meta = _decode(lib, p)
    assert rc == _lib.BL_OK and meta["channels"] == 1 and meta["resampled"] == 0
    assert np.array_equal(pcm, s16.astype(np.int16))
    # stereo 24-bit at 22 050 Hz: >> 16 of the left-justified word, resampled = 1
    s24 = rng.integers(-(1 << 23), 1 << 23, 2 * n)
    write_flac_verbatim(p, s24, 2, 22050, 24)
    rc, pcm, meta = _decode(lib, p)
    assert rc == _lib.BL_OK and meta["resampled"] == 1 and np.array_equal(pcm, ((s24 << 8) >> 16).astype(np.int16))


def test_32bit_flac_side_channel_is_refused(lib, tmp_path, capfd):
    """A 32-bit stereo FLAC frame with inter-channel decorrelation carries a 33-bit side channel:
    refused loudly (the frame CRCs would pass on wrongly decoded samples); independent channels
    at 32 bits decode."""
    rng = np.random.default_rng(22)
    s32 = rng.integers(-(1 << 31), 1 << 31, 2 * 5000)
    p = tmp_path / "v32.flac"
    md5 = write_flac_verbatim(p, s32, 2, 22050, 32)
    rc, pcm, meta = _decode(lib, p)
    assert rc == _lib.BL_OK and np.array_equal(pcm, (s32 >> 16).astype(np.int16))
    got = (C.c_uint8 * 16)()
    assert lib.bl_amd_flac_verify(str(p).encode(), got, None) == 1 and bytes(got) == md5
    write_flac_verbatim(p, s32, 2, 22050, 32, stereo_mode="left_side")
    capfd.readouterr()
    rc, pcm, meta = _decode(lib, p)
    assert rc == _lib.BL_UNEXPECTED
    assert "32-bit FLAC with inter-channel decorrelation" in capfd.readouterr().err
    # the same coding at 24 bits (a 25-bit side channel) decodes to the right samples
    s24 = rng.integers(-(1 << 23), 1 << 23, 2 * 5000)
    md5 = write_flac_verbatim(p, s24, 2, 22050, 24, stereo_mode="left_side")
    rc, pcm, meta = _decode(lib, p)
    assert rc == _lib.BL_OK and np.array_equal(pcm, ((s24 << 8) >> 16).astype(np.int16))
    assert lib.bl_amd_flac_verify(str(p).encode(), got, None) == 1 and bytes(got) == md5


def test_malformed_vorbis_comment_lengths(lib, tmp_path):
    """Comment / vendor lengths near 2^32 must not wrap the bounds checks (heap over-read)."""
    s16 = np.zeros(2 * 4096, dtype=np.int64)
    s16[::3] = 1000
    for vendor_len, comment_len in ((0, 0xFFFFFFFF), (0xFFFFFFF0, 5), (0, 0xFFFFFFF9), (3, 0x80000000)):
        blk = struct.pack("<I", vendor_len) + b"abc"[: min(vendor_len, 3)] + struct.pack("<I", 2)
        blk += struct.pack("<I", comment_len) + b"TITLE=x" + struct.pack("<I", 7) + b"GENRE=y"
        p = tmp_path / "bad.flac"
        write_flac_verbatim(p, s16, 2, 22050, 16, comment_block=blk)
        rc, pcm, meta = _decode(lib, p)             # tags are dropped, the audio still decodes
        assert rc == _lib.BL_OK and pcm.size == s16.size
    good = struct.pack("<I", 3) + b"abc" + struct.pack("<I", 1) + struct.pack("<I", 9) + b"title=Hey"
    p = tmp_path / "good.flac"
    write_flac_verbatim(p, s16, 2, 22050, 16, comment_block=good)
    rc, _, meta = _decode(lib, p)
    assert rc == _lib.BL_OK and meta["title"] == b"Hey"


def test_damaged_flac_frame_is_rejected(lib, tmp_path):
    """One flipped bit inside a frame's audio data breaks the frame's CRC-16; inside a frame header
    it breaks the CRC-8, the header is skipped as a false sync and the next frame's number no longer
    follows; a truncated file ends short of STREAMINFO's sample count.  In every case the decode
    fails instead of analysing altered or partial audio."""
    flac = bytearray(open(os.path.join(GOLD, "song.flac"), "rb").read())
    rc, ref, _ = _decode(lib, os.path.join(GOLD, "song.flac"))
    assert rc == _lib.BL_OK
    rng = np.random.default_rng(9)
    for trial in range(12):
        data = bytearray(flac)
        pos = int(rng.integers(20000, len(data) - 1000))   # well inside the audio frames
        data[pos] ^= 1 << int(rng.integers(0, 8))
        p = tmp_path / f"d{trial}.flac"
        p.write_bytes(bytes(data))
        rc, pcm, _ = _decode(lib, p)
        assert rc == _lib.BL_UNEXPECTED, (trial, pos)
    p = tmp_path / "short.flac"
    p.write_bytes(bytes(flac[: len(flac) * 2 // 3]))
    assert _decode(lib, p)[0] == _lib.BL_UNEXPECTED


def test_decoder_survives_corrupted_files(lib, tmp_path):
    """Byte flips, truncations and garbage tails of a real FLAC and of a WAV: bl_audio_decode may
    fail or succeed, but it returns (no crash, no hang) and leaves a struct that bl_free_song
    accepts."""
    rng = np.random.default_rng(77)
    flac = bytearray(open(os.path.join(GOLD, "song.flac"), "rb").read())
    wav_pcm = rng.integers(-3000, 3000, 2 * 6000).astype(np.int16)
    wav = bytearray(b"RIFF" + struct.pack("<I", 36 + wav_pcm.nbytes) + b"WAVE" + b"fmt " +
                    struct.pack("<IHHIIHH", 16, 1, 2, 22050, 22050 * 4, 4, 16) + b"data" +
                    struct.pack("<I", wav_pcm.nbytes) + wav_pcm.tobytes())
    ok = 0
    for trial in range(120):
        src = flac if trial % 3 else wav
        data = bytearray(src)
        kind = trial % 4
        if kind == 0:      # flips in the header / metadata region
            for _ in range(8):
                data[int(rng.integers(0, min(len(data), 9000)))] ^= int(rng.integers(1, 256))
        elif kind == 1:    # flips anywhere
            for _ in range(40):
                data[int(rng.integers(0, len(data)))] ^= int(rng.integers(1, 256))
        elif kind == 2:    # truncation
            data = data[: int(rng.integers(5, len(data)))]
        else:              # a length field blown up
            pos = int(rng.integers(4, 60))
            data[pos:pos + 4] = b"\xff\xff\xff\x7f"
        p = tmp_path / f"c{trial}.bin"
        p.write_bytes(bytes(data))
        song = _lib.BlSong()
        rc = lib.bl_audio_decode(str(p).encode(), C.byref(song))
        assert rc in (_lib.BL_OK, _lib.BL_UNEXPECTED)
        if rc == _lib.BL_OK:
            ok += 1
            assert song.nSamples > 0 and song.channels in (1, 2) and song.sample_array
            lib.bl_free_song(C.byref(song))
    assert ok > 0
