"""Parity of the HIP path (through the C-ABI) with the CPU oracle on the same inputs.
Bar: integer quantities bit-exact; f32 features within 1e-4 relative (north_star);
in practice tempo/amplitude/attack are expected to be bit-identical and asserted so."""
import ctypes as C
import hashlib
import os

import numpy as np
import pytest

import bliss_amd
from bliss_amd import _lib

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
INTS = ("start", "end", "mean", "variance", "n_frames", "nb_frames", "n_windows", "beat",
        "calm_or_loud")
FLOATS = ("tempo", "amplitude", "frequency", "attack", "force")
REL = 1e-4  # north_star tolerance for f32 features


def check_song(got, ref, tag):
    for k in INTS:
        assert int(got[k]) == int(ref[k]), (tag, k, int(got[k]), int(ref[k]))
    for k in FLOATS:
        a, b = float(got[k]), float(ref[k])
        assert abs(a - b) <= REL * max(abs(b), 1e-6), (tag, k, a, b)
    # the exactly-ordered parts of the path: bit-identical expected
    assert np.float32(got["amplitude"]) == np.float32(ref["amplitude"]), (tag, "amplitude bits")
    assert np.float32(got["tempo"]) == np.float32(ref["tempo"]), (tag, "tempo bits")
    # since round 6 that includes the frequency analysis: the kernel's f32 DFT evaluates libavcodec's operation order
    # node for node (bl_fft_lavc.h), as the oracle does (orc_fft_lavc.c) — every frame's power spectrum, their ordered
    # f32 sum, the peak, the rating and with it the force
    for k in ("frequency", "freq_peak", "force"):
        assert np.float32(got[k]).view(np.int32) == np.float32(ref[k]).view(np.int32), (tag, k + " bits", float(got[k]), float(ref[k]))
    assert abs(float(got["atk_sum"]) - ref["atk_sum"]) <= 1e-9 * abs(ref["atk_sum"]), (tag, "atk_sum")
    assert int(got["status"]) == 0


CASES = [  # (seed, rate, channels, seconds, extra_samples)
    (11, 22050, 2, 11, 0),
    (12, 44100, 2, 10, 0),
    (13, 44100, 1, 12, 0),
    (14, 22050, 1, 20, 333),     # n not a multiple of 8 / 512
    (15, 22050, 2, 9, 1022),
    (16, 8000, 1, 1, 0),         # 8000 samples: N = 30
    (17, 5120, 1, 1, 0),         # shortest input the reference supports (N = 20)
    (18, 48000, 2, 15, 6),
]


@pytest.fixture(scope="module")
def batch(gpu_lib, oracle):
    lengths = [r * c * s + e for (_, r, c, s, e) in CASES]
    chans = [c for (_, _, c, _, _) in CASES]
    durs = [s for (_, _, _, s, _) in CASES]
    pcms = [oracle.synth(seed, r, c, n) for (seed, r, c, _, _), n in zip(CASES, lengths)]
    # ragged edits the reference's trim / wrap corners are sensitive to
    pcms[1][:777] = 0                 # leading digital silence  -> start = 777...
    pcms[1][-4321:] = 0               # trailing silence
    pcms[2][5000:9000] = 0            # silence in the middle (zero bin, not trimmed)
    # DC offset on a long song: bl_mean's int32 accumulator wraps (ref src/helpers.c:31-36)
    pcms[4] = (pcms[4].astype(np.int32) // 2 + 15000).astype(np.int16)
    # DC offset on a short song: |mean| > 13571 -> int32 (v*v) wraps, k_variance_wrap path
    pcms[5] = (pcms[5].astype(np.int32) // 2 + 15000).astype(np.int16)
    corpus = bliss_amd.DeviceCorpus(lengths, chans, durs)
    for i, p in enumerate(pcms):
        corpus.upload(i, p)
    corpus.analyze()
    return corpus.fetch(), pcms, chans, durs


def test_batch_matches_oracle(batch, oracle):
    got, pcms, chans, durs = batch
    for i, p in enumerate(pcms):
        ref = oracle.analyze(p, chans[i], durs[i])
        check_song(got[i], ref, f"case{i}")
    assert int(got[1]["start"]) >= 777 and int(got[1]["end"]) <= len(pcms[1]) - 4322
    assert int(got[4]["mean"]) == 4206          # the wrapped int32 sum, as the reference computes it
    assert abs(int(got[5]["mean"])) > 13571     # exercised k_variance_wrap


def test_host_batch_equals_device_batch(batch, gpu_lib):
    got, pcms, chans, durs = batch
    host = bliss_amd.analyze_batch_host(pcms, chans, durs)
    for k in got.dtype.names:
        assert np.array_equal(got[k], host[k]), k


def test_device_synth_is_byte_identical(gpu_lib, oracle):
    lengths = [22050 * 2 * 3 + 5, 44100 * 1 * 2]
    corpus = bliss_amd.DeviceCorpus(lengths, [2, 1], [3, 2])
    corpus.synth(seed_base=40, sample_rate=22050)
    pcm = corpus.pcm.cpu().numpy()
    for i, n in enumerate(lengths):
        o = int(corpus.desc[i].pcm_offset)
        assert np.array_equal(pcm[o:o + n], oracle.synth(40 + i, 22050, [2, 1][i], n))


def test_reference_golden_through_bl_analyze(gpu_lib):
    """ref tests/test_analyze.c:26-57 run against the drop-in library."""
    song = _lib.BlSong()  # deliberately not initialised further (ref :27-28)
    rc = gpu_lib.bl_analyze(os.path.join(HERE, "golden", "song.flac").encode(), C.byref(song))
    assert rc == _lib.BL_CALM
    gold = dict(force=-20.777929, tempo=-8.945454, amplitude=-10.641844, frequency=-10.136086,
                attack=-15.560563)
    assert abs(song.force - gold["force"]) <= 1e-5
    assert "%.6f" % song.force == "%.6f" % gold["force"]
    for k in ("tempo", "amplitude", "frequency", "attack"):
        assert abs(getattr(song.force_vector, k) - gold[k]) <= 1e-5, k
        # to the last digit the reference's test prints (round 6: all five, through the HIP path)
        assert "%.6f" % getattr(song.force_vector, k) == "%.6f" % gold[k], (k, getattr(song.force_vector, k))
    assert (song.channels, song.nSamples, song.sample_rate, song.bitrate,
            song.nb_bytes_per_sample, song.duration) == (2, 488138, 22050, 233864, 2, 11)
    n = song.nSamples
    pcm = np.ctypeslib.as_array(C.cast(song.sample_array, C.POINTER(C.c_int16)), shape=(n,))
    assert hashlib.md5(pcm.tobytes()).hexdigest() == "8a1bd824951c0433cc47fec5bf41d0a9"
    # single analyzers of the public API on the same song (ref python/bliss/bl_song.py:179-199)
    amp = gpu_lib.bl_amplitude_sort(C.byref(song))
    frq = gpu_lib.bl_frequency_sort(C.byref(song))
    env = _lib.EnvelopeResult()
    gpu_lib.bl_envelope_sort(C.byref(song), C.byref(env))
    assert amp == song.force_vector.amplitude and frq == song.force_vector.frequency
    assert env.tempo == song.force_vector.tempo and env.attack == song.force_vector.attack
    gpu_lib.bl_free_song(C.byref(song))


def test_reference_golden_s32_through_bl_analyze(gpu_lib):
    """ref tests/test_analyze.c:59-89 (test_analyze_s32): the 48 kHz / 24-bit fixture goes through
    the rate converter (digest of ref tests/test_decode.c:35-36) and the GPU analyzers."""
    song = _lib.BlSong()
    rc = gpu_lib.bl_analyze(os.path.join(HERE, "golden", "song_s32.flac").encode(), C.byref(song))
    assert rc == _lib.BL_CALM
    gold = dict(force=-20.821571, tempo=-8.218182, amplitude=-10.641695, frequency=-10.179875,
                attack=-15.561186)
    assert abs(song.force - gold["force"]) <= 1e-5
    assert "%.6f" % song.force == "%.6f" % gold["force"]
    for k in ("tempo", "amplitude", "frequency", "attack"):
        assert abs(getattr(song.force_vector, k) - gold[k]) <= 1e-5, k
        assert "%.6f" % getattr(song.force_vector, k) == "%.6f" % gold[k], (k, getattr(song.force_vector, k))
    assert (song.channels, song.nSamples, song.sample_rate, song.nb_bytes_per_sample,
            song.duration, song.resampled) == (2, 488140, 22050, 2, 11, 1)
    assert (song.artist, song.title, song.album, song.tracknumber, song.genre) == \
        (b"David TMX", b"Renaissance", b"Renaissance", b"02", b"Pop")
    pcm = np.ctypeslib.as_array(C.cast(song.sample_array, C.POINTER(C.c_int16)), shape=(song.nSamples,))
    assert hashlib.md5(pcm.tobytes()).hexdigest() == "eb9f31a7b9ed022d66ff82b76e7c3c18"
    gpu_lib.bl_free_song(C.byref(song))


def test_distance_file_and_errors(gpu_lib, tmp_path):
    f = os.path.join(HERE, "golden", "song.flac").encode()
    s1, s2 = _lib.BlSong(), _lib.BlSong()
    d = gpu_lib.bl_distance_file(f, f, C.byref(s1), C.byref(s2))
    assert d == 0.0
    c = gpu_lib.bl_cosine_similarity_file(f, f, C.byref(s1), C.byref(s2))
    assert abs(c - 1.0) < 1e-6
    gpu_lib.bl_free_song(C.byref(s1)); gpu_lib.bl_free_song(C.byref(s2))
    # two different files (ref src/analyze.c:105-125,145-167): the 22.05 kHz fixture and its 48 kHz /
    # 24-bit sibling — both songs are filled, and the pair functions return exactly what the one-pair
    # functions give on the two force vectors (and what the oracle gives on them)
    g = os.path.join(HERE, "golden", "song_s32.flac").encode()
    from tests.oracle_py import Oracle
    orc = Oracle()
    d = gpu_lib.bl_distance_file(f, g, C.byref(s1), C.byref(s2))
    v1 = np.array([getattr(s1.force_vector, k) for k in ("tempo", "amplitude", "frequency", "attack")], dtype=np.float32)
    v2 = np.array([getattr(s2.force_vector, k) for k in ("tempo", "amplitude", "frequency", "attack")], dtype=np.float32)
    assert s1.nSamples == 488138 and s2.nSamples == 488140 and s2.resampled == 1
    assert d > 0 and d == gpu_lib.bl_distance(s1.force_vector, s2.force_vector) == orc.distance(v1, v2)
    assert abs(v2[0] - (-8.218182)) <= 1e-5 and abs(v1[0] - (-8.945454)) <= 1e-5      # the two goldens' tempi
    gpu_lib.bl_free_song(C.byref(s1)); gpu_lib.bl_free_song(C.byref(s2))
    c = gpu_lib.bl_cosine_similarity_file(g, f, C.byref(s1), C.byref(s2))
    assert 0.99 < c < 1.0 and c == gpu_lib.bl_cosine_similarity(s1.force_vector, s2.force_vector) == orc.cosine(v2, v1)
    gpu_lib.bl_free_song(C.byref(s1)); gpu_lib.bl_free_song(C.byref(s2))
    bad = str(tmp_path / "nope.flac").encode()
    assert gpu_lib.bl_analyze(bad, C.byref(s1)) == _lib.BL_UNEXPECTED
    assert gpu_lib.bl_distance_file(bad, f, C.byref(s1), C.byref(s2)) == float(_lib.BL_UNEXPECTED)


def test_all_zero_song_is_flagged(gpu_lib):
    res = bliss_amd.analyze_batch_host([np.zeros(22050 * 2, dtype=np.int16)], 2, 1)
    assert int(res[0]["status"]) == _lib.BL_UNEXPECTED  # reference: unbounded trim loop


def test_distance_matrix_bit_exact(gpu_lib, oracle):
    rng = np.random.default_rng(5)
    v = (rng.standard_normal((1000, 4)) * 10).astype(np.float32)
    dm = bliss_amd.distance_matrix(v)
    assert np.array_equal(dm, oracle.distance_matrix(v))
    cm = bliss_amd.cosine_matrix(v[:200])
    ref = np.array([[oracle.cosine(a, b) for b in v[:200]] for a in v[:200]], dtype=np.float32)
    assert np.array_equal(cm, ref)
    assert np.array_equal(cm, oracle.cosine_matrix(v[:200]))
    a, b = _lib.ForceVector(*v[0]), _lib.ForceVector(*v[1])
    assert gpu_lib.bl_distance(a, b) == oracle.distance(v[0], v[1]) == dm[0, 1]
    assert gpu_lib.bl_cosine_similarity(a, b) == oracle.cosine(v[0], v[1])


def test_distance_matrix_full_size_properties(gpu_lib):
    """BASELINE config 4 (N = 10 000): size-independent properties + sampled exactness."""
    rng = np.random.default_rng(6)
    v = (rng.standard_normal((10000, 4)) * 8).astype(np.float32)
    dm = bliss_amd.distance_matrix(v)
    assert dm.shape == (10000, 10000)
    assert np.array_equal(dm, dm.T) and not np.any(np.diag(dm))
    i = rng.integers(0, 10000, 2000); j = rng.integers(0, 10000, 2000)
    d = v[i] - v[j]
    s = d[:, 0] * d[:, 0]
    for k in (1, 2, 3):
        s = (s + d[:, k] * d[:, k]).astype(np.float32)
    assert np.array_equal(dm[i, j], np.sqrt(s).astype(np.float32))
    # triangle inequality on a sample (f32 slack)
    k = rng.integers(0, 10000, 2000)
    assert np.all(dm[i, j] <= dm[i, k] + dm[k, j] + 1e-3)


def test_helper_symbols(gpu_lib, oracle):
    pcm = oracle.synth(3, 22050, 2, 100001)
    pcm = (pcm.astype(np.int32) + 20000).clip(-32768, 32767).astype(np.int16)  # int32 v*v wrap
    p = pcm.ctypes.data_as(C.POINTER(C.c_int16))
    m = gpu_lib.bl_mean(p, pcm.size)
    assert m == oracle.mean(pcm)
    assert gpu_lib.bl_variance(p, pcm.size, m) == oracle.variance(pcm, m)
    rng = np.random.default_rng(1)
    inp = rng.standard_normal(500)
    old = rng.standard_normal(500)
    out = old.copy()
    dp = C.POINTER(C.c_double)
    gpu_lib.bl_rectangular_filter(out.ctypes.data_as(dp), inp.ctypes.data_as(dp), 500, 19)
    assert np.array_equal(out, oracle.rect_filter(old, inp, 19))


def test_window_energies_and_full_size_sample(gpu_lib, oracle):
    """BASELINE config 2 shape at reduced count (64 x 30 s, 44.1 kHz stereo): every song's
    integers + floats against the oracle for a sample, and batch-order independence."""
    n = 44100 * 2 * 30
    corpus = bliss_amd.DeviceCorpus([n] * 64, 2, 30)
    corpus.synth(seed_base=1000, sample_rate=44100)
    corpus.analyze()
    got = corpus.fetch()
    assert np.all(got["status"] == 0) and np.all(got["n_windows"] == 10332)
    pcm = corpus.pcm.cpu().numpy()
    for i in (0, 17, 63):
        o = int(corpus.desc[i].pcm_offset)
        ref = oracle.analyze(pcm[o:o + n], 2, 30)
        check_song(got[i], ref, f"s30[{i}]")
    # same songs analysed alone give identical records (no cross-song state)
    solo = bliss_amd.DeviceCorpus([n], 2, 30)
    solo.synth(seed_base=1017, sample_rate=44100)
    solo.analyze()
    one = solo.fetch()[0]
    for k in got.dtype.names:
        assert one[k] == got[17][k], k


def test_mixed_length_corpus(gpu_lib, oracle):
    """BASELINE configs[4] shape: lengths from 10 s to 10 min, mono and stereo, s16 and
    s32 sources.  An s32 source reaches the hot path as s16 through the library's own narrowing
    (arithmetic >> 16, bl_amd_narrow_s32_device / bl_amd_analyze_batch_host_s32 — the same-rate
    S32->S16 conversion of the reference's resampler, parity unpinned for that step); the oracle
    is given numpy's >> 16 of the same words.  Songs are analysed in one batch in caller order;
    internally they are processed longest first."""
    rng = np.random.default_rng(9)
    rate = 44100
    secs = [10, 600, 37, 75, 12, 240, 51, 18, 133, 10, 29, 64]
    chans = [2, 2, 1, 2, 1, 1, 2, 2, 1, 1, 2, 1]
    lengths = [s * rate * c + int(rng.integers(0, 3000)) for s, c in zip(secs, chans)]
    corpus = bliss_amd.DeviceCorpus(lengths, chans, secs)
    corpus.synth(seed_base=7000, sample_rate=rate)
    pcm = corpus.pcm.cpu().numpy()
    songs, wide = [], []
    for i, n in enumerate(lengths):
        o = int(corpus.desc[i].pcm_offset)
        s16 = pcm[o:o + n].copy()
        assert np.array_equal(s16[:4096], oracle.synth(7000 + i, rate, chans[i], 4096))
        s32 = s16.astype(np.int32) << 16
        if i % 2:  # a genuine 32-bit source: the low half carries data that must be dropped
            s32 |= rng.integers(0, 65536, n, dtype=np.int32)
            corpus.upload_s32(i, s32)
            s16 = (s32 >> 16).astype(np.int16)
        songs.append(s16)
        wide.append(s32)
    corpus.analyze()
    got = corpus.fetch()
    for i, s16 in enumerate(songs):
        ref = oracle.analyze(s16, chans[i], secs[i])
        check_song(got[i], ref, f"mixed[{i}] {secs[i]}s x{chans[i]}")
    # the same corpus through the host-pointer entry points (pinned staging, two streams)
    host = bliss_amd.analyze_batch_host(songs, chans, secs)
    host32 = bliss_amd.analyze_batch_host_s32(wide, chans, secs)
    for k in got.dtype.names:
        assert np.array_equal(got[k], host[k]), k
        assert np.array_equal(got[k], host32[k]), k


def test_quiet_sparse_and_clipped_material_bit_for_bit(gpu_lib, oracle):
    """Material at the edges of the f32 transform's range, every feature held to the oracle's bits (check_song): a song
    a few LSB loud (frames of tiny values: the power terms sit ~20 orders of magnitude below a loud song's, nowhere near
    the subnormals), isolated impulses in digital silence (almost every frame all zero, single Hann-weighted samples
    otherwise), full-scale square waves (the largest sums the transform can see), and a mono and a stereo song whose
    length leaves every remainder of the 64-frame iteration."""
    rate = 22050
    rng = np.random.default_rng(66)
    songs, chans, secs = [], [], []
    base = oracle.synth(6601, rate, 2, rate * 2 * 14)
    songs.append((base.astype(np.int32) // 900).astype(np.int16)); chans.append(2); secs.append(14)       # |s| <= 10
    sparse = np.zeros(rate * 12, dtype=np.int16)
    sparse[rng.integers(0, sparse.size, 400)] = rng.integers(-30000, 30000, 400, dtype=np.int64).astype(np.int16)
    sparse[0], sparse[-1] = 5, -7
    songs.append(sparse); chans.append(1); secs.append(12)
    t = np.arange(rate * 2 * 9)
    sq = np.where((t // 37) % 2 == 0, 32767, -32768).astype(np.int16)
    sq[::1001] = 0   # a few quiet samples: the reference's histogram needs the central bins non-empty for a finite rating
    songs.append(sq); chans.append(2); secs.append(9)
    for k, ch in ((5, 1), (37, 2)):   # n_frames = 64 m + k
        n = (64 * 9 + k) * 512 * ch + 77
        songs.append(oracle.synth(6610 + k, rate, ch, n)); chans.append(ch); secs.append(max(1, n // (rate * ch)))
    corpus = bliss_amd.DeviceCorpus([len(x) for x in songs], chans, secs)
    for i, x in enumerate(songs):
        corpus.upload(i, x)
    corpus.analyze()
    got = corpus.fetch()
    for i, x in enumerate(songs):
        ref = oracle.analyze(x, chans[i], secs[i])
        assert int(ref["n_frames"]) == int(got[i]["n_frames"])
        check_song(got[i], ref, f"edge[{i}]")


def test_repeatability_and_order_independence(gpu_lib):
    """Run-to-run determinism and independence from the position in the batch (no float
    atomics, fixed-order partial sums)."""
    n = 44100 * 2 * 20
    a = bliss_amd.DeviceCorpus([n] * 8, 2, 20)
    a.synth(seed_base=300, sample_rate=44100)
    a.analyze(); r1 = a.fetch()
    a.analyze(); r2 = a.fetch()
    b = bliss_amd.DeviceCorpus([n // 2 + 8, n, n, 3 * n // 4], [1, 2, 2, 2], [20, 20, 20, 15])
    b.synth(seed_base=299, sample_rate=44100)  # song 1 of b == song 0 of a... different slot
    b.analyze(); rb = b.fetch()
    for k in r1.dtype.names:
        assert np.array_equal(r1[k], r2[k]), k
        assert r1[k][0] == rb[k][1], k


def test_playlist_matches_reference_recipe(gpu_lib, oracle):
    """ref python/examples/make_m3u_playlist.py:62-72 (numpy L2 from the seed + argsort)."""
    rng = np.random.default_rng(12)
    v = (rng.standard_normal((5000, 4)) * 6).astype(np.float32)
    v[100] = v[7]; v[4000] = v[7]          # exact ties
    order, dist = bliss_amd.playlist(v, 7)
    want = np.array([oracle.distance(v[7], x) for x in v], dtype=np.float32)
    assert np.array_equal(dist, want)
    assert np.array_equal(order, np.argsort(want, kind="stable"))
    assert order[0] == 7 and list(order[1:3]) == [100, 4000]


def test_python_surface_mirrors_reference(gpu_lib):
    """ref python/bliss/bl_song.py + distance.py usage patterns."""
    f = os.path.join(HERE, "golden", "song.flac")
    with bliss_amd.bl_song(f) as song:
        assert song["duration"] == 11 and song["artist"] == "David TMX"
        fv = song["force_vector"]
        assert set(fv) == {"tempo", "amplitude", "frequency", "attack"}
        assert abs(fv["tempo"] - (-8.945454)) <= 1e-5
        assert len(song) == 17 and "force_vector" in list(song)
        env = song.envelope_analysis()
        assert env["tempo"] == fv["tempo"] and env["attack"] == fv["attack"]
        assert song.amplitude_analysis() == fv["amplitude"]
        other = bliss_amd.bl_song(initializer={"force_vector": fv})
        assert bliss_amd.distance.distance(song, other)["distance"] == 0.0
        assert abs(bliss_amd.distance.cosine_similarity(song, other)["similarity"] - 1.0) < 1e-6
    d = bliss_amd.distance.distance(f, f)
    assert d["distance"] == 0.0 and d["song1"]["title"] == "Renaissance"
    d["song1"].free(); d["song2"].free()
    assert bliss_amd.distance.distance(1, 2) == {"distance": None, "song1": None, "song2": None}
    assert abs(bliss_amd.version.version() - 1.2) < 1e-6
    # module-level wrappers of ref python/bliss/bl_song.py:212-254 and `bliss.lib`
    import importlib
    bl_song_module = importlib.import_module("bliss_amd.bl_song")   # the package attribute is the class, as in the reference
    g = os.path.join(HERE, "golden", "song_s32.flac")
    d = bl_song_module.distance(f, g)
    c = bl_song_module.cosine_similarity(f, g)
    assert d["distance"] > 0 and d["distance"] == bliss_amd.distance.distance(d["song1"], d["song2"])["distance"]
    assert 0.99 < c["similarity"] < 1.0 and c["song2"]["resampled"] == 1
    for r in (d, c):
        r["song1"].free(); r["song2"].free()
    assert bliss_amd.lib.bl_version() == bliss_amd.load().bl_version()


def test_extreme_amplitudes(gpu_lib, oracle):
    """Full-scale clipping, a very quiet song and a DC-heavy one: large / tiny variances
    stress the normalisation (exact quotient), the f32-rounded energy sums and the
    amplitude histogram saturation paths."""
    rng = np.random.default_rng(21)
    n = 22050 * 2 * 12
    loud = (rng.integers(0, 2, n) * 2 - 1).astype(np.int16) * 32767          # +-32767 square noise
    loud[::7] = -32768
    quiet = rng.integers(-3, 4, n).astype(np.int16)                          # +-3 LSB
    quiet[0] = 1; quiet[-1] = -2
    sparse = np.zeros(n, dtype=np.int16)                                     # mostly digital silence
    sparse[1000:200000:37] = rng.integers(-20000, 20000, len(range(1000, 200000, 37))).astype(np.int16)
    songs = [loud, quiet, sparse]
    got = bliss_amd.analyze_batch_host(songs, 2, 12)
    for i, s16 in enumerate(songs):
        ref = oracle.analyze(s16, 2, 12)
        for k in INTS:
            assert int(got[i][k]) == int(ref[k]), (i, k, int(got[i][k]), int(ref[k]))
        for k in FLOATS:
            a, b = float(got[i][k]), float(ref[k])
            assert (np.isnan(a) and np.isnan(b)) or abs(a - b) <= REL * max(abs(b), 1e-6), (i, k, a, b)


def test_histogram_out_of_range_samples_are_dropped(gpu_lib, oracle):
    """k_pcm_scan counts a sample into the central histogram without a range test: an out-of-range sample becomes an
    LDS address beyond the workgroup's allocation, which the hardware discards (scan_hist_word).  Songs that sit on
    the edges of the range (-2049 / -2048 / 2047 / 2048), at the ends of the 16-bit range, and just inside it, odd
    lengths included (the samples behind the last whole vector take the tested path): `amplitude` and the histogram
    integral bit-identical to the oracle's, `start` / `end` (k_trim) exact with silence on both sides."""
    rng = np.random.default_rng(77)
    n = 22050 * 2 * 6
    edges = np.array([-32768, -2049, -2048, -2047, -1, 0, 1, 2046, 2047, 2048, 32767], dtype=np.int16)
    songs = [
        edges[rng.integers(0, len(edges), n)],                       # only edge values
        rng.integers(-2048, 2048, n + 5).astype(np.int16),           # everything in range, n % 8 = 5
        rng.integers(-32768, 32768, n + 3).astype(np.int16),         # 6 % in range
        np.where(rng.random(n + 7) < 0.5, rng.integers(-2100, 2100, n + 7), rng.integers(-32768, 32768, n + 7)).astype(np.int16),
    ]
    songs[0][:1031] = 0          # leading silence ends inside a vector
    songs[0][-2050:] = 0
    songs[1][:8] = 0             # exactly one zero vector
    songs[2][-3:] = 0            # the samples behind the last whole vector are the silence
    songs[3][-9:] = 0
    songs[3][-10] = 2048
    got = bliss_amd.analyze_batch_host(songs, 2, 6)
    for i, s16 in enumerate(songs):
        ref = oracle.analyze(s16, 2, 6)
        for k in INTS:
            assert int(got[i][k]) == int(ref[k]), (i, k, int(got[i][k]), int(ref[k]))
        assert np.float32(got[i]["amplitude"]) == np.float32(ref["amplitude"]), (i, "amplitude bits")
        assert np.float32(got[i]["hist_integral"]) == np.float32(ref["hist_integral"]), (i, "histogram integral bits")


def test_trim_search_edges(gpu_lib, oracle):
    """k_trim finds the first / last non-zero sample by a search from both ends (ref src/amplitude_sort.c:26-31):
    the sample at index 0, the very last one when it lies behind the last whole 16-byte vector, silence longer than
    several search steps on either side, and a single non-zero sample (first == last)."""
    rng = np.random.default_rng(5)
    n = 22050 * 2 * 3 + 5
    base = rng.integers(-9000, 9000, n).astype(np.int16)
    base[base == 0] = 7
    songs = []
    a = base.copy(); songs.append(a)                                  # non-zero from index 0 to n - 1
    b = base.copy(); b[:5000] = 0; b[-7000:] = 0; songs.append(b)      # several search steps of silence on both sides
    c = base.copy(); c[:n - 3] = 0; c[n - 3] = 0; c[n - 2] = 12000; c[n - 1] = 0; songs.append(c)  # one sample, behind the last vector
    d = np.zeros(n, dtype=np.int16); d[0] = -15000; d[n - 1] = 9000; songs.append(d)               # both ends only
    e = np.zeros(n, dtype=np.int16); e[4097] = 20000; songs.append(e)                              # one sample in the middle
    got = bliss_amd.analyze_batch_host(songs, 2, 3)
    for i, pcm in enumerate(songs):
        ref = oracle.analyze(pcm, 2, 3)
        for k in ("start", "end", "mean", "variance"):
            assert int(got[i][k]) == int(ref[k]), (i, k, int(got[i][k]), int(ref[k]))
    assert int(got[2]["start"]) == n - 2 and int(got[2]["end"]) == n - 2
    assert int(got[3]["start"]) == 0 and int(got[3]["end"]) == n - 1


def test_fused_statistics_pass_equals_the_separate_kernels(gpu_lib, oracle):
    """bl_analyze / the batch calls run k_freq_scan (statistics riding along with the frequency pass); the single
    analyzers of the reference's API (bl_amplitude_sort, bl_frequency_sort, bl_envelope_sort: ref
    python/bliss/bl_song.py:179-199) run k_pcm_scan and k_freq_frames.  Same bits from both, on mono and stereo
    buffers whose lengths leave every kind of remainder (behind the last frame, behind the last 16-byte vector) and
    with silence at both ends (k_trim)."""
    cases = [(31, 22050, 2, 7, 0), (32, 22050, 1, 9, 333), (33, 44100, 2, 3, 1022), (34, 11025, 1, 5, 7)]
    songs, chans = [], []
    for seed, rate, ch, secs, extra in cases:
        pcm = oracle.synth(seed, rate, ch, rate * ch * secs + extra)
        pcm[:100 + seed] = 0
        pcm[-(50 + seed):] = 0
        songs.append(pcm)
        chans.append(ch)
    got = bliss_amd.analyze_batch_host(songs, chans, 5)
    for i, pcm in enumerate(songs):
        song = _lib.BlSong()
        song.sample_array = pcm.ctypes.data
        song.channels, song.nSamples, song.sample_rate = chans[i], len(pcm), 22050
        song.nb_bytes_per_sample, song.duration = 2, 5
        amp = gpu_lib.bl_amplitude_sort(C.byref(song))
        frq = gpu_lib.bl_frequency_sort(C.byref(song))
        env = _lib.EnvelopeResult()
        gpu_lib.bl_envelope_sort(C.byref(song), C.byref(env))
        assert np.float32(amp) == np.float32(got[i]["amplitude"]), (i, "amplitude")
        assert np.float32(frq) == np.float32(got[i]["frequency"]), (i, "frequency")
        assert np.float32(env.tempo) == np.float32(got[i]["tempo"]) and np.float32(env.attack) == np.float32(got[i]["attack"]), i
        check_song(got[i], oracle.analyze(pcm, chans[i], 5), f"case {i}")


def test_random_soak_small(gpu_lib, oracle):
    """tools/soak.py on 24 random songs (random rate / channels / level / spectrum / DC / silences): integers
    exact; tempo / amplitude / attack within 1e-4 relative (tempo and amplitude bit-identical); frequency and
    force within 1e-5 + 1e-4 |ref| (ref tests/test_analyze.c:30-35 uses 1e-5 absolute)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tools"))
    import soak
    songs = [soak.make_song(9000 + i, 12.0) for i in range(24)]
    corpus = bliss_amd.DeviceCorpus([p.size for p, _, _ in songs], [c for _, c, _ in songs],
                                    [d for _, _, d in songs])
    for i, (p, _, _) in enumerate(songs):
        corpus.upload(i, p)
    corpus.analyze()
    got = corpus.fetch()
    for i, (p, c, d) in enumerate(songs):
        ref = oracle.analyze(p, c, d)
        for k in INTS:
            assert int(got[i][k]) == int(ref[k]), (i, k, int(got[i][k]), int(ref[k]))
        for k in FLOATS:
            a, b = float(got[i][k]), float(ref[k])
            # frequency (and force, which contains it) is a difference of O(10) terms that crosses zero and
            # goes through an f32 DFT that is not the oracle's: the reference's own absolute 1e-5 applies;
            # the exactly-ordered parts of the path are held to the strict relative bound
            tol = 1e-5 + REL * abs(b) if k in ("frequency", "force") else REL * max(abs(b), 1e-6)
            assert abs(a - b) <= tol, (i, k, a, b)
        assert np.float32(got[i]["tempo"]) == np.float32(ref["tempo"])
        assert np.float32(got[i]["amplitude"]) == np.float32(ref["amplitude"])


def test_invalid_inputs_are_rejected(gpu_lib, tmp_path):
    """Inputs the reference leaves undefined fail loudly instead of reading out of bounds: too short
    (nSamples < 5120: ref src/tempo_atk_sort.c:63-67 would run -2 windows), channel counts other than
    1 / 2, duration 0 (division, ref :283), a truncated or non-audio file."""
    good = np.ones(8192, dtype=np.int16)
    for pcm, ch, dur in ((good[:5119], 1, 1), (good, 3, 1), (good, 0, 1), (good, 2, 0)):
        with pytest.raises(RuntimeError):
            bliss_amd.analyze_batch_host([pcm], ch, dur)
    song = _lib.BlSong()
    junk = tmp_path / "junk.flac"
    junk.write_bytes(b"fLaC" + bytes(range(200)))
    assert gpu_lib.bl_analyze(str(junk).encode(), C.byref(song)) == _lib.BL_UNEXPECTED
    flac = open(os.path.join(HERE, "golden", "song.flac"), "rb").read()
    cut = tmp_path / "cut.flac"
    cut.write_bytes(flac[: len(flac) // 3])
    rc = gpu_lib.bl_analyze(str(cut).encode(), C.byref(song))
    assert rc in (_lib.BL_UNEXPECTED, _lib.BL_LOUD, _lib.BL_CALM, _lib.BL_UNKNOWN)  # never a crash
    if rc != _lib.BL_UNEXPECTED:
        gpu_lib.bl_free_song(C.byref(song))
    txt = tmp_path / "notes.wav"
    txt.write_text("not audio at all")
    assert gpu_lib.bl_analyze(str(txt).encode(), C.byref(song)) == _lib.BL_UNEXPECTED


def test_c_caller_links_and_passes(gpu_lib, tmp_path):
    """tests/c/dropin_check.c: a C99 caller compiled against include/bliss.h only and linked against
    libbliss_amd.so, checking the reference's goldens for song.flac and song_s32.flac (ref
    tests/test_analyze.c:26-89) and the by-value struct ABI."""
    import subprocess
    root = os.path.dirname(HERE)
    exe = str(tmp_path / "dropin_check")
    subprocess.run(["gcc", "-std=c99", "-O1", "-Wall", "-I", os.path.join(root, "include"),
                    os.path.join(HERE, "c", "dropin_check.c"), "-o", exe,
                    "-L", os.path.join(root, "bliss_amd"), "-lbliss_amd", "-lm",
                    "-Wl,-rpath," + os.path.join(root, "bliss_amd")], check=True)
    r = subprocess.run([exe, os.path.join(HERE, "golden", "song.flac"), os.path.join(HERE, "golden", "song_s32.flac")],
                       stdout=subprocess.PIPE, text=True)
    assert r.returncode == 0, r.stdout


def test_batches_on_two_streams_do_not_race(gpu_lib, oracle):
    """The library's scratch workspace is shared by all calls: two device-resident batches enqueued back
    to back on different HIP streams must still give each its own results (device-side ordering through
    an event, not only the host mutex)."""
    import torch
    dev = torch.device("cuda", 0)
    s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    pcm_a = [oracle.synth(700 + i, 22050, 2, 22050 * 2 * 8) for i in range(6)]
    pcm_b = [oracle.synth(800 + i, 22050, 1, 22050 * 9) for i in range(6)]
    ca = bliss_amd.DeviceCorpus([p.size for p in pcm_a], 2, 8)
    cb = bliss_amd.DeviceCorpus([p.size for p in pcm_b], 1, 9)
    for i in range(6):
        ca.upload(i, pcm_a[i])
        cb.upload(i, pcm_b[i])
    torch.cuda.synchronize(dev)
    for _ in range(3):  # several rounds of interleaved enqueues
        with torch.cuda.stream(s1):
            ca.analyze()
        with torch.cuda.stream(s2):
            cb.analyze()
    torch.cuda.synchronize(dev)
    ra, rb = ca.fetch(), cb.fetch()
    for i in range(6):
        check_song(ra[i], oracle.analyze(pcm_a[i], 2, 8), ("stream1", i))
        check_song(rb[i], oracle.analyze(pcm_b[i], 1, 9), ("stream2", i))


def test_cosine_matrix_bit_exact_including_degenerate_vectors(gpu_lib, oracle):
    """bl_cosine_similarity for all pairs (ref src/analyze.c:135-140) through the guarded quotient of bl_cos.h: 3 000
    vectors of mixed scale — ordinary force vectors, tiny and huge norms, a zero vector (0 / 0), duplicates and
    sign flips (quotients of exactly +-1), orthogonal pairs (zero dot products) — every one of the 9 million
    outputs has the bits of the CPU restatement, NaNs included."""
    rng = np.random.default_rng(11)
    v = (rng.standard_normal((3000, 4)) * 10).astype(np.float32)
    v[100:200] *= np.float32(1e-18)
    v[200:300] *= np.float32(1e17)
    v[300] = 0
    v[301] = v[5]; v[302] = -v[5]; v[303] = v[5] * np.float32(3)
    v[304] = [1, 0, 0, 0]; v[305] = [0, 1, 0, 0]; v[306] = [0, 0, -2, 0]
    v[310:330, 1:] = 0           # collinear along x: quotients of exactly +-1 at many scales
    cm = bliss_amd.cosine_matrix(v)
    ref = oracle.cosine_matrix(v)
    assert np.array_equal(cm.view(np.int32), ref.view(np.int32)), int(np.count_nonzero(cm.view(np.int32) != ref.view(np.int32)))
    assert np.isnan(cm[300]).all() and cm[301, 5] == 1.0 and cm[302, 5] == -1.0 and cm[304, 305] == 0.0


def test_guarded_cosine_quotient_sweep(gpu_lib):
    """bl_cos.h is not taken on trust: 2^33 pseudo-random (dot, |a|^2, |b|^2) triples — each random dot with its
    eight neighbouring floats — on the device against the plain expression: no accepted fast result differs, the
    double quotients never differ by more than the 6 ulp that six roundings allow (the guard is 16), and the sweep does meet
    the float rounding boundaries the guard exists for (where the unguarded form is wrong)."""
    counts = (C.c_uint64 * 6)()
    assert gpu_lib.bl_amd_selftest_cos(counts, 1 << 33) == 0
    n, n_fast, bad, max_ulp, near, near_bad = (int(x) for x in counts)
    assert n >= 1 << 33 and bad == 0, (n, bad)
    assert n_fast > 0.999 * n * 0.5 and max_ulp <= 6, (n_fast, max_ulp)   # zero / tiny dots and norms take the plain form
    assert near > 100, near
    print("cosine sweep:", dict(triples=n, fast=n_fast, max_ulp=max_ulp, near_boundary=near, unguarded_wrong=near_bad))


def test_sqrt_of_the_distance_kernels_is_correctly_rounded(gpu_lib):
    """bl_distance's root (ref src/analyze.c:96-100, an IEEE sqrt): the five-instruction form of
    bliss_amd/csrc/bl_sqrt.h over EVERY f32 bit pattern of its domain, and the fallback over all
    2^32 patterns, against (float)sqrt((double)s) on the device — no mismatch anywhere."""
    counts = (C.c_uint64 * 3)()
    assert gpu_lib.bl_amd_selftest_sqrt(counts) == 0
    checked, bad_fast, bad_slow = int(counts[0]), int(counts[1]), int(counts[2])
    # [2^-100, 2^126]: 226 binades of 2^23 values and the upper end point
    assert checked == 226 * (1 << 23) + 1, checked
    assert bad_fast == 0 and bad_slow == 0, (bad_fast, bad_slow)


def test_checked_histogram_build_gives_the_same_records(gpu_lib):
    """INTEGRATION.md advertises `make checked` (-DBL_AMD_CHECKED_HIST: range-tested histogram adds instead of the
    out-of-range ds_add the LDS discards): the build must exist, load, and pass the very tests that pin the unchecked
    form — the out-of-range samples, the batch against the oracle — run here in a child process on
    libbliss_amd_checked.so."""
    import subprocess
    import sys
    lib = os.path.join(os.path.dirname(HERE), "bliss_amd", "libbliss_amd_checked.so")
    if not os.path.exists(lib):
        subprocess.run(["make", "-C", os.path.join(os.path.dirname(HERE), "bliss_amd", "csrc"), "checked"], check=True,
                       stdout=subprocess.DEVNULL)
    env = dict(os.environ, BLISS_AMD_LIB=lib)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-m", "gpu", "-k",
                        "test_histogram_out_of_range_samples_are_dropped or test_batch_matches_oracle or "
                        "test_reference_golden_through_bl_analyze", "-p", "no:cacheprovider"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900,
                       cwd=os.path.dirname(HERE))
    assert r.returncode == 0 and " passed" in r.stdout and "failed" not in r.stdout, r.stdout[-3000:]
