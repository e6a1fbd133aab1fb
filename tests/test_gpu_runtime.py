"""Runtime behaviour of the C-ABI: explicit contexts driven from two host threads, launch
groups, the asynchronous enqueue, registered vs staged host transfer, 32-bit sources, and the
multi-device corpus entry point (RCCL gather at world size 1, peer gather with two ranks on one
device)."""
import ctypes as C
import os
import subprocess
import sys
import threading
import time

import numpy as np
import pytest

import bliss_amd
from bliss_amd import _lib
from tests.test_gpu_parity import check_song

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _songs(oracle, seed0, count, rate=22050):
    out = []
    for i in range(count):
        ch = 1 + (i % 2)
        secs = 6 + (i * 5) % 9
        out.append((oracle.synth(seed0 + i, rate, ch, rate * ch * secs + 8 * (i % 3)), ch, secs))
    return out


def _same(a, b):
    for k in a.dtype.names:
        assert np.array_equal(a[k], b[k]), k


def test_two_contexts_from_two_threads(gpu_lib, oracle):
    """Two explicit contexts on device 0, each driven by its own host thread on its own stream,
    several rounds concurrently: each keeps its own workspace, results match the oracle."""
    import torch
    dev = torch.device("cuda", 0)
    sets = [_songs(oracle, 4000, 6), _songs(oracle, 4100, 7)]
    out, err = [None, None], []

    def work(t):
        try:
            torch.cuda.set_device(0)
            with bliss_amd.Context(0) as ctx:
                assert gpu_lib.bl_amd_ctx_device(ctx.handle) == 0
                pcms = [p for p, _, _ in sets[t]]
                corpus = bliss_amd.DeviceCorpus([p.size for p in pcms], [c for _, c, _ in sets[t]],
                                                [d for _, _, d in sets[t]])
                for i, p in enumerate(pcms):
                    corpus.upload(i, p)
                stream = torch.cuda.Stream(dev)
                with torch.cuda.stream(stream):
                    for _ in range(5):
                        corpus.analyze(ctx=ctx)
                stream.synchronize()
                out[t] = corpus.fetch()
        except Exception as e:  # surfaced below
            err.append(e)

    threads = [threading.Thread(target=work, args=(t,)) for t in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not err, err
    for t in range(2):
        for i, (p, c, d) in enumerate(sets[t]):
            check_song(out[t][i], oracle.analyze(p, c, d), ("ctx", t, i))


def test_launch_groups_and_async_enqueue(oracle, tmp_path):
    """More songs than one launch group holds (group size lowered through the environment for a
    fresh process) give the same records as a single group; and the device entry point returns
    long before the GPU is done."""
    code = r'''
import sys, time, json
import numpy as np
sys.path.insert(0, %r)
import torch, bliss_amd
n = 22050 * 2 * 7
c = bliss_amd.DeviceCorpus([n + 8 * (i %% 4) for i in range(13)], 2, 7)
c.synth(seed_base=5000, sample_rate=22050)
c.analyze(); r = c.fetch()
big = bliss_amd.DeviceCorpus([44100 * 2 * 30] * 256, 2, 30)
big.synth(seed_base=1, sample_rate=44100)
big.analyze(); torch.cuda.synchronize()
t0 = time.perf_counter(); big.analyze(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
np.save(sys.argv[1], r)
print(json.dumps({"enqueue": t1 - t0, "total": t2 - t0}))
''' % ROOT
    outs = {}
    for tag, env in (("one", {}), ("five", {"BL_AMD_GROUP_SONGS": "5"})):
        f = str(tmp_path / f"{tag}.npy")
        r = subprocess.run([sys.executable, "-c", code, f], env=dict(os.environ, **env), text=True,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[tag] = (np.load(f), __import__("json").loads(r.stdout.strip().splitlines()[-1]))
    _same(outs["one"][0], outs["five"][0])
    ref = oracle.analyze(oracle.synth(5012, 22050, 2, 22050 * 2 * 7), 2, 7)
    check_song(outs["five"][0][12], ref, "group-of-5, last song")
    t = outs["one"][1]
    assert t["enqueue"] < 0.5 * t["total"], t     # ~14 ms of kernels behind a sub-millisecond enqueue


def test_mixed_batch_with_early_tail_equals_the_plain_path(oracle, tmp_path):
    """A mixed-length group of >= 1 024 songs gives its longest songs their own window launch and starts their serial
    tail under the window kernel of the rest (blk_analyze, n_head).  Same records as the same corpus analysed in
    launch groups too small to be split (fresh process, BL_AMD_GROUP_SONGS=500), and a sample of them against the
    oracle."""
    code = r'''
import sys
import numpy as np
sys.path.insert(0, %r)
import torch, bliss_amd
rng = np.random.default_rng(77)
lengths = np.floor(np.exp(rng.uniform(np.log(6000), np.log(260000), 1400))).astype(np.int64)
chans = rng.integers(1, 3, 1400)
lengths = (lengths // chans) * chans
c = bliss_amd.DeviceCorpus(lengths.tolist(), chans.tolist(), [3] * 1400)
c.synth(seed_base=31000, sample_rate=22050)
c.analyze(); r = c.fetch()
assert int(r["status"].max()) == 0
np.save(sys.argv[1], r)
np.save(sys.argv[1] + ".len.npy", np.stack([lengths, chans]))
''' % ROOT
    outs = {}
    for tag, env in (("split", {}), ("groups", {"BL_AMD_GROUP_SONGS": "500"})):
        f = str(tmp_path / f"{tag}.npy")
        r = subprocess.run([sys.executable, "-c", code, f], env=dict(os.environ, **env), text=True,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[tag] = np.load(f)
    _same(outs["split"], outs["groups"])
    lengths, chans = np.load(str(tmp_path / "split.npy.len.npy"))
    order = np.argsort(-lengths)
    for i in [int(order[0]), int(order[255]), int(order[256]), int(order[700]), int(order[-1])]:
        ref = oracle.analyze(oracle.synth(31000 + i, 22050, int(chans[i]), int(lengths[i])), int(chans[i]), 3)
        check_song(outs["split"][i], ref, f"song {i} of the split batch ({lengths[i]} samples)")


def test_measurement_switches_are_ignored_by_the_product_build(tmp_path):
    """The result-invalidating measurement aids (BL_AMD_SQRT_VARIANT=3: the distance kernel's store stream alone,
    BL_AMD_NO_SIDE, BL_AMD_FUSED_SCAN, and the round-3 switches BL_AMD_ENV_DBG / BL_AMD_ENV_OLD, whose code is gone) only exist in `make measure` builds
    (-DBL_AMD_MEASURE); the shipped library must give the same records, window energies and distance matrix
    whether those variables are set or not."""
    code = r'''
import sys, hashlib
import numpy as np
sys.path.insert(0, %r)
import torch, bliss_amd
lengths = [22050 * 2 * 9 + 8 * i for i in range(5)] + [5120, 6000, 22050 * 1 * 40, 44100 * 2 * 33]
chans = [2, 2, 2, 2, 2, 2, 1, 1, 2]
c = bliss_amd.DeviceCorpus(lengths, chans, 9)
c.synth(seed_base=900, sample_rate=22050)
c.analyze(); r = c.fetch()
np.save(sys.argv[1], r)
lib = bliss_amd.load()
import ctypes as C
buf = (C.c_float * 400000)()
n = lib.bl_amd_last_energies(buf, 400000)
e = np.frombuffer(bytes(buf), dtype=np.float32)[: max(n, 0)]
# a song owns nb_frames slots and fills the first n_windows of them; the other two are never written
valid, off = [], 0
for nb, nw in zip(r["nb_frames"], r["n_windows"]):
    valid.append(e[off:off + int(nw)]); off += int(nb)
assert off == n
vecs = np.stack([r["tempo"], r["amplitude"], r["frequency"], r["attack"]], axis=1).astype(np.float32)
d = bliss_amd.distance_matrix(vecs)
print(n, hashlib.md5(np.concatenate(valid).tobytes()).hexdigest(), hashlib.md5(np.ascontiguousarray(d).tobytes()).hexdigest())
''' % ROOT
    outs = {}
    switches = {"BL_AMD_SQRT_VARIANT": "3", "BL_AMD_ENV_DBG": "3", "BL_AMD_ENV_OLD": "1", "BL_AMD_NO_SIDE": "1",
                "BL_AMD_FUSED_SCAN": "0"}
    for tag, env in (("plain", {}), ("switches", switches)):
        f = str(tmp_path / f"{tag}.npy")
        r = subprocess.run([sys.executable, "-c", code, f], env=dict(os.environ, **env), text=True,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[tag] = (np.load(f), r.stdout.strip().splitlines()[-1])
    _same(outs["plain"][0], outs["switches"][0])
    assert outs["plain"][1] == outs["switches"][1] and not outs["plain"][1].startswith("0 "), outs


def test_host_transfer_modes_and_s32(gpu_lib, oracle):
    songs = _songs(oracle, 4300, 9)
    pcms, chans, durs = [p for p, _, _ in songs], [c for _, c, _ in songs], [d for _, _, d in songs]
    staged = bliss_amd.analyze_batch_host(pcms, chans, durs)
    assert gpu_lib.bl_amd_set_host_transfer(1) == 0            # hipHostRegister on the caller's buffers
    try:
        reg = bliss_amd.analyze_batch_host(pcms, chans, durs)
    finally:
        assert gpu_lib.bl_amd_set_host_transfer(0) == 0
    _same(staged, reg)
    assert pcms[0][0] == oracle.synth(4300, 22050, 1, 1)[0]    # the caller's memory is still its own
    # 32-bit sources: narrowed by the library (host staging and device kernel), oracle gets numpy's >> 16
    rng = np.random.default_rng(2)
    s32 = [(p.astype(np.int32) << 16) | rng.integers(0, 65536, p.size, dtype=np.int32) for p in pcms]
    got = bliss_amd.analyze_batch_host_s32(s32, chans, durs)
    _same(staged, got)
    corpus = bliss_amd.DeviceCorpus([p.size for p in pcms], chans, durs)
    for i, q in enumerate(s32):
        corpus.upload_s32(i, q)
    corpus.analyze()
    _same(staged, corpus.fetch())
    for i in (0, 4, 8):
        check_song(got[i], oracle.analyze((s32[i] >> 16).astype(np.int16), chans[i], durs[i]), ("s32", i))
    odd = np.arange(-70000, 70001, 7, dtype=np.int32) * 30011   # unaligned views through the scalar path
    import torch
    t = torch.from_numpy(odd).cuda()
    o = torch.zeros(odd.size + 3, dtype=torch.int16, device="cuda")
    assert gpu_lib.bl_amd_narrow_s32_device(C.c_void_p(t.data_ptr() + 4), C.c_void_p(o.data_ptr() + 2),
                                            odd.size - 1, None) == 0
    torch.cuda.synchronize()
    assert np.array_equal(o.cpu().numpy()[1:odd.size], (odd[1:] >> 16).astype(np.int16))


def test_invalid_host_descriptors_fail_before_staging(gpu_lib):
    good = np.ones(8192, dtype=np.int16)
    ptrs = (C.c_void_p * 2)(good.ctypes.data, None)
    ns = (C.c_int32 * 2)(8192, 8192)
    ch = (C.c_int32 * 2)(1, 1)
    du = (C.c_uint64 * 2)(1, 1)
    out = (_lib.SongResult * 2)()
    assert gpu_lib.bl_amd_analyze_batch_host(ptrs, ns, ch, du, 2, out) == _lib.BL_UNEXPECTED   # NULL buffer
    ptrs[1] = good.ctypes.data
    ns[1] = -5
    assert gpu_lib.bl_amd_analyze_batch_host(ptrs, ns, ch, du, 2, out) == _lib.BL_UNEXPECTED   # negative length


def _all_devices():
    import torch
    return list(range(torch.cuda.device_count()))


def _check_corpus_multi(oracle, devices, gather):
    songs = _songs(oracle, 4500, 11)
    for equal in (False, True):
        use = songs if not equal else [(oracle.synth(4600 + i, 22050, 2, 22050 * 2 * 6), 2, 6) for i in range(7)]
        pcms, chans, durs = [p for p, _, _ in use], [c for _, c, _ in use], [d for _, _, d in use]
        res, mat = bliss_amd.analyze_corpus_multi(pcms, chans, durs, devices, gather=gather)
        single = bliss_amd.analyze_batch_host(pcms, chans, durs)
        _same(single, res)
        fv = np.stack([single[k] for k in ("tempo", "amplitude", "frequency", "attack")], axis=1)
        assert np.array_equal(mat, bliss_amd.distance_matrix(fv))
        assert np.array_equal(mat, oracle.distance_matrix(fv))
    res2, none = bliss_amd.analyze_corpus_multi(pcms, chans, durs, devices, gather=gather, matrix=False)
    _same(res, res2)
    assert none is None


@pytest.mark.parametrize("devices,gather", [([0], "rccl"), ([0], "peer"), ([0, 0], "peer"), ([0, 0, 0], "peer")])
def test_corpus_multi(gpu_lib, oracle, devices, gather):
    """The C-ABI batch-of-songs mode.  RCCL (ncclCommInitAll + ncclAllGather) runs at world size
    1; the peer gather also takes several ranks on one device, which exercises the sharding, the
    per-rank contexts and threads, the exchange and the row-block matrix with W = 2, 3."""
    _check_corpus_multi(oracle, devices, gather)


@pytest.mark.parametrize("gather", ["rccl", "peer"])
def test_corpus_multi_on_every_device(gpu_lib, oracle, gather):
    """The same on every HIP device the box has, one rank per device: ncclCommInitAll over N devices and an
    ncclAllGather that crosses xGMI, or hipMemcpyPeerAsync between devices.  Skipped on a one-GPU box; on an N-GPU
    box it runs without an edit."""
    devices = _all_devices()
    if len(devices) < 2:
        pytest.skip("one HIP device on this box: W > 1 is covered on device 0 by test_corpus_multi")
    _check_corpus_multi(oracle, devices, gather)
    _check_corpus_multi(oracle, devices[::-1], gather)     # rank r need not sit on device r


def _check_corpus_multi_device_resident(oracle, counts, gather, devices=None):
    import torch
    devices = devices or [0] * len(counts)
    songs = _songs(oracle, 4700, sum(counts))
    pcms, chans, durs = [p for p, _, _ in songs], [c for _, c, _ in songs], [d for _, _, d in songs]
    whole = bliss_amd.DeviceCorpus([p.size for p in pcms], chans, durs)
    for i, p in enumerate(pcms):
        whole.upload(i, p)
    whole.analyze()
    single = whole.fetch()
    fv = np.stack([single[k] for k in ("tempo", "amplitude", "frequency", "attack")], axis=1)
    want = bliss_amd.distance_matrix(fv)
    corpora, first = [], 0
    for cnt, d in zip(counts, devices):
        part = range(first, first + cnt)
        c = bliss_amd.DeviceCorpus([pcms[i].size for i in part] or [8], [chans[i] for i in part] or [1],
                                   [durs[i] for i in part] or [1], device=f"cuda:{d}")
        if cnt == 0:
            c.n_songs = 0          # an empty shard: the arena exists, nothing to analyse
        for k, i in enumerate(part):
            c.upload(k, pcms[i])
        corpora.append(c)
        first += cnt
    for _ in range(2):             # second call: the contexts' exchange buffers are reused
        res, mat, rows = bliss_amd.analyze_corpus_multi_device(corpora, gather=gather, keep_rows=True)
        _same(single, res)
        assert np.array_equal(mat, want) and np.array_equal(mat, oracle.distance_matrix(fv))
        for d in set(devices):
            torch.cuda.synchronize(d)
        assert np.array_equal(np.concatenate([r.cpu().numpy() for r in rows], axis=0), want)
    # the shard's own device records are the same records
    off = 0
    for c in corpora:
        if c.n_songs:
            _same(single[off:off + c.n_songs], c.fetch()[:c.n_songs])
        off += c.n_songs
    res2, none, _ = bliss_amd.analyze_corpus_multi_device(corpora, gather=gather, matrix=False)
    _same(single, res2)
    assert none is None


@pytest.mark.parametrize("counts,gather", [((5,), "rccl"), ((4, 3), "peer"), ((3, 0, 4), "peer"), ((2, 3, 2), "peer")])
def test_corpus_multi_device_resident(gpu_lib, oracle, counts, gather):
    """bl_amd_analyze_corpus_multi_device: the corpus already resident, one arena per rank (here
    all on device 0, separate contexts and host threads; an empty shard included).  Records and
    matrix must equal the single-context path bit for bit, the row blocks left in HBM too."""
    _check_corpus_multi_device_resident(oracle, counts, gather)


@pytest.mark.parametrize("gather", ["rccl", "peer"])
def test_corpus_multi_device_resident_on_every_device(gpu_lib, oracle, gather):
    """One resident shard per HIP device of the box (uneven counts, the last but one empty when there are more than
    two): the vectors cross xGMI by RCCL or by peer copies, every device computes its own row block.  Skipped on a
    one-GPU box."""
    devices = _all_devices()
    if len(devices) < 2:
        pytest.skip("one HIP device on this box: W > 1 is covered on device 0 by test_corpus_multi_device_resident")
    counts = [3 + (d % 3) for d in devices]
    if len(devices) > 2:
        counts[-2] = 0
    _check_corpus_multi_device_resident(oracle, tuple(counts), gather, devices)


def test_scalar_helpers_match_the_kernels(gpu_lib):
    """bl_distance / bl_cosine_similarity of one pair are host arithmetic (bl_api.c): the same
    bits as the all-pairs kernels."""
    rng = np.random.default_rng(31)
    v = (rng.standard_normal((300, 4)) * 9).astype(np.float32)
    dm, cm = bliss_amd.distance_matrix(v), bliss_amd.cosine_matrix(v)
    for i, j in rng.integers(0, 300, (400, 2)):
        a, b = _lib.ForceVector(*v[i]), _lib.ForceVector(*v[j])
        assert gpu_lib.bl_distance(a, b) == dm[i, j]
        assert gpu_lib.bl_cosine_similarity(a, b) == cm[i, j]


def test_analyze_files_equals_the_per_file_loop(gpu_lib, oracle, tmp_path):
    """bl_amd_analyze_files(filenames, ...) is `for f: bl_analyze(f, &song)` with the decoding on host
    threads and the songs batched to the GPU: every song struct — force vector, force, calm_or_loud,
    metadata — equals what the one-file call gives, bit for bit; a missing file and a clip that is
    too short are reported per file and do not stop the rest."""
    import shutil
    from tests.test_ingest import write_flac_verbatim, _write_wav16
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    files = []
    for name in ("song.flac", "song_s32.flac", "song_s32_mono.flac"):
        dst = tmp_path / name
        shutil.copy(os.path.join(gold, name), dst)
        files.append(str(dst))
    for i in range(9):
        rate, ch, secs = (22050, 44100, 48000)[i % 3], 1 + (i % 2), 5 + i % 5   # the test FLAC writer stops at 128 frames of 4096
        pcm = oracle.synth(6100 + i, rate, ch, rate * ch * secs)
        f = tmp_path / f"synth{i}.{'flac' if i % 2 else 'wav'}"
        if i % 2:
            write_flac_verbatim(f, pcm, ch, rate, 16)
        else:
            _write_wav16(f, pcm, ch, rate)
        files.append(str(f))
    missing = str(tmp_path / "missing.flac")
    short = tmp_path / "short.wav"
    _write_wav16(short, oracle.synth(1, 22050, 2, 3000), 2, 22050)
    order = files[:5] + [missing] + files[5:9] + [str(short)] + files[9:]
    recs, codes = bliss_amd.analyze_files(order, n_threads=4)
    assert len(recs) == len(order) == 14
    n_ok = 0
    for f, rec, code in zip(order, recs, codes):
        song = _lib.BlSong()
        rc = gpu_lib.bl_analyze(f.encode(), C.byref(song))
        assert rc == code, f
        if rc == _lib.BL_UNEXPECTED:
            assert rec is None and f in (missing, str(short))
            gpu_lib.bl_free_song(C.byref(song))   # bl_analyze leaves a decoded-but-unanalysable song allocated
            continue
        n_ok += 1
        for k in ("tempo", "amplitude", "frequency", "attack"):
            assert rec["force_vector"][k] == getattr(song.force_vector, k), (f, k)
        assert rec["force"] == song.force and rec["calm_or_loud"] == song.calm_or_loud
        for k in ("channels", "nSamples", "sample_rate", "bitrate", "nb_bytes_per_sample", "resampled", "duration"):
            assert rec[k] == getattr(song, k), (f, k)
        assert rec["title"] == song.title.decode() and rec["filename"] == f
        gpu_lib.bl_free_song(C.byref(song))
    assert n_ok == 12
    # keep_pcm: the decoded samples stay with the song
    recs2, _ = bliss_amd.analyze_files(order[:2], n_threads=1, keep_pcm=True)
    assert recs2[0]["pcm"].size == recs2[0]["nSamples"] == 488138
    # a NULL name among the files, song structs full of garbage (callers pass uninitialised structs, ref
    # tests/test_analyze.c:27-28): reported for that file only, every pointer field NULL afterwards
    names = (C.c_char_p * 3)(order[0].encode(), None, order[1].encode())
    songs = (_lib.BlSong * 3)()
    C.memset(songs, 0xAB, C.sizeof(songs))
    codes3 = (C.c_int * 3)()
    assert gpu_lib.bl_amd_analyze_files(names, 3, songs, codes3, 2, 0) == 2
    assert codes3[1] == _lib.BL_UNEXPECTED and codes3[0] == codes[0] and codes3[2] == codes[1]
    assert not songs[1].sample_array and songs[1].title is None and songs[1].filename is None
    assert songs[0].force_vector.tempo == recs[0]["force_vector"]["tempo"]
    for sg in songs:
        gpu_lib.bl_free_song(C.byref(sg))
