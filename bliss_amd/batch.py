"""Batch helpers over the C-ABI: device-resident corpora (torch owns the memory,
the library owns the arithmetic) and host-pointer conveniences."""
import ctypes as C
import os

import numpy as np

from . import _lib

RESULT_DTYPE = np.dtype([
    ("tempo", "<f4"), ("amplitude", "<f4"), ("frequency", "<f4"), ("attack", "<f4"),
    ("force", "<f4"), ("calm_or_loud", "<i4"), ("status", "<i4"), ("start", "<i4"),
    ("end", "<i4"), ("mean", "<i4"), ("variance", "<i4"), ("n_frames", "<i4"),
    ("nb_frames", "<i4"), ("n_windows", "<i4"), ("beat", "<i4"), ("hist_integral", "<f4"),
    ("freq_peak", "<f4"), ("atk_sum", "<f8")], align=True)
assert RESULT_DTYPE.itemsize == C.sizeof(_lib.SongResult) == 80


def results_to_numpy(raw_bytes):
    """bytes / uint8 array of bl_amd_song_result records -> structured numpy array."""
    return np.frombuffer(bytes(raw_bytes), dtype=RESULT_DTYPE).copy()


def _check(rc, what):
    if rc != _lib.BL_OK:
        raise RuntimeError(f"{what} failed with BL_UNEXPECTED ({rc}); see stderr")


class DeviceCorpus:
    """n_songs decoded songs laid out in one int16 arena in HBM.

    lengths: interleaved sample counts; channels / durations: per song (or scalars).
    The arena is a torch int16 CUDA tensor; every song starts at a multiple of 8 samples.
    """

    def __init__(self, lengths, channels, durations, device="cuda:0"):
        import torch
        self.torch = torch
        self.lib = _lib.load()
        self.device = torch.device(device)
        n = len(lengths)
        channels = [channels] * n if np.isscalar(channels) else list(channels)
        durations = [durations] * n if np.isscalar(durations) else list(durations)
        self.n_songs = n
        self.desc = (_lib.SongDesc * n)()
        off = 0
        for i, (ln, ch, du) in enumerate(zip(lengths, channels, durations)):
            self.desc[i].pcm_offset = off
            self.desc[i].n_samples = int(ln)
            self.desc[i].channels = int(ch)
            self.desc[i].duration = int(du)
            off += (int(ln) + 7) & ~7
        self.total_samples = off
        idx = self.device.index if self.device.index is not None else 0
        with torch.cuda.device(idx):
            _check(self.lib.bl_amd_init(idx), "bl_amd_init")
            self.pcm = torch.zeros(off + 64, dtype=torch.int16, device=self.device)
            self.results = torch.zeros(n * C.sizeof(_lib.SongResult), dtype=torch.uint8,
                                       device=self.device)

    @property
    def pcm_bytes(self):
        return 2 * sum(int(d.n_samples) for d in self.desc)

    def _stream(self):
        return C.c_void_p(self.torch.cuda.current_stream(self.device).cuda_stream)

    def synth(self, seed_base, sample_rate):
        _check(self.lib.bl_amd_synth_pcm_device(C.c_void_p(self.pcm.data_ptr()), self.desc,
                                                self.n_songs, seed_base, sample_rate,
                                                self._stream()), "bl_amd_synth_pcm_device")

    def upload(self, index, pcm_int16):
        t = self.torch.from_numpy(np.ascontiguousarray(pcm_int16, dtype=np.int16))
        o = int(self.desc[index].pcm_offset)
        assert t.numel() == self.desc[index].n_samples
        self.pcm[o:o + t.numel()].copy_(t)

    def analyze(self, ctx=None):
        """Enqueue the analysis on torch's current stream (asynchronous); ctx: an explicit
        Context instead of the thread's default one."""
        if ctx is None:
            _check(self.lib.bl_amd_analyze_batch_device(C.c_void_p(self.pcm.data_ptr()), self.desc,
                                                        self.n_songs,
                                                        C.c_void_p(self.results.data_ptr()),
                                                        self._stream()), "bl_amd_analyze_batch_device")
        else:
            _check(self.lib.bl_amd_ctx_analyze_batch_device(ctx.handle, C.c_void_p(self.pcm.data_ptr()),
                                                            self.desc, self.n_songs,
                                                            C.c_void_p(self.results.data_ptr()),
                                                            self._stream()),
                   "bl_amd_ctx_analyze_batch_device")

    def upload_s32(self, index, pcm_int32):
        """A 32-bit source: narrowed on the device (>> 16) straight into the song's slot."""
        t = self.torch.from_numpy(np.ascontiguousarray(pcm_int32, dtype=np.int32)).to(self.device)
        o = int(self.desc[index].pcm_offset)
        assert t.numel() == self.desc[index].n_samples
        dst = self.pcm[o:o + t.numel()]
        _check(self.lib.bl_amd_narrow_s32_device(C.c_void_p(t.data_ptr()), C.c_void_p(dst.data_ptr()),
                                                 t.numel(), self._stream()), "bl_amd_narrow_s32_device")
        self.torch.cuda.current_stream(self.device).synchronize()  # t may be freed on return

    def fetch(self):
        self.torch.cuda.synchronize(self.device)
        return results_to_numpy(self.results.cpu().numpy().tobytes())

    def force_vectors(self):
        """(n_songs, 4) float32 CUDA tensor view-copy of the force vectors."""
        rec = self.results.view(self.n_songs, C.sizeof(_lib.SongResult))
        return rec[:, :16].contiguous().view(self.torch.float32).view(self.n_songs, 4)


class Context:
    """An explicit library context: one device, its own workspace and internal streams.
    Independent of the default context and of other Contexts (also on the same device)."""

    def __init__(self, device=0):
        self.lib = _lib.load()
        self.handle = C.c_void_p()
        _check(self.lib.bl_amd_ctx_create(int(device), C.byref(self.handle)), "bl_amd_ctx_create")

    def close(self):
        if self.handle:
            self.lib.bl_amd_ctx_destroy(self.handle)
            self.handle = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def analyze_batch_host(self, pcm_list, channels, durations):
        args, out, _keep = _host_args(pcm_list, channels, durations, np.int16)
        _check(self.lib.bl_amd_ctx_analyze_batch_host(self.handle, *args, out),
               "bl_amd_ctx_analyze_batch_host")
        return results_to_numpy(bytes(out))


def _host_args(pcm_list, channels, durations, dtype):
    n = len(pcm_list)
    arrs = [np.ascontiguousarray(p, dtype=dtype) for p in pcm_list]
    channels = [channels] * n if np.isscalar(channels) else list(channels)
    durations = [durations] * n if np.isscalar(durations) else list(durations)
    ptrs = (C.c_void_p * n)(*[a.ctypes.data for a in arrs])
    ns = (C.c_int32 * n)(*[a.size for a in arrs])
    chs = (C.c_int32 * n)(*channels)
    dus = (C.c_uint64 * n)(*durations)
    out = (_lib.SongResult * n)()
    return (ptrs, ns, chs, dus, n), out, arrs


def analyze_batch_host(pcm_list, channels, durations):
    """pcm_list: list of 1-D int16 numpy arrays (interleaved).  Returns structured results."""
    lib = _lib.load()
    args, out, _keep = _host_args(pcm_list, channels, durations, np.int16)
    _check(lib.bl_amd_analyze_batch_host(*args, out), "bl_amd_analyze_batch_host")
    return results_to_numpy(bytes(out))


def analyze_files(filenames, n_threads=0, keep_pcm=False):
    """`for f in filenames: bl_analyze(f)` as one call (bl_amd_analyze_files): files decoded on host
    threads while earlier ones are transferred and analysed.  Returns (list of dicts with the
    song's fields, or None where the file could not be analysed; array of bl_analyze codes)."""
    lib = _lib.load()
    n = len(filenames)
    names = (C.c_char_p * n)(*[os.fsencode(f) for f in filenames])
    songs = (_lib.BlSong * n)()
    codes = (C.c_int * n)()
    got = lib.bl_amd_analyze_files(names, n, songs, codes, int(n_threads), int(bool(keep_pcm)))
    if got == _lib.BL_UNEXPECTED:
        for sg in songs:
            lib.bl_free_song(C.byref(sg))
        raise RuntimeError("bl_amd_analyze_files failed with BL_UNEXPECTED; see stderr")
    out = []
    for i in range(n):
        sg = songs[i]
        if codes[i] == _lib.BL_UNEXPECTED:
            out.append(None)
        else:
            rec = {k: getattr(sg, k) for k in ("force", "channels", "nSamples", "sample_rate", "bitrate",
                                                 "nb_bytes_per_sample", "calm_or_loud", "resampled", "duration")}
            rec["force_vector"] = {k: getattr(sg.force_vector, k) for k in ("tempo", "amplitude", "frequency", "attack")}
            for k in ("filename", "artist", "title", "album", "tracknumber", "genre"):
                v = getattr(sg, k)
                rec[k] = v.decode("utf-8", "replace") if v is not None else None
            if keep_pcm and sg.sample_array:
                rec["pcm"] = np.ctypeslib.as_array(C.cast(sg.sample_array, C.POINTER(C.c_int16)),
                                                   shape=(sg.nSamples,)).copy()
            out.append(rec)
        lib.bl_free_song(C.byref(sg))
    return out, np.array(list(codes), dtype=np.int32)


def analyze_batch_host_s32(pcm_list, channels, durations):
    """Same for 32-bit sources (1-D int32 arrays): narrowed with >> 16 while they are staged."""
    lib = _lib.load()
    args, out, _keep = _host_args(pcm_list, channels, durations, np.int32)
    _check(lib.bl_amd_analyze_batch_host_s32(*args, out), "bl_amd_analyze_batch_host_s32")
    return results_to_numpy(bytes(out))


def analyze_batch_host_rate(pcm_list, channels, durations, sample_rate):
    """Songs at `sample_rate` Hz (1-D int16 or int32 arrays, all of one dtype): each wave is
    converted to 22 050 Hz stereo on the device between its transfer and its analysis."""
    lib = _lib.load()
    dtype = np.asarray(pcm_list[0]).dtype
    if dtype not in (np.int16, np.int32):
        raise TypeError("int16 or int32 PCM expected")
    (ptrs, ns, chs, dus, n), out, _keep = _host_args(pcm_list, channels, durations, dtype)
    _check(lib.bl_amd_analyze_batch_host_rate(ptrs, int(dtype == np.int32), ns, chs, dus, n, int(sample_rate), out),
           "bl_amd_analyze_batch_host_rate")
    return results_to_numpy(bytes(out))


def analyze_corpus_multi(pcm_list, channels, durations, devices, gather="rccl", matrix=True):
    """Shard the corpus over `devices` (a rank per entry), analyse, all-gather the force vectors
    and compute the bl_distance matrix by row blocks (bl_amd_analyze_corpus_multi).  Returns
    (results, matrix or None), both in the caller's song order."""
    lib = _lib.load()
    args, out, _keep = _host_args(pcm_list, channels, durations, np.int16)
    n = args[-1]
    devs = (C.c_int * len(devices))(*[int(d) for d in devices])
    flags = {"rccl": 0, "peer": 1}[gather]
    mat = np.empty((n, n), dtype=np.float32) if matrix else None
    mp = mat.ctypes.data_as(C.POINTER(C.c_float)) if matrix else None
    _check(lib.bl_amd_analyze_corpus_multi(*args, devs, len(devices), flags, out, mp),
           "bl_amd_analyze_corpus_multi")
    return results_to_numpy(bytes(out)), mat


def analyze_corpus_multi_device(corpora, gather="rccl", matrix=True, keep_rows=False):
    """Resident corpora, one DeviceCorpus per rank (each on its rank's device; several on one
    device with gather="peer"): analyse every shard where it lies, all-gather the force vectors,
    row blocks of the bl_distance matrix (bl_amd_analyze_corpus_multi_device).  Returns
    (results in shard-major order, N x N matrix or None, list of per-shard row-block CUDA tensors
    or None)."""
    lib = _lib.load()
    torch = corpora[0].torch
    n = sum(c.n_songs for c in corpora)
    shards = (_lib.Shard * len(corpora))()
    rows = []
    for r, c in enumerate(corpora):
        c.torch.cuda.synchronize(c.device)   # the arena was filled on torch's stream
        shards[r].device = c.device.index or 0
        shards[r].n_songs = c.n_songs
        shards[r].d_pcm = c.pcm.data_ptr()
        shards[r].h_desc = C.cast(c.desc, C.POINTER(_lib.SongDesc))
        shards[r].d_results = c.results.data_ptr()
        if keep_rows:
            rows.append(torch.empty((c.n_songs, n), dtype=torch.float32, device=c.device))
            shards[r].d_rows = rows[-1].data_ptr()
    out = (_lib.SongResult * n)()
    mat = np.empty((n, n), dtype=np.float32) if matrix else None
    mp = mat.ctypes.data_as(C.POINTER(C.c_float)) if matrix else None
    flags = {"rccl": 0, "peer": 1}[gather]
    _check(lib.bl_amd_analyze_corpus_multi_device(shards, len(corpora), flags, out, mp),
           "bl_amd_analyze_corpus_multi_device")
    return results_to_numpy(bytes(out)), mat, (rows if keep_rows else None)


def resample_host(pcm, channels, in_rate):
    """Interleaved int16 / int32 (left-justified) numpy PCM at in_rate Hz -> interleaved stereo int16
    at 22 050 Hz, the conversion bl_audio_decode applies (include/bliss_amd.h)."""
    lib = _lib.load()
    pcm = np.ascontiguousarray(pcm)
    if pcm.dtype not in (np.int16, np.int32):
        raise TypeError("int16 or int32 PCM expected")
    out = C.POINTER(C.c_int16)()
    n = C.c_size_t(0)
    _check(lib.bl_amd_resample_host(pcm.ctypes.data, int(pcm.dtype == np.int32), pcm.size // channels,
                                    channels, in_rate, C.byref(out), C.byref(n)), "bl_amd_resample_host")
    try:
        return np.ctypeslib.as_array(out, shape=(2 * n.value,)).copy()
    finally:
        _libc_free(out)


def _libc_free(ptr):
    C.CDLL(None).free(C.cast(ptr, C.c_void_p))


def resample_batch_device(d_in, frames, channels, in_rate, stream=None):
    """Songs resident in HBM at in_rate Hz (one torch int16 or int32 CUDA tensor, song i = the next
    frames[i] * channels[i] elements, starts rounded up to 8 elements) -> (int16 CUDA arena at
    22 050 Hz stereo, SongDesc-ready list of (pcm_offset, n_samples)).  Asynchronous on `stream`."""
    import torch
    lib = _lib.load()
    n = len(frames)
    channels = [channels] * n if np.isscalar(channels) else list(channels)
    desc = (_lib.ResampleDesc * n)()
    in_off = out_off = 0
    placed = []
    for i, (fr, ch) in enumerate(zip(frames, channels)):
        of = lib.bl_amd_resample_out_frames(int(fr), in_rate)
        desc[i].in_offset, desc[i].out_offset = in_off, out_off
        desc[i].frames, desc[i].channels = int(fr), int(ch)
        placed.append((out_off, 2 * of))
        in_off += (int(fr) * int(ch) + 7) & ~7
        out_off += (2 * of + 7) & ~7
    out = torch.zeros(out_off + 64, dtype=torch.int16, device=d_in.device)
    s = stream.cuda_stream if stream is not None else torch.cuda.current_stream(d_in.device).cuda_stream
    idx = d_in.device.index or 0
    with torch.cuda.device(idx):
        _check(lib.bl_amd_init(idx), "bl_amd_init")
        _check(lib.bl_amd_resample_batch_device(d_in.data_ptr(), int(d_in.dtype == torch.int32), desc, n,
                                                in_rate, out.data_ptr(), C.c_void_p(s)),
               "bl_amd_resample_batch_device")
    return out, placed


def _matrix(fn_name, vecs):
    lib = _lib.load()
    v = np.ascontiguousarray(vecs, dtype=np.float32).reshape(-1, 4)
    n = v.shape[0]
    out = np.empty((n, n), dtype=np.float32)
    rc = getattr(lib, fn_name)(v.ctypes.data_as(C.POINTER(_lib.ForceVector)), n,
                               out.ctypes.data_as(C.POINTER(C.c_float)))
    _check(rc, fn_name)
    return out


def distance_matrix(vecs):
    """N x N bl_distance matrix (ref src/analyze.c:96-100) of (N, 4) force vectors."""
    return _matrix("bl_amd_distance_matrix_host", vecs)


def cosine_matrix(vecs):
    """N x N bl_cosine_similarity matrix (ref src/analyze.c:135-140)."""
    return _matrix("bl_amd_cosine_matrix_host", vecs)


def playlist(vecs, seed_index):
    """Song indices ordered by increasing bl_distance from song `seed_index`, and the
    distances (ref python/examples/make_m3u_playlist.py:62-72).  Stable for ties."""
    lib = _lib.load()
    v = np.ascontiguousarray(vecs, dtype=np.float32).reshape(-1, 4)
    n = v.shape[0]
    order = np.empty(n, dtype=np.int32)
    dist = np.empty(n, dtype=np.float32)
    rc = lib.bl_amd_playlist_host(v.ctypes.data_as(C.POINTER(_lib.ForceVector)), n, int(seed_index),
                                  order.ctypes.data_as(C.POINTER(C.c_int32)),
                                  dist.ctypes.data_as(C.POINTER(C.c_float)))
    _check(rc, "bl_amd_playlist_host")
    return order, dist
