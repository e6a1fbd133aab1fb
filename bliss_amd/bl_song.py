"""`bl_song` — dict-like wrapper of `struct bl_song`, the counterpart of the reference's
python/bliss/bl_song.py:9-209 on top of the ctypes binding (cffi is not available here).
Same surface: Mapping access to the struct fields, `force_vector` as a
{tempo, amplitude, frequency, attack} dict, `analyze` / `decode` / `envelope_analysis` /
`amplitude_analysis` / `frequency_analysis` / `free`, usable as a context manager."""
import ctypes as C
from collections.abc import Mapping

from . import _lib

_FIELDS = [f[0] for f in _lib.BlSong._fields_]
_STRINGS = {"filename", "artist", "title", "album", "tracknumber", "genre"}
_FV = ("tempo", "amplitude", "frequency", "attack")


class bl_song(Mapping):
    def __init__(self, filename=None, initializer=None, c_struct=None):
        """filename: file to load and analyze (ref bl_song.py:16-41); initializer: dict of
        field values; c_struct: an existing _lib.BlSong to wrap."""
        self._lib = _lib.load()
        self._c_struct = c_struct if c_struct is not None else _lib.BlSong()
        self._keepalive = {}
        if isinstance(initializer, dict):
            for k, v in initializer.items():
                self.set(k, v)
        if filename is not None:
            self.analyze(filename)

    # --- Mapping interface (ref bl_song.py:43-84) ---
    def __getitem__(self, key):
        return self.get(key)

    def __setitem__(self, key, value):
        return self.set(key, value)

    def __len__(self):
        return len(_FIELDS)

    def __iter__(self):
        return iter(_FIELDS)

    def __enter__(self):
        return self

    def __exit__(self, exc_type, exc_val, exc_tb):
        self.free()

    def __repr__(self):
        return {k: (self.get(k) if k != "sample_array" else "<%d samples>" % self._c_struct.nSamples)
                for k in _FIELDS}.__repr__()

    def get(self, key):
        """ref bl_song.py:86-112: char* -> str, force_vector -> dict, sample_array -> list of
        the nSamples int8 values the reference exposes, everything else as is."""
        if key not in _FIELDS:
            raise KeyError(key)
        value = getattr(self._c_struct, key)
        if key in _STRINGS:
            return value.decode("utf-8") if value is not None else None
        if key == "force_vector":
            return {k: getattr(value, k) for k in _FV}
        if key == "sample_array":
            if not value:
                return None
            n = self._c_struct.nSamples
            return list((C.c_int8 * n).from_address(value))
        return value

    def set(self, key, value):
        """ref bl_song.py:114-147."""
        if key not in _FIELDS:
            raise KeyError(key)
        if key in _STRINGS:
            buf = None if value is None else C.create_string_buffer(value.encode("utf-8"))
            self._keepalive[key] = buf
            value = None if buf is None else C.cast(buf, C.c_char_p)
        elif key == "force_vector":
            if value is None:
                return None
            if isinstance(value, dict):
                value = _lib.ForceVector(*[value[k] for k in _FV])
            elif not isinstance(value, _lib.ForceVector):
                value = _lib.ForceVector(*value)
        elif key == "sample_array":
            if value is not None:
                arr = (C.c_int8 * len(value))(*value)
                self._keepalive[key] = arr
                value = C.cast(arr, C.c_void_p)
        return setattr(self._c_struct, key, value)

    def decode(self, filename):
        """ref bl_song.py:149-159"""
        return self._lib.bl_audio_decode(filename.encode("utf-8"), C.byref(self._c_struct))

    def analyze(self, filename):
        """ref bl_song.py:161-169"""
        return self._lib.bl_analyze(filename.encode("utf-8"), C.byref(self._c_struct))

    def envelope_analysis(self):
        """ref bl_song.py:171-183"""
        result = _lib.EnvelopeResult()
        self._lib.bl_envelope_sort(C.byref(self._c_struct), C.byref(result))
        return {"tempo": result.tempo, "attack": result.attack}

    def amplitude_analysis(self):
        """ref bl_song.py:185-191 (the reference drops the score; it is returned here)"""
        return self._lib.bl_amplitude_sort(C.byref(self._c_struct))

    def frequency_analysis(self):
        """ref bl_song.py:193-199"""
        return self._lib.bl_frequency_sort(C.byref(self._c_struct))

    def free(self):
        """ref bl_song.py:201-209: drop Python-owned members, then bl_free_song for the
        malloc'd ones."""
        for k in list(self._keepalive):
            del self._keepalive[k]
            setattr(self._c_struct, k, None)
        self._lib.bl_free_song(C.byref(self._c_struct))


def distance(filename1, filename2):
    """ref python/bliss/bl_song.py:212-231: bl_distance_file on two files; returns
    {distance, song1, song2} with the two analysed songs."""
    lib = _lib.load()
    s1, s2 = _lib.BlSong(), _lib.BlSong()
    value = lib.bl_distance_file(filename1.encode("utf-8"), filename2.encode("utf-8"), C.byref(s1), C.byref(s2))
    return {"distance": value, "song1": bl_song(c_struct=s1), "song2": bl_song(c_struct=s2)}


def cosine_similarity(filename1, filename2):
    """ref python/bliss/bl_song.py:234-254.  The reference's wrapper passes four arguments to the
    two-argument bl_cosine_similarity (SURVEY.md section 8b notes the latent bug); the call it
    means is bl_cosine_similarity_file, which is what runs here."""
    lib = _lib.load()
    s1, s2 = _lib.BlSong(), _lib.BlSong()
    value = lib.bl_cosine_similarity_file(filename1.encode("utf-8"), filename2.encode("utf-8"), C.byref(s1),
                                          C.byref(s2))
    return {"similarity": value, "song1": bl_song(c_struct=s1), "song2": bl_song(c_struct=s2)}
