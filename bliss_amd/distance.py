"""`distance` / `cosine_similarity` — counterparts of ref python/bliss/distance.py:5-76:
two filenames -> bl_*_file (analyses both), two bl_song objects -> bl_distance /
bl_cosine_similarity on their force vectors, anything else -> None entries."""
import ctypes as C

from . import _lib
from .bl_song import bl_song


def _fv(song):
    v = song["force_vector"]
    return _lib.ForceVector(v["tempo"], v["amplitude"], v["frequency"], v["attack"])


def _pair(song1, song2, key, file_fn, vec_fn):
    lib = _lib.load()
    if isinstance(song1, str) and isinstance(song2, str):
        s1, s2 = _lib.BlSong(), _lib.BlSong()
        value = getattr(lib, file_fn)(song1.encode("utf-8"), song2.encode("utf-8"), C.byref(s1),
                                      C.byref(s2))
        return {key: value, "song1": bl_song(c_struct=s1), "song2": bl_song(c_struct=s2)}
    if isinstance(song1, bl_song) and isinstance(song2, bl_song):
        return {key: getattr(lib, vec_fn)(_fv(song1), _fv(song2)), "song1": song1, "song2": song2}
    return {key: None, "song1": None, "song2": None}


def distance(song1, song2):
    """ref python/bliss/distance.py:5-39"""
    return _pair(song1, song2, "distance", "bl_distance_file", "bl_distance")


def cosine_similarity(song1, song2):
    """ref python/bliss/distance.py:42-76"""
    return _pair(song1, song2, "similarity", "bl_cosine_similarity_file", "bl_cosine_similarity")
