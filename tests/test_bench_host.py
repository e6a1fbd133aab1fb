"""Host-side pieces of bench.py that need no GPU: what the bench line derives from the committed
PMC profile, and the arithmetic of the instruction floor."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402


def test_committed_profile_is_read_and_scaled():
    p = bench._committed_profile(8192, bench.SONG_SAMPLES)
    assert p["error"] is None and p["file"].startswith("profiles/") and p["file"].endswith("_hbm_traffic.json")
    raw = json.load(open(os.path.join(ROOT, p["file"])))
    k = raw["kernels"]["k_env_windows3"]
    # traffic: the profile's per-song HBM bytes times this launch's songs; ~1.02 x the algorithmic bytes
    assert p["traffic"] == k["hbm_bytes_per_song"] * 8192
    assert 1.0 <= p["traffic"] / (8192 * (2 * bench.SONG_SAMPLES + 16)) < 1.1
    # half the songs of half the length: a quarter of the bytes
    q = bench._committed_profile(4096, bench.SONG_SAMPLES // 2)
    assert abs(q["traffic"] / p["traffic"] - 0.25) < 1e-9
    assert p["fir_mode"] in (0, 1, 2) and 200 < p["valu_instr_per_window"] < 400
    assert 1.9 < p["whole_step_traffic_ratio"] < 2.3  # two passes over the PCM (k_freq_scan, k_env_windows3)


def test_committed_profile_belongs_to_the_committed_kernels():
    """The bench line scales the counters of the newest profiles/*_hbm_traffic.json to its launch: that profile must
    have been taken at (or after) the last commit that touched the kernels, or its numbers describe other code."""
    import subprocess
    import pytest
    p = bench._committed_profile(8192, bench.SONG_SAMPLES)
    head = (p.get("git_head") or "").split("+")[0]
    git = ["git", "-C", ROOT]
    try:
        last = subprocess.run(git + ["log", "-1", "--format=%H", "--", "bliss_amd/csrc/bl_kernels.hip", "bliss_amd/csrc/bl_fft.h"],
                              stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, check=True).stdout.strip()
    except (OSError, subprocess.CalledProcessError):
        pytest.skip("not a git checkout")
    if not last:
        pytest.skip("no history for the kernels")
    assert head, f"{p['file']} carries no git_head"
    assert "uncommitted" not in (p.get("git_head") or ""), f"{p['file']} was taken on uncommitted kernel sources"
    r = subprocess.run(git + ["merge-base", "--is-ancestor", last, head], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, (f"{p['file']} was taken at {head[:12]}, the kernels last changed in {last[:12]}: "
                               "regenerate it (tools/make_profiles.sh)")


def test_missing_profile_is_an_explicit_error(tmp_path, monkeypatch):
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    p = bench._committed_profile(8192, bench.SONG_SAMPLES)
    assert p["traffic"] is None and "no profiles" in p["error"]
    os.makedirs(tmp_path / "profiles")
    (tmp_path / "profiles" / "r99_hbm_traffic.json").write_text("{\"kernels\": {}}")
    p = bench._committed_profile(8192, bench.SONG_SAMPLES)
    assert p["traffic"] is None and p["error"].startswith("KeyError")


def test_instruction_floor_table():
    # per window: 272 FIR outputs (256 + 16 heads) on 64 lanes; 25 -> 17 operations per output, then
    # the two f64 operations per sample of the normalisation
    assert bench.F64_FLOOR_INSTR_PER_WINDOW[0] - bench.F64_FLOOR_INSTR_PER_WINDOW[1] == 272 * 8 // 64
    assert bench.F64_FLOOR_INSTR_PER_WINDOW[1] - bench.F64_FLOOR_INSTR_PER_WINDOW[2] == 256 * 2 // 64
