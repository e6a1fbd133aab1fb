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


def _full_record(n_gpus):
    """A full bench record shaped like the ones bench.py writes to bench_details.json (values of round 5's 8 192-song
    run), with every bulky part present: 32 verified songs, the CPU ladder, traffic by kernel, per-rank lists."""
    songs = [{"song": i, "beat": 350 + i, "max_rel_err": 1.3e-6, "rel_err_by_field": {k: 1.1e-9 for k in "abcde"},
              "value_by_field": {k: -3.48556519 for k in "abcde"}, "mismatch": []} for i in range(32)]
    return {
        "metric": "songs/sec bl_analyze (3-min 44.1kHz s16) + 10k x 10k distance-matrix sec", "value": 23688.0234567,
        "unit": "songs/s", "n_gpus": n_gpus, "steps": 20, "warmup": 5, "ms_per_step": 345.82891234, "higher_is_better": True,
        "value_fir_mode0": 19975.123456, "ms_per_step_fir_mode0": 410.1112345, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": "x" * 400, "workload_short": "configs[2] shard: 8192 synthetic 180-s 44.1 kHz s16 stereo songs/GPU "
                   "resident (260.1 GB), 65536 total; step = analyze + all-gather + row-block distance",
                   "songs_per_gpu": 8192, "song_samples": 15876000, "parallelism": f"shard{n_gpus}", "generator": "y" * 60},
        "distance_matrix_10k_s": 7.6391234e-05, "distance_matrix_10k_frac_hbm": 0.65512345, "cosine_matrix_10k_s": 7.729e-05,
        "whole_path_frac_hbm": 0.0903, "kernels_ms": {k: {"ms_total": 1.0, "launches": 20, "ms_avg": 0.05} for k in "abcdefgh"},
        "fir_modes": {"timed_mode": 2, "what": "z" * 200}, "results_ok": True, "verified_songs": 32,
        "verification": {"n_failing_strict_1e-4_rel": {k: 0 for k in ("tempo", "amplitude", "frequency", "attack", "force")},
                         "songs": songs},
        "collective": {"backend": "nccl", "all_gather_calls": 30, "bytes_per_rank": 131072},
        "per_rank": {"ms_per_step": [345.123456 + i for i in range(n_gpus)], "env_windows_ms": [274.2] * n_gpus},
        "device_state": {"sclk_mhz": {"mean": 2155.2}, "power_w": {"mean": 1347.2}, "power_cap_w": 1400.0,
                         "joules_per_song": 0.056912345, "per_rank_sclk_mhz": [2155.2] * n_gpus},
        "other_configs": {"configs1": {"ms_per_batch": 8.7641, "songs_per_s": 116841.2, "results_ok": True, "workload": "w" * 100},
                          "configs4_mixed": {"ms_per_batch": 219.31, "songs_per_s": 37355.1, "results_ok": True}},
        "roofline": {"bound": "hbm", "kernel": "k_env_windows3", "achieved": 948.6234, "peak": 8000.0, "unit": "GB/s",
                     "frac": 0.118577925, "traffic": 266357123456.0, "traffic_source": {"what": "collected in this run: ...",
                     "bytes_per_song_by_kernel": {f"k_{i}": 1.5 for i in range(12)}}, "ms_avg_launch": 274.2012,
                     "algorithmic_bytes_per_launch": 260112515072, "frac_of_f64_floor": 0.81723, "frac_of_f64_floor_nominal": 0.7449,
                     "whole_step_traffic_ratio": 2.0412, "frac_fir_mode0": 0.09606,
                     "other_pcm_pass": {"kernel": "k_freq_scan", "ms_avg_launch": 64.1, "frac": 0.507}},
        "cpu_baseline": {"value": 44.406524, "unit": "songs/s", "cores": 16, "kind": "port", "sample": "s" * 400,
                         "sample_short": "106 synthetic 3-min songs over 1/8/32/64 concurrent 1-thread oracle processes; rank 0 of 8",
                         "ladder": [{"processes": p, "songs": p, "songs_per_s": 40.0, "wall_s": 1.0} for p in (1, 8, 32, 64)],
                         "cpu_model": "AMD EPYC 9575F 64-Core Processor", "one_core_songs_per_s": 2.7286246},
    }


def test_compact_line_fits_the_drivers_tail():
    """BENCH_r05.json: `parsed: null` — the driver keeps the last 2 000 bytes of stdout + stderr and round 5's line had
    21 431.  The printed line is a digest of the full record: every key of the bench contract, `roofline` and
    `cpu_baseline` whole, at most LINE_BYTES_MAX bytes at N = 1 and N = 8, nothing bulky."""
    omp_note = ("W0929 10:11:12.123000 140 torch/distributed/run.py:803] \n" + "*" * 41 + "\nSetting OMP_NUM_THREADS environment "
                "variable for each process to be 1 in default, to avoid your system being overloaded, please further tune the "
                "variable for optimal performance in your application as needed. \n" + "*" * 41 + "\n")
    for n in (1, 8):
        rec = _full_record(n)
        assert len(json.dumps(rec)) > 8000
        limit = bench.LINE_BYTES_MAX if n == 1 else bench.LINE_BYTES_MAX_LAUNCHED
        line = bench.compact_line(rec, "/somewhere/bench_details.json", limit)
        text = json.dumps(line, separators=(",", ":"))
        assert len(text) <= limit <= 1800, len(text)
        # what the driver keeps: the ranks' own chatter is in their log files (QuietFds); at N = 1 the HIP runtime's
        # libdrm note precedes the redirection, at N > 1 the launcher adds its OMP_NUM_THREADS note
        stderr = "/opt/amdgpu/share/libdrm/amdgpu.ids: No such file or directory\n" if n == 1 else omp_note
        back = json.loads((text + "\n\n---- stderr ----\n" + stderr)[-2000:].splitlines()[0])
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                  "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "results_ok", "verified_songs"):
            assert k in back, k
        assert back["config"]["workload"].startswith("configs[2] shard") and back["config"]["parallelism"] == f"shard{n}"
        assert {"bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "ms_avg_launch", "algorithmic_bytes_per_launch",
                "frac_of_f64_floor", "frac_of_f64_floor_nominal", "whole_step_traffic_ratio"} <= set(back["roofline"])
        assert {"value", "unit", "cores", "kind", "sample", "cpu_model", "one_core_songs_per_s"} <= set(back["cpu_baseline"])
        assert back["roofline"]["traffic_src"] == "live_pmc" and back["details"] == "bench_details.json"
        assert back["value_fir_mode0"] == 19975.1 and back["value"] == 23688.02 and back["collective"] == "nccl"
        if n == 1:
            assert back["other_configs"]["ok"] is True and back["device_state"]["power_cap_w"] == 1400.0
            assert "per_rank_ms" not in back
        else:
            assert len(back["per_rank_ms"]) == n
    # a record too long for the limit gives up keys from the end of the optional list, never a contract key
    short = bench.compact_line(_full_record(8), "d.json", limit=1300)
    assert len(json.dumps(short, separators=(",", ":"))) <= 1300 and "roofline" in short and "cpu_baseline" in short
    assert "whole_path_frac_hbm" not in short


def test_f64_floors():
    # the nominal issue rate: 256 CUs x 4 SIMDs, one f64 wave-instruction per 4 cycles at 2.4 GHz = 78.6 TF as FMA
    assert abs(bench.F64_ISSUE_NOMINAL_TWAVEINSTR_S - 0.6144) < 1e-9
    assert abs(bench.F64_ISSUE_NOMINAL_TWAVEINSTR_S * 64 * 2 - bench.FP64_VALU_PEAK_TFLOPS) < 0.1
    assert bench.F64_ISSUE_TWAVEINSTR_S < bench.F64_ISSUE_NOMINAL_TWAVEINSTR_S
