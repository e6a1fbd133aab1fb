#!/usr/bin/env python3
"""bench.py — throughput of the bliss analysis hot path on MI355X.

One *step* = one pass of the hot path over one resident batch of synthetic decoded
songs: pcm_scan -> amplitude / frequency / envelope kernels -> force vectors
(bl_analyze after decode, ref src/analyze.c:40-80), then — as BASELINE.json's
batch-of-songs mode asks — an all-gather of the 16-byte force vectors over RCCL and this
rank's row block of the bl_distance matrix (ref src/analyze.c:96-100).

Workload: BASELINE.json configs[2] shape — 3-minute 44.1 kHz s16 stereo buffers
(15 876 000 interleaved int16 each), `--songs-per-gpu` of them resident in HBM per rank
(default: the configs[2] shard of 8 192 per GPU when it fits, else the largest count that
does; the count used is printed in config.workload).  Songs are sharded by index across
ranks (weak scaling, no data-path collective other than the vector all-gather).

Launch: `python bench.py` (1 GPU) or
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
 --master-port P bench.py --gpus N --steps K --warmup W`.
Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SONG_SAMPLES = 44100 * 2 * 180          # S180 of SURVEY.md §8
SONG_SECONDS = 180
SAMPLE_RATE = 44100
HBM_PEAK_GBS = 8000.0                   # spec, /opt/skills/guides/MI355X_MICROARCH.md
HBM_ACHIEVABLE_GBS = 6290.0             # measured float4 copy, same guide
FP64_VALU_PEAK_TFLOPS = 78.6            # spec (FMA = 2 flop); the faithful path's real ceiling


def _cpu_limits():
    """What the box lets this process use: affinity mask, cgroup quota."""
    info = {"os_cpu_count": os.cpu_count()}
    try:
        info["sched_affinity"] = len(os.sched_getaffinity(0))
    except Exception:
        info["sched_affinity"] = None
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                quota = None if txt[0] == "max" else float(txt[0]) / float(txt[1])
            else:
                q = float(txt[0])
                per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                quota = None if q <= 0 else q / per
            break
        except Exception:
            continue
    info["cgroup_cpu_quota"] = quota
    return info


def cpu_baseline():
    """Oracle (CPU restatement, kind 'port') timed on this box's host cores on a bounded sample of
    the same workload.  One process per worker (not threads: the reference is not thread
    re-entrant — FFTW planner globals, ref src/tempo_atk_sort.c:94,294-295), each analysing
    synthetic 3-minute songs; like the GPU number, the rate counts analysis time only (orc_cli
    times orc_analyze_pcm, not the integer synthesis).  A ladder of 1 / 8 / 32 / 64 / 128 / 256
    concurrent processes shows where the box stops scaling: `cores` is the smallest process count
    that reaches 90 % of the best throughput — or the cgroup CPU quota when that is smaller (the
    GPU boxes of this pool expose 256 hardware threads under a quota of 16 CPUs) — and `value`
    the best throughput."""
    from tests.oracle_py import build_oracle
    build_oracle()
    cli = os.path.join(ROOT, "oracle", "orc_cli")
    limits = _cpu_limits()
    avail = limits["sched_affinity"] or limits["os_cpu_count"] or 1

    def level(procs, per_proc, seed0):
        t0 = time.time()
        ps = [subprocess.Popen([cli, "time", str(seed0 + 16 * i), str(SAMPLE_RATE), "2", str(SONG_SECONDS),
                                str(per_proc)], stdout=subprocess.PIPE, text=True) for i in range(procs)]
        rate, done = 0.0, 0
        for p in ps:
            o, _ = p.communicate()
            try:
                r = json.loads(o.strip().splitlines()[-1])
                rate += r["songs"] / r["seconds"]
                done += r["songs"]
            except Exception:
                pass
        wall = time.time() - t0
        return {"processes": procs, "songs": done, "songs_per_s": rate, "wall_s": wall}

    ladder = [level(1, 2, 9000)]
    one = ladder[0]["songs_per_s"]
    for procs in (8, 32, 64, 128, 256):
        if procs > max(avail, 1):
            break
        ladder.append(level(procs, 1, 9100 + procs))
    if ladder[-1]["processes"] != avail and avail > 1 and avail not in (8, 32, 64, 128, 256):
        ladder.append(level(avail, 1, 9700))
    best = max(l["songs_per_s"] for l in ladder)
    eff = next(l["processes"] for l in ladder if l["songs_per_s"] >= 0.9 * best)
    # a cgroup CPU quota below that count is the real amount of CPU the processes shared
    quota = limits.get("cgroup_cpu_quota")
    if quota and quota < eff:
        eff = int(round(quota))
    model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    # second half of the metric: the 10 000 x 10 000 bl_distance matrix as the nested loop of
    # the reference's per-pair function, one core
    dm_cpu = None
    try:
        import numpy as np
        from tests.oracle_py import Oracle
        orc = Oracle()
        v = (np.random.default_rng(4).standard_normal((10000, 4)) * 8).astype(np.float32)
        out = np.empty((10000, 10000), dtype=np.float32)
        t1 = time.time()
        orc.lib.orc_distance_matrix(v.ctypes.data_as(C.POINTER(C.c_float)), 10000,
                                    out.ctypes.data_as(C.POINTER(C.c_float)))
        dm_cpu = time.time() - t1
    except Exception:
        pass
    songs = sum(l["songs"] for l in ladder)
    return {"value": best, "unit": "songs/s", "cores": eff, "kind": "port",
            "distance_matrix_10k_s_one_core": dm_cpu,
            "sample": f"{songs} synthetic 3-min 44.1 kHz s16 stereo songs over a ladder of "
                      f"{[l['processes'] for l in ladder]} concurrent single-threaded processes "
                      "(1 song each, 2 at the first level), analysis time only; cores = min(smallest "
                      "process count within 10 % of the best throughput, cgroup CPU quota)",
            "ladder": ladder, "limits": limits, "cpu_model": model,
            "one_core_songs_per_s": one, "scaling_vs_one_core": best / one if one else None}


def verify_songs(res, picks, seed_first, seconds):
    """Untimed: re-synthesise `picks` of the resident batch on the host (same integer generator,
    seeds = global song index) and analyse them with the CPU oracle (orc_cli); integers must be
    identical, f32 features within 1e-4 relative (north_star).  Returns (ok, details)."""
    from tests.oracle_py import build_oracle
    build_oracle()
    cli = os.path.join(ROOT, "oracle", "orc_cli")
    procs = [subprocess.Popen([cli, "synth", str(seed_first + i), str(SAMPLE_RATE), "2", str(seconds)],
                              stdout=subprocess.PIPE, text=True) for i in picks]
    ok, details = True, []
    for i, p in zip(picks, procs):
        o, _ = p.communicate()
        ref = json.loads(o.strip().splitlines()[-1])
        g = res[i]
        bad = [k for k in ("start", "end", "mean", "variance", "n_frames", "nb_frames", "n_windows", "beat",
                           "calm_or_loud") if int(g[k]) != int(ref[k])]
        worst = 0.0
        for k in ("tempo", "amplitude", "frequency", "attack", "force"):
            a, b = float(g[k]), float(ref[k])
            rel = abs(a - b) / max(abs(b), 1e-6)
            worst = max(worst, rel)
            if not rel <= 1e-4:
                bad.append(k)
        details.append({"song": int(i), "beat": int(g["beat"]), "max_rel_err": worst, "mismatch": bad})
        ok = ok and not bad
    return ok, details


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--songs-per-gpu", type=int, default=0,
                    help="0 = configs[2] shard (8192) if it fits in HBM, else the largest count that does")
    ap.add_argument("--seconds", type=int, default=SONG_SECONDS, help="song length (default 180)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--verify", type=int, default=4,
                    help="songs of the resident batch re-analysed by the CPU oracle after the timed region")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    under_launcher = "RANK" in os.environ and "MASTER_ADDR" in os.environ
    if world > 1 or under_launcher:  # torchrun: RCCL group even at world size 1
        dist.init_process_group("nccl", device_id=dev)

    import bliss_amd
    from bliss_amd import _lib
    lib = bliss_amd.load()
    assert lib.bl_amd_init(local_rank) == 0

    song_samples = SAMPLE_RATE * 2 * args.seconds
    song_bytes = 2 * song_samples
    # per-song scratch: 12 B per envelope slot + histogram + small records
    scratch_per_song = 12 * (2 * (song_samples // 512)) + 4 * 4096 + 4096
    free_b, total_b = torch.cuda.mem_get_info(dev)
    want = args.songs_per_gpu if args.songs_per_gpu > 0 else 8192
    margin = 8 << 30
    fit = int((free_b - margin) // (song_bytes + scratch_per_song + 16 + 4 * want * world))
    songs = want
    capped = False
    if songs > fit:
        songs = max(1, 1 << (max(fit, 1).bit_length() - 1))
        capped = True
    if dist.is_initialized():  # every rank uses the smallest count any rank can hold
        t = torch.tensor([songs], device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        songs = int(t.item())
    from bliss_amd.dist import gather_force_vectors, shard_range
    total_songs = songs * world
    my_first, my_count = shard_range(total_songs, rank, world)
    assert my_count == songs

    corpus = bliss_amd.DeviceCorpus([song_samples] * songs, 2, args.seconds, device=f"cuda:{local_rank}")
    corpus.synth(seed_base=my_first, sample_rate=SAMPLE_RATE)
    torch.cuda.synchronize(dev)

    all_vecs = torch.empty((total_songs, 4), dtype=torch.float32, device=dev)
    rows = torch.empty((songs, total_songs), dtype=torch.float32, device=dev)
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)

    n_gathers = [0]

    def step():
        corpus.analyze()
        mine = corpus.force_vectors()
        all_vecs.copy_(gather_force_vectors(mine, [songs] * world))   # RCCL all-gather, 16 B/song
        n_gathers[0] += 1
        rc = lib.bl_amd_distance_matrix_device(C.c_void_p(all_vecs.data_ptr()), total_songs, my_first,
                                               songs, C.c_void_p(rows.data_ptr()), stream)
        assert rc == 0

    def fence():
        torch.cuda.synchronize(dev)
        if dist.is_initialized():
            dist.barrier()
            torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    fence()
    lib.bl_amd_profile_reset()
    lib.bl_amd_profile(1)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    lib.bl_amd_profile(0)
    if dist.is_initialized():
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # per-kernel device time of the timed region (HIP events on the launch stream)
    kern = {}
    for name in ("pcm_scan", "amp_finish", "freq_frames", "freq_finish", "env_windows", "env_tail",
                 "distance"):
        n = C.c_int(0)
        ms = lib.bl_amd_profile_ms(name.encode(), C.byref(n))
        kern[name] = {"ms_total": ms, "launches": n.value,
                      "ms_avg": (ms / n.value) if n.value else None}

    res = corpus.fetch()
    ok = bool(np.all(res["status"] == 0) and np.all(np.isfinite(res["force"])))
    # the gathered vectors are the analysed ones, rank-major, and this rank's rows are distances
    ok = ok and bool(torch.equal(all_vecs[my_first:my_first + songs].cpu(),
                                 torch.from_numpy(np.stack([res[k] for k in ("tempo", "amplitude", "frequency",
                                                                             "attack")], axis=1))))
    verified, verify_details = 0, []
    if rank == 0 and args.verify > 0:
        k = min(args.verify, songs)
        picks = sorted(set(int(round(j * (songs - 1) / max(k - 1, 1))) for j in range(k)))
        v_ok, verify_details = verify_songs(res, picks, my_first, args.seconds)
        ok = ok and v_ok
        verified = len(picks)

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        value = total_songs * args.steps / elapsed
        alg_bytes_song = song_bytes + 16          # 2n read + 16 written
        dom = kern["env_windows"]
        launch_bytes = alg_bytes_song * songs     # one env_windows launch covers this rank's batch
        roof = None
        if dom["ms_avg"]:
            ach = launch_bytes / (dom["ms_avg"] * 1e-3) / 1e9
            # faithful-arithmetic count of the kernel: ~75 f64 VALU instructions per sample
            # (normalise 4, FIR 26, FFT+split 33, f32-ordered sum 3, log 0.1)
            f64_rate = 75.0 * song_samples * songs / (dom["ms_avg"] * 1e-3) / 1e12
            # HBM bytes per launch: FETCH_SIZE / WRITE_SIZE from the committed separate-pass PMC
            # profile (profiles/*_hbm_traffic.json, corrected as MI355X_MICROARCH.md prescribes),
            # scaled from its per-song figure to this launch's song count; None if absent
            traffic = None
            valu_busy = lds_busy = None   # SQ counters of the same committed profile, if collected
            try:
                import glob
                tj = json.load(open(sorted(glob.glob(os.path.join(ROOT, "profiles", "*_hbm_traffic.json")))[-1]))
                tk = tj["kernels"].get("k_env_windows3") or tj["kernels"]["k_env_windows2"]
                traffic = tk["hbm_bytes_per_song"] * songs * (song_samples / 15876000.0)
                valu_busy, lds_busy = tk.get("valu_busy_frac"), tk.get("lds_busy_frac")
            except Exception:
                pass
            roof = {"bound": "hbm", "kernel": "k_env_windows3", "achieved": ach, "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
                    "traffic_source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), "
                                      "profiles/*_hbm_traffic.json, scaled per song",
                    "frac_of_measured_copy_peak": ach / HBM_ACHIEVABLE_GBS,
                    "ms_avg_launch": dom["ms_avg"], "launches": dom["launches"],
                    "algorithmic_bytes_per_launch": launch_bytes,
                    "secondary_f64_valu": {"achieved_Tinstr_per_s": f64_rate,
                                           "peak_Tinstr_per_s": FP64_VALU_PEAK_TFLOPS / 2,
                                           "frac": f64_rate / (FP64_VALU_PEAK_TFLOPS / 2),
                                           "valu_busy_frac_profiled": valu_busy,
                                           "lds_busy_frac_profiled": lds_busy}}
        whole_path_gbs = value / world * alg_bytes_song / 1e9

        # BASELINE config 4: standalone 10 000 x 10 000 bl_distance matrix on one GPU
        g = torch.Generator(device="cpu").manual_seed(4)
        v10 = (torch.randn((10000, 4), generator=g) * 8).to(dev)
        m10 = torch.empty((10000, 10000), dtype=torch.float32, device=dev)
        for _ in range(2):
            lib.bl_amd_distance_matrix_device(C.c_void_p(v10.data_ptr()), 10000, 0, 10000,
                                              C.c_void_p(m10.data_ptr()), stream)
        torch.cuda.synchronize(dev)
        reps = 10
        t1 = time.perf_counter()
        for _ in range(reps):
            lib.bl_amd_distance_matrix_device(C.c_void_p(v10.data_ptr()), 10000, 0, 10000,
                                              C.c_void_p(m10.data_ptr()), stream)
        torch.cuda.synchronize(dev)
        dm_s = (time.perf_counter() - t1) / reps
        dm_bytes = 4 * 10000 * 10000 + 16 * 10000

        line = {
            "metric": "songs/sec bl_analyze (3-min 44.1kHz s16) + 10k x 10k distance-matrix sec",
            "value": value, "unit": "songs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"BASELINE configs[2] shape: {songs} synthetic {args.seconds}-s 44.1 kHz s16 "
                                   f"stereo songs resident per GPU ({songs * song_bytes / 1e9:.1f} GB PCM/GPU"
                                   + (", capped by free HBM" if capped else "")
                                   + f"), {total_songs} songs total, sharded by song; step = analyze + "
                                     "all-gather of force vectors + row-block distance matrix",
                       "songs_per_gpu": songs, "song_samples": song_samples, "parallelism": f"shard{world}",
                       "generator": "integer-only device synth, seeds = global song index"},
            "distance_matrix_10k_s": dm_s,
            "distance_matrix_10k_gbs": dm_bytes / dm_s / 1e9,
            "distance_matrix_10k_frac_hbm": dm_bytes / dm_s / 1e9 / HBM_PEAK_GBS,
            "whole_path_algorithmic_gbs_per_gpu": whole_path_gbs,
            "whole_path_frac_hbm": whole_path_gbs / HBM_PEAK_GBS,
            "kernels_ms": kern, "results_ok": ok, "verified_songs": verified,
            "verification": {"against": "CPU oracle (oracle/orc_cli) on the re-synthesised songs, untimed",
                             "bar": "integers identical, f32 features <= 1e-4 relative",
                             "songs": verify_details},
            "collective": {"backend": dist.get_backend() if dist.is_initialized() else None,
                           "all_gather_calls": n_gathers[0] if dist.is_initialized() else 0,
                           "bytes_per_rank": 16 * songs},
            "roofline": roof,
        }
        if not args.no_cpu_baseline and world == 1:
            try:
                line["cpu_baseline"] = cpu_baseline()
            except Exception as e:  # the baseline must never sink the GPU number
                line["cpu_baseline"] = {"value": None, "unit": "songs/s", "cores": None,
                                        "kind": "port", "sample": f"failed: {e}"}
        print(json.dumps(line), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
