"""Runs the device arithmetic codelets (bl_fft.h, bl_tail.h are __host__ __device__) on the
CPU: the FFT lane code against a long-double DFT, the streaming envelope tail against the
oracle (bit-exact beat / atk_sum, including the shortest arrays the reference supports)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "tests", "host")


def _run(tmp_path, src, extra):
    exe = str(tmp_path / (src + ".bin"))
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", os.path.join(HOST, src)] + extra
                   + ["-o", exe, "-lm"], check=True, stderr=subprocess.DEVNULL)
    out = subprocess.run([exe], stdout=subprocess.PIPE, text=True)
    assert out.returncode == 0, out.stdout
    assert out.stdout.strip().endswith("OK"), out.stdout


def test_fft_lane_code(tmp_path):
    _run(tmp_path, "test_fft_host.cpp", [])


def test_lavc_order_lane_code_is_bit_identical_to_the_oracle(tmp_path):
    """bl_fft_lavc.h (the frequency kernel's f32 DFT: libavcodec's operation order on 16 lanes x 16 registers) against
    oracle/orc_fft_lavc.c: 127 500 power values of 500 frames, bit for bit."""
    _run(tmp_path, "test_fft_lavc_host.cpp", ["-x", "c", os.path.join(ROOT, "oracle", "orc_fft_lavc.c")])


def test_streaming_tail_matches_oracle(tmp_path):
    orc = [os.path.join(ROOT, "oracle", f) for f in ("bliss_oracle.c", "orc_fft.c", "orc_fft_alt.c", "orc_fft_lavc.c", "orc_synth.c")]
    _run(tmp_path, "test_tail_host.cpp", ["-x", "c"] + orc)


def test_issue_microbenchmark_generator_assembles(tmp_path):
    """tools/gen_ubench_issue.py writes hand-assembled gfx950 instruction streams (the measurements DESIGN.md section
    4.1 rests on): the generated source must still assemble for gfx950 (device code only; no GPU needed)."""
    import shutil
    import sys
    import pytest
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    src = tmp_path / "ubench_issue.hip"
    gen = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_ubench_issue.py")], stdout=subprocess.PIPE,
                         text=True, check=True).stdout
    assert gen.count("__global__ void k_") >= 20
    src.write_text(gen)
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-w", "--offload-device-only", "-c", str(src), "-o",
                        str(tmp_path / "ubench_issue.o")], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
