#!/usr/bin/env python3
"""The 10 000 x 10 000 bl_distance / bl_cosine_similarity matrices (BASELINE configs[3]) timed
with HIP events around the kernel (bl_amd_profile), per square-root variant of the distance
kernel (BL_AMD_SQRT_VARIANT), plus the exhaustive self-test of the variant.  One JSON object.
The variants only exist in the measurement build (`make -C bliss_amd/csrc measure`, -DBL_AMD_MEASURE), which
this tool builds if needed and loads through BLISS_AMD_LIB; the product library ignores the variable.
usage: python tools/dist_bench.py [--n 10000] [--reps 50]"""
import argparse
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
MEASURE_LIB = os.path.join(ROOT, "bliss_amd", "libbliss_amd_measure.so")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=10000)
    ap.add_argument("--reps", type=int, default=50)
    ap.add_argument("--variants", default="0,1,2,3")
    a = ap.parse_args()
    import subprocess
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "bliss_amd", "csrc"), "measure"], check=True)
    os.environ["BLISS_AMD_LIB"] = MEASURE_LIB
    import torch
    import bliss_amd
    lib = bliss_amd.load()
    assert lib.bl_amd_init(0) == 0
    g = torch.Generator(device="cpu").manual_seed(4)
    v = (torch.randn((a.n, 4), generator=g) * 8).cuda()
    m = torch.empty((a.n, a.n), dtype=torch.float32, device="cuda")
    bytes_ = 4 * a.n * a.n + 16 * a.n
    out = {"n": a.n, "reps": a.reps, "algorithmic_bytes": bytes_, "variants": {}}
    ref = None

    def timed(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        lib.bl_amd_profile_reset()
        lib.bl_amd_profile(1)
        for _ in range(a.reps):
            fn()
        torch.cuda.synchronize()
        lib.bl_amd_profile(0)
        k = C.c_int(0)
        ms = lib.bl_amd_profile_ms(b"distance", C.byref(k))
        return 1e3 * ms / max(k.value, 1)

    for var in [int(x) for x in a.variants.split(",")]:
        os.environ["BL_AMD_SQRT_VARIANT"] = str(var)
        us = timed(lambda: lib.bl_amd_distance_matrix_device(C.c_void_p(v.data_ptr()), a.n, 0, a.n,
                                                             C.c_void_p(m.data_ptr()), None))
        got = m.clone()
        if ref is None:
            ref = got
        counts = (C.c_uint64 * 3)()
        assert lib.bl_amd_selftest_sqrt(counts) == 0
        out["variants"][str(var)] = {"distance_us": us, "TBps": bytes_ / us / 1e6, "frac_hbm_peak": bytes_ / us / 1e6 / 8.0,
                                     "matrix_equals_variant0": bool(torch.equal(got, ref)),
                                     "selftest": {"checked": int(counts[0]), "fast_mismatches": int(counts[1]),
                                                  "fallback_mismatches": int(counts[2])}}
    os.environ.pop("BL_AMD_SQRT_VARIANT", None)
    us = timed(lambda: lib.bl_amd_cosine_matrix_device(C.c_void_p(v.data_ptr()), a.n, 0, a.n,
                                                       C.c_void_p(m.data_ptr()), None))
    out["cosine_us"] = us
    out["cosine_frac_hbm_peak"] = bytes_ / us / 1e6 / 8.0
    print(json.dumps(out))


if __name__ == "__main__":
    main()
