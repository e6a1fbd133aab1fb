#!/usr/bin/env python3
"""Summarise a rocprofv3 (ROCm 7.2, rocpd SQLite output) kernel trace as text:
per-kernel calls / total / average / share (the `--stats` table) plus launch geometry
and register/LDS footprint of our kernels.  Usage: rocpd_summary.py results.db > summary.txt"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    print(f"# rocprofv3 --kernel-trace --stats summary of {path.split('/')[-1]} (durations in microseconds)")
    print(f"{'calls':>6} {'total_us':>14} {'avg_us':>13} {'pct':>7}  kernel")
    for name, calls, total, avg, pct in cur.execute(
            "select name,total_calls,total_duration,average,percentage from top_kernels"):
        short = name.split("(")[0] if name.startswith(("k_", "void k_")) else name[:70]
        print(f"{calls:>6} {total:>14.1f} {avg:>13.1f} {pct:>7.2f}  {short}")
    print("\n# launch geometry of bliss_amd kernels (first dispatch of each)")
    print(f"{'kernel':<24} {'grid':>18} {'wg':>6} {'lds_B':>8} {'vgpr':>5} {'agpr':>5} {'sgpr':>5} {'scratch':>8}")
    seen = set()
    for r in cur.execute("select name,grid_x,grid_y,grid_z,workgroup_x,lds_size,vgpr_count,"
                         "accum_vgpr_count,sgpr_count,scratch_size from kernels order by start"):
        k = r[0].split("(")[0].replace("void ", "")
        if not k.startswith("k_") or k in seen:
            continue
        seen.add(k)
        print(f"{k:<24} {f'{r[1]}x{r[2]}x{r[3]}':>18} {r[4]:>6} {r[5]:>8} {r[6]:>5} {r[7]:>5} {r[8]:>5} {r[9]:>8}")


if __name__ == "__main__":
    main(sys.argv[1])
