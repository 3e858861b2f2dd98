#!/usr/bin/env python3
"""Summarise a rocprofv3 (ROCm 7.2 rocpd SQLite) kernel trace: per-kernel calls / avg / min / max / total, as markdown.

  rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o bench -- python bench.py ...
  python tools/rocpd_summary.py gpurun_out/prof/bench_results.db > profiles/rNN_kernel_stats.md
"""
import sqlite3
import sys


def main(path, skip_first=0):
    c = sqlite3.connect(path)
    rows = c.execute("select name, duration, lds_size, scratch_size, vgpr_count, accum_vgpr_count, sgpr_count, grid_x, workgroup_x, start from kernels order by start").fetchall()
    per = {}
    for name, dur, lds, scr, vg, ag, sg, gx, wx, st in rows:
        per.setdefault(name, []).append((dur, lds, scr, vg, ag, sg, gx, wx))
    total = sum(d[0] for v in per.values() for d in v)
    print(f"# rocprofv3 kernel trace summary ({path})\n")
    print("| kernel | calls | avg us | min us | max us | total ms | % | grid | wg | LDS B | scratch B | VGPR | AGPR | SGPR |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|---|---|")
    for name, v in sorted(per.items(), key=lambda kv: -sum(d[0] for d in kv[1])):
        d = [x[0] for x in v][skip_first:] or [x[0] for x in v]
        short = name.split("(")[0].replace("qmk::", "")
        print(f"| {short} | {len(v)} | {sum(d) / len(d) / 1e3:.1f} | {min(d) / 1e3:.1f} | {max(d) / 1e3:.1f} | {sum(x[0] for x in v) / 1e6:.2f} | "
              f"{100 * sum(x[0] for x in v) / total:.1f} | {v[-1][6]} | {v[-1][7]} | {v[-1][1]} | {v[-1][2]} | {v[-1][3]} | {v[-1][4]} | {v[-1][5]} |")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 0)
