"""Start / end of the last kernels of a rocprofv3 --kernel-trace database (rocpd view `kernels`), relative to the first one printed: shows which launches overlap.
  python tools/kernel_timeline.py gpurun_out/prof_x/bench_results.db [count=40] [skip_at_end=0]"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
off = int(sys.argv[3]) if len(sys.argv) > 3 else 0      # skip that many kernels at the end (bench.py ends with a one-stream calibration pass of 20 cycles)
rows = c.execute("select name, start, start + duration, queue_id from kernels order by start").fetchall() if True else []
rows = rows[-(n + off):len(rows) - off]
t0 = rows[0][1]
for name, s, e, q in rows:
    print("%-28s q%-3s %9.1f -> %9.1f us  (%.1f)" % (name[:28], q, (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3))
