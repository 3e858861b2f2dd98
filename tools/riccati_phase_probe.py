"""Phase clocks of riccati_kernel (profiling build, -DQM_RICCATI_TIMING): where the cycles of a backward stage go, per wavefront.

  python tools/riccati_phase_probe.py --build      # here (hipcc): qm_door_amd/build/variants/ticks/libqmgpu_ticks.so
  python tools/riccati_phase_probe.py              # on the GPU box: runs the bench scenario, prints the s_memtime sums of workgroup 0
The product library carries no clocks (QM_TICK compiles to nothing)."""
import ctypes as C, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from qm_door_amd import abi, build
LIB = os.path.join(ROOT, "qm_door_amd", "build", "variants", "ticks", "libqmgpu_ticks.so")   # = tools/wbc_variants.py lib_path("ticks")
SLOTS = ["issue", "P1", "bar1", "P2", "bar2", "P3|P6a+gains", "commit", "bar3", "P6b", "bar4", "symm", "bar5", "tail", "fwd-tail", "F:top", "F:K.dx | issue", "F:sync | work", "F:B.du | commit", "F:barrier"]

if "--build" in sys.argv:
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    print(build.build_library(force=True, extra_flags=("-DQM_RICCATI_TIMING",) + tuple(f for f in sys.argv if f.startswith("-D")), out=LIB, obj_dir=os.path.dirname(LIB)))
    sys.exit(0)

abi.LIB_PATH = LIB
import torch, bench, gpu_harness as G
from qm_door_amd import api
B, N = 256, 100
itf = api.QMInterface()
sc = bench.build_scenario(itf, B, seed=0)
sol = G.make_solver(itf, B, N)
mb = G.MpcBatch(sc["x0"], sc["tt"], sc["ts"], np.full(B, sc["nev"], dtype=np.int32), np.tile(sc["ev"], (B, 1)), np.tile(sc["md"], (B, 1)), N)
for _ in range(3): sol.mpc(mb.args)
lib = sol.lib
lib.qmgpu_debug_riccati_ticks.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
buf = (C.c_ulonglong * 2048)()
assert lib.qmgpu_debug_riccati_ticks(buf, 1) == 0
wb = G.WbcBatch(sc["rbd"], np.full(B, 0.002), np.full(B, 20.0), np.zeros((B, 30)))
t_eval = G.dev(np.zeros(B), torch.float64)
sol.enable_timing(True)
R = 10
for _ in range(R): sol.cycle(mb.args, t_eval, wb.args)
torch.cuda.synchronize()
ms = sol.kernel_ms_mean(R)
assert lib.qmgpu_debug_riccati_ticks(buf, 0) == 0
raw = np.array(buf[:], dtype=np.float64) / R
t = raw[:128].reshape(4, 32)
tot = t[0, :24].sum()
print("riccati kernel ms (with clocks):", round(ms[2], 4), " ticks per launch, wavefront 0:", int(tot), " => ticks per ms:", round(tot / ms[2]))
print("per backward stage (ticks / %d stages), by wavefront:" % N)
for i, n in enumerate(SLOTS):
    per = t[:, i] / (1 if i in (12, 13) else N)
    print("  %-14s" % n, "  ".join("%8.0f" % v for v in per))
print("factorisation (wavefront 0): load columns %d, elimination loop %d, store L / W %d ticks per stage" % (raw[19] / N, raw[20] / N, raw[21] / N))
print(json.dumps({"kernel_ms": ms[2], "slots": SLOTS, "ticks": t[:, :len(SLOTS)].tolist()}))

LQ = ["inputs", "AD rows", "state cost", "flow-map Jacobian", "input cost", "QR projection", "Pall", "A~ B~ joint rows / transposes", "products (1)", "products (2)(3)", "padding / end"]
lq = raw[128:128 + len(LQ)]
print("lq_node_kernel, node 7 of instance 0 (two wavefronts share the SIMD): ticks per section, total", int(lq.sum()), " kernel ms", round(ms[1], 4))
for n, v in zip(LQ, lq): print("  %-34s %8.0f  %4.1f %%" % (n, v, 100 * v / lq.sum()))

IPM = ["setup (row norms, scale)", "interior point (all of it but its factorisations)", "factorise: transposition through LDS (rows of L out, rows of L^T back)", "T = L^-1 DZ' (forward substitutions, one per lane)", "S = T_P'T_P + small Cholesky", "pass: residuals D z, AZ'(AZ z + rhat), D^T t", "pass: forward, small solve, backward, D p", "decisions / reductions / bookkeeping (what no other slot holds)", "final checks (rounding bound, bounds)", "  factorise: weights + K tiles (fork-join)", "  factorise: columns of K into registers", "  factorise: elimination (in registers)", "  ipm: residuals, reductions, hand-over tests", "  ipm pass: t, D^T t", "  ipm pass: forward + backward", "  ipm pass: D dz", "  ipm pass: step lengths, update (+ loop exit)"]
for base, name in ((160, "NP = 36"), (256, "NP = 20"), (288, "NP = 8")):
    v = raw[base:base + 17]
    if v.sum() > 0:
        print("wbc level QP (all calls of the size), %s, instance 0: total %d ticks of %.0f (kernel %.4f ms)" % (name, v.sum(), ms[4] * tot / ms[2] if ms[2] else 0, ms[4]))
        for n_, x in zip(IPM, v): print("  %-40s %9.0f  %4.1f %%" % (n_, x, 100 * x / v.sum()))

WBC = ["S1-S2 inputs, coordinates", "S3 measured pass", "S4 nle, M, Jacobians", "S5 desired pass", "task 0 inequality rows + level-loop re-entry (null space of the previous level ends here)", "assemble level task", "reduced data", "interior point", "x update", "(after the last null space)", "torques", "  reduced data: A Z", "  reduced data: zero + D Z", "  reduced data: A x - b, margins", "  reduced data: zero + (A Z)^T A Z", "  null space: full-pivot LU of A Z", "  null space: kernel vectors (back substitution)", "  null space: Z N, copy", "    LU: max reduction", "    LU: pivot lane, permutations", "    LU: swap, pivot broadcast", "    LU: elimination"]   # (g and the vanishing-row test are what is left in "reduced data" above)
v = raw[192:192 + len(WBC)]
print("wbc_kernel, instance 0: total %d ticks (kernel %.4f ms)" % (v.sum(), ms[4]))
for n_, x in zip(WBC, v): print("  %-60s %9.0f  %4.1f %%" % (n_[:60], x, 100 * x / v.sum()))

NS = ["entry: row maxima", "step: max reduction, pivot lane", "step: swap, pivot broadcast", "step: elimination", "step: candidates of the chunks", "rank, free columns, zeroing", "kernel vectors (back substitution)"]
v = raw[352:352 + len(NS)]
print("wbcNullSpace (all calls of instance 0): total %d ticks" % v.sum())
for n_, x in zip(NS, v): print("  %-50s %8.0f  %4.1f %%" % (n_, x, 100 * x / max(v.sum(), 1)))

LS = ["two model sweeps + constraints + EE cost", "defect", "tracking cost (two 30 x 30 forms)", "barriers"]
v = raw[224:224 + len(LS)]
print("linesearch nodePerformance, node 5 of instance 0 (all calls of a launch): total %d ticks (kernel %.4f ms)" % (v.sum(), ms[3]))
for n_, x in zip(LS, v): print("  %-50s %9.0f  %4.1f %%" % (n_, x, 100 * x / max(v.sum(), 1)))

v = raw[240:247]
print("linesearch_kernel, thread 0 of instance 0: sections in ticks (total %d)" % v.sum())
for n_, x in zip(["weights / model into LDS, baseline sums", "trial iterates (axpy) + barrier", "node evaluations (nodePerformance)", "partial sums + barrier", "filter decision (thread 0) + barrier", "new iterate out", "step norms, convergence, statistics"], v): print("  %-50s %8.0f  %4.1f %%" % (n_, x, 100 * x / max(v.sum(), 1)))
# per-instance totals of the last wbc launch (not sums): ticks, interior-point iterations of the three levels
pi = np.array(buf[512:512 + 4 * B], dtype=np.float64).reshape(B, 4)
tk, its = pi[:, 0], pi[:, 1:]
print("wbc_kernel per instance (last launch): ticks mean %.0f  median %.0f  p90 %.0f  max %.0f  => max/mean %.3f; iterations per level mean %s max %s; total mean %.1f max %d"
      % (tk.mean(), np.median(tk), np.percentile(tk, 90), tk.max(), tk.max() / tk.mean(), np.round(its.mean(0), 2).tolist(), its.max(0).astype(int).tolist(), its.sum(1).mean(), int(its.sum(1).max())))
worst = np.argsort(-tk)[:6]
print("  slowest instances:", [(int(i), int(tk[i]), its[i].astype(int).tolist()) for i in worst])

AD = ["inputs (x, u, schedule, references)", "first sweep (21 tangents x 3 nodes)", "constraint rows + J1 rows into LDS", "second sweep", "J2 operand + chain rule (matrix cores)",
      "J1 += J2 in place", "phi rows out", "foot callbacks of both sweeps (parking in LDS)", "end-effector callback of both sweeps (pose error rows out)"]
v = raw[320:320 + len(AD)]
print("ad_node_kernel, workgroup 1000 (three nodes): total %d ticks (kernel %.4f ms)" % (v.sum(), ms[0]))
for n_, x in zip(AD, v): print("  %-50s %9.0f  %4.1f %%" % (n_, x, 100 * x / max(v.sum(), 1)))
# occupancy picture of the last ad_node launch: when every workgroup started / ended (100 MHz wall clock)
nwg = (B * (N + 1) + 2) // 3
cl = (C.c_ulonglong * (2 * nwg))()
lib.qmgpu_debug_ad_wg_clocks.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
assert lib.qmgpu_debug_ad_wg_clocks(cl, 2 * nwg) == 0
clk = np.array(cl[:], dtype=np.float64).reshape(nwg, 2)
t0 = clk[:, 0].min()
st_us, en_us = (clk[:, 0] - t0) / 100.0, (clk[:, 1] - t0) / 100.0
dur = en_us - st_us
print("ad_node_kernel launch: %d workgroups; workgroup duration mean %.1f us (min %.1f, max %.1f); last start %.1f us, last end %.1f us" % (nwg, dur.mean(), dur.min(), dur.max(), st_us.max(), en_us.max()))
edges = np.arange(0.0, en_us.max() + 20.0, 20.0)
run = [(int(((st_us < e + 10) & (en_us > e + 10)).sum())) for e in edges]
print("  workgroups in flight every 20 us:", run)
full = max(run)
tail = sum(1 for r in run if r < 0.5 * full) * 20.0
print("  peak %d in flight (1024 SIMDs x 1 wavefront); time below half of the peak: %.0f us of %.0f us" % (full, tail, en_us.max()))

# ---- the same per-instance picture of wbc_kernel on robots IN MOTION (tests/support.py: moving_inputs; main branch t >= 10 s, policy evaluated between nodes): the launch lasts as long
#      as its slowest instance, and the steady state of bench.py (config.steady_state) pays for the longer tail of interior-point iterations
import support as S
orc = S.Oracle(itf.problem)
mv = S.moving_inputs(orc, sc["x0"], itf.problem.settings.dt, seed=21)
mb2 = G.MpcBatch(mv["x0"], sc["tt"], sc["ts"], np.full(B, sc["nev"], dtype=np.int32), np.tile(sc["ev"], (B, 1)), np.tile(sc["md"], (B, 1)), N)
wb2 = G.WbcBatch(mv["rbd"], np.full(B, 0.002), np.full(B, 20.0), mv["input_last"])
te2 = G.dev(mv["t_eval"], torch.float64)
for _ in range(3): sol.cycle(mb2.args, te2, wb2.args)
torch.cuda.synchronize()
ms2 = sol.kernel_ms_mean(3)
assert lib.qmgpu_debug_riccati_ticks(buf, 0) == 0
pi = np.array(buf[512:512 + 4 * B], dtype=np.float64).reshape(B, 4)
tk, its = pi[:, 0], pi[:, 1:]
print("wbc_kernel per instance, robots in motion (kernel %.4f ms): ticks mean %.0f  median %.0f  p90 %.0f  max %.0f  => max/mean %.3f; iterations per level mean %s max %s; total mean %.1f max %d"
      % (ms2[4], tk.mean(), np.median(tk), np.percentile(tk, 90), tk.max(), tk.max() / tk.mean(), np.round(its.mean(0), 2).tolist(), its.max(0).astype(int).tolist(), its.sum(1).mean(), int(its.sum(1).max())))
worst = np.argsort(-tk)[:8]
print("  slowest instances:", [(int(i), int(tk[i]), its[i].astype(int).tolist()) for i in worst])
print("  histogram of total iterations:", np.bincount(its.sum(1).astype(int)).tolist())
