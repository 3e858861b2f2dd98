#!/usr/bin/env python3
"""Phase clocks of ONE WBC tick (profiling build, -DQM_RICCATI_TIMING = the `ticks` variant of tools/wbc_variants.py): the tick runs as a batch of one, so it is workgroup 0,
whose clocks the kernels record.  Made for the slow ticks tools/wbc_tail_probe.py dumps (gpurun_out/wbc_tail_cold.npz): where do the 40-46 working-set changes of a 36-variable
level spend their time?   GPU box: python tools/wbc_tick_probe.py gpurun_out/wbc_tail_cold.npz [index ...]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from qm_door_amd import abi  # noqa: E402
abi.LIB_PATH = os.path.join(ROOT, "qm_door_amd", "build", "variants", "ticks", "libqmgpu_ticks.so")
import torch  # noqa: E402
import gpu_harness as G  # noqa: E402
from qm_door_amd import api  # noqa: E402

QP = ["setup (row norms, scale)", "interior point (all of it but its factorisations)", "factorise: transposition through LDS", "T = L^-1 DZ' of the pinned rows", "S = T_P'T_P + small Cholesky",
      "pass: residuals D z, AZ'(AZ z + rhat), D^T t", "pass: forward, small solve, backward, D p", "decisions / reductions / bookkeeping", "final checks", "  factorise: weights + K tiles (fork-join)",
      "  factorise: columns of K into registers", "  factorise: elimination (in registers)", "  ipm: residuals, reductions, hand-over tests", "  ipm pass: t, D^T t", "  ipm pass: forward + backward", "  ipm pass: D dz",
      "  ipm pass: step lengths, update"]
d = np.load(sys.argv[1])
idx = [int(a) for a in sys.argv[2:]] or list(range(min(3, len(d["mode"]))))
itf = api.QMInterface()
sol = G.make_solver(itf, 1, 4)
lib = sol.lib
lib.qmgpu_debug_riccati_ticks.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
buf = (C.c_ulonglong * 2048)()
for i in idx:
    period = float(d["period"][i]) if "period" in d else 0.001
    wb = G.WbcBatch(d["rbd"][i][None], np.array([period]), np.array([float(d["time"][i])]), d["il"][i][None].copy(), d["xd"][i][None], d["ud"][i][None], np.array([int(d["mode"][i])], dtype=np.int32),
                    int(d["variant"][i]) if "variant" in d else 0, carry=True)
    sol.wbc(wb.args); torch.cuda.synchronize()
    assert lib.qmgpu_debug_riccati_ticks(buf, 1) == 0
    sol.enable_timing(True)
    wb.il.copy_(G.dev(d["il"][i][None])); wb.ws.zero_()
    sol.wbc(wb.args); torch.cuda.synchronize()
    ms = sol.last_kernel_ms()[4]
    assert lib.qmgpu_debug_riccati_ticks(buf, 0) == 0
    raw = np.array(buf[:], dtype=np.float64)
    w = wb.results()["working_set"][0]
    cb = np.ascontiguousarray(w[13:15]).view(np.uint8)
    print(f"tick {i}: kernel {ms:.3f} ms, status {int(wb.results()['status'][0])}, passes per solve {[int(v) for v in cb[:12]]}, whole kernel {raw[192:192 + 11].sum():.0f} ticks")
    WBC = ["S1-S2 inputs, coordinates", "S3 measured pass", "S4 nle, M, Jacobians", "S5 desired pass (join)", "task 0 inequality rows + level-loop re-entry", "assemble level task", "reduced data", "level QP", "x update",
           "(after the last null space)", "torques", "  reduced data: A Z", "  reduced data: zero + D Z", "  reduced data: A x - b, margins", "  reduced data: zero + (A Z)^T A Z", "  null space: full-pivot LU of A Z",
           "  null space: kernel vectors (back substitution)", "  null space: Z N, copy"]
    v = raw[192:192 + len(WBC)]
    print(f"  kernel sections ({v[:11].sum():.0f} ticks):")
    for n_, x in zip(WBC, v):
        print("    %-62s %9.0f  %5.1f %%" % (n_, x, 100 * x / v[:11].sum()))
    NS = ["entry: row maxima", "step: max reduction, pivot lane", "step: swap, pivot broadcast", "step: elimination", "step: candidates of the chunks", "rank, free columns, zeroing", "kernel vectors (back substitution)"]
    v = raw[352:352 + len(NS)]
    print(f"  wbcNullSpace, all calls ({v.sum():.0f} ticks):")
    for n_, x in zip(NS, v):
        print("    %-62s %9.0f  %5.1f %%" % (n_, x, 100 * x / max(v.sum(), 1)))
    for base, name in ((160, "NP = 36"), (256, "NP = 20"), (288, "NP = 8")):
        v = raw[base:base + 17]
        if v[:9].sum() > 0:
            print(f"  level QPs of {name}: {v[:9].sum():.0f} ticks")
            for n_, x in zip(QP, v):
                print("    %-62s %9.0f  %5.1f %%" % (n_, x, 100 * x / v[:9].sum()))
