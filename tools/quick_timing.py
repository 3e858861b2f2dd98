#!/usr/bin/env python3
"""Per-kernel HIP-event timing of the bench workload (developer loop; run through gpurun)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench, gpu_harness as G
from qm_door_amd import abi, api
if os.environ.get("QM_LIB"):      # A/B of two builds on the same box: QM_LIB=path/to/other/libqmgpu.so
    abi.LIB_PATH = os.environ["QM_LIB"]
itf = api.QMInterface(); B, N = 256, 100
sc = bench.build_scenario(itf, B, 0)
sol = G.make_solver(itf, B, N)
mb = G.MpcBatch(sc["x0"], sc["tt"], sc["ts"], np.full(B, sc["nev"], dtype=np.int32), np.tile(sc["ev"], (B, 1)), np.tile(sc["md"], (B, 1)), N)
wb = G.WbcBatch(sc["rbd"], np.full(B, 0.002), np.full(B, 20.0), np.zeros((B, 30)))
te = G.dev(np.zeros(B), torch.float64)
for _ in range(2): sol.cycle(mb.args, te, wb.args)
sol.enable_timing(True)
for _ in range(20): sol.cycle(mb.args, te, wb.args)
torch.cuda.synchronize()
ms = sol.kernel_ms_mean(20)
print("ms  ad %.3f  lq %.3f  riccati %.3f  linesearch %.3f  wbc %.3f  total %.3f  -> %.0f cycles/s" % (*ms, B / ms[5] * 1e3))
r = mb.results(); w = wb.results()
print("checksum X %.12e U %.12e tau %.12e status %s" % (np.abs(r["X"]).sum(), np.abs(r["U"]).sum(), np.abs(w["out"][:, 36:]).sum(), np.unique(w["status"])))
st = r["stats"]
print("line search: alpha", np.unique(st[:, 4], return_counts=True), "step type", np.unique(st[:, 5], return_counts=True))
