#!/usr/bin/env python3
"""PCIe-inclusive rate of the bench workload: host inputs -> device, qmgpu_cycle_batch, results -> host, every step (pinned buffers).
DESIGN.md quotes this next to the HBM-resident `value` of bench.py; it is never the headline number."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench, gpu_harness as G
from qm_door_amd import api, sharding
itf = api.QMInterface(); B, N = bench.BATCH_PER_GPU, bench.HORIZON_N
sc = bench.build_scenario(itf, B, 0)
sol = G.make_solver(itf, B, N)
mb = G.MpcBatch(sc["x0"], sc["tt"], sc["ts"], np.full(B, sc["nev"], dtype=np.int32), np.tile(sc["ev"], (B, 1)), np.tile(sc["md"], (B, 1)), N)
wb = G.WbcBatch(sc["rbd"], np.full(B, 0.002), np.full(B, 20.0), np.zeros((B, 30)))
te = G.dev(np.zeros(B), torch.float64)
h_x0 = torch.as_tensor(sc["x0"]).pin_memory(); h_rbd = torch.as_tensor(sc["rbd"]).pin_memory()
h_out = torch.empty((B, sharding.pack_len(N)), dtype=torch.float64).pin_memory()


def step(host):
    if host:
        mb.x0.copy_(h_x0, non_blocking=True); wb.rbd.copy_(h_rbd, non_blocking=True)
    sol.cycle(mb.args, te, wb.args)
    if host:
        h_out.copy_(sharding.pack(mb.oX, mb.oU, wb.out, mb.oM), non_blocking=True)


for host in (False, True):
    for _ in range(3):
        step(host)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        step(host)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print(("PCIe-inclusive" if host else "HBM-resident  "), "%.3f ms per step -> %.0f cycles/s" % (dt * 1e3, B / dt), "(%.1f MB out per step)" % (h_out.numel() * 8 / 1e6))
