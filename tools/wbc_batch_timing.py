"""WBC kernel time against the batch size (the launch is as slow as its slowest instance while the batch fits the 256 CUs once; beyond that the mean matters)."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench, gpu_harness as G
from qm_door_amd import api
itf = api.QMInterface()
for B in (256, 1024, 2048):
    sc = bench.build_scenario(itf, B, seed=0)
    sol = G.make_solver(itf, B, 100)
    mb = G.MpcBatch(sc["x0"], sc["tt"], sc["ts"], np.full(B, sc["nev"], dtype=np.int32), np.tile(sc["ev"], (B, 1)), np.tile(sc["md"], (B, 1)), 100)
    wb = G.WbcBatch(sc["rbd"], np.full(B, 0.002), np.full(B, 20.0), np.zeros((B, 30)))
    t_eval = G.dev(np.zeros(B), torch.float64)
    for _ in range(3): sol.cycle(mb.args, t_eval, wb.args)
    sol.enable_timing(True)
    for _ in range(10): sol.cycle(mb.args, t_eval, wb.args)
    torch.cuda.synchronize()
    ms = sol.kernel_ms_mean(10)
    print(B, dict(zip(["ad", "lq", "riccati", "ls", "wbc", "whole"], [round(m, 3) for m in ms])), "cycles/s", round(B / ms[5] * 1e3), flush=True)
    sol.close()
