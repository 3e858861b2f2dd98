#!/usr/bin/env python3
"""Experiment (run through gpurun): the bench batch as S sub-batches on S HIP streams, each with its own solver object, against ONE
solver on one stream.  Question: do the tails of one sub-batch's launches (the last partial round of the node kernels, the slowest
instance of wbc_kernel) fill with the other sub-batch's kernels, or does CU-time stay conserved?  Usage: split_stream_probe.py [B] [steps]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench, gpu_harness as G
from qm_door_amd import api

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
N = 100
itf = api.QMInterface()
sc = bench.build_scenario(itf, B, 0)


def leg(S):
    Bs = B // S
    streams = [torch.cuda.Stream() for _ in range(S)]
    parts = []
    for s in range(S):
        sl = slice(s * Bs, (s + 1) * Bs)
        with torch.cuda.stream(streams[s]):
            sol = G.make_solver(itf, Bs, N)
            mb = G.MpcBatch(sc["x0"][sl], sc["tt"][sl], sc["ts"][sl], np.full(Bs, sc["nev"], dtype=np.int32), np.tile(sc["ev"], (Bs, 1)), np.tile(sc["md"], (Bs, 1)), N)
            wb = G.WbcBatch(sc["rbd"][sl], np.full(Bs, 0.002), np.full(Bs, 20.0), np.zeros((Bs, 30)))
            te = G.dev(np.zeros(Bs), torch.float64)
        parts.append((sol, mb, wb, te))
    torch.cuda.synchronize()

    def step():
        for s in range(S):
            sol, mb, wb, te = parts[s]
            with torch.cuda.stream(streams[s]):
                sol.cycle(mb.args, te, wb.args)
    for _ in range(5): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps): step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    tau = np.concatenate([p[2].results()["out"][:, 36:] for p in parts])
    return dt, tau


ref = None
for S in (1, 2, 4, 1):
    dt, tau = leg(S)
    if ref is None: ref = tau
    print("B %d  sub-batches %d  ms/step %.4f  -> %.0f cycles/s   torques bit-identical to one stream: %s" % (B, S, dt * 1e3, B / dt, np.array_equal(tau, ref)), flush=True)
