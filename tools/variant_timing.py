"""Per-kernel times of the bench workload (256 instances, N = 100) for build variants of the library (tools/wbc_variants.py) next to the product build.
    python tools/variant_timing.py s_maxilp s_iterilp      # on the GPU box; the variants must have been built (wbc_variants.py --build ...)"""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench, gpu_harness as G, wbc_variants as V
from qm_door_amd import abi, api
B, N = 256, 100
ref = None
for name in ["product"] + sys.argv[1:]:
    path = abi.LIB_PATH if name == "product" else V.lib_path(name)
    if not os.path.exists(path):
        print(name, "not built"); continue
    itf = api.QMInterface(lib=abi.load_library(path))
    sc = bench.build_scenario(itf, B, seed=0)
    sol = G.make_solver(itf, B, N)
    mb = G.MpcBatch(sc["x0"], sc["tt"], sc["ts"], np.full(B, sc["nev"], dtype=np.int32), np.tile(sc["ev"], (B, 1)), np.tile(sc["md"], (B, 1)), N)
    wb = G.WbcBatch(sc["rbd"], np.full(B, 0.002), np.full(B, 20.0), np.zeros((B, 30)))
    t_eval = G.dev(np.zeros(B), torch.float64)
    for _ in range(5): sol.cycle(mb.args, t_eval, wb.args)
    sol.enable_timing(True)
    for _ in range(40): sol.cycle(mb.args, t_eval, wb.args)
    torch.cuda.synchronize()
    ms = sol.kernel_ms_mean(40)
    r, w = mb.results(), wb.results()
    if ref is None: ref = (r["X"].copy(), w["out"].copy())
    dx = np.abs(r["X"] - ref[0]).max(); dt = np.abs(w["out"][:, 36:] - ref[1][:, 36:]).max()
    print("%-10s" % name, dict(zip(["ad", "lq", "riccati", "ls", "wbc", "whole"], [round(m, 4) for m in ms])), "cycles/s", round(B / ms[5] * 1e3), " |X - product| %.1e |tau - product| %.1e" % (dx, dt), flush=True)
    sol.close()
