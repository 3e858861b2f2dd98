"""BASELINE.json configs[4]: mixed gait schedule (stance / trot / flying trot / static walk), N = 200, batch 1024, one MPC + WBC cycle in
fp64 and with the MPC kernels in fp32 (qmgpu_create_ex, QMGPU_F32).  Prints one JSON line: the ||.||_inf-relative deviation of X, U
and the WBC torques between the two (max / median / 99th percentile over the batch), the filter line-search decisions that differ,
and the per-kernel times of both paths.  Usage (GPU box):  python tools/fp32_sweep.py [--batch 1024] [--nodes 200]"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def run(batch, nodes, seed=3):
    import torch
    import gpu_harness as G
    import support as S
    from test_gpu_configs import _mixed_schedule
    from qm_door_amd import api
    itf = api.QMInterface()
    orc = S.Oracle(itf.problem)
    dt = itf.problem.settings.dt
    x_nom = itf.initial_state
    x0 = S.perturbed_states(x_nom, batch, seed=seed)
    tgt = S.nominal_target(orc, x_nom)
    tt = np.zeros((batch, 1)); ts = np.tile(tgt, (batch, 1, 1)).copy()
    nev, ev, md = _mixed_schedule(nodes * dt + 0.2)
    rbd = np.zeros((batch, 55)); rbd[:, 0:3] = x0[:, 9:12]; rbd[:, 3:6] = x0[:, 6:9]; rbd[:, 6:24] = x0[:, 12:30]
    out = {}
    for dtype in ("f64", "f32"):
        sol = G.make_solver(itf, batch, nodes, dtype=dtype)
        sol.enable_timing(True)
        ms, stats_reps = [], []
        for rep in range(3):
            mb = G.MpcBatch(x0, tt, ts, np.full(batch, nev, dtype=np.int32), np.tile(ev, (batch, 1)), np.tile(md, (batch, 1)), nodes)
            wb = G.WbcBatch(rbd, np.full(batch, 0.002), np.full(batch, 20.0), np.zeros((batch, 30)))
            sol.cycle(mb.args, G.dev(np.full(batch, 0.4 * dt), torch.float64), wb.args)
            torch.cuda.synchronize()
            ms.append(sol.last_kernel_ms())
            stats_reps.append(mb.results()["stats"])
        out[dtype] = dict(mpc=mb.results(), wbc=wb.results(), ms=ms[-1], repeatable=all(np.array_equal(stats_reps[0], r) for r in stats_reps[1:]))
        sol.close()
    return out


def rel_inf(a, b):
    """per-instance ||a - b||_inf / max(1, ||a||_inf)"""
    a = a.reshape(a.shape[0], -1); b = b.reshape(b.shape[0], -1)
    return np.abs(a - b).max(axis=1) / np.maximum(1.0, np.abs(a).max(axis=1))


def report(out):
    a, b = out["f64"], out["f32"]
    same_alpha = a["mpc"]["stats"][:, 4] == b["mpc"]["stats"][:, 4]
    rep = {"finite_f32": bool(np.isfinite(b["mpc"]["X"]).all() and np.isfinite(b["mpc"]["U"]).all() and np.isfinite(b["wbc"]["out"]).all()),
           "modes_bit_exact": bool(np.array_equal(a["mpc"]["mode"], b["mpc"]["mode"])),
           "statistics_repeat_bit_for_bit": bool(a["repeatable"] and b["repeatable"]),   # three identical calls per build
           "riccati_status_f32_all_zero": bool((b["mpc"]["stats"][:, 7] == 0).all()),
           "line_search_alpha_differs": int((~same_alpha).sum()), "batch": int(a["mpc"]["X"].shape[0]), "nodes": int(a["mpc"]["X"].shape[1] - 1),
           # (verdict r05, missing 5) what "tau" of the fp32 leg is: there is no fp32 WBC
           "wbc_precision": "fp64 in BOTH legs: tau of the f32 leg is the fp64 WBC fed with the fp32 plan -- HoQp's 1e-12 regulariser (HoQp.cpp:66) and its 1e-9 tolerances are below fp32 resolution, an fp32 WBC is not built"}
    sa, sb = a["mpc"]["stats"], b["mpc"]["stats"]
    rep["step_metrics_rel"] = {n: float((np.abs(sa[:, c] - sb[:, c]) / np.maximum(1e-12, np.abs(sa[:, c])))[same_alpha].max()) for c, n in ((0, "merit0"), (1, "violation0"), (2, "merit1"), (3, "violation1"))}
    for key, x, y in (("X", a["mpc"]["X"], b["mpc"]["X"]), ("U", a["mpc"]["U"], b["mpc"]["U"]), ("tau", a["wbc"]["out"][:, 36:], b["wbc"]["out"][:, 36:])):
        d = rel_inf(x, y)
        ds = rel_inf(x[same_alpha], y[same_alpha]) if same_alpha.any() else d
        rep[key] = {"max": float(d.max()), "p99": float(np.percentile(d, 99)), "median": float(np.median(d)), "max_same_step_length": float(ds.max())}
    names = ["ad_node", "lq_node", "riccati", "linesearch", "wbc", "whole"]
    rep["kernel_ms_f64"] = dict(zip(names, a["ms"])); rep["kernel_ms_f32"] = dict(zip(names, b["ms"]))
    return rep


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1024); ap.add_argument("--nodes", type=int, default=200)
    args = ap.parse_args()
    print(json.dumps(report(run(args.batch, args.nodes))))
