#!/usr/bin/env python3
"""Which instances make a WBC launch long?  Runs the record pass of bench.py's steady-state leg (receding horizon, 256 robots in motion, one WBC tick per MPC cycle) on the
GPU, reads the per-solve pass counts of every instance from the working-set records (qmgpu_wbc_args::working_set, words 13 / 14) and writes the WBC inputs of every
instance above --above passes to gpurun_out/wbc_tail.npz, so that the tick can be replayed on the CPU restatement with its per-iteration trace (tests/support.py:
Oracle.set_experiment(trace=True)).  Usage (GPU box): python tools/wbc_tail_probe.py [--steps 40] [--above 20] [--state cold|carry]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--above", type=int, default=20)
    ap.add_argument("--state", choices=["cold", "carry"], default="cold")
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--period", type=float, default=0.01, help="WbcBase::update's period (time since the previous tick: 10 ms in this loop; 1 ms reproduces round 5's bench)")
    args = ap.parse_args()
    import torch
    import bench
    from qm_door_amd import abi, api, harness as G
    itf = api.QMInterface()
    B, N = bench.BATCH_PER_GPU, bench.HORIZON_N
    sc = bench.build_scenario(itf, B, seed=0)
    t_end = args.steps * 0.01 + N * itf.problem.settings.dt + 0.5
    sc["nev"], sc["ev"], sc["md"] = api.GaitSchedule(lib=itf.lib).mode_schedule("trot", 0.0, 0.0, t_end)
    f64 = torch.float64
    dt = itf.problem.settings.dt
    dist_ = G.Disturbance(B, np.random.default_rng(5))
    sol = G.make_solver(itf, B, N)
    z = lambda *shape, dtype=f64: torch.zeros(shape, dtype=dtype, device=G.DEVICE)  # noqa: E731
    sets = [dict(T=z(B, N + 1), X=z(B, N + 1, 30), U=z(B, N, 30), M=z(B, N + 1, dtype=torch.int32), S=z(B, abi.NSTATS)) for _ in range(2)]
    wx, wu = z(B, N + 1, 30), z(B, N, 30)
    tt, ts = G.dev(sc["tt"], f64), G.dev(sc["ts"], f64)
    sn, se, sm = G.dev(np.full(B, sc["nev"], dtype=np.int32), torch.int32), G.dev(np.tile(sc["ev"], (B, 1)), f64), G.dev(np.tile(sc["md"], (B, 1)), torch.int32)
    kind, cmd, lastee, ftt, fts = z(B, dtype=torch.int32), z(B, 7), z(B, 7), z(B, 2), z(B, 2, 37)
    out, status, period = z(B, 54), z(B, dtype=torch.int32), G.dev(np.full(B, args.period), f64)
    ws = z(B, abi.WBC_STATE_WORDS, dtype=torch.int64)
    xd, ud, pm = z(B, 30), z(B, 30), z(B, dtype=torch.int32)
    il = z(B, 30)
    v0 = np.c_[np.random.default_rng(6).uniform(-0.1, 0.1, (B, 6)), np.random.default_rng(7).uniform(-0.2, 0.2, (B, 18))]
    rbd = G.pack_rbd(sc["x0"][:, 6:30] + dist_.dq(0.0), v0 + dist_.dv(0.0))
    plan, dump = None, []
    for k in range(args.steps):
        t0 = k * G.MPC_PERIOD
        if k > 0:
            rbd = G.measurement(dist_, plan, t0)
        cur, prev = sets[k & 1], sets[(k & 1) ^ 1]
        r = dict(rbd=G.dev(rbd, f64), x0=z(B, 30), grid=G.dev(np.tile(t0 + dt * np.arange(N + 1), (B, 1)), f64), t0=G.dev(np.full(B, t0), f64),
                 t_eval=G.dev(np.full(B, t0 + 0.3 * G.WBC_PERIOD), f64), time=G.dev(np.full(B, 20.0 + t0), f64))
        sol.frontend(sol.frontend_args(B, r["rbd"], r["t0"], kind, cmd, lastee, r["x0"], ftt, fts))
        if k > 0:
            sol.warm_start(B, N, prev["T"], prev["X"], prev["U"], N, r["grid"], r["x0"], wx, wu)
        a = api.GpuSolver.mpc_args(B, N, r["x0"], tt, ts, sn, se, sm, cur["T"], cur["X"], cur["U"], cur["M"], cur["S"], t0=r["t0"], time_grid=r["grid"],
                                   warm_x=wx if k > 0 else None, warm_u=wu if k > 0 else None)
        sol.mpc(a)
        sol.policy_eval(B, N, cur["T"], cur["X"], cur["U"], cur["M"], r["t_eval"], xd, ud, pm)
        il_in, ws_in = il.clone(), ws.clone()
        if args.state == "cold":
            ws.zero_(); ws_in.zero_()
        sol.wbc(api.GpuSolver.wbc_args(B, r["rbd"], period, r["time"], il, out, status, xd, ud, pm, args.variant, working_set=ws))
        torch.cuda.synchronize()
        plan = dict(T=cur["T"].cpu().numpy(), X=cur["X"].cpu().numpy(), U=cur["U"].cpu().numpy())
        w = ws.cpu().numpy().view(np.uint64)
        cb = np.ascontiguousarray(w[:, 13:15]).view(np.uint8).reshape(B, 16)
        tot = (cb & 127).astype(np.int64).sum(axis=1)
        print(f"step {k}: passes mean {tot.mean():.2f} max {tot.max()} (instance {tot.argmax()}: per solve {[int(v) for v in cb[tot.argmax()][:12]]}) status!=0: {int((status != 0).sum())}", flush=True)
        for i in np.nonzero(tot > args.above)[0]:
            dump.append(dict(step=k, instance=int(i), passes=cb[i].copy(), xd=xd[i].cpu().numpy(), ud=ud[i].cpu().numpy(), rbd=rbd[i].copy(), mode=int(pm[i]), time=20.0 + t0, period=args.period,
                             il=il_in[i].cpu().numpy(), ws_in=ws_in[i].cpu().numpy().view(np.uint64), out=out[i].cpu().numpy(), ws_out=w[i].copy()))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    if dump:
        np.savez(os.path.join(ROOT, "gpurun_out", f"wbc_tail_{args.state}.npz"), **{k: np.array([d[k] for d in dump]) for k in dump[0]})
    print(len(dump), "ticks above", args.above, "passes written")


if __name__ == "__main__":
    main()
