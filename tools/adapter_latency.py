#!/usr/bin/env python3
"""Latency of the drop-in at the plugin's OWN operating point: ONE robot, ~67 shooting nodes + event nodes, warm start, 100 Hz MPC /
1 kHz WBC, measured where the reference measures (mpcTimer_ around advanceMpc, QMController.cpp:322-324; wbcTimer_ around wbc_->update, :146-148) -- the
wall clock of GpuSqpSolver::run / GpuWbc::update through the adapters (pinned staging, H2D, batch-1 launch chain, D2H, one stream synchronisation), for both
plugin classes, with the per-kernel split of the same loop, next to the CPU oracle's time for one instance with the reference's 3 node threads.
Run on the GPU box:  python tools/adapter_latency.py  ->  gpurun_out/adapter_latency.json  (copied to profiles/r04_adapter_latency.json)."""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "adapters"))


def main(runs=200, ticks=10):
    import build_driver as BD
    import support as S
    from qm_door_amd import abi, api
    exe = BD.build_driver()
    d = abi.DATA_DIR
    out = {"operating_point": f"one robot, timeHorizon 1.0 s, dt 0.015, stance 0.2 s then trot (0.35 s phases), warm start, {runs} MPC runs at 100 Hz with {ticks} WBC ticks at 1 kHz each",
           "what_is_timed": "wall clock of mpc_->run(t, x) and wbc_->update(...) through qm_door_amd/adapters (std::chrono::steady_clock around the call, as the reference's RepeatedTimer)"}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    for variant in (0, 1):
        for cold in (True, False):     # the adapter's default (every tick cold: what rounds 1-5 measured), then GpuWbc::carryWorkingSet(true)
            path = os.path.join(ROOT, "gpurun_out", f"adapter_latency_v{variant}{'' if cold else '_carry'}.json")
            env = {k: v for k, v in os.environ.items() if k != "QM_WBC_CARRY"} if cold else dict(os.environ, QM_WBC_CARRY="1")
            p = subprocess.run([exe, f"{d}/task.info", f"{d}/aliengo_z1.urdf", f"{d}/reference.info", "--latency", path, str(variant), str(runs), str(ticks)], capture_output=True, text=True, timeout=900, env=env)
            if p.returncode != 0:
                raise SystemExit(p.stderr)
            out[("qm/QMGpuController" if variant == 0 else "qm/QMGpuMpcController") + ("" if cold else " (WBC working sets carried)")] = json.load(open(path))
    # the CPU restatement at the same size: cold-start cycles of one instance, N = 68, three worker threads over the nodes (task.info:78) and one thread
    itf = api.QMInterface()
    orc = S.Oracle(itf.problem, fast=True)
    x_nom = itf.initial_state
    tgt = np.r_[x_nom, x_nom[6] + 0.6, x_nom[7], x_nom[8] + 0.036, 0.0, 0.0, 0.0, 1.0]
    nev, ev, md = S.trot_schedule(2.0, phase0=0.2)
    rbd = np.zeros((1, 55)); rbd[0, 0:3] = x_nom[9:12]; rbd[0, 3:6] = x_nom[6:9]; rbd[0, 6:24] = x_nom[12:30]
    n = 40
    x0s = np.tile(x_nom, (n, 1)); rbds = np.tile(rbd, (n, 1))
    t3 = orc.time_cycles_node_threads(n, 68, x0s, np.zeros(1), tgt[None], nev, ev, md, rbds, node_threads=3)
    orc.time_split()
    t1 = orc.time_cycles(n, 68, x0s, np.zeros(1), tgt[None], nev, ev, md, rbds)
    split = orc.time_split(); tot = sum(split.values()) or 1.0
    out["cpu_oracle_one_instance_N68"] = {"ms_per_cycle_3_node_threads": 1e3 * t3 / n, "ms_per_cycle_1_thread": 1e3 * t1 / n, "cycles": n,
                                          "wbc_ms_1_thread": 1e3 * (split["wbc_model"] + split["wbc_qp"]) / n, "mpc_ms_1_thread": 1e3 * (tot - split["wbc_model"] - split["wbc_qp"]) / n,
                                          "note": "own CPU restatement (g++ -O3), not OCS2; cold-start cycle = one SQP iteration + one WBC update"}
    path = os.path.join(ROOT, "gpurun_out", "adapter_latency.json")
    json.dump(out, open(path, "w"), indent=1)
    for k in ("qm/QMGpuController", "qm/QMGpuController (WBC working sets carried)", "qm/QMGpuMpcController", "qm/QMGpuMpcController (WBC working sets carried)"):
        w = out[k]["wall_clock"]
        print(k, "mpc_run avg %.3f max %.3f ms | wbc_update avg %.3f max %.3f ms | nodes %.1f" % (w["mpc_run"]["avg_ms"], w["mpc_run"]["max_ms"], w["wbc_update"]["avg_ms"], w["wbc_update"]["max_ms"], w["mean_nodes"]))
        print("   kernels:", out[k]["with_kernel_timing"]["kernel_ms_mean"])
    print("oracle:", out["cpu_oracle_one_instance_N68"])


if __name__ == "__main__":
    main()
