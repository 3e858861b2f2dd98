"""Conditioning of the WBC cascade, measured on the CPU restatement ALONE (no GPU): the closed loop of tests/test_closed_loop.py (256 robots, 100 MPC cycles x 10 WBC
ticks, t = 9.5 .. 10.5 s) is run on the oracle, and on every tick of its second half the WBC is solved a second time with its inputs perturbed by 1e-13 relative -- the
size of the MPC-plan deviation between the GPU path and the oracle.  What comes out is the floor below which GPU / oracle torque parity cannot be asked for, tick by tick.

  python tools/oracle_sensitivity_closed_loop.py [variant=1] [cycles=100]   ->  profiles/<tag>_oracle_sensitivity_v<variant>.json  (tag: third argument, default r05)

HierarchicalWbc (variant 0): not one tick above 1e-7.  HierarchicalMpcWbc (variant 1, no arm task): 30 of 128,000 ticks above 1e-6, 10 above 1e-4, max 1.1e-2."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import closed_loop as CL  # noqa: E402
import support as S  # noqa: E402
from qm_door_amd import abi, api  # noqa: E402
from qm_door_amd.harness import MPC_PERIOD, WBC_PERIOD, measurement  # noqa: E402

variant = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cycles = int(sys.argv[2]) if len(sys.argv) > 2 else 100
tag = sys.argv[3] if len(sys.argv) > 3 else "r05"
B, first = 256, cycles // 2
itf = api.QMInterface(lib=abi.load_library())
sc = CL.Scenario(itf, B, cycles=cycles, gait_start=0.55)
o = S.Oracle(itf.problem, fast=True)
be = CL.OracleBackend(o, sc, variant)
rng = np.random.default_rng(5)
rbd = sc.first_measurement()
out, t00 = [], time.time()
for k in range(cycles):
    t0 = sc.t_start + k * MPC_PERIOD
    N, grid = sc.grid(t0)
    be.observe(rbd, t0)
    plan = be.mpc(t0, N, grid)
    for j in range(10):
        t = t0 + j * WBC_PERIOD
        if not (k == 0 and j == 0):
            rbd = measurement(sc, plan, t)
        xd, ud, md = o.policy_eval_batch(plan["T"], plan["X"], plan["U"], plan["mode"], t)
        il = be.il.copy()
        w = o.wbc_batch(xd, ud, rbd, md, WBC_PERIOD, t, il, variant)
        if k >= first:
            p = lambda a: a * (1 + 1e-13 * rng.standard_normal(a.shape))  # noqa: E731
            w2 = o.wbc_batch(p(xd), p(ud), p(rbd), md, WBC_PERIOD, t, p(il), variant)
            e = S.rel_inf(w["out"][:, 36:], w2["out"][:, 36:])
            for i in np.nonzero(e > 1e-7)[0]:
                out.append(dict(cycle=k, tick=j, instance=int(i), tau_move=float(e[i]), passes=w["iterations"][i].tolist()))
        be.il = w["input_last"]
    rbd = measurement(sc, plan, t0 + MPC_PERIOD)
    if k % 10 == 0:
        print(k, round(time.time() - t00), len(out), flush=True)
d = np.array([r["tau_move"] for r in out])
rep = dict(what="oracle vs oracle, WBC inputs perturbed by 1e-13 relative (standard normal), closed loop of tests/test_closed_loop.py", variant=variant, instances=B, cycles=cycles,
           ticks_probed=(cycles - first) * 10 * B, above_1e_7=len(d), above_1e_6=int((d > 1e-6).sum()), above_1e_4=int((d > 1e-4).sum()), max=float(d.max()) if len(d) else 0.0, ticks=out)
path = os.path.join(ROOT, "profiles", f"{tag}_oracle_sensitivity_v{variant}.json")
json.dump(rep, open(path, "w"), indent=0)
print({k: v for k, v in rep.items() if k != "ticks"}, "->", path)
