"""Regenerates tests/golden/wbc_degenerate_stance_tick.npz: the WBC inputs of instance 72, MPC cycle 0, tick 1 (t = 10.501 s, full stance) of the static-walk closed
loop of tests/test_closed_loop.py, as the ORACLE loop sees them (CPU only, this repository's own oracle; no reference code involved).
    python tests/golden/make_degenerate_tick.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import support as S, closed_loop as CL
from qm_door_amd import api
itf = api.QMInterface()
orc = S.Oracle(itf.problem, fast=True)
sc = CL.Scenario(itf, 128, cycles=60, t_start=10.5, gait_start=0.05, gait="static_walk", seed=37)
be = CL.OracleBackend(orc, sc, 0)
rbd = sc.first_measurement()
t0 = sc.t_start
N, grid = sc.grid(t0)
be.observe(rbd, t0)
plan = be.mpc(t0, N, grid)
be.tick(t0, rbd, t0)
il = be.il.copy()
t = t0 + CL.WBC_PERIOD
rbd1 = CL.measurement(sc, plan, t)
xd, ud, md = orc.policy_eval_batch(plan["T"], plan["X"], plan["U"], plan["mode"], t)
i = 72
np.savez(os.path.join(ROOT, "tests", "golden", "wbc_degenerate_stance_tick.npz"), xd=xd[i], ud=ud[i], rbd=rbd1[i], mode=md[i], t=t, il=il[i])
print("written; mode", md[i], "t", t)
