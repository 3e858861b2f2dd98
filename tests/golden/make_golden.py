#!/usr/bin/env python3
"""Generates tests/golden/oracle_vectors.npz.

These are outputs of THIS REPOSITORY'S OWN fp64 CPU oracle (oracle/), not of the reference: danisotelo/qm_door ships no
golden vectors and cannot be built or imported here -- parity is unpinned.  The vectors pin the oracle
against silent regressions and give the GPU tests a fixture that does not need the oracle at all.
Inputs are seeded; re-run with `python tests/golden/make_golden.py` after an intentional change of the oracle.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import support as S  # noqa: E402
from qm_door_amd import abi  # noqa: E402


def main():
    lib = abi.load_library(S.build_emu())        # host loaders only (qmgpu_load_problem)
    P = S.load_problem(lib)
    orc = S.Oracle(P)
    x_nom = np.array(P.settings.initial_state[:])
    rng = np.random.default_rng(2024)
    out = {}
    # flow map + Jacobians
    x = x_nom + rng.uniform(-1, 1, 30) * 0.1
    u = rng.uniform(-1, 1, 30) * np.r_[np.full(12, 20.0), np.full(18, 0.5)]; u[[2, 5, 8, 11]] += 68.0
    f, A, B = orc.flow_map_lin(x, u)
    out.update(flow_x=x, flow_u=u, flow_f=f, flow_A=A, flow_B=B)
    # one SQP iteration, N = 8, trot with a stance prefix
    N = 8
    x0 = S.perturbed_states(x_nom, 2, seed=77)
    tgt = S.nominal_target(orc, x_nom)
    nev, ev, md = S.trot_schedule(1.0, phase0=0.04)
    tt, ts = np.zeros(1), tgt[None, :].copy()
    sols = [orc.mpc_solve(N, 0.0, x0[i], tt, ts, nev, ev, md) for i in range(2)]
    out.update(mpc_x0=x0, mpc_target=tgt, mpc_nev=nev, mpc_ev=ev, mpc_md=md, mpc_X=np.array([s["X"] for s in sols]), mpc_U=np.array([s["U"] for s in sols]),
               mpc_mode=np.array([s["mode"] for s in sols]), mpc_stats=np.array([s["stats"] for s in sols]))
    # WBC
    cases = []
    for mode, t in ((15, 20.0), (9, 20.0), (6, 5.0), (0, 20.0)):
        flags = [(mode >> (3 - c)) & 1 for c in range(4)]
        uu = np.zeros(30)
        for c in range(4):
            if flags[c]:
                uu[3 * c + 2] = P.model.total_mass * 9.81 / sum(flags)
        uu[12:] = rng.uniform(-1, 1, 18) * 0.05
        xd = x_nom + rng.uniform(-1, 1, 30) * 0.02
        rbd = S.rbd_from_state(orc, x_nom + rng.uniform(-1, 1, 30) * 0.01, rng.uniform(-1, 1, 24) * 0.05)
        il = uu + rng.uniform(-1, 1, 30) * 0.001
        st, o, _ = orc.wbc_update(xd, uu, rbd, mode, 0.002, t, il)
        assert st == 0
        cases.append((xd, uu, rbd, mode, t, il, o))
    out.update(wbc_xd=np.array([c[0] for c in cases]), wbc_u=np.array([c[1] for c in cases]), wbc_rbd=np.array([c[2] for c in cases]),
               wbc_mode=np.array([c[3] for c in cases], dtype=np.int32), wbc_time=np.array([c[4] for c in cases]), wbc_il=np.array([c[5] for c in cases]),
               wbc_out=np.array([c[6] for c in cases]))
    np.savez(os.path.join(HERE, "oracle_vectors.npz"), **out)
    print("wrote", os.path.join(HERE, "oracle_vectors.npz"))


if __name__ == "__main__":
    main()
