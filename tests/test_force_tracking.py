"""BASELINE.json configs[3]: force tracking (door opening), OWN FORMULATION -- the reference's force-tracking branch is not in the mounted
tree (README.md:15,161,180): compliant end-effector contact f_e = -K_e (p_ee - p_env(t)) in the centroidal dynamics, the soft constraint
1/2 mu_f |f_e - f_ref(t)|^2 in the OCP, the external force in the WBC's equations of motion / torque limits / torque recovery with the
contact-force task at level 2 (include/qmgpu.h).  Kernels vs the CPU oracle of the same formulation."""
import numpy as np
import pytest

import support as S
from qm_door_amd import abi, api


def _ft_interface(lib=None):
    itf = api.QMInterface(lib=lib) if lib is not None else api.QMInterface()
    itf.problem.settings.ee_contact_stiffness = S.FT_STIFFNESS
    itf.problem.settings.ee_force_mu = S.FT_MU
    return itf


def test_oracle_force_tracking_terms_are_consistent():
    """Invariants of the formulation in the oracle: the contact force enters the momentum rate as f_e / m and (p_ee - com) x f_e / m, its
    Jacobian matches finite differences, and with the EE on its target the force equals the reference (the anchor is placed that way)."""
    itf = _ft_interface()
    orc = S.Oracle(itf.problem)
    x_nom = itf.initial_state
    x0, tt, ts, contact = S.door_opening_batch(orc, x_nom, 2)
    m = itf.robot_mass
    u = np.zeros(30); u[2] = u[5] = u[8] = u[11] = m * 9.81 / 4
    nev, ev, md = S.trot_schedule(0.0)              # pure stance
    # momentum rate with / without the contact at the second knot's anchor (time t_end: alpha = 0 -> right knot)
    orc.set_ee_contact_ref(None)
    h = 1e-5     # a tiny step: the defect difference divided by it is the momentum rate the contact adds, up to O(h)
    f_free = orc.lq_node(1.5, h, x0[0], u, x0[0], False, 0, ev, md, tt[0], ts[0])
    orc.set_ee_contact_ref(contact[0])
    f_ft = orc.lq_node(1.5, h, x0[0], u, x0[0], False, 0, ev, md, tt[0], ts[0])
    orc.set_ee_contact_ref(None)
    fe = S.ee_contact_force(orc, x0[0], contact[0, 1])
    _, _, ee, _, com = orc.kinematics(x0[0], u)
    # b = x + h/2 (k1 + k2) - xnext: the difference of the two defects is h * [f_e / m ; (p_ee - com) x f_e / m] + O(h^2)
    db = (f_ft["b"] - f_free["b"])[:6] / h
    expect = np.r_[fe / m, np.cross(ee - com, fe) / m]
    assert np.abs(db - expect).max() <= 1e-3 * np.abs(expect).max()
    # Gauss-Newton weight of the soft constraint on the EE position rows: Q gains mu_f K^2 Jp^T Jp (rank 3, positive semi-definite)
    dQ = (f_ft["Q"] - f_free["Q"]) / h      # costs are scaled by the step
    w = np.linalg.eigvalsh(0.5 * (dQ + dQ.T))
    assert w.min() >= -1e-9 * w.max() and (w > 1e-9 * w.max()).sum() == 3
    # on-target EE -> force equals the reference
    tgt_state = ts[0, 1, :30]
    _, _, ee_t, _, _ = orc.kinematics(tgt_state, u)
    shift = ts[0, 1, 30:33] - ee_t                    # move the base so that the EE sits on its target
    xs = tgt_state.copy(); xs[6:9] += shift
    assert np.abs(S.ee_contact_force(orc, xs, contact[0, 1]) - contact[0, 1, :3]).max() <= 1e-9


def test_emu_force_tracking_matches_oracle():
    """CPU tier: the kernels compiled for the host, with the contact on, against the oracle (LQ blocks, trajectories, WBC torques)."""
    lib = abi.load_library(S.build_emu())
    itf = _ft_interface(lib)
    orc = S.Oracle(itf.problem)
    B, N = 2, 6
    x_nom = itf.initial_state
    x0, tt, ts, contact = S.door_opening_batch(orc, x_nom, B, t_end=0.08)
    nev, ev, md = S.trot_schedule(2.0, phase0=0.03)
    sol = api.GpuSolver(itf, max_batch=B, max_nodes=N)
    sol.enable_debug(True)
    oT, oX, oU, oM, oS = np.zeros((B, N + 1)), np.zeros((B, N + 1, 30)), np.zeros((B, N, 30)), np.zeros((B, N + 1), dtype=np.int32), np.zeros((B, abi.NSTATS))
    a = sol.mpc_args(B, N, x0, tt, ts, np.full(B, nev, dtype=np.int32), np.tile(ev, (B, 1)).copy(), np.tile(md, (B, 1)).copy(), oT, oX, oU, oM, oS, t0=np.zeros(B),
                     ee_contact_ref=contact)
    sol.mpc(a)
    dt = itf.problem.settings.dt
    for inst in range(B):
        orc.set_ee_contact_ref(contact[inst])
        for k in (1, 4):
            g = sol.debug_lq(inst, k)
            mode = orc.node_mode_at(ev[:nev], md[:nev + 1], k * dt)
            flags = [(mode >> (3 - c)) & 1 for c in range(4)]
            u = np.zeros(30)
            for c in range(4):
                if flags[c]:
                    u[3 * c + 2] = itf.robot_mass * 9.81 / sum(flags)
            o = orc.lq_node(k * dt, dt, x0[inst], u, x0[inst], False, nev, ev, md, tt[inst], ts[inst])
            for key in ("A", "B", "b", "Q", "R", "q", "r", "C", "D", "e"):
                assert np.abs(g[key] - o[key]).max() <= 1e-10 * max(1.0, np.abs(o[key]).max()), (inst, k, key)
        ref = orc.mpc_solve(N, 0.0, x0[inst], tt[inst], ts[inst], nev, ev, md)
        assert np.abs(oX[inst] - ref["X"]).max() <= 1e-8 * max(1.0, np.abs(ref["X"]).max())
        assert np.abs(oU[inst] - ref["U"]).max() <= 1e-8 * max(1.0, np.abs(ref["U"]).max())
        assert np.allclose(oS[inst][:7], ref["stats"][:7], rtol=1e-8, atol=1e-10)
    orc.set_ee_contact_ref(None)
    # the contact really changed the solution
    freeX, keepX = np.zeros_like(oX), oX.copy()
    a0 = sol.mpc_args(B, N, x0, tt, ts, np.full(B, nev, dtype=np.int32), np.tile(ev, (B, 1)).copy(), np.tile(md, (B, 1)).copy(), oT.copy(), freeX, oU.copy(),
                      oM.copy(), oS.copy(), t0=np.zeros(B))
    sol.mpc(a0)
    assert np.abs(freeX - keepX).max() > 1e-6
    # WBC with the external end-effector force
    fe = np.array([S.ee_contact_force(orc, x0[i], contact[i, 0]) + np.array([3.0, -2.0, 1.0]) for i in range(B)])
    rbd = np.array([S.rbd_from_state(orc, x0[i]) for i in range(B)])
    out, st, il = np.zeros((B, 54)), np.zeros(B, dtype=np.int32), np.zeros((B, 30))
    wa = sol.wbc_args(B, rbd, np.full(B, 0.002), np.full(B, 20.0), il, out, st, keepX[:, 0].copy(), oU[:, 0].copy(), oM[:, 0].copy(), 0, ee_force=fe)
    sol.wbc(wa)
    for i in range(B):
        orc.set_wbc_ee_force(fe[i])
        s, ref, _ = orc.wbc_update(keepX[i, 0], oU[i, 0], rbd[i], int(oM[i, 0]), 0.002, 20.0, np.zeros(30))
        orc.set_wbc_ee_force(None)
        s0, ref0, _ = orc.wbc_update(keepX[i, 0], oU[i, 0], rbd[i], int(oM[i, 0]), 0.002, 20.0, np.zeros(30))
        assert st[i] == 0 and s == 0
        assert np.abs(out[i] - ref).max() <= 1e-6 * max(1.0, np.abs(ref).max())
        assert np.abs(ref[36:] - ref0[36:]).max() > 1e-3          # the arm torques carry the external load


@pytest.mark.gpu
def test_config4_force_tracking_n100_batch1024():
    """BASELINE.json configs[3]: batch 1024, N = 100, door-opening reference, half the instances standing and half trotting (seed 2):
    whole batch finite / factorised, EVERY instance against the oracle (trajectories, modes, step lengths, and the WBC torques with the
    external end-effector force of the contact model) at the north_star tolerance."""
    import torch
    import gpu_harness as G
    itf = _ft_interface()
    orc = S.Oracle(itf.problem)
    B, N = 1024, 100
    dt = itf.problem.settings.dt
    x_nom = itf.initial_state
    x0, tt, ts, contact = S.door_opening_batch(orc, x_nom, B, seed=2, t_end=N * dt)
    nev_t, ev_t, md_t = S.trot_schedule(N * dt + 1.0, phase0=0.2)
    nev_s, ev_s, md_s = S.trot_schedule(0.0)          # STANCE throughout
    trot = (np.arange(B) % 2) == 1
    nev = np.where(trot, nev_t, nev_s).astype(np.int32)
    ev = np.where(trot[:, None], ev_t[None, :], ev_s[None, :]); md = np.where(trot[:, None], md_t[None, :], md_s[None, :]).astype(np.int32)
    sol = G.make_solver(itf, B, N)
    cdev = G.dev(contact, torch.float64)
    # robots in motion (support.moving_inputs): measured twist / joint rates, momentum-consistent x0, non-zero inputLast_, t < 10 and t >= 10, t_eval between nodes
    mv = S.moving_inputs(orc, x0, dt, seed=23)
    x0, rbd = mv["x0"], mv["rbd"]
    mb = G.MpcBatch(x0, tt, ts, nev, ev, md, N)
    mb.args.ee_contact_ref = cdev.data_ptr()
    fe = np.array([S.ee_contact_force(orc, x0[i], contact[i, 0]) for i in range(B)])
    wb = G.WbcBatch(rbd, np.full(B, 0.002), mv["time"], mv["input_last"])
    fdev = G.dev(fe, torch.float64)
    wb.args.ee_force = fdev.data_ptr()
    sol.debug_poison()
    sol.cycle(mb.args, G.dev(mv["t_eval"], torch.float64), wb.args)
    r, w = mb.results(), wb.results()
    assert np.isfinite(r["X"]).all() and np.isfinite(r["U"]).all() and np.isfinite(w["out"]).all()
    assert (r["stats"][:, 7] == 0).all() and (w["status"] == 0).all()
    r.update(w)
    ref = S.Oracle(itf.problem, fast=True).cycle_batch(N, x0, tt, ts, nev, ev, md, contact=contact, rbd=rbd, ee_force=fe, t_eval=mv["t_eval"], time=mv["time"],
                                                       input_last=mv["input_last"])   # all 1024 instances
    S.assert_parity(S.parity_report("configs3_force_tracking_1024xN100_moving", r, ref))
    # the force soft constraint does its job: at the end of the horizon the planned contact force is closer to the reference than without it
    from numpy.linalg import norm
    k_end = N
    errs = []
    for i in (0, 1, 510, 1023):
        fe_end = S.ee_contact_force(orc, r["X"][i][k_end], contact[i, 1])
        errs.append(norm(fe_end - contact[i, 1, :3]) / max(1.0, norm(contact[i, 1, :3])))
    assert max(errs) < 0.5, errs
