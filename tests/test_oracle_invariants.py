import os
"""Pins the CPU oracle (PARITY UNPINNED: the reference ships no golden vectors) by the invariant suite of SURVEY.md 8(c):
model identities, finite-difference checks of every derivative, rigid-body dynamics identities, QP optimality (KKT),
hierarchy properties of the HoQP cascade, bit-exact mode tables, swing-spline boundary conditions, SQP convergence."""
import itertools

import numpy as np
import pytest

import support as S


@pytest.fixture(scope="module")
def rng():
    return np.random.default_rng(11)


def _state(interface, rng, scale=1.0):
    x = interface.initial_state + rng.uniform(-1, 1, 30) * 0.1 * scale
    u = np.zeros(30)
    u[[2, 5, 8, 11]] = interface.robot_mass * 9.81 / 4
    u += rng.uniform(-1, 1, 30) * np.r_[np.full(12, 10.0), np.full(18, 0.5)] * scale
    return x, u


# ------------------------------------------------------------------------------------------------ model / dynamics
def test_nominal_flow_map(interface, oracle):
    x = interface.initial_state
    u = np.zeros(30); u[[2, 5, 8, 11]] = interface.robot_mass * 9.81 / 4
    f = oracle.flow_map(x, u)
    assert np.abs(f[:3]).max() < 1e-13            # weight compensation: no linear momentum rate
    assert np.abs(f[6:]).max() < 1e-13            # zero momentum, zero joint velocity: nothing moves
    fp, _, _, _, com = oracle.kinematics(x, u)
    assert np.allclose(f[3:6], sum(np.cross(fp[c] - com, u[3 * c:3 * c + 3]) for c in range(4)) / interface.robot_mass, atol=1e-13)


def test_centroidal_matrix_identities(interface, oracle, rng):
    x, u = _state(interface, rng)
    A = oracle.centroidal_matrix(x[6:])
    m = interface.robot_mass
    assert np.allclose(A[:3, :3], m * np.eye(3), atol=1e-12)      # A_G[0:3,0:3] = m I
    assert np.allclose(A[3:, :3], 0, atol=1e-12)                  # translation carries no angular momentum about the com
    # flow map base velocity is consistent with A_G: A_G [v_b; v_j] = m h
    f = oracle.flow_map(x, u)
    v = np.r_[f[6:12], u[12:]]
    assert np.allclose(A @ v, m * x[:6], atol=1e-10)
    # linear momentum = m * com velocity (finite difference of the com along v)
    eps = 1e-6
    com = lambda q: oracle.kinematics(np.r_[x[:6], q], u)[4]
    dcom = (com(x[6:] + eps * v) - com(x[6:] - eps * v)) / (2 * eps)
    assert np.allclose(A[:3] @ v, m * dcom, atol=1e-7)


def test_flow_map_derivatives_vs_finite_differences(interface, oracle, rng):
    x, u = _state(interface, rng)
    f, A, B = oracle.flow_map_lin(x, u)
    eps = 1e-6
    for j in range(30):
        d = np.zeros(30); d[j] = eps
        assert np.allclose(A[:, j], (oracle.flow_map(x + d, u) - oracle.flow_map(x - d, u)) / (2 * eps), rtol=1e-6, atol=1e-7)
        assert np.allclose(B[:, j], (oracle.flow_map(x, u + d) - oracle.flow_map(x, u - d)) / (2 * eps), rtol=1e-6, atol=1e-7)


def test_lq_node_against_finite_differences(interface, oracle, rng):
    """Constraint Jacobians, cost gradient/Hessian structure and the RK2 sensitivities of one trot node."""
    x, u = _state(interface, rng, 0.5)
    nev, ev, md = S.trot_schedule(2.0)
    tgt = S.nominal_target(oracle, interface.initial_state)
    tt, ts = np.zeros(1), tgt[None, :].copy()
    t, dt = 0.1, 0.015
    xn = x + 0.01
    o = oracle.lq_node(t, dt, x, u, xn, False, nev, ev, md, tt, ts)
    assert o["nc"] == 14                                           # trot: 2 stance (3 rows) + 2 swing (3 + 1 rows)
    eps = 1e-6
    def pert(j, s):
        d = np.zeros(60); d[j] = s
        return oracle.lq_node(t, dt, x + d[:30], u + d[30:], xn, False, nev, ev, md, tt, ts)
    for j in range(60):
        p, q = pert(j, eps), pert(j, -eps)
        col = (p["e"] - q["e"]) / (2 * eps)
        ref = o["C"][:, j] if j < 30 else o["D"][:, j - 30]
        assert np.allclose(col, ref, rtol=1e-6, atol=1e-7)
        colb = (p["b"] - q["b"]) / (2 * eps)
        refb = o["A"][:, j] if j < 30 else o["B"][:, j - 30]
        assert np.allclose(colb, refb, rtol=1e-6, atol=1e-7)
        g = (p["cost"] - q["cost"]) / (2 * eps)
        refg = o["q"][j] if j < 30 else o["r"][j - 30]
        assert abs(g - refg) <= 1e-5 * max(1.0, abs(refg))
    assert np.allclose(o["Q"], o["Q"].T) and np.allclose(o["R"], o["R"].T)
    assert np.linalg.eigvalsh(o["R"]).min() > 0 and np.linalg.eigvalsh(o["Q"]).min() > -1e-9


def test_rigid_body_dynamics_identities(interface, oracle, rng):
    x, _ = _state(interface, rng)
    v = rng.uniform(-1, 1, 24) * 0.5
    rbd = S.rbd_from_state(oracle, x)
    # measured velocities: world angular velocity from Euler rates
    e = x[9:12]
    sz, cz, sy, cy = np.sin(e[0]), np.cos(e[0]), np.sin(e[1]), np.cos(e[1])
    w = np.array([-sz * v[4] + cy * cz * v[5], cz * v[4] + cy * sz * v[5], v[3] - sy * v[5]])
    rbd[24:27] = w; rbd[27:30] = v[:3]; rbd[30:48] = v[6:]
    u = np.zeros(30)
    mdl = oracle.wbc_model(x, u, rbd, 0.002, np.zeros(30))
    M, nle = mdl["M"], mdl["nle"]
    assert np.allclose(mdl["qv"][1], v, atol=1e-12)
    assert np.allclose(M, M.T, atol=1e-12) and np.linalg.eigvalsh(M).min() > 0
    assert np.allclose(M[:3] @ v, oracle.centroidal_matrix(x[6:])[:3] @ v, atol=1e-10)      # base rows of M v = linear momentum
    # gravity part: nle(q, 0) = dV/dq by finite differences of the potential energy
    rbd0 = rbd.copy(); rbd0[24:48] = 0
    g = oracle.wbc_model(x, u, rbd0, 0.002, np.zeros(30))["nle"]
    m_tot, eps = interface.robot_mass, 1e-6
    V = lambda q: m_tot * 9.81 * oracle.kinematics(np.r_[x[:6], q], u)[4][2]
    dV = np.array([(V(x[6:] + eps * np.eye(24)[k]) - V(x[6:] - eps * np.eye(24)[k])) / (2 * eps) for k in range(24)])
    assert np.allclose(g, dV, atol=1e-6)
    # Coriolis part: v^T C(q,v) v = 1/2 v^T dM/dt v   (M_dot - 2C skew symmetric)
    def Mat(q):
        r2 = rbd0.copy(); r2[3:6] = q[:3]; r2[0:3] = q[3:6]; r2[6:24] = q[6:]
        return oracle.wbc_model(np.r_[x[:6], q], u, r2, 0.002, np.zeros(30))["M"]
    dM = (Mat(x[6:] + eps * v) - Mat(x[6:] - eps * v)) / (2 * eps)
    assert abs(v @ (nle - g) - 0.5 * v @ dM @ v) < 1e-6
    # dJ v = d/dt (J v) with v constant
    def Jv(q):
        r2 = rbd.copy(); r2[3:6] = q[:3]; r2[0:3] = q[3:6]; r2[6:24] = q[6:]
        # keep the generalized velocity v fixed: rebuild the world angular velocity for the perturbed Euler angles
        s0, c0, s1, c1 = np.sin(q[3]), np.cos(q[3]), np.sin(q[4]), np.cos(q[4])
        r2[24:27] = [-s0 * v[4] + c1 * c0 * v[5], c0 * v[4] + c1 * s0 * v[5], v[3] - s1 * v[5]]
        mm = oracle.wbc_model(np.r_[x[:6], q], u, r2, 0.002, np.zeros(30))
        return np.r_[mm["J"] @ v, mm["armJ"] @ v, mm["baseJ"] @ v]
    fd = (Jv(x[6:] + eps * v) - Jv(x[6:] - eps * v)) / (2 * eps)
    an = np.r_[mdl["dJ"] @ v, mdl["armDJ"] @ v, mdl["baseDJ"] @ v]
    assert np.allclose(fd, an, atol=1e-6)


# ------------------------------------------------------------------------------------------------ schedule
def test_mode_numbers_and_lookup(oracle):
    ev = np.array([0.35, 0.70]); md = np.array([9, 6, 15], dtype=np.int32)
    assert oracle.mode_at(ev, md, 0.0) == 9 and oracle.mode_at(ev, md, 0.35) == 9      # an event time still belongs to the phase before it
    assert oracle.mode_at(ev, md, 0.35 + 1e-12) == 6 and oracle.mode_at(ev, md, 0.70) == 6 and oracle.mode_at(ev, md, 5.0) == 15
    # a SHOOTING NODE placed on an event time is upstream's PostEvent node: it takes the mode that starts there
    assert oracle.node_mode_at(ev, md, 0.35) == 6 and oracle.node_mode_at(ev, md, 0.70) == 15 and oracle.node_mode_at(ev, md, 0.35 - 1e-12) == 9
    assert oracle.node_mode_at(ev, md, 0.0) == 9 and oracle.node_mode_at(ev, md, 0.5) == 6


def test_event_node_switches_constraint_rows(interface, oracle):
    """On an event-aligned grid the node AT the switch carries the post-event mode: its equality rows are those of the new contact set."""
    from qm_door_amd import api
    dt = interface.problem.settings.dt
    x_nom = interface.initial_state
    tgt = S.nominal_target(oracle, x_nom)
    nev, ev, md = S.trot_schedule(2.0, phase0=0.04)
    N, grid = api.time_grid_with_events(0.0, 0.3, dt, ev[:nev], lib=interface.lib)
    r = oracle.mpc_solve(N, 0.0, x_nom, np.zeros(1), tgt[None, :].copy(), nev, ev, md, time_grid=grid, line_search=False)
    k = int(np.argmin(np.abs(grid - ev[0])))
    assert grid[k] == ev[0]
    assert r["mode"][k - 1] == md[0] and r["mode"][k] == md[1] and md[0] != md[1]
    nc = lambda m: sum(3 if (m >> (3 - c)) & 1 else 4 for c in range(4))
    lq = oracle.lq_node(grid[k], grid[k + 1] - grid[k], r["X"][k], r["U"][k], r["X"][k + 1], False, nev, ev, md, np.zeros(1), tgt[None, :].copy())
    assert lq["nc"] == nc(int(md[1]))


def test_swing_spline_boundary_conditions(interface, oracle):
    st = interface.problem.settings
    nev, ev, md = S.trot_schedule(3.0)          # events 0, .35, .70, ...; RF/LH swing on [0, .35], LF/RH on [.35, .70]
    eps = 1e-9
    zp, zv = oracle.swing_reference(nev, ev, md, 0.0 + eps)
    assert abs(zp[1]) < 1e-8 and abs(zv[1] - st.liftoff_velocity) < 1e-6 and zp[0] == 0 and zv[0] == 0     # lift-off of RF, LF in stance
    zp, zv = oracle.swing_reference(nev, ev, md, 0.175)
    assert abs(zp[1] - st.swing_height) < 1e-12 and abs(zv[1]) < 1e-12                                    # apex at mid swing
    zp, zv = oracle.swing_reference(nev, ev, md, 0.35 - eps)
    assert abs(zp[2]) < 1e-8 and abs(zv[2] - st.touchdown_velocity) < 1e-6                                # touch-down


# ------------------------------------------------------------------------------------------------ QP
def _kkt_residual(H, c, D, f, z, act_tol=1e-6):
    """max violation of the KKT conditions; multipliers of the (near-)active rows by non-negative least squares."""
    from scipy.optimize import nnls
    r = D @ z - f
    act = r > -act_tol * max(1.0, np.abs(f).max())
    g = H @ z + c
    stat = np.abs(g).max()
    if act.any():
        lam, _ = nnls(D[act].T, -g)
        stat = np.abs(g + D[act].T @ lam).max()
    return max(stat, max(r.max(), 0))


def test_qp_solver_kkt_and_enumeration(oracle, rng):
    for trial in range(5):
        n, m = 6, 8
        L = rng.normal(size=(n, n)); H = L @ L.T + 0.1 * np.eye(n)
        c = rng.normal(size=n); D = rng.normal(size=(m, n)); f = rng.uniform(0.1, 1.0, m)
        it, z, res = oracle.qp_solve(H, c, D, f)
        assert it >= 0
        scale = max(1, np.abs(c).max())
        assert _kkt_residual(H, c, D, f, z) < 1e-6 * scale
        # brute force over active sets
        best, bestz = np.inf, None
        for k in range(0, n + 1):
            for act in itertools.combinations(range(m), k):
                Da = D[list(act)]
                K = np.block([[H, Da.T], [Da, np.zeros((k, k))]])
                try:
                    sol = np.linalg.solve(K, np.r_[-c, f[list(act)]])
                except np.linalg.LinAlgError:
                    continue
                zz, lam = sol[:n], sol[n:]
                if (D @ zz - f).max() < 1e-9 and (lam > -1e-9).all():
                    val = 0.5 * zz @ H @ zz + c @ zz
                    if val < best:
                        best, bestz = val, zz
        assert bestz is not None and np.allclose(z, bestz, atol=1e-6)


# ------------------------------------------------------------------------------------------------ HoQP
def test_hoqp_hierarchy_properties(interface, oracle, rng):
    x_nom, m = interface.initial_state, interface.robot_mass
    u = np.zeros(30); u[[2, 11]] = m * 9.81 / 2; u[12:] = rng.uniform(-1, 1, 18) * 0.05
    xd = x_nom + rng.uniform(-1, 1, 30) * 0.01
    rbd = S.rbd_from_state(oracle, x_nom + rng.uniform(-1, 1, 30) * 0.005, rng.uniform(-1, 1, 24) * 0.02)
    il = u.copy()
    st, out, il2 = oracle.wbc_update(xd, u, rbd, 9, 0.002, 20.0, il)
    assert st == 0 and np.array_equal(il2, u)
    x, tau = out[:36], out[36:]
    t0 = oracle.wbc_task(0, xd, u, rbd, 9, 0.002, 20.0, il)
    assert t0["A"].shape == (18, 36) and t0["D"].shape == (36 + 10 + 6, 36)                 # trot: (88,104) QP of SURVEY.md 8a
    assert np.abs(t0["A"] @ x - t0["b"]).max() < 1e-7                                    # highest priority equalities hold
    assert (t0["D"] @ x - t0["f"]).max() < 1e-7                                          # and its inequalities (feasible case)
    assert np.abs(x[24 + 3:24 + 9]).max() < 1e-7                                         # swing feet carry no force
    assert (np.abs(tau) <= np.r_[np.tile([35.278, 35.278, 44.4], 4), [30, 60, 30, 30, 30, 30]] + 1e-6).all()
    # level solutions: each lower level keeps the residual of the levels above
    l0, l1, l2 = (oracle.wbc_level(k, xd, u, rbd, 9, 0.002, 20.0, il) for k in range(3))
    assert (l0["H"].shape[0], l0["D"].shape[0]) == (88, 104) and (l1["H"].shape[0], l1["D"].shape[0]) == (18, 52) and (l2["H"].shape[0], l2["D"].shape[0]) == (2, 52)
    t1 = oracle.wbc_task(1, xd, u, rbd, 9, 0.002, 20.0, il)
    r1 = lambda xx: np.linalg.norm(t1["A"] @ xx - t1["b"])
    assert abs(r1(l2["x"]) - r1(l1["x"])) < 1e-6 * max(1.0, r1(l1["x"]))
    for lv in (l0, l1, l2):
        assert _kkt_residual(lv["H"], lv["c"], lv["D"], lv["f"], lv["sol"]) < 1e-5 * max(1.0, np.abs(lv["c"]).max())


def test_wbc_standing_is_physically_consistent(interface, oracle):
    x = interface.initial_state
    u = np.zeros(30); u[[2, 5, 8, 11]] = interface.robot_mass * 9.81 / 4
    rbd = S.rbd_from_state(oracle, x)
    st, out, _ = oracle.wbc_update(x, u, rbd, 15, 0.002, 20.0, u)
    assert st == 0
    F = out[24:36].reshape(4, 3)
    mu = interface.problem.settings.wbc_friction_coefficient
    assert (F[:, 2] > 0).all() and (np.abs(F[:, 0]) <= mu * F[:, 2] + 1e-9).all() and (np.abs(F[:, 1]) <= mu * F[:, 2] + 1e-9).all()
    # Newton for the whole robot: sum F = m (a_com + g); a_com from the linear rows of the centroidal map (zero velocity: no bias)
    A = oracle.centroidal_matrix(x[6:])
    assert np.allclose(F.sum(0), A[:3] @ out[:24] + np.array([0, 0, interface.robot_mass * 9.81]), atol=1e-6)


# ------------------------------------------------------------------------------------------------ SQP
def test_sqp_iterations_converge(interface, oracle):
    x0 = S.perturbed_states(interface.initial_state, 1, seed=4)[0]
    tgt = S.nominal_target(oracle, interface.initial_state)
    tt, ts = np.zeros(1), tgt[None, :].copy()
    nev, ev, md = S.trot_schedule(2.0, phase0=0.05)
    r = oracle.mpc_solve(20, 0.0, x0, tt, ts, nev, ev, md)
    v = [r["stats"][1], r["stats"][3]]
    for _ in range(4):
        r = oracle.mpc_solve(20, 0.0, x0, tt, ts, nev, ev, md, warm=(r["X"], r["U"]))
        v.append(r["stats"][3])
        assert r["stats"][4] > 0
    assert v[-1] < 1e-3 * v[0]                          # constraint violation collapses
    assert np.allclose(r["X"][0], x0)
    assert list(r["mode"][:5]) == [15, 15, 15, 15, 9]   # nodes at t <= 0.05 are STANCE, then LF_RH


def test_riccati_step_equals_dense_kkt_solve(interface, oracle):
    """Projection + Riccati recursion == one dense KKT solve of the same equality-constrained QP."""
    N, dt, m = 5, interface.problem.settings.dt, interface.robot_mass
    x0 = S.perturbed_states(interface.initial_state, 1, seed=11)[0]
    tgt = S.nominal_target(oracle, interface.initial_state)
    tt, ts = np.zeros(1), tgt[None, :].copy()
    nev, ev, md = S.trot_schedule(1.0, phase0=2.5 * dt)          # nodes 0..2 STANCE, 3.. LF_RH
    r = oracle.mpc_solve(N, 0.0, x0, tt, ts, nev, ev, md, line_search=False)
    # cold start (QMInitializer): x_k = x0, u_k = weight compensation of the node's mode
    lq = []
    for k in range(N + 1):
        mode = oracle.node_mode_at(ev[:nev], md[:nev + 1], k * dt)
        flags = [(mode >> (3 - c)) & 1 for c in range(4)]
        u = np.zeros(30)
        for c in range(4):
            if flags[c]:
                u[3 * c + 2] = m * 9.81 / sum(flags)
        lq.append((oracle.lq_node(k * dt, dt if k < N else 0.0, x0, u if k < N else None, x0, k == N, nev, ev, md, tt, ts), u))
    nz = 30 * (N + 1) + 30 * N
    ix = lambda k: slice(30 * k, 30 * k + 30)
    iu = lambda k: slice(30 * (N + 1) + 30 * k, 30 * (N + 1) + 30 * k + 30)
    H, g, rows, rhs = np.zeros((nz, nz)), np.zeros(nz), [], []
    E = np.zeros((30, nz)); E[:, ix(0)] = np.eye(30); rows.append(E); rhs.append(np.zeros(30))
    for k, (o, _) in enumerate(lq):
        H[ix(k), ix(k)] += o["Q"]; g[ix(k)] += o["q"]
        if k == N:
            break
        H[iu(k), iu(k)] += o["R"]; g[iu(k)] += o["r"]
        E = np.zeros((30, nz)); E[:, ix(k)] = o["A"]; E[:, iu(k)] = o["B"]; E[:, ix(k + 1)] = -np.eye(30); rows.append(E); rhs.append(-o["b"])
        E = np.zeros((o["nc"], nz)); E[:, ix(k)] = o["C"]; E[:, iu(k)] = o["D"]; rows.append(E); rhs.append(-o["e"])
    Aeq, beq = np.vstack(rows), np.concatenate(rhs)
    KKT = np.block([[H, Aeq.T], [Aeq, np.zeros((Aeq.shape[0],) * 2)]])
    sol = np.linalg.lstsq(KKT, np.r_[-g, beq], rcond=None)[0][:nz]
    assert np.abs(KKT[:nz] @ np.linalg.lstsq(KKT, np.r_[-g, beq], rcond=None)[0] + g).max() < 1e-7
    dX = np.array([sol[ix(k)] for k in range(N + 1)]); dU = np.array([sol[iu(k)] for k in range(N)])
    U0 = np.array([u for _, u in lq[:N]])
    assert np.abs(r["X"] - (x0[None, :] + dX)).max() <= 1e-9 * max(1.0, np.abs(dX).max())
    assert np.abs(r["U"] - (U0 + dU)).max() <= 1e-8 * max(1.0, np.abs(dU).max())


def test_two_derivative_routes_of_the_oracle_agree(interface, oracle):
    """flowMap<Dual<60>> (every direction carried through the kinematics) against the structured route of the timing-grade build (21
    configuration directions as dual numbers + closed-form columns for momentum / joint rates / forces / base position): all sixty columns of
    the flow map, the foot positions / velocities and the end-effector pose, with and without the force-tracking contact."""
    rng = np.random.default_rng(5)
    x_nom, m = interface.initial_state, interface.robot_mass
    for trial in range(4):
        x = x_nom + rng.uniform(-1, 1, 30) * np.r_[np.full(6, 0.2), np.full(3, 0.3), np.full(3, 0.3), np.full(18, 0.3)]
        u = np.r_[rng.uniform(-30, 30, 12), rng.uniform(-1, 1, 18)]; u[2::3][:4] += m * 9.81 / 4
        assert oracle.structured_vs_dual60(x, u) <= 1e-11
        assert oracle.structured_vs_dual60(x, u, contact_stiffness=500.0, env=rng.uniform(-1, 1, 3)) <= 1e-9


def test_fast_oracle_build_reproduces_the_checker(interface, oracle):
    """The -O3 structured-derivative build that bench.py times is the same algorithm: one MPC solve and one WBC update agree with the checker
    build to round-off (the derivative route and the compiler's contraction of multiply-adds differ, nothing else)."""
    fast = S.Oracle(interface.problem, fast=True)
    assert fast.lib.qmo_is_fast_build() == 1 and oracle.lib.qmo_is_fast_build() == 0
    x_nom = interface.initial_state
    x0 = S.perturbed_states(x_nom, 1, seed=4)[0]
    tgt = S.nominal_target(oracle, x_nom)
    nev, ev, md = S.trot_schedule(1.0, phase0=0.04)
    N = 20
    a = oracle.mpc_solve(N, 0.0, x0, np.zeros(1), tgt[None, :].copy(), nev, ev, md)
    b = fast.mpc_solve(N, 0.0, x0, np.zeros(1), tgt[None, :].copy(), nev, ev, md)
    assert np.array_equal(a["mode"], b["mode"]) and a["stats"][4] == b["stats"][4]
    assert np.abs(a["X"] - b["X"]).max() <= 1e-9 and np.abs(a["U"] - b["U"]).max() <= 1e-8 * max(1.0, np.abs(a["U"]).max())
    rbd = S.rbd_from_state(oracle, x0)
    _, oa, _ = oracle.wbc_update(a["X"][0], a["U"][0], rbd, int(a["mode"][0]), 0.002, 20.0, np.zeros(30))
    _, ob, _ = fast.wbc_update(a["X"][0], a["U"][0], rbd, int(a["mode"][0]), 0.002, 20.0, np.zeros(30))
    assert np.abs(oa - ob).max() <= 1e-7 * max(1.0, np.abs(oa).max())
    split = fast.time_split()
    assert split["lq"] > 0 and split["riccati"] > 0 and split["linesearch"] > 0 and split["wbc_qp"] > 0


def test_model_against_independent_fixture(interface, oracle):
    """tests/golden/model_independent.npz (tests/golden/make_model_fixture.py): mass matrix as the Hessian of the kinetic energy, non-linear
    effects from Lagrange's equations, centroidal momentum matrix from body momenta -- all built from an independent parse of the URDF and
    complex-step differentiation of forward kinematics only.  Pins the oracle's rigid-body model AND the product's URDF loader (fixed-link
    merging, frame offsets, joint order), which the oracle's inputs come through."""
    import os
    fx = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "model_independent.npz"))
    assert abs(float(fx["total_mass"]) - interface.robot_mass) <= 1e-9
    for k in range(fx["q"].shape[0]):
        q, v = fx["q"][k], fx["v"][k]
        x = np.r_[np.zeros(6), q]
        rbd = np.zeros(55)
        rbd[0:3] = q[3:6]; rbd[3:6] = q[0:3]; rbd[6:24] = q[6:]
        rbd[24:27] = fx["omega_world"][k]; rbd[27:30] = v[0:3]; rbd[30:48] = v[6:]
        mm = oracle.wbc_model(x, np.zeros(30), rbd, 0.002, np.zeros(30))
        assert np.abs(mm["M"] - fx["M"][k]).max() <= 1e-10 * np.abs(fx["M"][k]).max()
        assert np.abs(mm["nle"] - fx["nle"][k]).max() <= 1e-7 * max(1.0, np.abs(fx["nle"][k]).max())
        A = oracle.centroidal_matrix(q)
        assert np.abs(A - fx["AG"][k]).max() <= 1e-10 * np.abs(fx["AG"][k]).max()
        fp, _, ee, _, com = oracle.kinematics(x, np.zeros(30))
        assert np.abs(fp.reshape(4, 3) - fx["feet"][k]).max() <= 1e-12 and np.abs(ee - fx["ee"][k]).max() <= 1e-12 and np.abs(com - fx["com"][k]).max() <= 1e-12


def test_full_size_hoqp_levels_against_a_primal_active_set_method(interface, oracle, rng):
    """Every level of the hierarchy, at the sizes the reference hands to qpOASES ((92,112), (18,56), (8,56) in stance), for all ten
    contact modes of the gait files: the oracle's interior point + polish against an independent textbook active-set method (tests/active_set_qp.py),
    the solver class the reference uses.  Compared: objective value and the decision part of the minimiser (the slack part is determined by it)."""
    import active_set_qp as AS
    x_nom, m = interface.initial_state, interface.robot_mass
    checked = undecided = 0
    for mode in (15, 9, 6, 10, 5, 13, 7, 14, 11, 0):
        flags = [(mode >> (3 - c)) & 1 for c in range(4)]
        u = np.zeros(30)
        for c in range(4):
            if flags[c]:
                u[3 * c:3 * c + 3] = [rng.uniform(-8, 8), rng.uniform(-8, 8), m * 9.81 / max(1, sum(flags))]
        u[12:] = rng.uniform(-1, 1, 18) * 0.1
        xd = x_nom + rng.uniform(-1, 1, 30) * 0.02
        rbd = S.rbd_from_state(oracle, x_nom + rng.uniform(-1, 1, 30) * 0.01, rng.uniform(-1, 1, 24) * 0.05)
        il = u + rng.uniform(-1, 1, 30) * 0.002
        for level in range(3):
            lv = oracle.wbc_level(level, xd, u, rbd, mode, 0.002, 20.0, il)
            nz = lv["H"].shape[0]
            if nz == 0 or lv["num_dec"] == 0:
                continue                                  # FLY: nothing left to decide below level 1
            H, c, D, f = lv["H"], lv["c"], lv["D"], lv["f"]
            z0 = AS.feasible_start(D, f, lv["num_dec"])
            if (D @ z0 - f).max() > 1e-7 * max(1.0, np.abs(f).max()):
                z0 = lv["sol"].copy(); z0[lv["num_dec"]:] += 1e-6     # inherited rows need the previous levels' margins: start next to the oracle's point
            try:
                z, iters = AS.solve(H, c, D, f, z0)
            except RuntimeError:
                undecided += 1        # the textbook method cycled at a degenerate vertex under every perturbation tried: no verdict on this problem
                continue
            obj = lambda v: 0.5 * v @ H @ v + c @ v
            sc = max(1.0, abs(obj(lv["sol"])))
            # 1e-5: the lowest level of three-leg stances is a nearly degenerate LP in the directions its Hessian does not see
            assert abs(obj(z) - obj(lv["sol"])) <= 1e-5 * sc, (mode, level, obj(z), obj(lv["sol"]))
            # the minimiser of a convex QP need not be unique (level 0 decides 36 variables with 18 equality rows and a 1e-12 regulariser), but H z and
            # c^T z are the same for every minimiser
            assert np.abs(H @ z - H @ lv["sol"]).max() <= 1e-4 * max(1.0, np.abs(H @ lv["sol"]).max()), (mode, level)
            assert abs(c @ z - c @ lv["sol"]) <= 1e-5 * sc
            if np.linalg.cond(H) < 1e8:   # a definite level Hessian pins the minimiser itself
                nd = lv["num_dec"]
                assert np.abs(z[:nd] - lv["sol"][:nd]).max() <= 1e-5 * max(1.0, np.abs(lv["sol"][:nd]).max()), (mode, level)
            checked += 1
    assert checked >= 25 and undecided <= 3


def _eigen_full_piv_lu(A, column_major=True):
    """Independent numpy restatement of Eigen 3.3's FullPivLU::computeInPlace + kernel() (upstream: bottomRightCorner(..).cwiseAbs().maxCoeff(&row, &col) is a scalar
    visitor over a column-major expression: column by column, first strict maximum).  column_major=False: the row-by-row scan of round 4, kept to show that the test
    tells the two orders apart.  Returns (pivot positions, kernel basis with a 1 on each free column, free columns)."""
    A = np.array(A, dtype=np.float64); r, c = A.shape; perm = list(range(c)); seq = []
    for k in range(min(r, c)):
        best, pr, pc = 0.0, k, k
        scan = ((i, j) for j in range(k, c) for i in range(k, r)) if column_major else ((i, j) for i in range(k, r) for j in range(k, c))
        for i, j in scan:
            if abs(A[i, j]) > best:
                best, pr, pc = abs(A[i, j]), i, j
        if best == 0.0:
            break
        seq.append((pr, pc))
        A[[k, pr]] = A[[pr, k]]; A[:, [k, pc]] = A[:, [pc, k]]; perm[k], perm[pc] = perm[pc], perm[k]
        for i in range(k + 1, r):
            f = A[i, k] / A[k, k]
            A[i, k + 1:] = A[i, k + 1:] - f * A[k, k + 1:]
    piv = np.abs(np.array([A[k, k] for k in range(len(seq))]))
    ok = [k for k in range(len(seq)) if piv[k] > piv.max() * 2.220446049250313e-16 * min(r, c)]
    freep = [j for j in range(c) if j not in ok]
    N = np.zeros((c, len(freep)))
    for col, fp in enumerate(freep):
        x = {}
        for k in reversed(ok):
            s_ = -A[k, fp] if fp >= k else 0.0
            for k2 in ok:
                if k2 > k:
                    s_ -= A[k, k2] * x[k2]
            x[k] = s_ / A[k, k]
        for k in ok:
            N[perm[k], col] = x[k]
        N[perm[fp], col] = 1.0
    return seq, N, [perm[fp] for fp in freep]


def test_kernel_basis_follows_eigens_pivot_order(interface, oracle):
    """HoQp.cpp:129 takes Eigen's fullPivLu().kernel(): the basis depends on the pivot order wherever magnitudes tie, and the level tasks carry unit rows -- ties are the
    rule.  Three matrices with HAND-DERIVED pivot sequences (Eigen's column-major visitor: first strict maximum column by column):
      M1 = [0 0 0 1; 1 0 1 0]: column 0 holds the first 1 at row 1 -> pivot (1, 0) [a row-by-row scan takes (0, 3)]; after the row swap the corner row is [0 0 1] ->
           pivot position (1, 3); pivot columns {0, 3}, free {2, 1} in position order; basis [-1 0 1 0], [0 1 0 0]  (row-by-row: free {0, 1}, basis [1 0 -1 0], [0 1 0 0])
      M2 = [1 1 0; 1 0 1]: pivot (0, 0); row 1 becomes [0 -1 1]: |-1| = |1| ties -> the smaller column, position (1, 1); free {2}; basis [-1 1 1]
      M3 = [0 1 1; 0 1 -1; 2 0 0] / tied unit rows around a larger entry: pivot (2, 0) (the 2), then the corner [1 1; 1 -1] -> (1, 1)... derived below
    and one A Z of a real stance tick (18 x 36, level 0) against the independent numpy restatement above -- same pivots, same basis to 1e-12."""
    M1 = np.array([[0, 0, 0, 1], [1, 0, 1, 0]], dtype=float)
    ker, free, seq = oracle.kernel_full_piv_lu(M1)
    assert seq == [(1, 0), (1, 3)] and free == [2, 1]
    assert np.array_equal(ker, np.array([[-1, 0], [0, 1], [1, 0], [0, 0]], dtype=float))
    M2 = np.array([[1, 1, 0], [1, 0, 1]], dtype=float)
    ker, free, seq = oracle.kernel_full_piv_lu(M2)
    assert seq == [(0, 0), (1, 1)] and free == [2] and np.array_equal(ker[:, 0], np.array([-1.0, 1.0, 1.0]))
    # M3: step 0: the 2 at (2, 0).  Rows 0 and 2 swap: [2 0 0 0; 0 1 -1 0; 0 1 1 1].  Step 1: corner columns 1..3, rows 1..2, column by column: |1| at (1, 1) first
    # -> pivot (1, 1), no swap; row 2 becomes [0 0 2 1].  Step 2: corner [2 1] -> (2, 2).  Pivot columns {0, 1, 2}, free {3}: 2 x2 = -1 -> x2 = -1/2, x1 = x2 = -1/2, x0 = 0
    M3 = np.array([[0, 1, 1, 1], [0, 1, -1, 0], [2, 0, 0, 0]], dtype=float)
    ker, free, seq = oracle.kernel_full_piv_lu(M3)
    assert seq == [(2, 0), (1, 1), (2, 2)] and free == [3] and np.array_equal(ker[:, 0], np.array([0.0, -0.5, -0.5, 1.0]))
    for M in (M1, M2, M3):      # the numpy restatement agrees with the hand derivation as well
        s_, N_, f_ = _eigen_full_piv_lu(M)
        k_, fr_, sq_ = oracle.kernel_full_piv_lu(M)
        assert s_ == sq_ and f_ == fr_ and np.array_equal(N_, k_)
    # a real level-0 task of a trot stance tick (LF_RH): equations of motion, contact rows, zero-force rows of the swing legs
    xd = S.perturbed_states(interface.initial_state, 1, seed=8)[0]
    ud = np.zeros(30); ud[[2, 11]] = interface.robot_mass * 9.81 / 2
    rbd = S.rbd_from_state(oracle, xd)
    tk = oracle.wbc_task(0, xd, ud, rbd, 9, 0.002, 20.0, np.zeros(30))
    A = tk["A"]
    assert A.shape == (18, 36)
    ker, free, seq = oracle.kernel_full_piv_lu(A)
    s_, N_, f_ = _eigen_full_piv_lu(A)
    assert seq == s_ and free == f_ and ker.shape == (36, 18)
    assert np.abs(ker - N_).max() <= 1e-12 * max(1.0, np.abs(N_).max()) and np.abs(A @ ker).max() <= 1e-9
    # (on THIS matrix the two scan orders agree -- its unit rows are ordered like their columns; M1 above is the case that tells them apart)
    assert _eigen_full_piv_lu(M1, column_major=False)[0] == [(0, 3), (1, 2)]


def test_every_level_ends_at_the_same_vertex_whatever_the_path(interface, oracle):
    """Every HoQP level ends with a primal active-set method (oracle/qmo_wbc.h activeSetPhase): the point it returns satisfies the KKT conditions on its working set and is
    THE minimiser -- so it cannot depend on how the method got there.  512 random instances over every contact mode, robots in motion, both controllers, three paths: the
    product's (interior point from 0.5 sqrt(scale) -> active set), another interior-point start (0.15 sqrt(scale)), and no interior point at all (the active-set method cold from z = 0, one
    working-set change at a time).  Stated bound: torques equal to 1e-9 rel-inf on all but 1 % of the instances (measured on 2 x 2048: 99.7 % within 1e-13, the rest nearly
    degenerate level problems -- a direction whose curvature sits at the rounding of the normal equations, or a multiplier at the rounding of its gradient -- where a
    path-dependent decision is unavoidable in any arithmetic), median <= 1e-13, and no level is ever flagged."""
    import test_gpu_wbc as TW
    for variant in (0, 1):
        B = 512
        c = TW.stress_batch(interface, variant, B)
        per = np.full(B, 0.002)

        def run(**kw):
            try:
                oracle.set_experiment(**kw)
                return oracle.wbc_batch(c["xd"], c["u"], c["rbd"], c["mode"], per, c["t"], c["il"].copy(), variant=variant)
            finally:
                oracle.set_experiment()
        ref, alt, cold = run(), run(lower_level_start=0.15), run(no_interior_point=True)
        assert (ref["status"] == 0).all() and (alt["status"] == 0).all() and (cold["status"] == 0).all()
        assert (ref["polished"][ref["iterations"] > 0] == 1).all()             # every level that ran ended at a verified vertex
        tau = lambda r: r["out"][:, 36:]  # noqa: E731
        for name, other in (("start 0.15 sqrt(scale)", alt), ("cold active set", cold)):
            dev = S.rel_inf(tau(ref), tau(other))
            print("variant", variant, name, "max", dev.max(), "p99", np.percentile(dev, 99), "median", np.median(dev), "above 1e-9:", int((dev > 1e-9).sum()))
            assert np.median(dev) <= 1e-13 and (dev > 1e-9).sum() <= B // 100, (variant, name, dev.max(), int((dev > 1e-9).sum()))
        # the cold method needs more working-set changes, the product's path fewer passes in the tail
        assert cold["iterations"][:, 1].max() >= ref["iterations"][:, 1].max() - 5


def test_degenerate_lowest_level_is_solved_on_its_face(interface, oracle):
    """tests/golden/wbc_degenerate_stance_tick.npz: the WBC inputs of ONE tick of the static-walk closed loop (tests/test_closed_loop.py, instance 72, t = 10.501 s, full
    stance) on which the round-4 implementations -- an interior point with a relaxed re-solve behind it -- returned torques 12 % apart for inputs 1e-11 apart.  The
    contact-force level inherits rows the level above left strongly active: positively dependent in the variables that are left, a cone without interior.  They are
    equalities for this level (oracle/qmo_wbc.h eliminateImpliedEqualities) and are removed exactly before anything is solved.  Pinned here: the level's answer is the
    exact one (nothing moves: z = 0 up to rounding, no inherited row violated), it does not depend on the path (interior-point start, no interior point), and it is
    STABLE: input perturbations of 1e-9 relative move the torques by a small multiple of that."""
    c = np.load(os.path.join(S.ROOT, "tests", "golden", "wbc_degenerate_stance_tick.npz"))
    args = (c["xd"], c["ud"], c["rbd"], int(c["mode"]), 0.001, float(c["t"]))
    outs = []
    try:
        for kw in (dict(), dict(lower_level_start=0.15), dict(no_interior_point=True)):
            oracle.set_experiment(**kw)
            st, out, _ = oracle.wbc_update(*args, c["il"].copy())
            assert st == 0
            outs.append(out)
    finally:
        oracle.set_experiment()
    tau_scale = max(1.0, np.abs(outs[0][36:]).max())
    for o in outs[1:]:
        assert np.abs(o[36:] - outs[0][36:]).max() <= 1e-10 * tau_scale
    lv = oracle.wbc_level(2, *args, c["il"].copy())
    assert np.abs(lv["sol"]).max() <= 1e-9 and (lv["D"] @ lv["sol"] - lv["f"]).max() <= 1e-9        # the exact level: nothing to gain inside the inherited rows
    rng = np.random.default_rng(5)
    worst = 0.0
    for _ in range(8):
        xd = c["xd"] * (1 + 1e-9 * rng.uniform(-1, 1, 30)); rbd = c["rbd"].copy(); rbd[:48] *= 1 + 1e-9 * rng.uniform(-1, 1, 48)
        st, out, _ = oracle.wbc_update(xd, c["ud"], rbd, int(c["mode"]), 0.001, float(c["t"]), c["il"].copy())
        assert st == 0
        worst = max(worst, np.abs(out[36:] - outs[0][36:]).max() / tau_scale)
    assert worst <= 1e-6, worst          # (an amplification of <= 1e3 of the input perturbation; the round-4 oracle moved by 1.2e-1 here)
