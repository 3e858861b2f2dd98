"""-m gpu: argument validation and edge sizes of the C ABI (the reference throws C++ exceptions; the ABI returns status codes)."""
import ctypes as C

import numpy as np
import pytest

import support as S

pytestmark = pytest.mark.gpu


def _clone(args):
    """Field-by-field copy of a ctypes argument struct (structs holding pointers cannot be pickled / copy.copy'd)."""
    c = type(args)()
    C.memmove(C.byref(c), C.byref(args), C.sizeof(args))
    return c


def _batch(G, interface, oracle, B, N, seed=5, K=1):
    x0 = S.perturbed_states(interface.initial_state, B, seed=seed)
    tgt = S.nominal_target(oracle, interface.initial_state)
    tt = np.tile(np.linspace(0.0, 1.0, K), (B, 1)); ts = np.tile(tgt, (B, K, 1)).copy()
    nev, ev, md = S.trot_schedule(N * interface.problem.settings.dt + 1.0, phase0=0.03)
    mb = G.MpcBatch(x0, tt, ts, np.full(B, nev, dtype=np.int32), np.tile(ev, (B, 1)), np.tile(md, (B, 1)), N)
    return mb, (x0, tt, ts, nev, ev, md)


def test_bad_arguments_are_reported_not_executed(interface, oracle):
    import gpu_harness as G
    from qm_door_amd import abi
    sol = G.make_solver(interface, 4, 8)
    mb, _ = _batch(G, interface, oracle, 2, 8)
    lib = sol.lib

    def status(args):
        return lib.qmgpu_mpc_solve_batch(sol.handle, C.byref(args))

    assert status(mb.args) == 0
    for field, value, code in (("batch", 0, 1), ("batch", 5, 7), ("num_nodes", 0, 1), ("num_nodes", 9, 7), ("num_target_knots", 0, 1),
                               ("x0", None, 1), ("out_x", None, 1), ("sched_modes", None, 1), ("target_states", None, 1)):
        a = _clone(mb.args)
        setattr(a, field, value)
        st = status(a)
        assert st == code, (field, st)
        assert lib.qmgpu_strerror(st) and lib.qmgpu_last_error()
    a = _clone(mb.args)
    a.t0 = None; a.time_grid = None
    assert status(a) == 1                                            # neither t0 nor a grid
    assert lib.qmgpu_mpc_solve_batch(None, C.byref(mb.args)) == 1     # null handle
    with pytest.raises(abi.QmGpuError):
        sol.mpc(a)                                                    # the Python mirror raises, as the reference throws
    # WBC: capacity and missing pointers
    B = 2
    rbd = np.zeros((B, 55)); rbd[:, 6:24] = interface.initial_state[12:30]; rbd[:, 2] = interface.initial_state[8]
    wb = G.WbcBatch(rbd, np.full(B, 0.002), np.full(B, 20.0), np.zeros((B, 30)), np.tile(interface.initial_state, (B, 1)), np.zeros((B, 30)), np.full(B, 15, dtype=np.int32))
    assert lib.qmgpu_wbc_solve_batch(sol.handle, C.byref(wb.args)) == 0
    for field, value, code in (("batch", 0, 1), ("batch", 5, 7), ("out", None, 1), ("rbd_measured", None, 1)):
        w = _clone(wb.args)
        setattr(w, field, value)
        assert lib.qmgpu_wbc_solve_batch(sol.handle, C.byref(w)) == code, field
    # the failed calls left nothing behind: the good call still gives the same answer
    r0 = mb.results()
    assert status(mb.args) == 0
    r1 = mb.results()
    assert np.array_equal(r0["X"], r1["X"]) and np.array_equal(r0["U"], r1["U"])


@pytest.mark.parametrize("B,N,K", [(1, 1, 1), (1, 2, 3), (3, 5, 2)])
def test_smallest_horizons_and_partial_batches(interface, oracle, B, N, K):
    """One- and two-node horizons, several target knots, a batch smaller than the handle's capacity."""
    import gpu_harness as G
    sol = G.make_solver(interface, 8, 16)
    mb, (x0, tt, ts, nev, ev, md) = _batch(G, interface, oracle, B, N, seed=6, K=K)
    sol.mpc(mb.args)
    r = mb.results()
    for i in range(B):
        ref = oracle.mpc_solve(N, 0.0, x0[i], tt[i], ts[i], nev, ev, md)
        assert np.array_equal(r["mode"][i], ref["mode"])
        assert np.abs(r["X"][i] - ref["X"]).max() <= 1e-6 * max(1.0, np.abs(ref["X"]).max())
        assert np.abs(r["U"][i] - ref["U"]).max() <= 1e-6 * max(1.0, np.abs(ref["U"]).max())


def test_maximum_number_of_mode_switches(interface, oracle):
    """QMGPU_MAX_EVENTS switches inside the horizon (every node in a different phase of a fast trot)."""
    import gpu_harness as G
    from qm_door_amd import abi
    B, N = 2, 44
    dt = interface.problem.settings.dt
    nev = abi.MAX_EVENTS
    ev = (np.arange(nev) + 0.5) * dt * 1.05
    md = np.array([15 if k % 4 == 0 else (9 if k % 4 == 1 else (15 if k % 4 == 2 else 6)) for k in range(nev + 1)], dtype=np.int32)
    x0 = S.perturbed_states(interface.initial_state, B, seed=7)
    tgt = S.nominal_target(oracle, interface.initial_state)
    tt = np.zeros((B, 1)); ts = np.tile(tgt, (B, 1, 1)).copy()
    sol = G.make_solver(interface, B, N)
    mb = G.MpcBatch(x0, tt, ts, np.full(B, nev, dtype=np.int32), np.tile(ev, (B, 1)), np.tile(md, (B, 1)), N)
    sol.mpc(mb.args)
    r = mb.results()
    ref = oracle.mpc_solve(N, 0.0, x0[0], tt[0], ts[0], nev, ev, md)
    assert np.array_equal(r["mode"][0], ref["mode"]) and len(set(ref["mode"].tolist())) == 3
    assert np.abs(r["X"][0] - ref["X"]).max() <= 1e-6 * max(1.0, np.abs(ref["X"]).max())
    assert np.abs(r["U"][0] - ref["U"]).max() <= 1e-6 * max(1.0, np.abs(ref["U"]).max())


def test_relaxed_barrier_quadratic_branches_and_distinct_target_knots(interface, oracle):
    """The edge cases of the reference's soft constraints (QMInterface.cpp:177-259, 344-358) that nominal scenarios never reach, on the GPU:
      * friction cone in the quadratic extension of the relaxed barrier (h = mu f_z - sqrt(f_x^2 + f_y^2 + reg) <= delta = 5: a stance foot that carries
        a few newtons, and one pulling on the ground);
      * arm joint positions within delta = 1e-3 of a URDF limit, exactly ON it and beyond it; arm joint velocities beyond their bounds;
      * three DIFFERENT target knots (position lerp + a genuine quaternion slerp) straight through qmgpu_mpc_solve_batch.
    The iterate comes in as a warm start, so the LQ blocks are formed exactly there; blocks and the resulting step are compared with the oracle."""
    import gpu_harness as G
    B, N = 4, 12
    dt = interface.problem.settings.dt
    md_ = interface.problem.model
    x_nom, m = interface.initial_state, interface.robot_mass
    x0 = S.perturbed_states(x_nom, B, seed=9)
    up = np.array([md_.q_upper[12 + i] for i in range(6)]); lo = np.array([md_.q_lower[12 + i] for i in range(6)])
    x0[0, 24] = up[0] - 5e-4; x0[0, 25] = lo[1] + 2e-4           # inside the delta band
    x0[1, 26] = up[2]; x0[1, 27] = lo[3]                         # exactly on the limits
    x0[2, 24] = up[0] + 0.02; x0[2, 28] = lo[4] - 0.01           # beyond
    tgt = S.nominal_target(oracle, x_nom)
    # three distinct knots: base moves and yaws, the EE target moves and rotates about two different axes
    K = 3
    tt = np.tile(np.array([0.0, 0.08, 0.2]), (B, 1))
    ts = np.tile(tgt, (B, K, 1)).copy()
    ts[:, 1, 6] += 0.05; ts[:, 2, 6] += 0.12; ts[:, 2, 9] += 0.2
    ts[:, 1, 30:33] += [0.03, -0.02, 0.04]; ts[:, 2, 30:33] += [0.08, 0.05, -0.03]
    def quat(axis, ang):
        a = np.asarray(axis, float); a /= np.linalg.norm(a)
        return np.r_[a * np.sin(ang / 2), np.cos(ang / 2)]
    ts[:, 1, 33:37] = quat([0, 1, 0.2], 0.5); ts[:, 2, 33:37] = quat([1, 0.3, 0], -0.8)
    nev, ev, md = S.trot_schedule(N * dt + 1.0, phase0=0.05)
    # warm start: states = x0 held, inputs with starved / pulling stance feet and arm rates beyond their bounds
    X = np.repeat(x0[:, None, :], N + 1, axis=1)
    U = np.zeros((B, N, 30))
    for i in range(B):
        for k in range(N):
            mode = oracle.node_mode_at(ev[:nev], md[:nev + 1], k * dt)
            flags = [(mode >> (3 - c)) & 1 for c in range(4)]
            st = [c for c in range(4) if flags[c]]
            for c in st:
                U[i, k, 3 * c + 2] = m * 9.81 / len(st)
            U[i, k, 3 * st[0] + 2] = 3.0 + i                    # h = 0.7 * 3 - 5 < delta: quadratic branch
            U[i, k, 3 * st[0]] = 4.0
            if i == 3:
                U[i, k, 3 * st[-1] + 2] = -6.0                  # pulling on the ground: h < 0
    al = np.array([interface.problem.settings.arm_vel_lower[i] for i in range(6)]); au = np.array([interface.problem.settings.arm_vel_upper[i] for i in range(6)])
    U[0, :, 24] = au[0] + 0.05; U[1, :, 26] = al[2] - 0.2; U[2, :, 29] = au[5] - 4e-4
    sol = G.make_solver(interface, B, N)
    sol.enable_debug(True)
    mb = G.MpcBatch(x0, tt, ts, np.full(B, nev, dtype=np.int32), np.tile(ev, (B, 1)), np.tile(md, (B, 1)), N, warm=(X, U))
    sol.mpc(mb.args)
    r = mb.results()
    assert np.isfinite(r["X"]).all() and (r["stats"][:, 7] == 0).all()
    for i in range(B):
        for k in (0, 5, 9):
            g = sol.debug_lq(i, k)
            o = oracle.lq_node(k * dt, dt, X[i, k], U[i, k], X[i, k + 1], False, nev, ev, md, tt[i], ts[i])
            for key in ("A", "B", "b", "Q", "R", "q", "r", "C", "D", "e"):
                assert np.abs(g[key] - o[key]).max() <= 1e-10 * max(1.0, np.abs(o[key]).max()), (i, k, key)
        g = sol.debug_lq(i, N)
        o = oracle.lq_node(N * dt, 0.0, X[i, N], None, X[i, N], True, nev, ev, md, tt[i], ts[i])
        assert np.abs(g["Q"] - o["Q"]).max() <= 1e-10 * max(1.0, np.abs(o["Q"]).max()) and np.abs(g["q"] - o["q"]).max() <= 1e-10 * max(1.0, np.abs(o["q"]).max())
        ref = oracle.mpc_solve(N, 0.0, x0[i], tt[i], ts[i], nev, ev, md, warm=(X[i], U[i]))
        assert np.array_equal(r["mode"][i], ref["mode"]) and r["stats"][i][4] == ref["stats"][4]
        assert np.abs(r["X"][i] - ref["X"]).max() <= 1e-6 * max(1.0, np.abs(ref["X"]).max())
        assert np.abs(r["U"][i] - ref["U"]).max() <= 1e-6 * max(1.0, np.abs(ref["U"]).max())
    # the branches really were the quadratic ones: the barrier's second derivative equals mu / delta^2 there
    st_ = interface.problem.settings
    o = oracle.lq_node(0.0, dt, X[2, 0], U[2, 0], X[2, 1], False, nev, ev, md, tt[2], ts[2])
    curv = st_.joint_pos_barrier_mu / st_.joint_pos_barrier_delta ** 2          # 1e5; the end-effector Gauss-Newton term adds a few hundred on top
    assert curv <= o["Q"][24, 24] / dt <= curv + 2e3


def test_horizon_beyond_the_line_search_lds_budget(interface, oracle):
    """N = 300: the trial trajectories of the line search no longer fit its dynamic LDS (linesearch_kernel.h: lsTrialLdsBytes) and go through
    the HBM scratch instead; the Riccati sweeps walk 300 stages.  fp64 against the oracle, and the fp32 build against fp64."""
    import gpu_harness as G
    B, N = 2, 300
    mb, (x0, tt, ts, nev, ev, md) = _batch(G, interface, oracle, B, N, seed=8)
    sol = G.make_solver(interface, B, N)
    sol.mpc(mb.args)
    r = mb.results()
    assert (r["stats"][:, 7] == 0).all()
    for i in range(B):
        ref = oracle.mpc_solve(N, 0.0, x0[i], tt[i], ts[i], nev, ev, md)
        assert np.array_equal(r["mode"][i], ref["mode"])
        assert np.abs(r["X"][i] - ref["X"]).max() <= 1e-6 * max(1.0, np.abs(ref["X"]).max())
        assert np.abs(r["U"][i] - ref["U"]).max() <= 1e-6 * max(1.0, np.abs(ref["U"]).max())
        assert r["stats"][i][4] == ref["stats"][4]
    sol32 = G.make_solver(interface, B, N, dtype="f32")
    mb32, _ = _batch(G, interface, oracle, B, N, seed=8)
    sol32.mpc(mb32.args)
    r32 = mb32.results()
    assert np.array_equal(r32["mode"], r["mode"]) and (r32["stats"][:, 4] == r["stats"][:, 4]).all()
    assert np.abs(r32["X"] - r["X"]).max() <= 1e-4 * max(1.0, np.abs(r["X"]).max())
    assert np.abs(r32["U"] - r["U"]).max() <= 1e-4 * max(1.0, np.abs(r["U"]).max())


def test_barrier_constants_follow_a_settings_update(interface, oracle):
    """The relaxed-barrier constants that input_weight_kernel derives once per settings update (log delta, value(-lower) + value(upper) per arm joint; layout.h:
    QM_RW_DERIVED) must be re-derived by qmgpu_update_settings: the next MPC solve is the oracle's solve WITH the new barrier parameters, and differs from the old."""
    import gpu_harness as G
    from qm_door_amd import abi
    B, N = 2, 8
    sol = G.make_solver(interface, B, N)
    mb, (x0, tt, ts, nev, ev, md) = _batch(G, interface, oracle, B, N, seed=11)
    sol.mpc(mb.args)
    before = {k: v.copy() for k, v in mb.results().items()}
    P2 = type(interface.problem).from_buffer_copy(interface.problem)
    P2.settings.joint_vel_barrier_delta *= 3.0; P2.settings.joint_pos_barrier_mu *= 4.0
    P2.settings.friction_barrier_delta *= 0.5; P2.settings.friction_barrier_mu *= 2.0
    abi.check(interface.lib, interface.lib.qmgpu_update_settings(sol.handle, C.byref(P2.settings)))
    sol.mpc(mb.args)
    after = mb.results()
    assert np.abs(after["U"] - before["U"]).max() > 1e-6
    o2 = S.Oracle(P2)
    for i in range(B):
        ref = o2.mpc_solve(N, 0.0, x0[i], tt[i], ts[i], nev, ev, md)
        assert np.array_equal(after["mode"][i], ref["mode"])
        assert np.abs(after["X"][i] - ref["X"]).max() <= 1e-6 * max(1.0, np.abs(ref["X"]).max())
        assert np.abs(after["U"][i] - ref["U"]).max() <= 1e-6 * max(1.0, np.abs(ref["U"]).max())
    abi.check(interface.lib, interface.lib.qmgpu_update_settings(sol.handle, C.byref(interface.problem.settings)))   # (handles are per test, but leave it as found)


def test_policy_evaluation_and_warm_start_over_long_grids(interface, oracle):
    """qmgpu_policy_eval_batch / qmgpu_warm_start_batch on a 150-node grid holding two events' pairs of equal times (their grid intervals come from ballots
    over the grid, schedule_dev.h: timeSegmentWave): evaluation times before the first node, on nodes, far into the horizon, past the end; new grids that
    start later and end past the old horizon.  Against the oracle: planned modes exact, interpolated values to 1e-12 (fused multiply-adds differ between the builds)."""
    import torch
    import gpu_harness as G
    rng = np.random.default_rng(17)
    N = 150
    steps = rng.uniform(0.005, 0.02, N); steps[40] = 0.0; steps[97] = 0.0
    grid = 0.3 + np.r_[0.0, np.cumsum(steps)]
    te = np.r_[grid[0] - 0.01, grid[0], grid[1], grid[40], grid[41], grid[64], grid[65], grid[128], grid[-1], grid[-1] + 0.02, rng.uniform(grid[0], grid[-1], 54)]
    B = len(te)
    T = np.tile(grid, (B, 1)); X = rng.normal(size=(B, N + 1, 30)); U = rng.normal(size=(B, N, 30))
    M = rng.integers(0, 16, (B, N + 1)).astype(np.int32)
    sol = G.make_solver(interface, B, N)
    f64 = torch.float64
    dT, dX, dU, dM = G.dev(T, f64), G.dev(X, f64), G.dev(U, f64), G.dev(M, torch.int32)
    xo, uo, mo = G.dev(np.zeros((B, 30)), f64), G.dev(np.zeros((B, 30)), f64), G.dev(np.zeros(B), torch.int32)
    sol.policy_eval(B, N, dT, dX, dU, dM, G.dev(te, f64), xo, uo, mo)
    xr, ur, mr = oracle.policy_eval_batch(T, X, U, M, te)
    assert np.abs(xo.cpu().numpy() - xr).max() <= 1e-12 and np.abs(uo.cpu().numpy() - ur).max() <= 1e-12 and np.array_equal(mo.cpu().numpy(), mr)
    Nn = 141
    gn = te[:, None] * 0.0 + grid[0] + rng.uniform(0.0, 0.05, (B, 1)) + np.cumsum(np.concatenate([np.zeros((B, 1)), rng.uniform(0.008, 0.02, (B, Nn)) * 1.3], axis=1), axis=1)
    x0 = rng.normal(size=(B, 30))
    wx, wu = G.dev(np.zeros((B, Nn + 1, 30)), f64), G.dev(np.zeros((B, Nn, 30)), f64)
    sol.warm_start(B, N, dT, dX, dU, Nn, G.dev(gn, f64), G.dev(x0, f64), wx, wu)
    rx, ru = oracle.warm_start_batch(T, X, U, gn, x0)
    assert np.abs(wx.cpu().numpy() - rx).max() <= 1e-12 and np.abs(wu.cpu().numpy() - ru).max() <= 1e-12


def test_a_failed_factorisation_flags_the_instance_and_leaves_the_iterate(interface, oracle):
    """H = R~ + B~' S B~ that is not positive definite -- here: the input weights negated by a settings update -- fails the Cholesky of the backward sweep.  The reference's
    HPIPM call would return an error and ocs2_sqp would throw; the library flags the instance (stats[7] = 1), leaves its iterate where it was (for a cold start: the initial
    guess, exactly what the oracle returns) and goes on: riccati_kernel runs its elimination on with the failed pivot floored (round 6: one fmax on the dependent chain instead
    of a select), so whatever it wrote into its own buffers must neither reach X / U nor survive into the next solve."""
    import gpu_harness as G
    from qm_door_amd import abi
    B, N = 3, 8
    sol = G.make_solver(interface, B, N)
    mb, (x0, tt, ts, nev, ev, md) = _batch(G, interface, oracle, B, N, seed=23)
    sol.mpc(mb.args)
    good = {k: v.copy() for k, v in mb.results().items()}
    assert (good["stats"][:, 7] == 0).all()
    P2 = type(interface.problem).from_buffer_copy(interface.problem)
    for k in range(900):
        P2.settings.R_task[k] = -P2.settings.R_task[k]
    abi.check(interface.lib, interface.lib.qmgpu_update_settings(sol.handle, C.byref(P2.settings)))
    sol.mpc(mb.args)
    bad = {k: v.copy() for k, v in mb.results().items()}
    assert (bad["stats"][:, 7] == 1).all(), bad["stats"][:, 7]
    assert np.isfinite(bad["X"]).all() and np.isfinite(bad["U"]).all()
    o2 = S.Oracle(P2)
    for i in range(B):
        ref = o2.mpc_solve(N, 0.0, x0[i], tt[i], ts[i], nev, ev, md)
        assert ref["status"] != 0 or ref["stats"][7] != 0
        assert np.array_equal(bad["mode"][i], ref["mode"])
        assert np.abs(bad["X"][i] - ref["X"]).max() <= 1e-12 * max(1.0, np.abs(ref["X"]).max())
        assert np.abs(bad["U"][i] - ref["U"]).max() <= 1e-12 * max(1.0, np.abs(ref["U"]).max())
    abi.check(interface.lib, interface.lib.qmgpu_update_settings(sol.handle, C.byref(interface.problem.settings)))
    sol.mpc(mb.args)
    again = mb.results()
    assert np.array_equal(again["X"], good["X"]) and np.array_equal(again["U"], good["U"]) and (again["stats"][:, 7] == 0).all()
