"""CPU tier: the SAME kernel sources compiled against tests/emu/simt_emu.h (workgroups as host threads) are checked against the
oracle.  This validates the kernels' lane roles, LDS hand-offs and index math without a GPU; the -m gpu tests repeat the
comparison on the real hipcc build.  The emulation library is test infrastructure and is never loaded by qm_door_amd."""
import ctypes as C

import os
import numpy as np
import pytest

import support as S
from qm_door_amd import abi, api


@pytest.fixture(scope="module")
def emu():
    lib = abi.load_library(S.build_emu())
    itf = api.QMInterface(lib=lib)
    return itf, S.Oracle(itf.problem)


def test_emu_lq_and_sqp_iteration(emu):
    itf, orc = emu
    B, N = 2, 6
    x_nom = itf.initial_state
    x0 = S.perturbed_states(x_nom, B, seed=0)
    tgt = S.nominal_target(orc, x_nom)
    tt = np.zeros((B, 1)); ts = np.tile(tgt, (B, 1, 1)).copy()
    nev, ev, md = S.trot_schedule(2.0, phase0=0.03)
    sol = api.GpuSolver(itf, max_batch=B, max_nodes=N)
    assert np.abs(sol.input_weight() - orc.input_weight()).max() < 1e-13
    sol.enable_debug(True)
    oT, oX, oU, oM, oS = np.zeros((B, N + 1)), np.zeros((B, N + 1, 30)), np.zeros((B, N, 30)), np.zeros((B, N + 1), dtype=np.int32), np.zeros((B, abi.NSTATS))
    a = sol.mpc_args(B, N, x0, tt, ts, np.full(B, nev, dtype=np.int32), np.tile(ev, (B, 1)).copy(), np.tile(md, (B, 1)).copy(), oT, oX, oU, oM, oS, t0=np.zeros(B))
    sol.mpc(a)
    dt = itf.problem.settings.dt
    for inst in range(B):
        for k in (0, 3, N):
            g = sol.debug_lq(inst, k)
            mode = orc.node_mode_at(ev[:nev], md[:nev + 1], k * dt)
            flags = [(mode >> (3 - c)) & 1 for c in range(4)]
            u = np.zeros(30)
            for c in range(4):
                if flags[c]:
                    u[3 * c + 2] = itf.robot_mass * 9.81 / sum(flags)
            o = orc.lq_node(k * dt, dt if k < N else 0.0, x0[inst], u if k < N else None, x0[inst], k == N, nev, ev, md, tt[inst], ts[inst])
            assert g["nc"] == o["nc"]
            for key in (["Q", "q"] if k == N else ["A", "B", "b", "Q", "R", "q", "r", "C", "D", "e"]):
                assert np.abs(g[key] - o[key]).max() <= 1e-10 * max(1.0, np.abs(o[key]).max()), (inst, k, key)
        ref = orc.mpc_solve(N, 0.0, x0[inst], tt[inst], ts[inst], nev, ev, md)
        assert np.array_equal(oM[inst], ref["mode"])
        assert np.abs(oX[inst] - ref["X"]).max() <= 1e-8 * max(1.0, np.abs(ref["X"]).max())
        assert np.abs(oU[inst] - ref["U"]).max() <= 1e-8 * max(1.0, np.abs(ref["U"]).max())
        assert np.allclose(oS[inst][:7], ref["stats"][:7], rtol=1e-8, atol=1e-10)


def test_emu_line_search_launch_shape_for_batches_beyond_the_cus(emu):
    """linesearch_kernel runs 128 threads per instance when the batch exceeds the device's CUs (lsThreads, linesearch_kernel.h): same results, bit for bit,
    as the 256-thread launch -- iterate, merit, violation, alpha, step type.  The emulated device's CU count is QMGPU_EMU_CUS (simt_emu.h)."""
    itf, orc = emu
    B, N = 3, 7
    x0 = S.perturbed_states(itf.initial_state, B, seed=11)
    tgt = S.nominal_target(orc, itf.initial_state)
    tt = np.zeros((B, 1)); ts = np.tile(tgt, (B, 1, 1)).copy()
    nev, ev, md = S.trot_schedule(2.0, phase0=0.03)
    res = {}
    for cus in ("256", "2"):
        os.environ["QMGPU_EMU_CUS"] = cus
        try:
            sol = api.GpuSolver(itf, max_batch=B, max_nodes=N)
        finally:
            del os.environ["QMGPU_EMU_CUS"]
        oT, oX, oU, oM, oS = np.zeros((B, N + 1)), np.zeros((B, N + 1, 30)), np.zeros((B, N, 30)), np.zeros((B, N + 1), dtype=np.int32), np.zeros((B, abi.NSTATS))
        a = sol.mpc_args(B, N, x0, tt, ts, np.full(B, nev, dtype=np.int32), np.tile(ev, (B, 1)).copy(), np.tile(md, (B, 1)).copy(), oT, oX, oU, oM, oS, t0=np.zeros(B))
        sol.mpc(a)
        res[cus] = (oX.copy(), oU.copy(), oS.copy(), oM.copy())
    for a_, b_ in zip(res["256"], res["2"]):
        assert np.array_equal(a_, b_)
    assert np.isfinite(res["2"][0]).all() and (res["2"][2][:, 4] > 0).all()   # a step was taken in every instance


def test_emu_wbc(emu):
    itf, orc = emu
    rng = np.random.default_rng(3)
    x_nom, m = itf.initial_state, itf.robot_mass
    cases = []
    for mode, t in ((9, 20.0), (15, 5.0)):
        flags = [(mode >> (3 - c)) & 1 for c in range(4)]
        u = np.zeros(30)
        for c in range(4):
            if flags[c]:
                u[3 * c + 2] = m * 9.81 / sum(flags)
        u[12:] = rng.uniform(-1, 1, 18) * 0.05
        xd = x_nom + rng.uniform(-1, 1, 30) * 0.02
        rbd = S.rbd_from_state(orc, x_nom + rng.uniform(-1, 1, 30) * 0.01, rng.uniform(-1, 1, 24) * 0.05)
        cases.append((xd, u, rbd, mode, t, u + rng.uniform(-1, 1, 30) * 0.001))
    B = len(cases)
    sol = api.GpuSolver(itf, max_batch=B, max_nodes=4)
    out, st = np.zeros((B, 54)), np.zeros(B, dtype=np.int32)
    il = np.array([c[5] for c in cases])
    a = sol.wbc_args(B, np.array([c[2] for c in cases]), np.full(B, 0.002), np.array([c[4] for c in cases]), il, out, st, np.array([c[0] for c in cases]),
                     np.array([c[1] for c in cases]), np.array([c[3] for c in cases], dtype=np.int32), 0)
    sol.wbc(a)
    assert (st == 0).all()
    for i, (xd, u, rbd, mode, t, il0) in enumerate(cases):
        s, ref, il_ref = orc.wbc_update(xd, u, rbd, mode, 0.002, t, il0)
        assert s == 0
        assert np.abs(out[i] - ref).max() <= 1e-6 * max(1.0, np.abs(ref).max())
        assert np.array_equal(il[i], il_ref)


def test_emu_wbc_on_the_degenerate_stance_tick(emu):
    """tests/golden/wbc_degenerate_stance_tick.npz (test_oracle_invariants.py says what it is): the lowest level inherits a cone without interior; both implementations
    remove its strongly active rows as equalities and return THE answer -- one, not one of two."""
    itf, orc = emu
    c = np.load(os.path.join(S.ROOT, "tests", "golden", "wbc_degenerate_stance_tick.npz"))
    sol = api.GpuSolver(itf, max_batch=1, max_nodes=4)
    out, st = np.zeros((1, 54)), np.zeros(1, dtype=np.int32)
    il = c["il"][None].copy()
    a = sol.wbc_args(1, c["rbd"][None], np.full(1, 0.001), np.array([float(c["t"])]), il, out, st, c["xd"][None], c["ud"][None], np.array([int(c["mode"])], dtype=np.int32), 0)
    sol.wbc(a)
    s, ref, il_ref = orc.wbc_update(c["xd"], c["ud"], c["rbd"], int(c["mode"]), 0.001, float(c["t"]), c["il"].copy())
    assert s == 0 and st[0] == 0
    assert np.abs(out[0, 36:] - ref[36:]).max() <= 1e-9 * max(1.0, np.abs(ref[36:]).max())
    assert np.abs(out[0, :36] - ref[:36]).max() <= 1e-9 * max(1.0, np.abs(ref[:36]).max())


def test_emu_wbc_with_the_working_sets_carried_from_tick_to_tick(emu):
    """qmgpu_wbc_args::working_set: the rows every level of the hierarchical QP ended on and the point it ended at travel from tick to tick next to inputLast_ and are
    the next tick's starting guess (the reference cold-starts qpOASES every tick, HoQp.cpp:136-149).  One robot through two MPC cycles x three 1 kHz ticks on the host-
    emulated kernel and on the CPU restatement, each carrying its own record: same torques on every tick, and the same torques as the cold solve of the same tick -- the
    vertex does not depend on the path."""
    import closed_loop as CL
    itf, orc = emu
    sc = CL.Scenario(itf, 1, cycles=2, gait_start=0.02, t_start=10.2)
    a = CL.OracleBackend(orc, sc, 0, carry=True)
    sol = api.GpuSolver(itf, max_batch=1, max_nodes=4)
    ws = np.zeros((1, abi.WBC_STATE_WORDS), dtype=np.uint64)
    rbd = sc.first_measurement()
    warm_ticks = 0
    for k in range(sc.cycles):
        t0 = sc.t_start + k * CL.MPC_PERIOD
        N, grid = sc.grid(t0)
        a.observe(rbd, t0)
        plan = a.mpc(t0, N, grid)
        for j in range(3):
            t = t0 + j * CL.WBC_PERIOD
            if not (k == 0 and j == 0):
                rbd = CL.measurement(sc, plan, t)
            w = a.tick(t, rbd, t)
            xd, ud, rb, md, tm, il = a.last
            out, st, il_e = np.zeros((1, 54)), np.zeros(1, dtype=np.int32), il.copy()
            sol.wbc(sol.wbc_args(1, rb, np.full(1, CL.WBC_PERIOD), np.full(1, tm), il_e, out, st, xd, ud, md.astype(np.int32), 0, working_set=ws))
            assert st[0] == 0 and w["status"][0] == 0
            assert S.rel_inf(out[:, 36:], w["out"][:, 36:]).max() <= 1e-9 and S.rel_inf(out[:, :36], w["out"][:, :36]).max() <= 1e-9
            assert ws[0, 0] >> np.uint64(63) == 1 and ws[0, 0] == a.ws[0, 0]                 # both records carry this tick's key (contact mode, controller, task set)
            assert (ws[0, 3] >> np.uint64(63)) == 1 and (a.ws[0, 3] >> np.uint64(63)) == 1     # the second level's word is valid on both sides
            if not (k == 0 and j == 0):
                warm_ticks += 1
                cold = orc.wbc_update(xd[0], ud[0], rb[0], int(md[0]), CL.WBC_PERIOD, tm, il[0].copy())[1]
                assert np.abs(out[0] - cold).max() <= 1e-9 * max(1.0, np.abs(cold).max())
        rbd = CL.measurement(sc, plan, t0 + CL.MPC_PERIOD)
    assert warm_ticks == 5


def test_emu_first_level_of_a_diverged_robot_goes_through_the_interior_point(emu):
    """One of the slowest ticks of round 6's steady-state leg (tests/golden/wbc_slow_ticks.npz, index 1: 40 working-set changes of the 36-variable level from z = 0 until the
    end of round 6) on the host-emulated kernel: held-variable form given up after QP_HELD_CAP iterations, the interior point with the own rows as penalised slacks,
    the active-set method behind it -- same torques as the CPU restatement on the same path AND on its cold path, a bounded number of passes."""
    import os
    itf, orc = emu
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "wbc_slow_ticks.npz"))
    i = 1
    sol = api.GpuSolver(itf, max_batch=1, max_nodes=4)
    out, st = np.zeros((1, 54)), np.zeros(1, dtype=np.int32)
    ws = np.zeros((1, abi.WBC_STATE_WORDS), dtype=np.uint64)
    sol.wbc(sol.wbc_args(1, d["rbd"][i][None], np.array([float(d["period"][i])]), np.array([float(d["time"][i])]), d["il"][i][None].copy(), out, st, d["xd"][i][None], d["ud"][i][None],
                         np.array([int(d["mode"][i])], dtype=np.int32), 0, working_set=ws))
    a = (d["xd"][i], d["ud"][i], d["rbd"][i], int(d["mode"][i]), float(d["period"][i]), float(d["time"][i]))
    s_now, now, _ = orc.wbc_update(*a, d["il"][i].copy())
    try:
        orc.set_experiment(own_interior_point=0)
        s_cold, cold, _ = orc.wbc_update(*a, d["il"][i].copy())
    finally:
        orc.set_experiment()
    assert st[0] == 0 and s_now == 0 and s_cold == 0
    for ref in (now, cold):
        assert all(v.max() <= 1e-9 for v in S.rel_inf_blocks(out, ref[None]).values())       # measured 2e-15 / 4e-14
    assert (int(np.ascontiguousarray(ws[0, 13:14]).view(np.uint8)[0]) & 127) <= 12              # measured 8 (interior-point iterations + working-set changes)


def test_emu_mixed_modes_and_event_grid(emu):
    """Flight / three-leg / stance nodes (m~ = 14, 17, 18 tile paths of the MFMA kernels) on an event-aligned, non-uniform grid."""
    itf, orc = emu
    from qm_door_amd import abi as _abi
    gs = api.GaitSchedule(lib=itf.lib)
    ev, md, t = [], [15], 0.06
    for name in ("flying_trot", "static_walk"):
        g = gs.template(name)
        for i in range(g.num_modes):
            m = int(g.modes[i]); d = 0.25 * (g.switching_times[i + 1] - g.switching_times[i])     # compressed so that 24 nodes see every mode
            if m != md[-1]:
                ev.append(t); md.append(m)
            t += d
    ev.append(t); md.append(15)
    nev = len(ev)
    evp = np.full(_abi.MAX_EVENTS, 1e300); evp[:nev] = ev
    mdp = np.full(_abi.MAX_EVENTS + 1, 15, dtype=np.int32); mdp[:len(md)] = md
    N, grid = api.time_grid_with_events(0.0, 0.36, itf.problem.settings.dt, ev, max_nodes=64, lib=itf.lib)
    B = 1
    x0 = S.perturbed_states(itf.initial_state, B, seed=8)
    tgt = S.nominal_target(orc, itf.initial_state)
    tt = np.zeros((B, 1)); ts = np.tile(tgt, (B, 1, 1)).copy()
    sol = api.GpuSolver(itf, max_batch=B, max_nodes=N)
    oT, oX, oU, oM, oS = np.zeros((B, N + 1)), np.zeros((B, N + 1, 30)), np.zeros((B, N, 30)), np.zeros((B, N + 1), dtype=np.int32), np.zeros((B, abi.NSTATS))
    a = sol.mpc_args(B, N, x0, tt, ts, np.full(B, nev, dtype=np.int32), evp[None, :].copy(), mdp[None, :].copy(), oT, oX, oU, oM, oS, time_grid=grid[None, :].copy())
    sol.mpc(a)
    ref = orc.mpc_solve(N, 0.0, x0[0], tt[0], ts[0], nev, evp, mdp, time_grid=grid)
    assert {0, 15}.issubset(set(oM[0].tolist())) and len(set(oM[0].tolist())) >= 4 and (np.diff(grid) < 0.9 * itf.problem.settings.dt).any()
    assert np.array_equal(oM[0], ref["mode"]) and np.array_equal(oT[0], grid)
    assert np.abs(oX[0] - ref["X"]).max() <= 1e-8 * max(1.0, np.abs(ref["X"]).max())
    assert np.abs(oU[0] - ref["U"]).max() <= 1e-8 * max(1.0, np.abs(ref["U"]).max())


def test_emu_frontend(emu):
    itf, orc = emu
    import test_frontend as TF
    cs = TF._cases(itf, orc, 8, seed=21)
    B = len(cs)
    rbd = np.array([c["rbd"] for c in cs]); tm = np.array([c["time"] for c in cs]); yl = np.array([c["yaw_last"] for c in cs])
    kd = np.array([c["kind"] for c in cs], dtype=np.int32); cmd = np.array([c["cmd"] for c in cs]); le = np.array([c["last_ee"] for c in cs]); fh = np.array([c["feet"] for c in cs])
    x0, tt, ts = np.zeros((B, 30)), np.zeros((B, 2)), np.zeros((B, 2, 37))
    sol = api.GpuSolver(itf, max_batch=B, max_nodes=4)
    sol.frontend(sol.frontend_args(B, rbd, tm, kd, cmd, le, x0, tt, ts, yaw_last=yl, feet_height=fh))
    for i, c in enumerate(cs):
        rx, rt, rs, rl = orc.frontend(c["rbd"], c["time"], c["kind"], c["cmd"], c["last_ee"], yaw_last=c["yaw_last"], feet_height=c["feet"])
        assert np.abs(x0[i] - rx).max() <= 1e-12 * max(1.0, np.abs(rx).max()) and np.abs(ts[i] - rs).max() <= 1e-12 * max(1.0, np.abs(rs).max())
        assert np.abs(tt[i] - rt).max() <= 1e-12 * max(1.0, np.abs(rt).max()) and np.abs(le[i] - rl).max() <= 1e-15


def test_emu_sqp_convergence_test_skips_converged_instances():
    """sqp.sqpIteration = 5 with upstream's convergence test: one instance starts at the optimum's doorstep (warm start = a solved
    trajectory) and stops on the primal-step criterion, the other runs on; statistics slots 8 / 9 carry the count and the reason."""
    lib = abi.load_library(S.build_emu())
    itf = api.QMInterface(lib=lib)
    itf.problem.settings.sqp_iterations = 5
    itf.problem.settings.delta_tol = 5.0
    orc = S.Oracle(itf.problem)
    B, N = 2, 8
    x0 = S.perturbed_states(itf.initial_state, B, seed=9)
    tgt = S.nominal_target(orc, itf.initial_state)
    tt = np.zeros((B, 1)); ts = np.tile(tgt, (B, 1, 1)).copy()
    nev, ev, md = S.trot_schedule(2.0, phase0=0.03)
    # warm start: instance 0 from an already converged trajectory, instance 1 cold (x_k = x0, weight compensation)
    solved = orc.mpc_solve(N, 0.0, x0[0], tt[0], ts[0], nev, ev, md)
    cold = orc.mpc_solve(N, 0.0, x0[1], tt[1], ts[1], nev, ev, md)
    wx = np.stack([solved["X"], np.tile(x0[1], (N + 1, 1))]); wu = np.stack([solved["U"], cold["U"] * 0.0])
    for k in range(N):
        mode = orc.node_mode_at(ev[:nev], md[:nev + 1], k * itf.problem.settings.dt)
        flags = [(mode >> (3 - c)) & 1 for c in range(4)]
        for c in range(4):
            if flags[c]:
                wu[1, k, 3 * c + 2] = itf.robot_mass * 9.81 / sum(flags)
    sol = api.GpuSolver(itf, max_batch=B, max_nodes=N)
    oT, oX, oU, oM, oS = np.zeros((B, N + 1)), np.zeros((B, N + 1, 30)), np.zeros((B, N, 30)), np.zeros((B, N + 1), dtype=np.int32), np.zeros((B, abi.NSTATS))
    a = sol.mpc_args(B, N, x0, tt, ts, np.full(B, nev, dtype=np.int32), np.tile(ev, (B, 1)).copy(), np.tile(md, (B, 1)).copy(), oT, oX, oU, oM, oS, t0=np.zeros(B),
                     warm_x=wx, warm_u=wu)
    sol.mpc(a)
    counts = []
    for i in range(B):
        ref = orc.mpc_solve(N, 0.0, x0[i], tt[i], ts[i], nev, ev, md, warm=(wx[i], wu[i]))
        assert oS[i][8] == ref["stats"][8] and oS[i][9] == ref["stats"][9], (oS[i][8:], ref["stats"][8:])
        assert np.abs(oX[i] - ref["X"]).max() <= 1e-8 * max(1.0, np.abs(ref["X"]).max())
        assert np.abs(oU[i] - ref["U"]).max() <= 1e-8 * max(1.0, np.abs(ref["U"]).max())
        counts.append((int(oS[i][8]), int(oS[i][9])))
    assert counts[0][0] < counts[1][0] and counts[0][1] in (3, 4), counts   # the warm-started instance stopped earlier, on a tolerance


def test_emu_warm_start_resamples_the_previous_solution(emu):
    """qmgpu_warm_start_batch: previous (grid, X, U) -> initial guess on a shifted / non-uniform grid, against numpy interpolation."""
    itf, orc = emu
    rng = np.random.default_rng(3)
    _warm_start_case(itf, orc, rng, 2, 9, 12, event_pair=False)
    # a previous grid longer than two wavefronts' worth of entries (the interval search counts the entries below t with one ballot per 64) that holds an
    # event's (pre, post) pair of equal times; the new nodes run four to a wavefront pass with a ragged last pass (142 = 35 x 4 + 2)
    _warm_start_case(itf, orc, rng, 3, 150, 141, event_pair=True)


def _warm_start_case(itf, orc, rng, B, Np, Nn, event_pair):
    steps = rng.uniform(0.01, 0.02, (B, Np))
    if event_pair:
        steps[:, 70] = 0.0
    gp = np.cumsum(np.concatenate([np.zeros((B, 1)), steps], axis=1), axis=1)
    Xp = rng.normal(size=(B, Np + 1, 30)); Up = rng.normal(size=(B, Np, 30))
    gn = 0.013 + np.cumsum(np.concatenate([np.zeros((B, 1)), rng.uniform(0.008, 0.02, (B, Nn)) * (1.3 if Nn > 100 else 1.0)], axis=1), axis=1)   # starts later, ends past the old horizon
    assert (gn[:, -1] > gp[:, -1]).all()
    x0 = rng.normal(size=(B, 30))
    wx, wu = np.zeros((B, Nn + 1, 30)), np.zeros((B, Nn, 30))
    sol = api.GpuSolver(itf, max_batch=B, max_nodes=Nn)
    sol.warm_start(B, Np, gp, Xp, Up, Nn, gn, x0, wx, wu)
    for b in range(B):
        tu = gp[b][:-1]                                    # input k holds on [t_k, t_{k+1}); the last value is kept to the end
        for i in range(30):
            ex = np.interp(gn[b], gp[b], Xp[b][:, i]); ex[0] = x0[b][i]
            # inputs: interpolation between entries k and k + 1 over [t_k, t_{k+1}], the last entry beyond t_{N-1} (upstream pads U with its last value)
            eu = np.interp(gn[b][:-1], np.append(tu, gp[b][-1]), np.append(Up[b][:, i], Up[b][-1, i]))
            assert np.abs(wx[b][:, i] - ex).max() <= 1e-12 and np.abs(wu[b][:, i] - eu).max() <= 1e-12
    # the result is accepted as a warm start
    tgt = S.nominal_target(orc, itf.initial_state)
    assert sol.lib.qmgpu_warm_start_batch(sol.handle, B, Np, None, None, None, Nn, None, None, None, None) == 1


def test_emu_policy_evaluation_over_the_whole_horizon(emu):
    """qmgpu_policy_eval_batch against the oracle (MPC_MRT_Interface::evaluatePolicy, QMController.cpp:134-142) for evaluation times before the first node, on
    nodes, between nodes far into a 150-node horizon (the interval and the mode come from ballots over the grid: more than two wavefronts' worth of entries),
    on an event's pair of equal times and past the end: state, input and planned mode bit for bit."""
    itf, orc = emu
    rng = np.random.default_rng(17)
    N = 150
    steps = rng.uniform(0.005, 0.02, N); steps[40] = 0.0; steps[97] = 0.0
    grid = 0.3 + np.r_[0.0, np.cumsum(steps)]
    te = np.r_[grid[0] - 0.01, grid[0], grid[1], grid[40], grid[41], grid[64], grid[65], grid[128], grid[-1], grid[-1] + 0.02,
               rng.uniform(grid[0], grid[-1], 22)]
    B = len(te)
    T = np.tile(grid, (B, 1)); X = rng.normal(size=(B, N + 1, 30)); U = rng.normal(size=(B, N, 30))
    M = rng.integers(0, 16, (B, N + 1)).astype(np.int32)
    sol = api.GpuSolver(itf, max_batch=B, max_nodes=N)
    xo, uo, mo = np.zeros((B, 30)), np.zeros((B, 30)), np.zeros(B, dtype=np.int32)
    sol.policy_eval(B, N, T, X, U, M, te, xo, uo, mo)
    xr, ur, mr = orc.policy_eval_batch(T, X, U, M, te)
    assert np.array_equal(xo, xr) and np.array_equal(uo, ur) and np.array_equal(mo, mr)


def test_emu_fp32_build_of_the_mpc_chain(emu):
    """The fp32 build (qmgpu_mpc32.hip: the same kernel sources with real = float, namespace qmk32) on the CPU tier.  The emulation reproduces the
    hardware's fp32 accumulator map (row 4 (l / 16) + r instead of fp64's l / 16 + 4 r), so the permutation of the A-operand rows that keeps the kernels'
    accumulator code unchanged is exercised here: a wrong row anywhere scrambles the projected stages.  SQP and DDP variants, against the fp64 build."""
    itf, orc = emu
    B, N = 2, 10
    x_nom = itf.initial_state
    x0 = S.perturbed_states(x_nom, B, seed=8)
    tgt = S.nominal_target(orc, x_nom)
    tt = np.zeros((B, 1)); ts = np.tile(tgt, (B, 1, 1)).copy()
    nev, ev, md = S.trot_schedule(2.0, phase0=0.045)         # a switch falls between nodes 3 and 4
    out = {}
    for dtype in ("f64", "f32"):
        sol = api.GpuSolver(itf, max_batch=B, max_nodes=N, dtype=dtype)
        for alg in (0, 1):
            oT, oX, oU, oM, oS = np.zeros((B, N + 1)), np.zeros((B, N + 1, 30)), np.zeros((B, N, 30)), np.zeros((B, N + 1), dtype=np.int32), np.zeros((B, abi.NSTATS))
            a = sol.mpc_args(B, N, x0, tt, ts, np.full(B, nev, dtype=np.int32), np.tile(ev, (B, 1)).copy(), np.tile(md, (B, 1)).copy(), oT, oX, oU, oM, oS, t0=np.zeros(B),
                             algorithm=alg)
            sol.mpc(a)
            out[dtype, alg] = (oT, oX, oU, oM, oS)
        sol.close()
    for alg in (0, 1):
        T64, X64, U64, M64, S64 = out["f64", alg]
        T32, X32, U32, M32, S32 = out["f32", alg]
        assert np.array_equal(M64, M32) and np.array_equal(T64.astype(np.float32), T32.astype(np.float32))
        assert (S32[:, 7] == 0).all() and np.array_equal(S64[:, 4], S32[:, 4])          # factorised; same step length accepted
        assert 1e-9 < np.abs(X64 - X32).max() <= 2e-5 * max(1.0, np.abs(X64).max())
        assert np.abs(U64 - U32).max() <= 2e-5 * max(1.0, np.abs(U64).max())
    ref = orc.mpc_solve(N, 0.0, x0[0], tt[0], ts[0], nev, ev, md)
    assert np.abs(out["f64", 0][1][0] - ref["X"]).max() <= 1e-8


def test_emu_dense_tracking_weights_take_the_general_path():
    """The line search evaluates the tracking cost with structured forms when Q is diagonal and R' = diag + leg block + diag (what the reference's task.info
    produces) and with the dense forms otherwise; the kernel decides from the actual values.  A Q with off-diagonal entries must take the dense path
    and still match the oracle (merit of the accepted step, trajectories); the default weights give the same solve as before (all other tests)."""
    lib = abi.load_library(S.build_emu())
    itf = api.QMInterface(lib=lib)
    Q = np.array(itf.problem.settings.Q[:]).reshape(30, 30)
    assert np.count_nonzero(Q - np.diag(np.diag(Q))) == 0            # the reference's Q is diagonal: the structured path is the default
    Q[6, 7] = Q[7, 6] = 120.0; Q[12, 15] = Q[15, 12] = 1.5; Q[0, 9] = Q[9, 0] = -3.0
    for k, v in enumerate(Q.ravel()):
        itf.problem.settings.Q[k] = float(v)
    orc = S.Oracle(itf.problem)
    B, N = 2, 5
    x0 = S.perturbed_states(itf.initial_state, B, seed=21)
    tgt = S.nominal_target(orc, itf.initial_state)
    tt = np.zeros((B, 1)); ts = np.tile(tgt, (B, 1, 1)).copy()
    nev, ev, md = S.trot_schedule(2.0, phase0=0.03)
    sol = api.GpuSolver(itf, max_batch=B, max_nodes=N)
    oT, oX, oU, oM, oS = np.zeros((B, N + 1)), np.zeros((B, N + 1, 30)), np.zeros((B, N, 30)), np.zeros((B, N + 1), dtype=np.int32), np.zeros((B, abi.NSTATS))
    a = sol.mpc_args(B, N, x0, tt, ts, np.full(B, nev, dtype=np.int32), np.tile(ev, (B, 1)).copy(), np.tile(md, (B, 1)).copy(), oT, oX, oU, oM, oS, t0=np.zeros(B))
    sol.mpc(a)
    for i in range(B):
        ref = orc.mpc_solve(N, 0.0, x0[i], tt[i], ts[i], nev, ev, md)
        assert np.abs(oX[i] - ref["X"]).max() <= 1e-8 * max(1.0, np.abs(ref["X"]).max())
        assert np.abs(oU[i] - ref["U"]).max() <= 1e-8 * max(1.0, np.abs(ref["U"]).max())
        assert np.allclose(oS[i][:7], ref["stats"][:7], rtol=1e-8, atol=1e-10)      # merit before / after the step includes the off-diagonal terms


def test_emu_cycle_with_the_overlap_option(emu):
    """qmgpu_set_overlap / qmgpu_join_wbc through the host-emulated library: the code path of the second stream (events, joins, the carried inputLast_) executes and two
    cycles give the torques of two cycles without it, bit for bit (on the host every launch is synchronous: this checks the plumbing, the GPU test checks the schedule)."""
    itf, orc = emu
    B, N = 1, 4
    x0 = S.perturbed_states(itf.initial_state, B, seed=5)
    tgt = S.nominal_target(orc, itf.initial_state)
    tt = np.zeros((B, 1)); ts = np.tile(tgt, (B, 1, 1)).copy()
    nev, ev, md = api.GaitSchedule(lib=itf.lib).mode_schedule("trot", 0.0, 0.0, 1.0)
    rbd = np.zeros((B, 55)); rbd[:, 0:3] = x0[:, 9:12]; rbd[:, 3:6] = x0[:, 6:9]; rbd[:, 6:24] = x0[:, 12:30]
    outs = []
    for overlap in (False, True):
        sol = api.GpuSolver(itf, max_batch=B, max_nodes=N)
        sol.set_overlap(overlap)
        oT, oX, oU, oM, oS = np.zeros((B, N + 1)), np.zeros((B, N + 1, 30)), np.zeros((B, N, 30)), np.zeros((B, N + 1), dtype=np.int32), np.zeros((B, abi.NSTATS))
        a = sol.mpc_args(B, N, x0, tt, ts, np.full(B, nev, dtype=np.int32), ev[None, :].copy(), md[None, :].copy(), oT, oX, oU, oM, oS, t0=np.zeros(B))
        il, out, st = np.zeros((B, 30)), np.zeros((B, 54)), np.zeros(B, dtype=np.int32)
        w = sol.wbc_args(B, rbd, np.full(B, 0.002), np.full(B, 20.0), il, out, st)
        for _ in range(2):
            sol.cycle(a, np.zeros(B), w)
        sol.join_wbc(); sol.synchronize()
        outs.append((out.copy(), il.copy(), int(st[0])))
        sol.close()
    assert outs[0][2] == 0 and np.isfinite(outs[0][0]).all() and np.abs(outs[0][1]).max() > 0
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])


def test_emu_failed_factorisation_flags_the_instance_and_leaves_the_iterate(emu):
    """The failure path of riccati_kernel on the CPU tier (tests/test_gpu_edges.py has the GPU twin): input weights negated by a settings update -> the Cholesky of the
    backward sweep fails, the instance is flagged (stats[7] = 1), X / U stay the oracle's initial guess, and the next solve with the weights restored is the first one again."""
    itf, orc = emu
    B, N = 1, 4
    x0 = S.perturbed_states(itf.initial_state, B, seed=3)
    tgt = S.nominal_target(orc, itf.initial_state)
    tt = np.zeros((B, 1)); ts = np.tile(tgt, (B, 1, 1)).copy()
    nev, ev, md = S.trot_schedule(2.0, phase0=0.03)
    sol = api.GpuSolver(itf, max_batch=B, max_nodes=N)

    def solve():
        oT, oX, oU, oM, oS = np.zeros((B, N + 1)), np.zeros((B, N + 1, 30)), np.zeros((B, N, 30)), np.zeros((B, N + 1), dtype=np.int32), np.zeros((B, abi.NSTATS))
        sol.mpc(sol.mpc_args(B, N, x0, tt, ts, np.full(B, nev, dtype=np.int32), np.tile(ev, (B, 1)).copy(), np.tile(md, (B, 1)).copy(), oT, oX, oU, oM, oS, t0=np.zeros(B)))
        return oX, oU, oS

    gX, gU, gS = solve()
    assert gS[0, 7] == 0
    P2 = type(itf.problem).from_buffer_copy(itf.problem)
    for k in range(900):
        P2.settings.R_task[k] = -P2.settings.R_task[k]
    abi.check(itf.lib, itf.lib.qmgpu_update_settings(sol.handle, C.byref(P2.settings)))
    try:
        bX, bU, bS = solve()
        assert bS[0, 7] == 1 and np.isfinite(bX).all() and np.isfinite(bU).all()
        ref = S.Oracle(P2).mpc_solve(N, 0.0, x0[0], tt[0], ts[0], nev, ev, md)
        assert ref["status"] != 0 or ref["stats"][7] != 0
        assert np.abs(bX[0] - ref["X"]).max() <= 1e-12 * max(1.0, np.abs(ref["X"]).max()) and np.abs(bU[0] - ref["U"]).max() <= 1e-12 * max(1.0, np.abs(ref["U"]).max())
    finally:
        abi.check(itf.lib, itf.lib.qmgpu_update_settings(sol.handle, C.byref(itf.problem.settings)))
    aX, aU, aS = solve()
    assert np.array_equal(aX, gX) and np.array_equal(aU, gU) and aS[0, 7] == 0
