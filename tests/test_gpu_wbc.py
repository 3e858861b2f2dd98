"""-m gpu: HIP whole-body controller (model update + three-level HoQP) against the CPU oracle, through the C ABI."""
import numpy as np
import pytest

import support as S

pytestmark = pytest.mark.gpu


def _cases(interface, oracle, variant, seed=3):
    rng = np.random.default_rng(seed)
    x_nom, m = interface.initial_state, interface.robot_mass
    out = []
    for mode, t in ((15, 20.0), (15, 5.0), (9, 20.0), (6, 12.0), (0, 20.0), (13, 20.0), (7, 20.0), (10, 20.0)):
        flags = [(mode >> (3 - c)) & 1 for c in range(4)]
        u = np.zeros(30)
        for c in range(4):
            if flags[c]:
                u[3 * c + 2] = m * 9.81 / max(1, sum(flags))
        u[12:] = rng.uniform(-1, 1, 18) * 0.05
        xd = x_nom + rng.uniform(-1, 1, 30) * 0.02
        xm = x_nom + rng.uniform(-1, 1, 30) * 0.01
        vm = rng.uniform(-1, 1, 24) * 0.05
        out.append(dict(xd=xd, u=u, rbd=S.rbd_from_state(oracle, xm, vm), mode=mode, t=t, il=u + rng.uniform(-1, 1, 30) * 0.001, variant=variant))
    return out


@pytest.mark.parametrize("variant", [0, 1])
def test_wbc_matches_oracle(interface, oracle, variant):
    """Torques / decision vector within 1e-6 rel-inf of the CPU restatement; inputLast state updated identically."""
    import gpu_harness as G
    cs = _cases(interface, oracle, variant)
    B = len(cs)
    sol = G.make_solver(interface, B, 4)
    wb = G.WbcBatch(np.array([c["rbd"] for c in cs]), np.full(B, 0.002), np.array([c["t"] for c in cs]), np.array([c["il"] for c in cs]),
                    np.array([c["xd"] for c in cs]), np.array([c["u"] for c in cs]), np.array([c["mode"] for c in cs], dtype=np.int32), variant)
    sol.wbc(wb.args)
    r = wb.results()
    assert (r["status"] == 0).all()
    for i, c in enumerate(cs):
        st, ref, il = oracle.wbc_update(c["xd"], c["u"], c["rbd"], c["mode"], 0.002, c["t"], c["il"], variant)
        assert st == 0
        # HierarchicalMpcWbc leaves the arm accelerations to the contact-force level only: they reach 1e4 rad/s^2 and are
        # conditioned accordingly, hence the looser bound on the decision vector (torques keep 1e-6)
        xtol = 1e-6 if variant == 0 else 2e-5
        assert np.abs(r["out"][i][:36] - ref[:36]).max() <= xtol * max(1.0, np.abs(ref[:36]).max()), (i, c["mode"])
        assert np.abs(r["out"][i][36:] - ref[36:]).max() <= 1e-6 * max(1.0, np.abs(ref[36:]).max()), (i, c["mode"])
        assert np.array_equal(r["input_last"][i], il)


def test_cycle_matches_oracle(interface, oracle):
    """qmgpu_cycle_batch = MPC solve -> policy evaluation at t_eval -> WBC, all on device."""
    import gpu_harness as G
    import torch
    B, N = 3, 20
    x_nom = interface.initial_state
    x0 = S.perturbed_states(x_nom, B, seed=5)
    tgt = S.nominal_target(oracle, x_nom)
    tt = np.zeros((B, 1)); ts = np.tile(tgt, (B, 1, 1)).copy()
    nev, ev, md = S.trot_schedule(2.0, phase0=0.03)
    sol = G.make_solver(interface, B, N)
    mb = G.MpcBatch(x0, tt, ts, np.full(B, nev, dtype=np.int32), np.tile(ev, (B, 1)), np.tile(md, (B, 1)), N)
    rbd = np.array([S.rbd_from_state(oracle, x0[i]) for i in range(B)])
    il = np.zeros((B, 30))
    wb = G.WbcBatch(rbd, np.full(B, 0.002), np.full(B, 20.0), il)
    dt = interface.problem.settings.dt
    t_eval = G.dev(np.full(B, 0.4 * dt), torch.float64)
    sol.cycle(mb.args, t_eval, wb.args)
    r, w = mb.results(), wb.results()
    for i in range(B):
        ref = oracle.mpc_solve(N, 0.0, x0[i], tt[i], ts[i], nev, ev, md)
        a = 1.0 - 0.4
        xd = a * ref["X"][0] + (1 - a) * ref["X"][1]
        ud = a * ref["U"][0] + (1 - a) * ref["U"][1]
        st, out, _ = oracle.wbc_update(xd, ud, rbd[i], int(ref["mode"][0]), 0.002, 20.0, il[i])
        assert np.abs(w["out"][i][36:] - out[36:]).max() <= 1e-6 * max(1.0, np.abs(out[36:]).max())


stress_batch = S.wbc_stress_batch


def oracle_sensitivity(orc, c, idx, variant, eps=1e-9, draws=8, seed=5):
    """How far the ORACLE's own torques move (||.||_inf relative, per instance of idx) under two kinds of change that leave the mathematical problem (almost) alone:
    (input) the desired state / measurement perturbed by eps relative, a few seeded directions; (path) another path to the same vertex -- the interior point that
    runs in front of the active-set method started from 0.15 sqrt(scale) instead of 0.5 sqrt(scale), and no interior point at all (the active-set method cold from z = 0).  A well-posed
    instance moves by ~1e2 * eps and not at all; an instance one of whose level problems is nearly degenerate -- a direction whose curvature sits at the rounding of
    the normal equations, a multiplier at the rounding of its gradient: a path-dependent decision no arithmetic can avoid -- moves by orders of magnitude more, and
    GPU / oracle agreement there cannot be better than that."""
    rng = np.random.default_rng(seed)
    args = lambda xd, rbd: (xd, c["u"][idx], rbd, c["mode"][idx], 0.002, c["t"][idx], c["il"][idx], variant)  # noqa: E731
    base = orc.wbc_batch(*args(c["xd"][idx], c["rbd"][idx]))["out"][:, 36:]
    worst = np.zeros(len(idx))
    for _ in range(draws):
        xd = c["xd"][idx] * (1 + eps * rng.uniform(-1, 1, (len(idx), 30))); rbd = c["rbd"][idx].copy()
        rbd[:, :48] *= 1 + eps * rng.uniform(-1, 1, (len(idx), 48))
        worst = np.maximum(worst, S.rel_inf(orc.wbc_batch(*args(xd, rbd))["out"][:, 36:], base))
    path = np.zeros(len(idx))
    # (the checker against ITSELF compiled differently -- the checker build next to the timing-grade one, other contractions and vector widths: where its two builds part,
    #  a third implementation cannot be pinned any better; instance 221 of the HierarchicalMpcWbc batch is such a tick: 2.5e-3 between the builds, 1.5e-8 between GPU and the other one)
    other = S.Oracle(orc.P, fast=not orc.lib.qmo_is_fast_build())
    path = np.maximum(path, S.rel_inf(other.wbc_batch(*args(c["xd"][idx], c["rbd"][idx]))["out"][:, 36:], base))
    for kw in (dict(lower_level_start=0.15), dict(no_interior_point=True)):
        orc.set_experiment(**kw)
        try:
            path = np.maximum(path, S.rel_inf(orc.wbc_batch(*args(c["xd"][idx], c["rbd"][idx]))["out"][:, 36:], base))
        finally:
            orc.set_experiment()
    return worst, path


def test_first_level_of_diverged_robots_goes_through_the_interior_point(interface, oracle):
    """tests/golden/wbc_slow_ticks.npz: the ten slowest WBC ticks of round 6's steady-state leg -- robots whose plan has diverged, the first level's minimum-norm point
    violates 7-16 limits.  Until the end of round 6 three of them took 23-46 working-set changes of a 36-variable level from z = 0 (3.1-3.5 ms, the tail of every
    256-instance launch they were in).  Now a held-variable form that is rejected, or still running after QP_HELD_CAP iterations, hands the level to the interior point with the
    own rows as penalised slacks: same vertex (the CPU restatement's cold path is the reference here: own_interior_point = 0), a bounded number of passes."""
    import os
    import gpu_harness as G
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "wbc_slow_ticks.npz"))
    n = len(d["mode"])
    sol = G.make_solver(interface, n, 4)
    wb = G.WbcBatch(d["rbd"], d["period"].astype(np.float64), d["time"].astype(np.float64), d["il"].copy(), d["xd"], d["ud"], d["mode"].astype(np.int32), 0, carry=True)
    sol.wbc(wb.args)
    r = wb.results()
    assert (r["status"] == 0).all()
    passes = np.ascontiguousarray(r["working_set"][:, 13:15]).view(np.uint8).astype(int) & 127
    oracle.set_experiment(own_interior_point=0)      # cold from z = 0, as until round 6
    try:
        cold = np.array([oracle.wbc_update(d["xd"][i], d["ud"][i], d["rbd"][i], int(d["mode"][i]), float(d["period"][i]), float(d["time"][i]), d["il"][i].copy())[1] for i in range(n)])
    finally:
        oracle.set_experiment()
    now = np.array([oracle.wbc_update(d["xd"][i], d["ud"][i], d["rbd"][i], int(d["mode"][i]), float(d["period"][i]), float(d["time"][i]), d["il"][i].copy())[1] for i in range(n)])
    for name, ref in (("the oracle on the same path", now), ("the oracle cold from z = 0", cold)):
        dev = S.rel_inf_blocks(r["out"], ref)
        assert all(v.max() <= 1e-9 for v in dev.values()), (name, {k: float(v.max()) for k, v in dev.items()})       # measured 4e-13 / 4e-13
    assert passes[:, 0].max() <= 12, passes[:, 0]          # first level: interior-point iterations + working-set changes (measured: 1, 8, 8, 5, 6, 6, 5, 6, 5, 8; cold: up to 46)
    sol.close()


@pytest.mark.parametrize("variant", [0, 1])
def test_wbc_fast_robots_whose_limits_cannot_hold(interface, variant):
    """support.wbc_fast_robots_batch (512 instances, every contact mode, both controllers): joint rates of +-15 rad/s -- the first level's limits cannot hold, 40 % of the instances
    go through the interior point with the own rows as penalised slacks.  GPU against the oracle on the same path and against the oracle's cold path (own_interior_point = 0,
    the algorithm until round 6): status words zero, every block within 1e-9, first-level passes bounded."""
    import gpu_harness as G
    B = 512
    c = S.wbc_fast_robots_batch(interface, variant, B)
    sol = G.make_solver(interface, B, 4)
    wb = G.WbcBatch(c["rbd"], np.full(B, 0.002), c["t"], c["il"].copy(), c["xd"], c["u"], c["mode"], variant, carry=True)
    sol.wbc(wb.args)
    r = wb.results()
    assert (r["status"] == 0).all(), np.nonzero(r["status"])[0][:10]
    passes = np.ascontiguousarray(r["working_set"][:, 13:15]).view(np.uint8).astype(int) & 127
    orc = S.Oracle(interface.problem, fast=True)
    w = orc.wbc_batch(c["xd"], c["u"], c["rbd"], c["mode"], 0.002, c["t"], c["il"], variant=variant)
    ref, st = w["out"], w["status"]
    assert (st == 0).all()
    try:
        orc.set_experiment(own_interior_point=0)
        w = orc.wbc_batch(c["xd"], c["u"], c["rbd"], c["mode"], 0.002, c["t"], c["il"], variant=variant)
        cold, st_c = w["out"], w["status"]
    finally:
        orc.set_experiment()
    assert (st_c == 0).all()
    rep = {}
    for name, o in (("same_path", ref), ("cold_path", cold)):
        dev = S.rel_inf_blocks(r["out"], o)
        rep[name] = {k: float(v.max()) for k, v in dev.items()}
        assert all(v.max() <= 1e-9 for v in dev.values()), (name, rep[name])          # measured 6e-12 (HierarchicalWbc) / 2e-12 (HierarchicalMpcWbc) on either path
    assert passes[:, 0].max() <= 25, passes[:, 0].max()          # (cold: up to 15 on this batch, 46 on the bench's diverged robots)
    import json, os
    os.makedirs(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out"), exist_ok=True)
    with open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "wbc_fast_robots_v%d.json" % variant), "w") as f:
        json.dump({"instances": B, "first_level_passes_max": int(passes[:, 0].max()), "first_level_passes_mean": float(passes[:, 0].mean()), "gpu_vs_oracle": rep}, f, indent=1)
    sol.close()


@pytest.mark.parametrize("variant", [0, 1])
def test_wbc_stress_all_modes_converge(interface, variant):
    """2048 random instances over every contact mode of gait.info, robots in motion (NOT MPC-consistent desired states, 20 % on the start-up branch, swing legs during it:
    the canonical second pass of the cascade runs): no level is flagged anywhere, and EVERY instance is compared with the (multi-threaded) oracle -- no sampling.  The
    distribution goes to gpurun_out/parity.json.  Stated bound: torques within 1e-9 rel-inf at the 99th percentile (measured: 1e-14) and within 1e-6 on every instance
    that is well posed, where ill posed is MEASURED on the oracle alone, not assumed: its own torques move by at least a third of the deviation when only its path to
    the vertex changes (another interior-point start, no interior point) or under a 1e-9 relative perturbation of the inputs; at most 0.5 % of the instances are."""
    import json
    import os
    import gpu_harness as G
    B = 2048
    os.makedirs(os.path.join(S.ROOT, "gpurun_out"), exist_ok=True)
    c = stress_batch(interface, variant, B)
    sol = G.make_solver(interface, B, 4)
    wb = G.WbcBatch(c["rbd"], np.full(B, 0.002), c["t"], c["il"], c["xd"], c["u"], c["mode"], variant)
    sol.debug_poison()
    sol.wbc(wb.args)
    r = wb.results()
    assert np.isfinite(r["out"]).all()
    assert (r["status"] == 0).all(), (np.nonzero(r["status"])[0][:10], r["status"][np.nonzero(r["status"])[0][:10]])
    orc = S.Oracle(interface.problem, fast=True)
    ref = orc.wbc_batch(c["xd"], c["u"], c["rbd"], c["mode"], 0.002, c["t"], c["il"], variant)
    assert (ref["status"] == 0).all()
    assert np.array_equal(r["input_last"], ref["input_last"])
    blocks = S.rel_inf_blocks(r["out"], ref["out"])
    err = np.maximum(S.rel_inf(r["out"][:, 36:], ref["out"][:, 36:]), np.maximum(blocks["tau_legs"], blocks["tau_arm"]))     # the 18-wide norm and each block's own: the worst
    errx = S.rel_inf(r["out"][:, :36], ref["out"][:, :36])
    np.savez(os.path.join(S.ROOT, "gpurun_out", f"wbc_stress_v{variant}.npz"), gpu=r["out"], oracle=ref["out"], iterations=ref["iterations"], as_iterations=ref["as_iterations"])   # scratch, for offline analysis
    above = np.nonzero(err > 1e-6)[0]
    sens_in, sens_path = oracle_sensitivity(orc, c, above, variant) if len(above) else (np.zeros(0), np.zeros(0))
    rep = {"instances": B, "tau": {"max": float(err.max()), "p99": float(np.percentile(err, 99)), "p90": float(np.percentile(err, 90)), "median": float(np.median(err))},
           "x": {"max": float(errx.max()), "p99": float(np.percentile(errx, 99)), "median": float(np.median(errx))},
           "blocks": S.block_summary(r["out"], ref["out"]),
           "count_above_1e-6": int(len(above)), "count_above_1e-9": int((err > 1e-9).sum()), "count_above_1e-12": int((err > 1e-12).sum()),
           "oracle_passes_mean_max_per_level": [[float(ref["iterations"][:, l].mean()), int(ref["iterations"][:, l].max())] for l in range(4)],
           "oracle_active_set_iterations_mean_max_per_level": [[float(ref["as_iterations"][:, l].mean()), int(ref["as_iterations"][:, l].max())] for l in range(4)],
           "above_1e-6": [{"instance": int(i), "mode": int(c["mode"][i]), "time": float(c["t"][i]), "tau_dev": float(err[i]), "x_dev": float(errx[i]),
                           "tau_legs_dev": float(blocks["tau_legs"][i]), "tau_arm_dev": float(blocks["tau_arm"][i]), "tau_legs_abs_dev_Nm": float(np.abs(r["out"][i, 36:48] - ref["out"][i, 36:48]).max()),
                           "oracle_passes": ref["iterations"][i].tolist(),
                           "oracle_tau_move_under_1e-9_input_perturbation": float(sens_in[k]), "oracle_tau_move_on_another_path_to_the_vertex": float(sens_path[k])}
                          for k, i in enumerate(above)]}
    path = os.path.join(S.ROOT, "gpurun_out", "parity.json")
    try:
        allrep = json.load(open(path))
    except (OSError, ValueError):
        allrep = {}
    allrep[f"wbc_stress_2048_all_modes_variant{variant}"] = rep
    json.dump(allrep, open(path, "w"), indent=1)
    assert np.median(err) <= 1e-12 and np.percentile(err, 99) <= 1e-9, rep["tau"]
    assert len(above) <= B // 200, rep["count_above_1e-6"]
    assert err.max() <= 0.1, rep["above_1e-6"]            # gross cap, whatever the checker's own sensitivity says (advisor r05): a shared regression in an ill-conditioned class must not pass as "explained"
    for k, i in enumerate(above):
        assert max(sens_in[k], sens_path[k]) >= err[i] / 3, rep["above_1e-6"][k]


def test_settings_update_and_dtype_handles(interface, oracle):
    """qmgpu_update_settings (run-time gain changes, what the reference's dynamic_reconfigure callbacks do): the next WBC call uses the new gains and matches
    the oracle evaluated with them; qmgpu_create_ex validates the dtype and an fp32 handle refuses the fp64-only LQ dump."""
    import copy
    import ctypes as C
    import gpu_harness as G
    from qm_door_amd import abi
    c = _cases(interface, oracle, 0)[2]
    sol = G.make_solver(interface, 1, 4)
    def wbc():
        wb = G.WbcBatch(c["rbd"][None], np.array([0.002]), np.array([c["t"]]), c["il"][None].copy(), c["xd"][None], c["u"][None], np.array([c["mode"]], dtype=np.int32), 0)
        sol.wbc(wb.args)
        return wb.results()["out"][0]
    before = wbc()
    P2 = type(interface.problem).from_buffer_copy(interface.problem)
    P2.settings.kp_base_height *= 3.0; P2.settings.kd_swing *= 0.5
    abi.check(interface.lib, interface.lib.qmgpu_update_settings(sol.handle, C.byref(P2.settings)))
    after = wbc()
    assert np.abs(after - before).max() > 1e-3
    st, ref, _ = S.Oracle(P2).wbc_update(c["xd"], c["u"], c["rbd"], c["mode"], 0.002, c["t"], c["il"])
    assert st == 0 and np.abs(after[36:] - ref[36:]).max() <= 1e-6 * max(1.0, np.abs(ref[36:]).max())
    h = C.c_void_p()
    assert interface.lib.qmgpu_create_ex(C.byref(interface.problem), 0, 1, 4, 7, C.byref(h)) == abi.ERR_INVALID_ARGUMENT
    s32 = G.make_solver(interface, 1, 4, dtype="f32")
    assert interface.lib.qmgpu_enable_debug(s32.handle, 1) == abi.ERR_INVALID_ARGUMENT
