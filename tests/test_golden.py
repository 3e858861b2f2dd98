"""Golden vectors (tests/golden/oracle_vectors.npz): outputs of this repository's own CPU oracle, NOT of the reference
(parity unpinned -- see tests/golden/make_golden.py).  CPU tier: the oracle still reproduces them.  GPU tier: the HIP path
reproduces them without touching the oracle.

tests/golden/hoqp_exact_ticks.npz / hoqp_exact_offenders.npz are of another kind: WBC ticks with the 50-digit solution of the reference's LITERAL hierarchical QP
(tools/hoqp_exact.py: HoQp.cpp:60-134 restated in mpmath, 1e-12 I in the matrix, Eigen's kernel basis, KKT conditions of the dense QP verified to 1e-38) -- computed
by a method that shares nothing with the oracle's or the kernels' level solver, so a regression both of those share cannot hide behind it."""
import os

import numpy as np
import pytest

import support as S

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_vectors.npz"))


def test_oracle_reproduces_golden(oracle):
    f, A, B = oracle.flow_map_lin(G["flow_x"], G["flow_u"])
    assert np.allclose(f, G["flow_f"], rtol=1e-12, atol=1e-12) and np.allclose(A, G["flow_A"], rtol=1e-11, atol=1e-12) and np.allclose(B, G["flow_B"], rtol=1e-11, atol=1e-12)
    tt, ts = np.zeros(1), G["mpc_target"][None, :].copy()
    for i in range(2):
        r = oracle.mpc_solve(8, 0.0, G["mpc_x0"][i], tt, ts, int(G["mpc_nev"]), G["mpc_ev"], G["mpc_md"])
        assert np.array_equal(r["mode"], G["mpc_mode"][i])
        assert np.allclose(r["X"], G["mpc_X"][i], rtol=1e-9, atol=1e-10) and np.allclose(r["U"], G["mpc_U"][i], rtol=1e-9, atol=1e-9)
    for i in range(len(G["wbc_mode"])):
        st, out, _ = oracle.wbc_update(G["wbc_xd"][i], G["wbc_u"][i], G["wbc_rbd"][i], int(G["wbc_mode"][i]), 0.002, float(G["wbc_time"][i]), G["wbc_il"][i])
        assert st == 0 and np.abs(out - G["wbc_out"][i]).max() <= 1e-7 * max(1.0, np.abs(G["wbc_out"][i]).max())


@pytest.mark.gpu
def test_hip_reproduces_golden(interface):
    import gpu_harness as H
    B, N = 2, 8
    sol = H.make_solver(interface, 4, N)
    tt = np.zeros((B, 1)); ts = np.tile(G["mpc_target"], (B, 1, 1)).copy()
    nev = int(G["mpc_nev"])
    mb = H.MpcBatch(G["mpc_x0"], tt, ts, np.full(B, nev, dtype=np.int32), np.tile(G["mpc_ev"], (B, 1)), np.tile(G["mpc_md"], (B, 1)), N)
    sol.mpc(mb.args)
    r = mb.results()
    assert np.array_equal(r["mode"], G["mpc_mode"])                                                     # contact modes bit exact
    assert np.abs(r["X"] - G["mpc_X"]).max() <= 1e-6 * max(1.0, np.abs(G["mpc_X"]).max())
    assert np.abs(r["U"] - G["mpc_U"]).max() <= 1e-6 * max(1.0, np.abs(G["mpc_U"]).max())
    nb = len(G["wbc_mode"])
    wb = H.WbcBatch(G["wbc_rbd"], np.full(nb, 0.002), G["wbc_time"], G["wbc_il"], G["wbc_xd"], G["wbc_u"], G["wbc_mode"])
    sol.wbc(wb.args)
    w = wb.results()
    for i in range(nb):
        assert np.abs(w["out"][i][36:] - G["wbc_out"][i][36:]).max() <= 1e-6 * max(1.0, np.abs(G["wbc_out"][i][36:]).max())


# ------------------------------------------------------------------------------------------------ against the 50-digit solution of the reference's own level QPs
EXACT = {name: np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name + ".npz")) for name in ("hoqp_exact_ticks", "hoqp_exact_offenders", "hoqp_exact_gaits")
         if os.path.exists(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name + ".npz"))}
# Stated bounds, rel-inf per block (each block its own norm; legs = what the separated-system plugin commands, QMController.cpp:428-431).
#   HierarchicalWbc: every block of every tick within 1e-8 (measured: median 3e-10, max 2e-9 -- the size of 1e-12 / curvature, i.e. of taking HoQp's regulariser in the limit).
#   HierarchicalMpcWbc: the arm has no task, its accelerations (1e3 .. 1e4 rad/s^2) hang on curvatures of 1e-10 next to the regulariser: median within 1e-6 on the torques,
#   every tick of the seeded sample within 1e-3 (measured: legs median 7e-8 / max 9e-5, arm 4e-7 / 1.5e-4); the offenders' file holds the worst ticks a GPU run found.
EXACT_BOUNDS = {0: dict(every=1e-8), 1: dict(median_tau=1e-6, every=1e-3)}


def exact_deviations(out, d):
    """per tick and block: rel-inf deviation of out [n][54] from the exact solutions of fixture d"""
    return S.rel_inf_blocks(out, d["exact"])


def check_against_exact(out, d, who):
    dev = exact_deviations(out, d)
    rep = {}
    for variant in (0, 1):
        idx = np.nonzero(d["variant"] == variant)[0]
        if len(idx) == 0:
            continue
        b = EXACT_BOUNDS[variant]
        rep[variant] = {k: (float(np.median(e[idx])), float(e[idx].max())) for k, e in dev.items()}
        for k, e in dev.items():
            assert e[idx].max() <= b["every"], (who, variant, k, float(e[idx].max()), str(d["source"][idx[e[idx].argmax()]]))
        if "median_tau" in b:
            assert np.median(dev["tau_legs"][idx]) <= b["median_tau"] and np.median(dev["tau_arm"][idx]) <= b["median_tau"], (who, rep[variant])
    return rep


@pytest.mark.parametrize("name", [n for n in ("hoqp_exact_ticks", "hoqp_exact_gaits") if n in EXACT])
def test_oracle_against_the_50_digit_solution_of_the_reference_qp(oracle, name):
    """hoqp_exact_ticks: trot / stance ticks of both controllers; hoqp_exact_gaits: the start-up branch (t < 10 s: the canonical second pass), flying trot (flight phases) and static walk
    (three-leg stances), both controllers -- measured round 6: every block within 2.6e-7 (legs), 1.1e-6 (arm, HierarchicalMpcWbc three-leg stances)"""
    d = EXACT[name]
    out = np.zeros((len(d["mode"]), 54))
    for i in range(len(out)):
        st, out[i], _ = oracle.wbc_update(d["xd"][i], d["ud"][i], d["rbd"][i], int(d["mode"][i]), float(d["period"][i]), float(d["time"][i]), d["il"][i].copy(), variant=int(d["variant"][i]))
        assert st == 0
    check_against_exact(out, d, "oracle")


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(EXACT))
def test_hip_against_the_50_digit_solution_of_the_reference_qp(interface, oracle, name):
    """The kernel on the fixture's ticks against the exact answer -- no oracle in the loop.  The offenders' file is reported, and bounded by the loose figure only: those are
    the ticks on which GPU and oracle disagreed most in a full-size closed loop / stress run, kept to show where BOTH stand against the reference's own problem."""
    import json
    import gpu_harness as H
    d = EXACT[name]
    n = len(d["mode"])
    out = np.zeros((n, 54))
    sol = H.make_solver(interface, n, 4)
    for variant in (0, 1):
        idx = np.nonzero(d["variant"] == variant)[0]
        if len(idx) == 0:
            continue
        wb = H.WbcBatch(d["rbd"][idx], d["period"][idx], d["time"][idx], d["il"][idx].copy(), d["xd"][idx], d["ud"][idx], d["mode"][idx].astype(np.int32), variant)
        sol.wbc(wb.args)
        r = wb.results()
        assert (r["status"] == 0).all()
        out[idx] = r["out"]
    orc_out = np.zeros((n, 54))
    for i in range(n):
        orc_out[i] = oracle.wbc_update(d["xd"][i], d["ud"][i], d["rbd"][i], int(d["mode"][i]), float(d["period"][i]), float(d["time"][i]), d["il"][i].copy(), variant=int(d["variant"][i]))[1]
    dev_gpu, dev_orc = exact_deviations(out, d), exact_deviations(orc_out, d)
    rep = {"ticks": n, "what": "rel-inf per block against the 50-digit solution of the reference's literal level QPs (tools/hoqp_exact.py); oracle = the CPU restatement (checker build) on the same inputs"}
    for variant in (0, 1):
        idx = np.nonzero(d["variant"] == variant)[0]
        if len(idx):
            rep["HierarchicalWbc" if variant == 0 else "HierarchicalMpcWbc"] = {
                "ticks": int(len(idx)),
                "gpu_vs_exact": {k: {"median": float(np.median(e[idx])), "max": float(e[idx].max())} for k, e in dev_gpu.items()},
                "oracle_vs_exact": {k: {"median": float(np.median(e[idx])), "max": float(e[idx].max())} for k, e in dev_orc.items()},
                "gpu_vs_oracle": {k: {"median": float(np.median(e[idx])), "max": float(e[idx].max())} for k, e in S.rel_inf_blocks(out, orc_out).items()}}
    if name == "hoqp_exact_offenders":
        rep["per_tick"] = [{"source": str(d["source"][i]), "gpu_vs_exact": {k: float(e[i]) for k, e in dev_gpu.items()}, "oracle_vs_exact": {k: float(e[i]) for k, e in dev_orc.items()},
                            "tau_legs_abs_Nm": [float(np.abs(out[i, 36:48] - d["exact"][i, 36:48]).max()), float(np.abs(orc_out[i, 36:48] - d["exact"][i, 36:48]).max())]} for i in range(n)]
    os.makedirs(os.path.join(S.ROOT, "gpurun_out"), exist_ok=True)
    json.dump(rep, open(os.path.join(S.ROOT, "gpurun_out", name + "_gpu.json"), "w"), indent=1)
    if name in ("hoqp_exact_ticks", "hoqp_exact_gaits"):
        check_against_exact(out, d, "gpu")
    else:
        # the ticks a full-size closed loop / stress run found hardest: the kernel within 1e-6 of the reference's exact answer on all but three of them (measured round 6: 23 of 26;
        # the checker on all 26), and nowhere grossly off
        for k, e in dev_gpu.items():
            assert np.median(e) <= 1e-6 and (e > 1e-5).sum() <= 3 and e.max() <= 0.5, (k, float(np.median(e)), int((e > 1e-5).sum()), float(e.max()))
        for k, e in dev_orc.items():
            assert e.max() <= 1e-5, (k, float(e.max()))


def test_first_level_behind_the_interior_point_ends_at_the_cold_vertex(oracle):
    """tests/golden/wbc_slow_ticks.npz (the ten slowest WBC ticks of round 6's steady-state leg: robots whose torque limits cannot hold): the first level through the interior
    point with its own rows as penalised slacks (default) against the same level cold from z = 0 (own_interior_point = 0, the algorithm until round 6): same torques,
    a bounded number of passes instead of up to 46 working-set changes."""
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "wbc_slow_ticks.npz"))
    n = len(d["mode"])

    def run():
        outs, passes = [], []
        for i in range(n):
            ws = np.zeros((1, 48), dtype=np.uint64)      # (qmgpu.h: QMGPU_WBC_STATE_WORDS)
            oracle.set_working_set(ws)
            st, out, _ = oracle.wbc_update(d["xd"][i], d["ud"][i], d["rbd"][i], int(d["mode"][i]), float(d["period"][i]), float(d["time"][i]), d["il"][i].copy())
            assert st == 0
            outs.append(out); passes.append(int(np.ascontiguousarray(ws[0, 13:14]).view(np.uint8)[0]) & 127)
        oracle.set_working_set(None)
        return np.array(outs), np.array(passes)
    try:
        oracle.set_experiment(own_interior_point=0)
        cold, p_cold = run()
    finally:
        oracle.set_experiment()
    now, p_now = run()
    dev = S.rel_inf_blocks(now, cold)
    assert all(v.max() <= 1e-9 for v in dev.values()), {k: float(v.max()) for k, v in dev.items()}        # measured 2.6e-13
    assert p_cold.max() >= 40 and p_now.max() <= 12, (p_cold, p_now)                                      # measured: up to 46 cold, at most 8 now


@pytest.mark.parametrize("variant", [0, 1])
def test_first_level_paths_agree_on_fast_robots(oracle, interface, variant):
    """support.wbc_fast_robots_batch (128 instances): robots moving so fast that the first level's limits cannot hold.  The default path (held-variable form for at most four
    iterations, then the interior point with the own rows as penalised slacks) and the cold path (own_interior_point = 0) end at the same torques; a third of the instances
    take the new path."""
    b = S.wbc_fast_robots_batch(interface, variant, 128)

    def run():
        outs, passes = [], []
        for i in range(128):
            ws = np.zeros((1, 48), dtype=np.uint64)
            oracle.set_working_set(ws)
            st, out, _ = oracle.wbc_update(b["xd"][i], b["u"][i], b["rbd"][i], int(b["mode"][i]), 0.002, float(b["t"][i]), b["il"][i].copy(), variant=variant)
            assert st == 0
            outs.append(out); passes.append(int(np.ascontiguousarray(ws[0, 13:14]).view(np.uint8)[0]) & 127)
        oracle.set_working_set(None)
        return np.array(outs), np.array(passes)
    try:
        oracle.set_experiment(own_interior_point=0)
        cold, p_cold = run()
    finally:
        oracle.set_experiment()
    now, p_now = run()
    dev = S.rel_inf_blocks(now, cold)
    assert all(v.max() <= 1e-9 for v in dev.values()), {k: float(v.max()) for k, v in dev.items()}        # measured 4e-13 / 7e-13
    assert (p_now != p_cold).sum() >= 128 // 5, (p_now != p_cold).sum()                                 # the batch exercises the new path (measured: 40 %)
