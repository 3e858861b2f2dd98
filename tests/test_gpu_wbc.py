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


@pytest.mark.parametrize("variant", [0, 1])
def test_wbc_stress_all_modes_converge(interface, oracle, variant):
    """2048 random instances over every contact mode of gait.info: all three levels converge everywhere; 48 samples vs the oracle."""
    import gpu_harness as G
    rng = np.random.default_rng(17 + variant)
    B = 2048
    x_nom, m = interface.initial_state, interface.robot_mass
    modes_all = np.array([15, 9, 6, 0, 10, 5, 13, 7, 14, 11], dtype=np.int32)
    mode = modes_all[rng.integers(0, len(modes_all), B)]
    xd = x_nom[None, :] + rng.uniform(-1, 1, (B, 30)) * 0.03
    xm = x_nom[None, :] + rng.uniform(-1, 1, (B, 30)) * 0.02
    vm = rng.uniform(-1, 1, (B, 24)) * 0.1
    u = np.zeros((B, 30))
    for i in range(B):
        flags = [(int(mode[i]) >> (3 - c)) & 1 for c in range(4)]
        for c in range(4):
            if flags[c]:
                u[i, 3 * c + 2] = m * 9.81 / max(1, sum(flags))
    u[:, :12] += rng.uniform(-1, 1, (B, 12)) * 2.0 * (u[:, :12] != 0)
    u[:, 12:] = rng.uniform(-1, 1, (B, 18)) * 0.1
    il = u + rng.uniform(-1, 1, (B, 30)) * 0.002
    t = np.where(rng.uniform(size=B) < 0.2, 5.0, 20.0)
    rbd = np.zeros((B, 55))
    rbd[:, 0:3] = xm[:, 9:12]; rbd[:, 3:6] = xm[:, 6:9]; rbd[:, 6:24] = xm[:, 12:30]; rbd[:, 24:48] = vm
    sol = G.make_solver(interface, B, 4)
    wb = G.WbcBatch(rbd, np.full(B, 0.002), t, il, xd, u, mode, variant)
    sol.debug_poison()
    sol.wbc(wb.args)
    r = wb.results()
    assert np.isfinite(r["out"]).all()
    assert (r["status"] == 0).all(), (np.nonzero(r["status"])[0][:10], r["status"][np.nonzero(r["status"])[0][:10]])
    errs = []
    for i in rng.choice(B, 48, replace=False):
        st, ref, _ = oracle.wbc_update(xd[i], u[i], rbd[i], int(mode[i]), 0.002, float(t[i]), il[i], variant)
        assert st == 0
        errs.append(np.abs(r["out"][i][36:] - ref[36:]).max() / max(1.0, np.abs(ref[36:]).max()))
    # Random (not MPC-consistent) desired states make some lowest-priority levels nearly degenerate LPs whose minimiser moves by 1e-2
    # when an inherited bound moves by 1e-5; the 1e-6 bar is asserted on the MPC-driven workloads (test_gpu_fullsize, test_gpu_configs).
    assert np.median(errs) <= 1e-8 and np.quantile(errs, 0.85) <= 1e-6 and max(errs) <= 1e-4, np.sort(errs)[-8:]


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
