"""-m gpu: the other BASELINE.json configurations as parity cases.

  config 3: batch 2048, randomised base pose + EE target (the 8-GPU weak-scaling workload, here all on one GPU)
  config 5: mixed gait schedule stance -> trot -> flying_trot -> static_walk (nc in {12, 14, 16, 13}, FLY nodes), N = 200, batch 1024,
            in fp64 against the oracle, and the fp32-vs-fp64 sweep of the same batch (MPC kernels in fp32, qmgpu_create_ex)
EVERY instance is compared with the oracle (qmo_cycle_batch_mt on all host threads) at the north_star tolerance; the numbers go to
gpurun_out/parity.json (-> profiles/r03_parity.json).
"""
import numpy as np
import pytest

import support as S

pytestmark = pytest.mark.gpu


def _mixed_schedule(t_end):
    """Templates of gait.info laid end to end: 0.4 s stance, then trot / flying_trot / static_walk cycles until t_end."""
    from qm_door_amd import abi, api
    gs = api.GaitSchedule()
    ev, md = [], [15]
    t = 0.4
    names = ["trot", "flying_trot", "static_walk"]
    k = 0
    while t < t_end:
        g = gs.template(names[k % 3]); k += 1
        for i in range(g.num_modes):
            m = int(g.modes[i])
            if m == md[-1]:
                t += g.switching_times[i + 1] - g.switching_times[i]
                continue
            ev.append(t); md.append(m)
            t += g.switching_times[i + 1] - g.switching_times[i]
    ev.append(t); md.append(15)
    assert len(ev) <= abi.MAX_EVENTS
    evp = np.full(abi.MAX_EVENTS, 1e300); evp[:len(ev)] = ev
    mdp = np.full(abi.MAX_EVENTS + 1, 15, dtype=np.int32); mdp[:len(md)] = md
    return len(ev), evp, mdp


def test_config5_mixed_gaits_n200_batch1024(interface, oracle):
    import gpu_harness as G
    B, N = 1024, 200
    dt = interface.problem.settings.dt
    x_nom = interface.initial_state
    x0 = S.perturbed_states(x_nom, B, seed=3)
    tgt = S.nominal_target(oracle, x_nom)
    tt = np.zeros((B, 1)); ts = np.tile(tgt, (B, 1, 1)).copy()
    nev, ev, md = _mixed_schedule(N * dt + 0.2)
    modes = np.array([oracle.node_mode_at(ev[:nev], md[:nev + 1], k * dt) for k in range(N + 1)], dtype=np.int32)
    assert {15, 9, 6, 0}.issubset(set(modes.tolist())) and len(set(modes.tolist())) >= 6      # stance, trot pair, FLY, three-leg phases
    sol = G.make_solver(interface, B, N)
    mb = G.MpcBatch(x0, tt, ts, np.full(B, nev, dtype=np.int32), np.tile(ev, (B, 1)), np.tile(md, (B, 1)), N)
    sol.debug_poison()          # every tile path (m~ = 14, 16, 17, 18) runs on NaN-filled scratch / LDS: nothing may read leftovers
    sol.mpc(mb.args)
    r = mb.results()
    assert np.isfinite(r["X"]).all() and np.isfinite(r["U"]).all() and (r["stats"][:, 7] == 0).all()
    assert np.array_equal(r["mode"], np.tile(modes, (B, 1)))
    ref = S.Oracle(interface.problem, fast=True).cycle_batch(N, x0, tt, ts, nev, ev, md)     # all 1024 instances, MPC only
    S.assert_parity(S.parity_report("configs4_fp64_1024xN200_mixed_gaits", r, ref, tau=False))


def test_config3_randomised_pose_and_targets_batch2048(interface, oracle):
    import gpu_harness as G
    import torch
    import bench
    B, N = 2048, 100
    sc = bench.build_config3(interface)        # the ONE global batch `bench.py --gpus 8` shards (seed 1), here all of it on one GPU
    x0, tt, ts, nev, ev, md, rbd = sc["x0"], sc["tt"], sc["ts"], sc["nev"], sc["ev"], sc["md"], sc["rbd"]
    assert x0.shape == (B, 30) and np.abs(x0[:, 6:8]).max() <= 0.5 and np.abs(x0[:, 9]).max() <= 0.5
    # the same 2048 poses and targets with the robots in motion (support.moving_inputs: measured twist / joint rates, momentum-consistent x0, non-zero
    # inputLast_, controller times on both sides of the start-up branch, the policy evaluated between nodes)
    mv = S.moving_inputs(oracle, x0, interface.problem.settings.dt, seed=22)
    x0, rbd = mv["x0"], mv["rbd"]
    sol = G.make_solver(interface, B, N)
    mb = G.MpcBatch(x0, tt, ts, np.full(B, nev, dtype=np.int32), np.tile(ev, (B, 1)), np.tile(md, (B, 1)), N)
    wb = G.WbcBatch(rbd, np.full(B, 0.002), mv["time"], mv["input_last"])
    sol.cycle(mb.args, G.dev(mv["t_eval"], torch.float64), wb.args)
    r, w = mb.results(), wb.results()
    assert np.isfinite(r["X"]).all() and np.isfinite(r["U"]).all() and np.isfinite(w["out"]).all()
    assert (r["stats"][:, 7] == 0).all() and (w["status"] == 0).all()
    r.update(w)
    ref = S.Oracle(interface.problem, fast=True).cycle_batch(N, x0, tt, ts, nev, ev, md, rbd=rbd, t_eval=mv["t_eval"], time=mv["time"],
                                                             input_last=mv["input_last"])     # all 2048 instances, MPC + policy + WBC
    S.assert_parity(S.parity_report("configs2_2048xN100_random_pose_moving", r, ref))
    assert S.rel_inf(r["input_last"], ref["input_last"]).max() <= 1e-9            # inputLast_ <- the evaluated policy input


def test_config5_fp32_vs_fp64_sweep(interface, oracle):
    """BASELINE.json configs[4], second half: the same 1024 x 200-node mixed-gait batch with the MPC kernels in fp32 (v_mfma_f32_16x16x4_f32, fp32
    scratch) next to the fp64 path; one MPC + policy evaluation + WBC cycle each.  Contact modes must be bit-exact (they are decided on the fp64
    times in both builds); X, U and the WBC torques are held to STATED ||.||_inf-relative bounds per instance:
        X, U   1e-4   (measured on MI355X: max 0.9e-5 .. 1.1e-5 / 0.8e-5 .. 0.9e-5, median 2.5e-6 -- DESIGN.md section 5.1)
        tau    2e-3   (measured: 99th percentile 2.3e-5 .. 2.6e-5, median 2.7e-6, max 4e-5 .. 7.5e-4: the WBC runs in fp64 on either policy and is
                       piecewise linear in it -- an instance next to an active-set change amplifies the 1e-5 policy difference)
    fp32 is a tolerance study, not a parity path: 1e-6 against the reference is only claimed for fp64."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
    import fp32_sweep
    out = fp32_sweep.run(1024, 200)
    rep = fp32_sweep.report(out)
    assert rep["finite_f32"] and rep["riccati_status_f32_all_zero"] and rep["modes_bit_exact"]
    assert rep["statistics_repeat_bit_for_bit"]        # merits / violations of three identical calls: a stray write into a neighbouring buffer shows here
    assert (out["f32"]["wbc"]["status"] == 0).all() and (out["f64"]["wbc"]["status"] == 0).all()
    assert max(rep["step_metrics_rel"].values()) <= 2e-4, rep      # merit and constraint violation before / after the step agree between the builds
    assert rep["line_search_alpha_differs"] == 0, rep    # (a differing step length would show as an O(1) deviation below)
    assert rep["X"]["max"] <= 1e-4 and rep["U"]["max"] <= 1e-4 and rep["tau"]["max"] <= 2e-3 and rep["tau"]["p99"] <= 1e-4, rep
    assert rep["X"]["max"] > 1e-9           # the two paths really are different arithmetic
    # the fp64 leg of the same run is the parity path: sampled against the oracle at the north_star tolerance
    from test_gpu_configs import _mixed_schedule
    dt = interface.problem.settings.dt
    x0 = S.perturbed_states(interface.initial_state, 1024, seed=3)
    tgt = S.nominal_target(oracle, interface.initial_state)
    nev, ev, md = _mixed_schedule(200 * dt + 0.2)
    ref = S.Oracle(interface.problem, fast=True).cycle_batch(200, x0, np.zeros((1024, 1)), np.tile(tgt, (1024, 1, 1)), nev, ev, md)
    S.assert_parity(S.parity_report("configs4_fp64_leg_of_the_sweep_1024xN200", out["f64"]["mpc"], ref, tau=False))
