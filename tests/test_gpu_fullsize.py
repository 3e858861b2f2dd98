"""-m gpu: BASELINE.json configs[1] at full size (256 instances, N = 100) through the C ABI -- size-independent properties.

EVERY instance is compared value by value with the oracle (qmo_cycle_batch_mt: the timing-grade build of the oracle on all host threads,
~1 s for the 256 cycles), plus properties that need no second implementation: batch independence and permutation equivariance (bit-exact),
mode tables (bit-exact), filter acceptance.
"""
import numpy as np
import pytest

import support as S

pytestmark = pytest.mark.gpu

B, N = 256, 100


@pytest.fixture(scope="module")
def solved(interface, oracle):
    import gpu_harness as G
    import torch
    x_nom = interface.initial_state
    x0 = S.perturbed_states(x_nom, B, seed=0)
    tgt = S.nominal_target(oracle, x_nom)
    tt = np.zeros((B, 1)); ts = np.tile(tgt, (B, 1, 1)).copy()
    nev, ev, md = S.trot_schedule(N * interface.problem.settings.dt + 1.0, phase0=0.0)
    # robots in motion on an arbitrary controller tick: measured twist / joint rates, momentum-consistent x0, non-zero inputLast_, t < 10 and t >= 10,
    # the policy evaluated strictly between nodes (support.moving_inputs)
    mv = S.moving_inputs(oracle, x0, interface.problem.settings.dt, seed=21)
    x0, rbd = mv["x0"], mv["rbd"]

    def run(idx):
        n = len(idx)
        sol = G.make_solver(interface, n, N)
        mb = G.MpcBatch(x0[idx], tt[idx], ts[idx], np.full(n, nev, dtype=np.int32), np.tile(ev, (n, 1)), np.tile(md, (n, 1)), N)
        wb = G.WbcBatch(rbd[idx], np.full(n, 0.002), mv["time"][idx], mv["input_last"][idx])
        sol.cycle(mb.args, G.dev(mv["t_eval"][idx], torch.float64), wb.args)
        r = mb.results(); r.update(wb.results())
        return r

    return dict(run=run, full=run(np.arange(B)), x0=x0, tt=tt, ts=ts, sched=(nev, ev, md), rbd=rbd, mv=mv)


def test_finite_converged_and_modes_bit_exact(solved, interface, oracle):
    r = solved["full"]; nev, ev, md = solved["sched"]
    assert np.isfinite(r["X"]).all() and np.isfinite(r["U"]).all() and np.isfinite(r["out"]).all()
    assert (r["stats"][:, 7] == 0).all() and (r["status"] == 0).all()
    dt = interface.problem.settings.dt
    assert np.array_equal(r["T"], np.tile(np.arange(N + 1) * dt, (B, 1)))
    modes = np.array([oracle.node_mode_at(ev[:nev], md[:nev + 1], k * dt) for k in range(N + 1)], dtype=np.int32)
    assert np.array_equal(r["mode"], np.tile(modes, (B, 1)))
    assert np.array_equal(r["X"][:, 0], solved["x0"])      # the initial state is never moved


def test_filter_line_search_acceptance(solved):
    st = solved["full"]["stats"]
    alpha = st[:, 4]
    assert ((alpha > 0) & (alpha <= 1)).all() and np.array_equal(np.log2(alpha), np.round(np.log2(alpha)))
    # an accepted step never leaves the filter's outer box (g_max = 1e-2) unless it reduced the violation it started from
    assert ((st[:, 3] <= 1e-2) | (st[:, 3] < st[:, 1])).all()


def test_batch_independence_and_permutation(solved):
    full = solved["full"]
    idx = np.array([255, 17, 0, 128])
    sub = solved["run"](idx)
    for key in ("X", "U", "mode", "stats", "out"):
        assert np.array_equal(sub[key], full[key][idx]), key


def test_every_instance_against_oracle(solved, interface):
    """All 256 MPC solves + policy evaluations + WBC updates against the oracle's batch entry: X, U, tau 1e-6 rel-inf, modes / step length / step type exact."""
    full = solved["full"]; nev, ev, md = solved["sched"]
    mv = solved["mv"]
    ref = S.Oracle(interface.problem, fast=True).cycle_batch(N, solved["x0"], solved["tt"], solved["ts"], nev, ev, md, rbd=solved["rbd"], t_eval=mv["t_eval"], time=mv["time"],
                                                             input_last=mv["input_last"])
    rep = S.parity_report("configs1_256xN100_trot_moving", full, ref)
    S.assert_parity(rep)
    assert S.rel_inf(full["input_last"], ref["input_last"]).max() <= 1e-9          # inputLast_ carried to the next tick: the evaluated policy's input
    assert (mv["time"] < 10).sum() > 20 and (mv["time"] >= 10).sum() > 100 and np.abs(mv["rbd"][:, 24:48]).min(axis=1).max() > 0
    # with the active-set polish after the interior point both implementations land on the same vertex of every level's QP and agree far below
    # the north_star tolerance
    assert rep["tau"]["median"] <= 1e-9, rep


def test_fast_and_checker_builds_of_the_oracle_agree_on_a_sample(solved, interface, oracle):
    """The whole-batch comparisons use the timing-grade build of the oracle (-O3, FMA contraction, structured derivative route); the checker build
    (-O2, no contraction, Dual<60>) is the one the CPU suite pins.  Same sources, results within 1e-9 of each other."""
    idx = np.array([0, 101, 255])
    nev, ev, md = solved["sched"]
    mv = solved["mv"]
    kw = dict(rbd=solved["rbd"][idx], t_eval=mv["t_eval"][idx], time=mv["time"][idx], input_last=mv["input_last"][idx])
    a = S.Oracle(interface.problem, fast=True).cycle_batch(N, solved["x0"][idx], solved["tt"][idx], solved["ts"][idx], nev, ev, md, **kw)
    b = oracle.cycle_batch(N, solved["x0"][idx], solved["tt"][idx], solved["ts"][idx], nev, ev, md, **kw)
    for k in ("X", "U", "out"):
        assert S.rel_inf(a[k], b[k]).max() <= 1e-9, k
    assert np.array_equal(a["mode"], b["mode"]) and np.array_equal(a["stats"][:, 4:6], b["stats"][:, 4:6])
