"""Closed loop, test side: the oracle backend and the lock-step comparison of two backends.  The scenario, the plan-following measurement and the GPU backend are
in qm_door_amd/harness.py (bench.py's steady-state leg drives them without touching tests/)."""
import numpy as np

from qm_door_amd.harness import *  # noqa: F401,F403
from qm_door_amd.harness import MPC_PERIOD, WBC_PERIOD, HORIZON, measurement  # noqa: F401


class OracleBackend:
    """The same loop on the CPU oracle (tests only)."""

    def __init__(self, oracle, sc, variant=0, carry=False, other_build=None):
        """carry: the working sets of the hierarchical QP travel from tick to tick next to inputLast_ (qmgpu_wbc_args::working_set on the other side).
        other_build: the same restatement compiled differently (support.Oracle(fast=...)): sensitivity() also reports how far the checker's two builds are apart on a tick"""
        self.o, self.sc, self.variant, self.other = oracle, sc, variant, other_build
        self.prev = None
        self.il = np.zeros((sc.B, 30))
        self.ws = np.zeros((sc.B, 48), dtype=np.uint64) if carry else None

    def observe(self, rbd, t):
        z7 = np.zeros(7)
        self.x0 = np.stack([self.o.frontend(rbd[i], t, 0, z7, z7)[0] for i in range(self.sc.B)])
        return self.x0

    def mpc(self, t0, N, grid):
        sc, B = self.sc, self.sc.B
        g = np.tile(grid, (B, 1))
        warm = None
        if self.prev is not None:
            warm = self.o.warm_start_batch(self.prev["T"], self.prev["X"], self.prev["U"], g, self.x0)
        r = self.o.cycle_batch(N, self.x0, sc.tt, sc.ts, sc.nev, sc.ev, sc.md, t0=np.full(B, t0), time_grid=g, warm=warm)
        self.plan = dict(T=g, X=r["X"], U=r["U"], mode=r["mode"], stats=r["stats"], x0=self.x0)
        self.prev = self.plan
        return self.plan

    def tick(self, t, rbd, time):
        xd, ud, md = self.o.policy_eval_batch(self.plan["T"], self.plan["X"], self.plan["U"], self.plan["mode"], t)
        self.last = (xd, ud, np.array(rbd, dtype=np.float64), md, time, self.il.copy())
        self.ws_before = None if self.ws is None else self.ws.copy()
        self.o.set_working_set(self.ws)
        try:
            w = self.o.wbc_batch(xd, ud, rbd, md, WBC_PERIOD, time, self.il, self.variant)
        finally:
            self.o.set_working_set(None)
        self.last_out = w["out"]
        self.il = w["input_last"]
        return dict(out=w["out"], status=w["status"], mode=md, input_last=self.il, attempts=w["attempts"], polished=w["polished"], iterations=w["iterations"],
                    working_set=None if self.ws is None else self.ws.copy())

    def sensitivity(self, idx, eps=1e-9, draws=5, seed=5):
        """How far the ORACLE's own torques of the last tick move (rel-inf, per instance of idx) when its inputs -- desired state and input, measurement, inputLast_ --
        are perturbed by eps relative (a few seeded directions): the conditioning of the tick's cascade, measured on the checker alone.  Two backends whose MPC plans
        agree to 1e-13 cannot agree on the torques by better than this scaled to 1e-13 -- or, where a level sits on a discrete decision (a direction at the
        exclusion floor of the factorisation: solved for, or left where the interior point put it), by better than the jump itself."""
        from support import rel_inf
        xd, ud, rbd, md, time, il = self.last
        idx = np.asarray(idx, dtype=np.int64)
        rng = np.random.default_rng(seed)
        base = self.last_out[idx, 36:]
        worst = np.zeros(len(idx))
        for _ in range(draws):
            p = lambda a: a[idx] * (1 + eps * rng.uniform(-1, 1, a[idx].shape))  # noqa: E731
            ws = None if self.ws_before is None else np.ascontiguousarray(self.ws_before[idx])     # (the same carried working sets as the tick itself)
            self.o.set_working_set(ws)
            try:
                w = self.o.wbc_batch(p(xd), p(ud), p(rbd), md[idx], WBC_PERIOD, time, p(il), self.variant)
            finally:
                self.o.set_working_set(None)
            worst = np.maximum(worst, rel_inf(w["out"][:, 36:], base))
        if self.other is not None:      # the checker against ITSELF, compiled differently (-O3 -march=x86-64-v3 with other contractions): a tick on which its two builds part cannot pin a third implementation any better
            w = self.other.wbc_batch(xd[idx], ud[idx], rbd[idx], md[idx], WBC_PERIOD, time, il[idx], self.variant)
            worst = np.maximum(worst, rel_inf(w["out"][:, 36:], base))
        return worst


def passes_of(ws):
    """(interior-point + active-set iterations summed over the solves of the last tick, number of refuted guesses) per instance from the working-set records
    (words 13 / 14: one byte per solve, bit 7 = the carried guess was refuted; include/qmgpu.h)"""
    b = np.ascontiguousarray(ws[:, 13:15]).view(np.uint8).reshape(ws.shape[0], 16)
    return (b & 127).astype(np.int64).sum(axis=1), (b >> 7).astype(np.int64).sum(axis=1)


def run_lockstep(sc, a, b, ticks=10, on_cycle=None, offenders=None, tol=1e-6):
    """Both backends through sc.cycles MPC cycles x `ticks` WBC ticks, each on its OWN plan and measurements; returns per-cycle deviations (rel-inf per
    instance, worst over the batch) and the agreement of the discrete outcomes.  `offenders` (a list) receives one record per (cycle, tick, instance)
    whose torques deviate by more than `tol` or whose WBC status words are non-zero / differ, with the second backend's diagnostics when it has any
    (the oracle's re-solve attempts, polish flags and iteration counts per level)."""
    from support import rel_inf
    rows = []
    rbd = [sc.first_measurement(), sc.first_measurement()]
    for k in range(sc.cycles):
        t0 = sc.t_start + k * MPC_PERIOD
        N, grid = sc.grid(t0)
        plans = []
        for s, be in enumerate((a, b)):
            be.observe(rbd[s], t0)
            plans.append(be.mpc(t0, N, grid))
        pa, pb = plans
        row = dict(cycle=k, N=int(N), x0=float(rel_inf(pa["x0"], pb["x0"]).max()), X=float(rel_inf(pa["X"], pb["X"]).max()), U=float(rel_inf(pa["U"], pb["U"]).max()),
                   modes_equal=bool(np.array_equal(pa["mode"], pb["mode"])), alpha_differs=int((pa["stats"][:, 4] != pb["stats"][:, 4]).sum()),
                   step_type_differs=int((pa["stats"][:, 5] != pb["stats"][:, 5]).sum()), alpha_min=float(pb["stats"][:, 4].min()),
                   riccati_status=[int((pa["stats"][:, 7] != 0).sum()), int((pb["stats"][:, 7] != 0).sum())], tau=0.0, xacc=0.0, tau_legs=0.0, tau_arm=0.0, wbc_status=[0, 0], policy_mode_differs=0,
                   working_sets_differ=0, wbc_passes=[0, 0], wbc_passes_max=[0, 0], guesses_refuted=[0, 0])
        for j in range(ticks):
            t = t0 + j * WBC_PERIOD
            ws = []
            for s, be in enumerate((a, b)):
                if not (k == 0 and j == 0):
                    rbd[s] = measurement(sc, plans[s], t)
                ws.append(be.tick(t, rbd[s], t))
            wa, wb = ws
            row["tau"] = max(row["tau"], float(rel_inf(wa["out"][:, 36:], wb["out"][:, 36:]).max()))
            row["xacc"] = max(row["xacc"], float(rel_inf(wa["out"][:, :36], wb["out"][:, :36]).max()))
            # per block, each with its own norm: the separated-system plugin commands the LEG torques only (QMController.cpp:428-431)
            e_legs, e_arm = rel_inf(wa["out"][:, 36:48], wb["out"][:, 36:48]), rel_inf(wa["out"][:, 48:54], wb["out"][:, 48:54])
            row["tau_legs"] = max(row["tau_legs"], float(e_legs.max())); row["tau_arm"] = max(row["tau_arm"], float(e_arm.max()))
            row["wbc_status"] = [row["wbc_status"][0] + int((wa["status"] != 0).sum()), row["wbc_status"][1] + int((wb["status"] != 0).sum())]
            row["policy_mode_differs"] += int((wa["mode"] != wb["mode"]).sum())
            if wa.get("working_set") is not None and wb.get("working_set") is not None:     # the carried solver state: the rows every solve ended pinned, the same on both sides
                row["working_sets_differ"] += int((wa["working_set"][:, :13] != wb["working_set"][:, :13]).any(axis=1).sum())
            for s_, w_ in enumerate(ws):
                if w_.get("working_set") is not None:
                    per, ref_ = passes_of(w_["working_set"])
                    row["wbc_passes"][s_] += int(per.sum()); row["wbc_passes_max"][s_] = max(row["wbc_passes_max"][s_], int(per.max())); row["guesses_refuted"][s_] += int(ref_.sum())
            if offenders is not None:
                e = np.maximum(rel_inf(wa["out"][:, 36:], wb["out"][:, 36:]), np.maximum(e_legs, e_arm))     # (the worst of the 18-wide norm and the two blocks' own)
                bad = np.nonzero((e > tol) | (wa["status"] != 0) | (wb["status"] != 0))[0]
                sens = b.sensitivity(bad) if len(bad) and hasattr(b, "sensitivity") else None
                for n_, i in enumerate(bad):
                    rec = dict(cycle=k, tick=j, instance=int(i), time=float(t), tau_dev=float(e[i]), tau_legs_dev=float(e_legs[i]), tau_arm_dev=float(e_arm[i]),
                               tau_legs_abs_dev=float(np.abs(wa["out"][i, 36:48] - wb["out"][i, 36:48]).max()), status=[int(wa["status"][i]), int(wb["status"][i])], mode=int(wb["mode"][i]))
                    for key in ("attempts", "polished", "iterations"):
                        if key in wb:
                            rec[key] = wb[key][i].tolist()
                    if sens is not None:
                        rec["oracle_tau_move_under_1e-9_input_perturbation"] = float(sens[n_])
                    offenders.append(rec)
        # the measurement the next cycle starts from: each loop's own plan at the next MPC time
        for s in range(2):
            rbd[s] = measurement(sc, plans[s], t0 + MPC_PERIOD)
        rows.append(row)
        if on_cycle:
            on_cycle(row)
    return rows
