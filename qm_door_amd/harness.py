"""Device-side plumbing around the C ABI that is NOT test logic: torch tensors as device memory for the batched calls (MpcBatch / WbcBatch / make_solver), and the
receding-horizon closed loop of a batch of robots through the library's kernels (Scenario, measurement, GpuBackend) -- what bench.py's steady-state leg, smoke() and
the -m gpu tests all drive.  No oracle here: the CPU restatement lives under oracle/ and is only ever reached from tests/, smoke() and bench.py's cpu_baseline leg.

Closed loop (the controller's steady state: QMController::update / advanceMpc, qm_controllers/src/QMController.cpp:116-157,316-327).  Per MPC cycle (100 Hz,
mpcDesiredFrequency task.info:147): measurement -> centroidal observation (front end) -> event-aligned shooting grid (timeHorizon 1.0 s, dt 0.015: ~67 nodes + one per
mode switch, task.info:79,141) -> previous solution resampled as the initial guess (coldStart false, task.info:143) -> one SQP iteration.  Per WBC tick (1 kHz, ten per
MPC cycle): policy evaluation at the tick's time -> WBC update with the measured state of that tick and inputLast_ carried from the previous tick.  The "robot" is a
plan follower: the measured configuration is the backend's OWN plan at the tick's time plus a seeded, smooth disturbance, the measured velocities are the plan's
finite-difference base rates / planned joint rates plus a disturbance (measurement(), plain numpy)."""
import numpy as np
import torch

from qm_door_amd import abi, api

# "cuda": the product path.  "cpu" is set only by bench.py --emulate (the CPU test of the multi-rank bench: host-emulated kernels, tests/emu).
DEVICE = "cuda"


def dev(a, dtype=None):
    t = torch.as_tensor(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.to(DEVICE).contiguous()


def _sync():
    if DEVICE == "cuda":
        torch.cuda.synchronize()


class MpcBatch:
    """Device-resident inputs/outputs of one batched MPC(+WBC) call."""

    def __init__(self, x0, target_times, target_states, sched_num, sched_times, sched_modes, N, t0=None, warm=None, line_search=True, time_grid=None):
        B = x0.shape[0]
        self.B, self.N = B, N
        f64 = torch.float64
        self.x0 = dev(x0, f64)
        self.t0 = dev(np.zeros(B) if t0 is None else t0, f64)
        self.tt = dev(target_times, f64); self.ts = dev(target_states, f64)
        self.sn = dev(sched_num, torch.int32); self.se = dev(sched_times, f64); self.sm = dev(sched_modes, torch.int32)
        self.wx = dev(warm[0], f64) if warm else None
        self.wu = dev(warm[1], f64) if warm else None
        self.tg = dev(time_grid, f64) if time_grid is not None else None
        self.oT = torch.zeros((B, N + 1), dtype=f64, device=DEVICE); self.oX = torch.zeros((B, N + 1, 30), dtype=f64, device=DEVICE)
        self.oU = torch.zeros((B, N, 30), dtype=f64, device=DEVICE); self.oM = torch.zeros((B, N + 1), dtype=torch.int32, device=DEVICE)
        self.oS = torch.zeros((B, abi.NSTATS), dtype=f64, device=DEVICE)
        K = target_times.shape[1]
        assert target_states.shape == (B, K, 37)
        self.args = api.GpuSolver.mpc_args(B, N, self.x0, self.tt, self.ts, self.sn, self.se, self.sm, self.oT, self.oX, self.oU, self.oM, self.oS, t0=self.t0,
                                           time_grid=self.tg, warm_x=self.wx, warm_u=self.wu, line_search=line_search)

    def results(self):
        _sync()
        return dict(T=self.oT.cpu().numpy(), X=self.oX.cpu().numpy(), U=self.oU.cpu().numpy(), mode=self.oM.cpu().numpy(), stats=self.oS.cpu().numpy())


class WbcBatch:
    def __init__(self, rbd, period, time, input_last, state_desired=None, input_desired=None, mode=None, variant=0, carry=False):
        """carry: a working-set record per instance travels from call to call next to input_last (qmgpu_wbc_args::working_set); off: every call cold"""
        B = rbd.shape[0]
        f64 = torch.float64
        self.rbd = dev(rbd, f64); self.period = dev(period, f64); self.time = dev(time, f64); self.il = dev(input_last, f64)
        self.xd = dev(state_desired, f64) if state_desired is not None else None
        self.ud = dev(input_desired, f64) if input_desired is not None else None
        self.mode = dev(mode, torch.int32) if mode is not None else None
        self.out = torch.zeros((B, 54), dtype=f64, device=DEVICE); self.status = torch.zeros(B, dtype=torch.int32, device=DEVICE)
        self.ws = torch.zeros((B, abi.WBC_STATE_WORDS), dtype=torch.int64, device=DEVICE) if carry else None
        self.args = api.GpuSolver.wbc_args(B, self.rbd, self.period, self.time, self.il, self.out, self.status, self.xd, self.ud, self.mode, variant, working_set=self.ws)

    def results(self):
        _sync()
        r = dict(out=self.out.cpu().numpy(), status=self.status.cpu().numpy(), input_last=self.il.cpu().numpy())
        if self.ws is not None:
            r["working_set"] = self.ws.cpu().numpy().view(np.uint64)
        return r


def make_solver(interface, max_batch, max_nodes, dtype="f64"):
    if DEVICE != "cuda":
        return api.GpuSolver(interface, max_batch, max_nodes, device=0, dtype=dtype)
    s = api.GpuSolver(interface, max_batch, max_nodes, device=torch.cuda.current_device(), dtype=dtype)
    s.set_stream(torch.cuda.current_stream().cuda_stream)
    return s


MPC_PERIOD = 0.01        # mpcDesiredFrequency 100 (task.info:147)
WBC_PERIOD = 0.001       # ros_control update rate of the controller (1 kHz)
HORIZON = 1.0            # timeHorizon (task.info:141)


class Disturbance:
    """a sin(w t + phi) per instance and coordinate on the measured configuration / velocities (smooth, so that the measurement of a plan follower stays
    differentiable); amplitudes 2-4 mm / mrad and 2-4 cm/s / crad/s"""

    def __init__(self, batch, rng):
        self.amp_q = np.c_[np.full((batch, 3), 0.002), np.full((batch, 3), 0.003), np.full((batch, 18), 0.004)] * rng.uniform(0.3, 1.0, (batch, 24))
        self.amp_v = np.c_[np.full((batch, 3), 0.02), np.full((batch, 3), 0.02), np.full((batch, 18), 0.04)] * rng.uniform(0.3, 1.0, (batch, 24))
        self.om = rng.uniform(2.0, 9.0, (batch, 24)); self.ph_q = rng.uniform(0, 2 * np.pi, (batch, 24)); self.ph_v = rng.uniform(0, 2 * np.pi, (batch, 24))

    def dq(self, t):
        return self.amp_q * np.sin(self.om * t + self.ph_q)

    def dv(self, t):
        return self.amp_v * np.sin(self.om * t + self.ph_v)


class Scenario(Disturbance):
    """Seeded batch: initial poses (xy, yaw, joints perturbed), two target knots spanning the run (base + end-effector displaced), stance then trot,
    a start time just before the WBC's start-up branch ends (t = 10 s, HierarchicalWbc.cpp:23), smooth per-coordinate disturbances."""

    def __init__(self, itf, batch, seed=31, t_start=9.5, cycles=100, gait_start=0.12, max_nodes=96, horizon=HORIZON, gait="trot"):
        from qm_door_amd import abi, api
        rng = np.random.default_rng(seed)
        self.B, self.t_start, self.cycles, self.max_nodes, self.horizon = batch, t_start, cycles, max_nodes, horizon
        self.dt = itf.problem.settings.dt
        x_nom = itf.initial_state
        q0 = np.tile(x_nom[6:30], (batch, 1))
        q0[:, 0:2] = rng.uniform(-0.5, 0.5, (batch, 2)); q0[:, 3] = rng.uniform(-0.5, 0.5, batch)
        q0[:, 6:] += rng.uniform(-1, 1, (batch, 18)) * 0.05
        self.q0 = q0
        self.v0 = np.c_[rng.uniform(-0.1, 0.1, (batch, 6)), rng.uniform(-0.2, 0.2, (batch, 18))]     # [w_world, v_lin, dq_j] at the first tick
        t_end = t_start + cycles * MPC_PERIOD + horizon + 0.5
        # targets: knot 0 = the initial pose at t_start, knot 1 = displaced base (and the end-effector with it) 2.5 s later
        yaw = q0[:, 3]
        ts = np.zeros((batch, 2, 37)); tt = np.tile(np.array([t_start, t_start + 2.5]), (batch, 1))
        move = np.c_[rng.uniform(-0.3, 0.3, (batch, 2)), np.zeros(batch), rng.uniform(-0.3, 0.3, batch)]     # dx, dy, dz = 0, dyaw
        for i in range(batch):
            for k in range(2):
                base = np.r_[q0[i, 0:2] + k * move[i, 0:2], x_nom[8], yaw[i] + k * move[i, 3], 0.0, 0.0]
                c, s = np.cos(base[3]), np.sin(base[3])
                ee = np.r_[base[0] + c * 0.6, base[1] + s * 0.6, x_nom[8] + 0.036 + 0.05 * k]
                ts[i, k] = np.r_[np.zeros(6), base, x_nom[12:30], ee, 0.0, 0.0, np.sin(base[3] / 2), np.cos(base[3] / 2)]
        self.tt, self.ts = tt, ts
        # stance until t_start + gait_start, then trot (gait.info) tiled past the last horizon
        g = api.GaitSchedule(lib=itf.lib)
        nev, ev, md = g.mode_schedule(gait, t_start + gait_start, t_start + gait_start, t_end)
        ev = np.array(ev); md = np.array(md, dtype=np.int32)
        assert md[0] == 15 and nev <= abi.MAX_EVENTS          # the tiler puts the default STANCE mode in front of the template's first phase
        self.nev, self.ev, self.md = int(nev), ev, md
        Disturbance.__init__(self, batch, rng)
        self.lib = itf.lib

    def grid(self, t0):
        """event-aligned shooting grid of one MPC cycle (the same for every instance: they share the gait): (N, grid[N + 1])"""
        from qm_door_amd import api
        return api.time_grid_with_events(t0, t0 + self.horizon, self.dt, self.ev[:self.nev], max_nodes=self.max_nodes, lib=self.lib)

    def first_measurement(self):
        return pack_rbd(self.q0 + self.dq(self.t_start), self.v0 + self.dv(self.t_start))


def pack_rbd(q, v):
    """q = [p(3), zyx(3), q_j(18)], v = [w_world(3), v_lin(3), dq_j(18)] -> rbdState[55] (the end-effector pose slots are only read by the target front end: unit quaternion)"""
    B = q.shape[0]
    r = np.zeros((B, 55))
    r[:, 0:3] = q[:, 3:6]; r[:, 3:6] = q[:, 0:3]; r[:, 6:24] = q[:, 6:24]; r[:, 24:48] = v; r[:, 54] = 1.0
    return r


def interp_plan(T, X, U, t):
    """(x, u) of every instance at time t: LinearInterpolation as MPC_MRT_Interface::evaluatePolicy applies it (end values held, U holds its last entry).
    T [B][N+1], X [B][N+1][30], U [B][N][30]."""
    B, N = U.shape[0], U.shape[1]
    lb = (T < t).sum(axis=1)                      # first index with T >= t
    interval = lb - 1
    idx = np.clip(interval, 0, N - 1)
    rows = np.arange(B)
    t0, t1 = T[rows, idx], T[rows, idx + 1]
    length = t1 - t0
    alpha = np.where(length > 2 * np.finfo(float).eps, (t1 - t) / np.where(length == 0, 1.0, length), 1.0)
    alpha = np.where(interval < 0, 1.0, np.where(interval >= N, 0.0, alpha))[:, None]
    x = alpha * X[rows, idx] + (1 - alpha) * X[rows, idx + 1]
    u = alpha * U[rows, np.minimum(idx, N - 1)] + (1 - alpha) * U[rows, np.minimum(idx + 1, N - 1)]
    return x, u


def measurement(sc, plan, t, h=1e-3):
    """rbdState [B][55] at time t of robots that follow `plan` (dict T, X, U): configuration = plan + disturbance, base twist = finite difference of the
    planned base pose over h (ZYX Euler rates mapped to the world angular velocity), joint rates = planned joint velocities, + disturbance."""
    x, u = interp_plan(plan["T"], plan["X"], plan["U"], t)
    xh, _ = interp_plan(plan["T"], plan["X"], plan["U"], t + h)
    q = x[:, 6:30] + sc.dq(t)
    rate = (xh[:, 6:12] - x[:, 6:12]) / h
    z, y = x[:, 9], x[:, 10]
    sz, cz, sy, cy = np.sin(z), np.cos(z), np.sin(y), np.cos(y)
    zd, yd, xd = rate[:, 3], rate[:, 4], rate[:, 5]
    w = np.c_[-sz * yd + cz * cy * xd, cz * yd + sz * cy * xd, zd - sy * xd]
    v = np.c_[w, rate[:, 0:3], u[:, 12:30]] + sc.dv(t)
    return pack_rbd(q, v)


class GpuBackend:
    """The loop through the C ABI: front end, warm start, MPC, policy evaluation and WBC are the library's kernels; buffers stay on the device, the plan is
    downloaded once per cycle for the plan-following measurement."""

    def __init__(self, itf, sc, variant=0, carry=False):
        """carry: the working sets of the hierarchical QP travel from tick to tick next to inputLast_ (qmgpu_wbc_args::working_set)"""
        import torch
        import qm_door_amd.harness as G
        from qm_door_amd import abi
        self.torch, self.G, self.abi, self.itf, self.sc, self.variant = torch, G, abi, itf, sc, variant
        B, Nmax, f64 = sc.B, sc.max_nodes, torch.float64
        self.sol = G.make_solver(itf, B, Nmax)
        z = lambda *shape, dtype=f64: torch.zeros(shape, dtype=dtype, device=G.DEVICE)  # noqa: E731
        self.sets = [dict(S=z(B, abi.NSTATS)) for _ in range(2)]      # two output sets: the previous solution stays resident for the warm start
        self.cur, self.prevN = 0, 0
        self.x0, self.ftt, self.fts = z(B, 30), z(B, 2), z(B, 2, 37)
        self.wx, self.wu = z(B, Nmax + 1, 30), z(B, Nmax, 30)
        self.tt, self.ts = G.dev(sc.tt, f64), G.dev(sc.ts, f64)
        self.sn, self.se, self.sm = G.dev(np.full(B, sc.nev, dtype=np.int32), torch.int32), G.dev(np.tile(sc.ev, (B, 1)), f64), G.dev(np.tile(sc.md, (B, 1)), torch.int32)
        self.kind, self.cmd, self.lastee = z(B, dtype=torch.int32), z(B, 7), z(B, 7)
        self.il = z(B, 30)
        self.ws = z(B, abi.WBC_STATE_WORDS, dtype=torch.int64) if carry else None
        self.xd, self.ud, self.pm = z(B, 30), z(B, 30), z(B, dtype=torch.int32)
        self.out, self.status = z(B, 54), z(B, dtype=torch.int32)
        self.period = G.dev(np.full(B, WBC_PERIOD), f64)
        self.N = 0

    def _views(self, s, N):
        """exact-size output tensors of buffer set s for N nodes (the event-aligned grid changes N from cycle to cycle), cached"""
        cache = s.setdefault("_v", {})
        if N not in cache:
            t, B = self.torch, self.sc.B
            z = lambda *shape, dtype=t.float64: t.zeros(shape, dtype=dtype, device=self.G.DEVICE)  # noqa: E731
            cache[N] = dict(T=z(B, N + 1), X=z(B, N + 1, 30), U=z(B, N, 30), M=z(B, N + 1, dtype=t.int32))
        return cache[N]

    def observe(self, rbd, t):
        """rbdState -> centroidal state on the device (qmgpu_frontend_batch; the target outputs of the front end are not used: the scenario's targets are)"""
        G, t64 = self.G, self.torch.float64
        self.rbd = G.dev(rbd, t64)
        a = self.sol.frontend_args(self.sc.B, self.rbd, G.dev(np.full(self.sc.B, t), t64), self.kind, self.cmd, self.lastee, self.x0, self.ftt, self.fts)
        self.sol.frontend(a)
        return self.x0

    def mpc(self, t0, N, grid):
        from qm_door_amd import api
        G, t64, B = self.G, self.torch.float64, self.sc.B
        nxt = self.cur ^ 1
        o = self._views(self.sets[nxt], N)
        gd = G.dev(np.tile(grid, (B, 1)), t64)
        warm = self.prevN > 0
        if warm:
            p = self._views(self.sets[self.cur], self.prevN)
            wx = self.wx.view(-1)[:B * (N + 1) * 30].view(B, N + 1, 30); wu = self.wu.view(-1)[:B * N * 30].view(B, N, 30)
            self.sol.warm_start(B, self.prevN, p["T"], p["X"], p["U"], N, gd, self.x0, wx, wu)
        args = api.GpuSolver.mpc_args(B, N, self.x0, self.tt, self.ts, self.sn, self.se, self.sm, o["T"], o["X"], o["U"], o["M"], self.sets[nxt]["S"],
                                      t0=G.dev(np.full(B, t0), t64), time_grid=gd, warm_x=wx if warm else None, warm_u=wu if warm else None)
        self.sol.mpc(args)
        self.cur, self.prevN, self.N = nxt, N, N
        self.torch.cuda.synchronize() if G.DEVICE == "cuda" else None
        self.plan = dict(T=o["T"].cpu().numpy(), X=o["X"].cpu().numpy(), U=o["U"].cpu().numpy(), mode=o["M"].cpu().numpy(), stats=self.sets[nxt]["S"].cpu().numpy(),
                         x0=self.x0.cpu().numpy())
        return self.plan

    def tick(self, t, rbd, time):
        """policy evaluation at t + WBC update with the measurement of this tick; inputLast_ stays on the device between ticks"""
        from qm_door_amd import api
        G, t64, B = self.G, self.torch.float64, self.sc.B
        o = self._views(self.sets[self.cur], self.N)
        self.sol.policy_eval(B, self.N, o["T"], o["X"], o["U"], o["M"], G.dev(np.full(B, t), t64), self.xd, self.ud, self.pm)
        rb = G.dev(rbd, t64)
        a = api.GpuSolver.wbc_args(B, rb, self.period, G.dev(np.full(B, time), t64), self.il, self.out, self.status, self.xd, self.ud, self.pm, self.variant, working_set=self.ws)
        self.sol.wbc(a)
        self.torch.cuda.synchronize() if G.DEVICE == "cuda" else None
        return dict(out=self.out.cpu().numpy(), status=self.status.cpu().numpy(), mode=self.pm.cpu().numpy(), input_last=self.il.cpu().numpy(),
                    working_set=None if self.ws is None else self.ws.cpu().numpy().view(np.uint64))


