"""Shared test plumbing: oracle (ctypes), emulated kernel library, seeded scenario generators.

The oracle (oracle/libqm_oracle.so) is TEST INFRASTRUCTURE: it is only ever loaded from here, from
__graft_entry__.smoke() and from bench.py's cpu_baseline leg.  PARITY UNPINNED (see oracle/qmo_core.h).
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from qm_door_amd import abi  # noqa: E402

ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_LIB = os.path.join(ORACLE_DIR, "libqm_oracle.so")
ORACLE_FAST_LIB = os.path.join(ORACLE_DIR, "libqm_oracle_fast.so")
EMU_DIR = os.path.join(ROOT, "tests", "emu")
EMU_LIB = os.path.join(EMU_DIR, "build", "libqmgpu_emu.so")


def build_oracle():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])
    return ORACLE_LIB


def build_emu():
    """g++ build of the kernel sources against tests/emu/simt_emu.h (host threads instead of lanes)."""
    csrc = os.path.join(ROOT, "qm_door_amd", "csrc")
    srcs = [os.path.join(csrc, "qmgpu_api.hip"), os.path.join(csrc, "host", "host_config.cpp")]
    deps = srcs + [os.path.join(csrc, "qmgpu_mpc32.hip"), os.path.join(csrc, "mpc32.h")] + [os.path.join(ROOT, "qm_door_amd", "csrc", "kernels", f) for f in os.listdir(os.path.join(ROOT, "qm_door_amd", "csrc", "kernels"))]
    deps += [os.path.join(EMU_DIR, "simt_emu.h"), os.path.join(ROOT, "include", "qmgpu.h")]
    if os.path.exists(EMU_LIB) and all(os.path.getmtime(EMU_LIB) >= os.path.getmtime(d) for d in deps):
        return EMU_LIB
    os.makedirs(os.path.dirname(EMU_LIB), exist_ok=True)
    # as the product: the kernel sources twice, fp64 (everything) and fp32 (the MPC chain in namespace qmk32)
    base = ["g++", "-std=c++20", "-O2", "-fPIC", "-DQMGPU_HOST_EMULATION", "-I", EMU_DIR, "-x", "c++", "-c"]
    objs = [os.path.join(os.path.dirname(EMU_LIB), n) for n in ("api.o", "host.o", "mpc32.o")]
    procs = [subprocess.Popen(base + [srcs[0], "-o", objs[0]]), subprocess.Popen(base + [srcs[1], "-o", objs[1]]),
             subprocess.Popen(base + ["-DQM_REAL=float", "-Dqmk=qmk32", os.path.join(csrc, "qmgpu_mpc32.hip"), "-o", objs[2]])]
    for pr in procs:
        if pr.wait() != 0:
            raise subprocess.CalledProcessError(pr.returncode, "g++ (emulation build)")
    subprocess.check_call(["g++", "-shared", "-o", EMU_LIB, *objs, "-lpthread"])
    return EMU_LIB


def p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Oracle:
    def __init__(self, problem, fast=False):
        """fast=True: the timing-grade build of the same sources (oracle/Makefile: -O3, structured derivatives, phase timers)"""
        build_oracle()
        self.lib = C.CDLL(ORACLE_FAST_LIB if fast else ORACLE_LIB)
        self.P = problem
        self.lib.qmo_time_cycles.restype = C.c_double

    def flow_map(self, x, u):
        f = np.zeros(30)
        self.lib.qmo_flow_map(C.byref(self.P), p(x), p(u), p(f))
        return f

    def flow_map_lin(self, x, u):
        f, A, B = np.zeros(30), np.zeros((30, 30)), np.zeros((30, 30))
        self.lib.qmo_flow_map_lin(C.byref(self.P), p(x), p(u), p(f), p(A), p(B))
        return f, A, B

    def kinematics(self, x, u):
        fp, fv, ee, eq, com = np.zeros(12), np.zeros(12), np.zeros(3), np.zeros(4), np.zeros(3)
        self.lib.qmo_kinematics(C.byref(self.P), p(x), p(u), p(fp), p(fv), p(ee), p(eq), p(com))
        return fp.reshape(4, 3), fv.reshape(4, 3), ee, eq, com

    def centroidal_matrix(self, q):
        A = np.zeros((6, 24))
        self.lib.qmo_centroidal_matrix(C.byref(self.P), p(q), p(A))
        return A

    def input_weight(self):
        R = np.zeros((30, 30))
        self.lib.qmo_input_weight(C.byref(self.P), p(R))
        return R

    def mode_at(self, ev, modes, t):
        ev = np.ascontiguousarray(ev, dtype=np.float64); modes = np.ascontiguousarray(modes, dtype=np.int32)
        self.lib.qmo_mode_at.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_double]
        return self.lib.qmo_mode_at(len(ev), p(ev), p(modes), t)

    def structured_vs_dual60(self, x, u, contact_stiffness=0.0, env=None):
        """max |difference| of the oracle's two derivative routes at (x, u), optionally with the force-tracking contact on"""
        self.lib.qmo_set_flow_contact.argtypes = [C.c_double, C.c_void_p]
        self.lib.qmo_set_flow_contact(contact_stiffness, p(None if env is None else np.ascontiguousarray(env, dtype=np.float64)))
        self.lib.qmo_structured_vs_dual60.restype = C.c_double
        self.lib.qmo_structured_vs_dual60.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        d = self.lib.qmo_structured_vs_dual60(C.byref(self.P), p(np.ascontiguousarray(x)), p(np.ascontiguousarray(u)))
        self.lib.qmo_set_flow_contact(0.0, None)
        return d

    def time_split(self):
        out = np.zeros(5)
        self.lib.qmo_time_split.argtypes = [C.c_void_p]
        self.lib.qmo_time_split(p(out))
        return dict(zip(("lq", "riccati", "linesearch", "wbc_model", "wbc_qp"), out.tolist()))

    def set_ee_contact_ref(self, ref):
        """force tracking: contact reference [K][6] (f_ref, p_env per target knot) used by the following mpc / lq calls; None clears it"""
        self._contact_ref = None if ref is None else np.ascontiguousarray(ref, dtype=np.float64)
        self.lib.qmo_set_ee_contact_ref.argtypes = [C.c_void_p]
        self.lib.qmo_set_ee_contact_ref(p(self._contact_ref))

    def set_wbc_ee_force(self, f):
        self._wbc_force = None if f is None else np.ascontiguousarray(f, dtype=np.float64)
        self.lib.qmo_set_wbc_ee_force.argtypes = [C.c_void_p]
        self.lib.qmo_set_wbc_ee_force(p(self._wbc_force))

    def node_mode_at(self, ev, modes, t):
        """mode of a shooting node at time t (a node on an event time takes the post-event mode)"""
        ev = np.ascontiguousarray(ev, dtype=np.float64); modes = np.ascontiguousarray(modes, dtype=np.int32)
        self.lib.qmo_node_mode_at.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_double]
        return self.lib.qmo_node_mode_at(len(ev), p(ev), p(modes), t)

    def swing_reference(self, nev, ev, modes, t):
        zp, zv = np.zeros(4), np.zeros(4)
        self.lib.qmo_swing_reference.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p]
        self.lib.qmo_swing_reference(C.byref(self.P), nev, p(ev), p(modes), t, p(zp), p(zv))
        return zp, zv

    def lq_node(self, t, dt, x, u, xnext, terminal, nev, ev, modes, ttimes, tstates):
        A, B, Q, R = (np.zeros((30, 30)) for _ in range(4))
        b, q, r = (np.zeros(30) for _ in range(3))
        Cm, Dm, e = np.zeros((16, 30)), np.zeros((16, 30)), np.zeros(16)
        nc = C.c_int32(0); cost = C.c_double(0)
        self.lib.qmo_lq_node.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                         C.c_void_p, C.c_void_p] + [C.c_void_p] * 10 + [C.c_void_p, C.c_void_p]
        u_ = u if u is not None else np.zeros(30)
        xn = xnext if xnext is not None else x
        self.lib.qmo_lq_node(C.byref(self.P), t, dt, p(x), p(u_), p(xn), int(terminal), nev, p(ev), p(modes), len(ttimes), p(ttimes), p(tstates), p(A), p(B), p(b), p(Q),
                             p(R), p(q), p(r), p(Cm), p(Dm), p(e), C.byref(nc), C.byref(cost))
        n = nc.value
        return dict(A=A, B=B, b=b, Q=Q, R=R, q=q, r=r, C=Cm[:n], D=Dm[:n], e=e[:n], nc=n, cost=cost.value)

    def mpc_solve(self, N, t0, x0, ttimes, tstates, nev, ev, modes, warm=None, line_search=True, time_grid=None):
        T, X, U, M, st = np.zeros(N + 1), np.zeros((N + 1, 30)), np.zeros((N, 30)), np.zeros(N + 1, dtype=np.int32), np.zeros(abi.NSTATS)
        self.lib.qmo_mpc_solve.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 5
        wx = p(np.ascontiguousarray(warm[0])) if warm else None
        wu = p(np.ascontiguousarray(warm[1])) if warm else None
        rc = self.lib.qmo_mpc_solve(C.byref(self.P), N, t0, p(x0), p(time_grid), len(ttimes), p(ttimes), p(tstates), nev, p(ev), p(modes), wx, wu, int(line_search), p(T),
                                    p(X), p(U), p(M), p(st))
        return dict(status=rc, T=T, X=X, U=U, mode=M, stats=st)

    def performance(self, N, tgrid, x0, X, U, ttimes, tstates, nev, ev, modes):
        """(merit, constraint violation) of a trajectory: sum of the dt-scaled costs, sqrt(dt |defects|^2 + dt |equalities|^2)"""
        m, v = C.c_double(0), C.c_double(0)
        self.lib.qmo_performance.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                             C.c_void_p, C.c_void_p]
        self.lib.qmo_performance(C.byref(self.P), N, p(np.ascontiguousarray(tgrid)), p(x0), p(np.ascontiguousarray(X)), p(np.ascontiguousarray(U)), len(ttimes), p(ttimes),
                                 p(tstates), nev, p(ev), p(modes), C.byref(m), C.byref(v))
        return m.value, v.value

    def ddp_solve(self, N, t0, x0, ttimes, tstates, nev, ev, modes, warm_u=None, time_grid=None, warm_x=None):
        T, X, U, M, st = np.zeros(N + 1), np.zeros((N + 1, 30)), np.zeros((N, 30)), np.zeros(N + 1, dtype=np.int32), np.zeros(abi.NSTATS)
        self.lib.qmo_ddp_solve.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_void_p] + [C.c_void_p] * 5
        wu = p(np.ascontiguousarray(warm_u)) if warm_u is not None else None
        wx = p(np.ascontiguousarray(warm_x)) if warm_x is not None else None
        rc = self.lib.qmo_ddp_solve(C.byref(self.P), N, t0, p(x0), p(time_grid), len(ttimes), p(ttimes), p(tstates), nev, p(ev), p(modes), wx, wu, p(T), p(X), p(U), p(M), p(st))
        return dict(status=rc, T=T, X=X, U=U, mode=M, stats=st)

    def wbc_update(self, x_des, u_des, rbd, mode, period, time, input_last, variant=0):
        out = np.zeros(54)
        il = np.array(input_last, dtype=np.float64)
        self.lib.qmo_wbc_update.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_void_p, C.c_void_p]
        st = self.lib.qmo_wbc_update(C.byref(self.P), variant, p(x_des), p(u_des), p(rbd), int(mode), period, time, p(il), p(out))
        return st, out, il

    def wbc_model(self, x_des, u_des, rbd, period, input_last):
        il = np.array(input_last, dtype=np.float64)
        o = dict(M=np.zeros((24, 24)), nle=np.zeros(24), J=np.zeros((12, 24)), dJ=np.zeros((12, 24)), baseJ=np.zeros((6, 24)), baseDJ=np.zeros((6, 24)),
                 armJ=np.zeros((6, 24)), armDJ=np.zeros((6, 24)), qv=np.zeros((4, 24)), baseAcc=np.zeros(6), feet=np.zeros((4, 4, 3)), ee=np.zeros(33))
        self.lib.qmo_wbc_model.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double] + [C.c_void_p] * 13
        self.lib.qmo_wbc_model(C.byref(self.P), p(x_des), p(u_des), p(rbd), period, p(il), p(o["M"]), p(o["nle"]), p(o["J"]), p(o["dJ"]), p(o["baseJ"]), p(o["baseDJ"]),
                               p(o["armJ"]), p(o["armDJ"]), p(o["qv"]), p(o["baseAcc"]), p(o["feet"]), p(o["ee"]))
        return o

    def qp_solve(self, H, c, D, f):
        n, m = H.shape[0], D.shape[0]
        z = np.zeros(n); res = C.c_double(0)
        self.lib.qmo_qp_solve.argtypes = [C.c_int, C.c_int] + [C.c_void_p] * 5 + [C.c_void_p]
        it = self.lib.qmo_qp_solve(n, m, p(np.ascontiguousarray(H)), p(np.ascontiguousarray(c)), p(np.ascontiguousarray(D)), p(np.ascontiguousarray(f)), p(z), C.byref(res))
        return it, z, res.value

    def kernel_full_piv_lu(self, A):
        """oracle/qmo_core.h kernelFullPivLU: (basis [cols][dimker], free columns, pivot positions [(row, column), ...])"""
        A = np.ascontiguousarray(A, dtype=np.float64)
        r, c = A.shape
        ker, free, seq, npiv = np.zeros((c, c)), np.zeros(c, dtype=np.int32), np.zeros(2 * min(r, c), dtype=np.int32), C.c_int32(0)
        self.lib.qmo_kernel_full_piv_lu.argtypes = [C.c_int, C.c_int] + [C.c_void_p] * 5
        dim = self.lib.qmo_kernel_full_piv_lu(r, c, p(A), p(ker), p(free), p(seq), C.byref(npiv))
        return ker[:, :dim].copy(), free[:dim].tolist(), [(int(seq[2 * k]), int(seq[2 * k + 1])) for k in range(npiv.value)]

    def wbc_level(self, level, x_des, u_des, rbd, mode, period, time, input_last, variant=0):
        dims = np.zeros(3, dtype=np.int32)
        H, c, D, f, sol, xl = np.zeros(128 * 128), np.zeros(128), np.zeros(160 * 128), np.zeros(160), np.zeros(128), np.zeros(36)
        self.lib.qmo_wbc_levels.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_void_p, C.c_int] + [C.c_void_p] * 7
        it = self.lib.qmo_wbc_levels(C.byref(self.P), variant, p(x_des), p(u_des), p(rbd), int(mode), period, time, p(np.array(input_last, dtype=np.float64)), level,
                                     p(dims), p(H), p(c), p(D), p(f), p(sol), p(xl))
        nz, rows, nd = (int(v) for v in dims)
        return dict(iters=it, H=H[:nz * nz].reshape(nz, nz), c=c[:nz], D=D[:rows * nz].reshape(rows, nz), f=f[:rows], sol=sol[:nz], x=xl, num_dec=nd)

    def wbc_task(self, level, x_des, u_des, rbd, mode, period, time, input_last, variant=0):
        dims = np.zeros(2, dtype=np.int32)
        A, b, D, f = np.zeros((64, 36)), np.zeros(64), np.zeros((64, 36)), np.zeros(64)
        self.lib.qmo_wbc_task.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_void_p, C.c_int] + [C.c_void_p] * 5
        self.lib.qmo_wbc_task(C.byref(self.P), variant, p(x_des), p(u_des), p(rbd), int(mode), period, time, p(np.array(input_last, dtype=np.float64)), level, p(dims),
                              p(A), p(b), p(D), p(f))
        ra, rd = int(dims[0]), int(dims[1])
        return dict(A=A[:ra], b=b[:ra], D=D[:rd], f=f[:rd])

    def frontend(self, rbd, time, kind, cmd, last_ee, yaw_last=None, feet_height=0.0, arm_dist=0.6, start=(-2.0, 0.0, 0.0)):
        x0, tt, ts = np.zeros(30), np.zeros(2), np.zeros((2, 37))
        le = np.array(last_ee, dtype=np.float64)
        self.lib.qmo_frontend.restype = None
        self.lib.qmo_frontend.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_int, C.c_double, C.c_int, C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_double,
                                          C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]
        self.lib.qmo_frontend(C.byref(self.P), p(np.ascontiguousarray(rbd)), time, int(yaw_last is not None), float(yaw_last or 0.0), int(kind), p(np.ascontiguousarray(cmd)),
                              p(le), feet_height, arm_dist, start[0], start[1], start[2], p(x0), p(tt), p(ts))
        return x0, tt, ts, le

    def cycle_batch(self, N, x0, ttimes, tstates, nev, ev, modes, t0=None, contact=None, line_search=True, t_eval=None, rbd=None, period=None, time=None,
                    input_last=None, ee_force=None, variant=0, threads=None, time_grid=None, warm=None):
        """qmo_cycle_batch_mt: the whole control cycle (MPC solve -> policy evaluation at t_eval -> WBC update) of EVERY instance of a batch on
        `threads` host threads (default: the CPUs this process may use).  Shapes as the C ABI: x0 [B][30], ttimes [B][K], tstates [B][K][37],
        nev [B] (or a scalar), ev [B][MAX_EVENTS] / modes [B][MAX_EVENTS + 1] (or one schedule for all).  rbd None: MPC only."""
        B = x0.shape[0]
        f64 = lambda a: np.ascontiguousarray(a, dtype=np.float64)  # noqa: E731
        x0 = f64(x0); ttimes = f64(ttimes); tstates = f64(tstates)
        K = ttimes.shape[1]
        assert tstates.shape == (B, K, 37)
        nev = np.ascontiguousarray(np.broadcast_to(np.asarray(nev, dtype=np.int32), (B,)))
        ev = f64(np.broadcast_to(ev, (B, abi.MAX_EVENTS))); modes = np.ascontiguousarray(np.broadcast_to(modes, (B, abi.MAX_EVENTS + 1)), dtype=np.int32)
        t0 = None if t0 is None else f64(t0)
        contact = None if contact is None else f64(contact)
        X, U, M, st = np.zeros((B, N + 1, 30)), np.zeros((B, N, 30)), np.zeros((B, N + 1), dtype=np.int32), np.zeros((B, abi.NSTATS))
        pol, pm, wout, wst = np.zeros((B, 60)), np.zeros(B, dtype=np.int32), np.zeros((B, 54)), np.zeros(B, dtype=np.int32)
        il = None
        if rbd is not None:
            rbd = f64(rbd); t_eval = f64(np.broadcast_to(0.0 if t_eval is None else t_eval, (B,)))
            period = f64(np.broadcast_to(0.002 if period is None else period, (B,))); time = f64(np.broadcast_to(20.0 if time is None else time, (B,)))
            il = np.zeros((B, 30)) if input_last is None else np.array(input_last, dtype=np.float64)
            ee_force = None if ee_force is None else f64(ee_force)
        threads = host_threads() if threads is None else threads
        time_grid = None if time_grid is None else f64(np.broadcast_to(time_grid, (B, N + 1)))
        wx = None if warm is None else f64(warm[0]); wu = None if warm is None else f64(warm[1])
        assert warm is None or (wx.shape == (B, N + 1, 30) and wu.shape == (B, N, 30))
        fn = self.lib.qmo_cycle_batch_warm_mt
        fn.restype = C.c_int
        fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 11 + [C.c_int] + [C.c_void_p] * 6 + [C.c_int] + [C.c_void_p] * 8
        bad = fn(C.byref(self.P), B, N, K, int(threads), p(t0), p(x0), p(time_grid), p(wx), p(wu), p(ttimes), p(tstates), p(contact), p(nev), p(ev), p(modes), int(line_search),
                 p(t_eval), p(rbd), p(period), p(time), p(il), p(ee_force), int(variant), p(X), p(U), p(M), p(st), p(pol), p(pm), p(wout), p(wst))
        T = time_grid if time_grid is not None else (np.zeros(B) if t0 is None else t0)[:, None] + self.P.settings.dt * np.arange(N + 1)[None, :]
        out = dict(X=X, U=U, mode=M, stats=st, failed=bad, T=T)
        if rbd is not None:
            out.update(x_des=pol[:, :30], u_des=pol[:, 30:], policy_mode=pm, out=wout, status=wst, input_last=il)
        return out

    def set_working_set(self, ws):
        """ws: np.uint64 [B][48] (kept alive by the caller) or None.  The solver state the WBC entry points of this library carry from tick to tick (instance i of a batch
        call uses block i): the counterpart of qmgpu_wbc_args::working_set.  None: every tick cold."""
        assert ws is None or (ws.dtype == np.uint64 and ws.flags["C_CONTIGUOUS"] and ws.shape[-1] == 48)
        self.lib.qmo_set_wbc_working_set.argtypes = [C.c_void_p]
        self.lib.qmo_set_wbc_working_set(p(ws))
        self._ws_keep = ws

    def set_experiment(self, lower_level_start=0.5, no_interior_point=False, trace=False, no_warm_start=False, literal_reg_max_n=12, own_interior_point=3, ipm_start_delta=0.0):
        """experiment knobs of the WBC restatement (process-wide for this library; the defaults are the product's algorithm).  Both change only the PATH to the vertex
        every level ends at: another starting value of the interior point that runs in front of the active-set method, or no interior point at all (the active-set
        method cold from z = 0)."""
        self.lib.qmo_set_experiment.argtypes = [C.c_int, C.c_double]
        self.lib.qmo_set_experiment(0, float(lower_level_start)); self.lib.qmo_set_experiment(3, float(bool(no_interior_point))); self.lib.qmo_set_experiment(9, float(bool(trace)))
        self.lib.qmo_set_experiment(8, float(bool(no_warm_start))); self.lib.qmo_set_experiment(12, float(literal_reg_max_n)); self.lib.qmo_set_experiment(14, float(int(own_interior_point))); self.lib.qmo_set_experiment(15, float(ipm_start_delta))

    def wbc_batch(self, x_des, u_des, rbd, mode, period, time, input_last, variant=0, ee_force=None, threads=None):
        """qmo_wbc_batch_mt: the WBC update of EVERY instance of a batch on `threads` host threads; returns out [B][54], status [B], input_last [B][30] (updated)"""
        B = rbd.shape[0]
        f64 = lambda a: np.ascontiguousarray(a, dtype=np.float64)  # noqa: E731
        x_des, u_des, rbd = f64(x_des), f64(u_des), f64(rbd)
        mode = np.ascontiguousarray(np.broadcast_to(mode, (B,)), dtype=np.int32)
        period = f64(np.broadcast_to(period, (B,))); time = f64(np.broadcast_to(time, (B,)))
        il = np.array(input_last, dtype=np.float64)
        ee_force = None if ee_force is None else f64(ee_force)
        out, st, diag = np.zeros((B, 54)), np.zeros(B, dtype=np.int32), np.zeros((B, 8), dtype=np.int32)
        fn = self.lib.qmo_wbc_batch_mt
        fn.restype = C.c_int
        fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 11
        fn(C.byref(self.P), B, int(host_threads() if threads is None else threads), int(variant), p(x_des), p(u_des), p(rbd), p(mode), p(period), p(time), p(il), p(ee_force),
           p(out), p(st), p(diag))
        # diag per level: [verified vertex (0 / 1) + 10 x active-set iterations + 1000 x rows released / guesses dropped | interior-point + active-set iterations]
        return dict(out=out, status=st, input_last=il, attempts=np.zeros_like(diag[:, :4]), polished=diag[:, :4] % 10, as_iterations=(diag[:, :4] // 10) % 100, drops=diag[:, :4] // 1000, iterations=diag[:, 4:])

    def warm_start_batch(self, T, X, U, new_grid, x0):
        """previous solutions (T [B][Np+1], X, U) resampled on new_grid [B][Nn+1], x[0] = x0: the oracle's counterpart of qmgpu_warm_start_batch"""
        B, Np, Nn = U.shape[0], U.shape[1], new_grid.shape[1] - 1
        f64 = lambda a: np.ascontiguousarray(a, dtype=np.float64)  # noqa: E731
        wx, wu = np.zeros((B, Nn + 1, 30)), np.zeros((B, Nn, 30))
        fn = self.lib.qmo_warm_start_batch
        fn.restype = None
        fn.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        fn(B, Np, p(f64(T)), p(f64(X)), p(f64(U)), Nn, p(f64(new_grid)), p(f64(x0)), p(wx), p(wu))
        return wx, wu

    def policy_eval_batch(self, T, X, U, modes, t):
        """MPC_MRT_Interface::evaluatePolicy for every instance: (x_des [B][30], u_des [B][30], planned mode [B]) at the times t [B]"""
        B, N = U.shape[0], U.shape[1]
        f64 = lambda a: np.ascontiguousarray(a, dtype=np.float64)  # noqa: E731
        xu, md = np.zeros((B, 60)), np.zeros(B, dtype=np.int32)
        fn = self.lib.qmo_policy_eval_batch
        fn.restype = None
        fn.argtypes = [C.c_int, C.c_int] + [C.c_void_p] * 7
        fn(B, N, p(f64(T)), p(f64(X)), p(f64(U)), p(np.ascontiguousarray(modes, dtype=np.int32)), p(f64(np.broadcast_to(t, (B,)))), p(xu), p(md))
        return xu[:, :30].copy(), xu[:, 30:].copy(), md

    def time_cycles_node_threads(self, count, N, x0s, ttimes, tstates, nev, ev, modes, rbds, node_threads=3, line_search=True):
        """Seconds for `count` MPC+WBC cycles, one instance at a time, `node_threads` workers over the shooting nodes (task.info:78)."""
        f = self.lib.qmo_time_cycles_node_threads
        f.restype = C.c_double
        f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        return f(C.byref(self.P), count, N, p(x0s), len(ttimes), p(ttimes), p(tstates), nev, p(ev), p(modes), p(rbds), int(line_search), int(node_threads))

    def time_cycles(self, count, N, x0s, ttimes, tstates, nev, ev, modes, rbds, line_search=True, threads=1):
        """Seconds of wall clock for `count` MPC+WBC cycles on `threads` host threads (instances interleaved over the threads)."""
        self.lib.qmo_time_cycles_mt.restype = C.c_double
        self.lib.qmo_time_cycles_mt.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                                C.c_int]
        return self.lib.qmo_time_cycles_mt(C.byref(self.P), count, N, p(x0s), len(ttimes), p(ttimes), p(tstates), nev, p(ev), p(modes), p(rbds), int(line_search), int(threads))


def host_threads():
    """CPUs this process may keep busy: the affinity mask capped by a cgroup quota (the GPU boxes: 256 hardware threads, 16 CPUs granted)"""
    threads = len(os.sched_getaffinity(0))
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            threads = max(1, min(threads, int(np.ceil(int(q) / int(per)))))
    except (OSError, ValueError):
        pass
    return threads


def load_problem(lib):
    P = abi.Problem()
    d = abi.DATA_DIR.encode()
    st = lib.qmgpu_load_problem(d + b"/task.info", d + b"/aliengo_z1.urdf", d + b"/reference.info", d + b"/wbc_gains.info", C.byref(P))
    assert st == 0, lib.qmgpu_last_error()
    return P


# ------------------------------------------------------------------------------------------------ scenarios
def trot_schedule(t_end, period=0.70, phase0=0.0):
    """STANCE until phase0, then LF_RH / RF_LH (gait.info trot) tiled past t_end, then the default final STANCE."""
    ev, md = [phase0], [15]
    t = phase0
    while t < t_end:
        for mode in (9, 6):
            md.append(mode); t += period / 2; ev.append(t)
    md.append(15)
    evp = np.full(abi.MAX_EVENTS, 1e300); evp[:len(ev)] = ev
    mdp = np.full(abi.MAX_EVENTS + 1, 15, dtype=np.int32); mdp[:len(md)] = md
    return len(ev), evp, mdp


def nominal_target(oracle, x_nom):
    _, _, ee, eq, _ = oracle.kinematics(x_nom, np.zeros(30))
    return np.r_[x_nom, ee, eq]


def perturbed_states(x_nom, batch, seed=0):
    """Config 2 of SURVEY.md section 8(d): x0 = x_nom + U(-1,1) * s per component."""
    rng = np.random.default_rng(seed)
    s = np.r_[np.full(6, 0.1), np.full(3, 0.05), np.full(3, 0.05), np.full(18, 0.1)]
    return x_nom[None, :] + rng.uniform(-1, 1, (batch, 30)) * s[None, :]


def rbd_from_state(oracle, x, v=None):
    r = np.zeros(55)
    r[0:3] = x[9:12]; r[3:6] = x[6:9]; r[6:24] = x[12:30]
    if v is not None:
        r[24:48] = v
    _, _, ee, eq, _ = oracle.kinematics(x, np.zeros(30))
    r[48:51] = ee; r[51:55] = eq
    return r


def moving_inputs(oracle, x0, dt, seed=11, t0=0.0):
    """A batch of robots IN MOTION on an arbitrary tick of the controller (the whole-batch comparisons used to run the WBC on
    robots standing still at t = 20 s on their first tick).  For every instance: a seeded measured twist + joint rates (rbd[24:48] != 0, so Jdot v, the
    Coriolis part of nle, Adot_G v and the Euler-rate maps of WbcBase.cpp:154-155,180-202,279-299 contribute), the centroidal momentum of x0 replaced by
    A(q) v / m of that measurement (the observation QMController.cpp:239-244 would hand the MPC), a non-zero inputLast_ (WbcBase.cpp:224-225), controller
    times on both sides of the start-up branch (t < 10, HierarchicalWbc.cpp:23) and a policy-evaluation time strictly between shooting nodes.
    Returns dict(x0, rbd, input_last, time, t_eval)."""
    rng = np.random.default_rng(seed)
    B = x0.shape[0]
    v = np.c_[rng.uniform(-0.3, 0.3, (B, 3)), rng.uniform(-0.3, 0.3, (B, 3)), rng.uniform(-0.5, 0.5, (B, 18))]
    rbd = np.zeros((B, 55)); x0m = np.array(x0, dtype=np.float64)
    for i in range(B):
        rbd[i] = rbd_from_state(oracle, x0[i], v[i])
        x0m[i, :6] = oracle.frontend(rbd[i], 0.0, 0, np.zeros(7), rbd[i, 48:55])[0][:6]
    il = np.zeros((B, 30))
    il[:, 2:12:3] = 9.81 * 7.0 + rng.uniform(-5, 5, (B, 4))                 # never read by the update (only the joint part is), non-zero all the same
    il[:, 12:] = v[:, 6:] + rng.uniform(-1, 1, (B, 18)) * 0.02
    time = np.where(rng.uniform(size=B) < 0.25, rng.uniform(0.5, 9.5, B), rng.uniform(10.5, 30.0, B))
    t_eval = t0 + (rng.integers(0, 4, B) + rng.uniform(0.1, 0.9, B)) * dt
    return dict(x0=x0m, rbd=rbd, input_last=il, time=time, t_eval=t_eval, v=v)


# ------------------------------------------------------------------------------------------------ force tracking (BASELINE.json configs[3], own formulation)
FT_STIFFNESS, FT_MU = 500.0, 0.02      # K_e [N/m], mu_f of the force soft constraint


def door_opening_batch(oracle, x_nom, batch, seed=2, t_end=1.5, stiffness=FT_STIFFNESS):
    """Door-opening reference for `batch` instances: the end-effector pushes 10 cm along a random horizontal direction n while the door
    pushes back with a force that ramps from 0 to F in [5, 25] N; the base follows by 5 cm.  Two target knots (t = 0, t_end).
    Returns x0 [B][30], target_times [B][2], target_states [B][2][37], contact [B][2][6] (f_ref, p_env per knot).
    The anchor of the compliant environment is placed so that tracking the EE position exactly produces exactly f_ref."""
    rng = np.random.default_rng(seed)
    x0 = perturbed_states(x_nom, batch, seed=seed)
    tgt = nominal_target(oracle, x_nom)
    tt = np.tile(np.array([0.0, t_end]), (batch, 1))
    ts = np.tile(tgt, (batch, 2, 1)).copy()
    contact = np.zeros((batch, 2, 6))
    psi = rng.uniform(-0.6, 0.6, batch)
    force = rng.uniform(5.0, 25.0, batch)
    for i in range(batch):
        n = np.array([np.cos(psi[i]), np.sin(psi[i]), 0.0])
        ts[i, 1, 30:33] += 0.10 * n
        ts[i, 1, 6:9] += 0.05 * n
        f1 = -force[i] * n                           # force ON the end-effector at the end of the push
        contact[i, 0, 0:3] = 0.0; contact[i, 1, 0:3] = f1
        for k in range(2):
            contact[i, k, 3:6] = ts[i, k, 30:33] + contact[i, k, 0:3] / stiffness
    return x0, tt, ts, contact


def ee_contact_force(oracle, x, contact_knot, stiffness=FT_STIFFNESS):
    """f_e = -K (p_ee(x) - p_env) of the contact model at state x for one knot [f_ref(3), p_env(3)]"""
    _, _, ee, _, _ = oracle.kinematics(x, np.zeros(30))
    return -stiffness * (ee - contact_knot[3:6])


def wbc_stress_batch(interface, variant, B=2048):
    """2048 random (NOT MPC-consistent) instances over every contact mode of gait.info, robots in motion, non-zero inputLast_, 20 % on the start-up branch"""
    rng = np.random.default_rng(17 + variant)
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
    return dict(xd=xd, u=u, rbd=rbd, mode=mode, t=t, il=il)


def wbc_fast_robots_batch(interface, variant, B=512, velocity=150.0):
    """The stress batch with the robots MOVING FAST (joint rates of +-15 rad/s, base +-15 m/s; planned forces +-150 N off the weight, desired joint rates +-30 rad/s, joint
    accelerations +-2500 rad/s^2): the stance feet's -Jdot v asks for accelerations the torque limits cannot deliver, the first level's minimum-norm point is rejected and in
    40 % of the instances its held-variable form too -- the regime of the diverged robots of the bench's steady-state leg (tests/golden/wbc_slow_ticks.npz), as a batch.
    Beyond velocity ~ 250 (25 rad/s) a few instances per hundred end with 18 limits violated, every lower level eliminated and one direction of the first level unseen by any row:
    its value then depends on the path (cold and interior-point path 1e-2 apart, the reference's 1e-12 I would pick the minimum-norm one) -- stated in DESIGN.md section 5, not tested."""
    b = wbc_stress_batch(interface, variant, B)
    rng = np.random.default_rng(99 + variant)
    u = b["u"].copy()
    u[:, :12] += rng.uniform(-1, 1, (B, 12)) * 150.0 * (u[:, :12] != 0)
    u[:, 12:] = rng.uniform(-1, 1, (B, 18)) * 30.0
    rbd = b["rbd"].copy(); rbd[:, 24:48] *= velocity
    il = u + rng.uniform(-1, 1, (B, 30)) * np.r_[np.zeros(12), np.full(18, 5.0)]
    return dict(xd=b["xd"], u=u, rbd=rbd, mode=b["mode"], t=b["t"], il=il)


# ------------------------------------------------------------------------------------------------ whole-batch parity (every instance, not a sample)
def rel_inf(got, ref):
    """||got - ref||_inf / max(1, ||ref||_inf) per instance (axis 0 = instance)"""
    B = ref.shape[0]
    d = np.abs(got.reshape(B, -1) - ref.reshape(B, -1)).max(axis=1)
    return d / np.maximum(1.0, np.abs(ref.reshape(B, -1)).max(axis=1))


WBC_BLOCKS = {"tau_legs": (36, 48), "tau_arm": (48, 54), "accelerations": (0, 24), "contact_forces": (24, 36)}


def rel_inf_blocks(got, ref):
    """rel-inf deviation per instance and per block of the WBC output [B][54] = [ddq (24), F (12); tau (18)].  The blocks are normalised SEPARATELY: the separated-system
    plugin commands only the leg torques tau[0:12] (qm_controllers/src/QMController.cpp:428-431; the arm runs on position PIDs), and an arm torque of 30 N m in the
    same norm would hide a leg deviation 30 times smaller."""
    return {k: rel_inf(got[:, a:b], ref[:, a:b]) for k, (a, b) in WBC_BLOCKS.items()}


def block_summary(got, ref):
    return {k: {"max": float(e.max()), "p99": float(np.percentile(e, 99)), "median": float(np.median(e)), "argmax": int(e.argmax()), "above_1e-6": int((e > 1e-6).sum())}
            for k, e in rel_inf_blocks(got, ref).items()}


def parity_report(name, got, ref, keys=("X", "U"), tau=True, record=True):
    """Per-instance deviations of a GPU batch against the oracle batch: max / p99 / median per block, exact agreement of modes and step lengths.
    Appends the numbers to gpurun_out/parity.json (copied to profiles/r03_parity.json for the record)."""
    rep = {"instances": int(ref["X"].shape[0])}
    for k in keys:
        e = rel_inf(got[k], ref[k])
        rep[k] = {"max": float(e.max()), "p99": float(np.percentile(e, 99)), "median": float(np.median(e)), "argmax": int(e.argmax())}
    if tau and "out" in ref:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        np.savez(os.path.join(ROOT, "gpurun_out", f"wbc_{name}.npz"), gpu=got["out"], oracle=ref["out"])     # scratch, for offline analysis
        e = rel_inf(got["out"][:, 36:], ref["out"][:, 36:])
        rep["tau"] = {"max": float(e.max()), "p99": float(np.percentile(e, 99)), "median": float(np.median(e)), "argmax": int(e.argmax()), "above_tol": int((e > 1e-6).sum())}
        rep["wbc_blocks"] = block_summary(got["out"], ref["out"])
        rep["wbc_status_nonzero"] = [int((got["status"] != 0).sum()), int((ref["status"] != 0).sum())]
    rep["modes_equal"] = bool(np.array_equal(got["mode"], ref["mode"]))
    rep["alpha_equal"] = int((got["stats"][:, 4] == ref["stats"][:, 4]).sum())
    rep["step_type_equal"] = int((got["stats"][:, 5] == ref["stats"][:, 5]).sum())
    rep["riccati_status_nonzero"] = [int((got["stats"][:, 7] != 0).sum()), int((ref["stats"][:, 7] != 0).sum())]
    if record:
        import json
        path = os.path.join(ROOT, "gpurun_out", "parity.json")
        os.makedirs(os.path.dirname(path), exist_ok=True)
        try:
            allrep = json.load(open(path))
        except (OSError, ValueError):
            allrep = {}
        allrep[name] = rep
        json.dump(allrep, open(path, "w"), indent=1)
    return rep


def assert_parity(rep, tol=1e-6, tau_tol=1e-6):
    """north_star bar: X, U, tau within 1e-6 rel-inf on EVERY instance, modes / step lengths / step types exact (no outlier allowance: every WBC level ends at the
    vertex of its QP on both sides, oracle/qmo_wbc.h activeSetPhase)."""
    B = rep["instances"]
    assert rep["modes_equal"], rep
    assert rep["alpha_equal"] == B and rep["step_type_equal"] == B, rep
    assert rep["riccati_status_nonzero"] == [0, 0], rep
    for k in ("X", "U"):
        assert rep[k]["max"] <= tol, (k, rep)
    if "tau" in rep:
        assert rep["wbc_status_nonzero"] == [0, 0], rep
        assert rep["tau"]["max"] <= tau_tol, rep
        assert rep["wbc_blocks"]["tau_legs"]["max"] <= tau_tol and rep["wbc_blocks"]["tau_arm"]["max"] <= tau_tol, rep["wbc_blocks"]
