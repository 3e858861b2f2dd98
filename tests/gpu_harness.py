"""Runs the C ABI on the GPU with torch tensors as device memory (plumbing shared by the -m gpu tests, smoke and bench)."""
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
    def __init__(self, rbd, period, time, input_last, state_desired=None, input_desired=None, mode=None, variant=0):
        B = rbd.shape[0]
        f64 = torch.float64
        self.rbd = dev(rbd, f64); self.period = dev(period, f64); self.time = dev(time, f64); self.il = dev(input_last, f64)
        self.xd = dev(state_desired, f64) if state_desired is not None else None
        self.ud = dev(input_desired, f64) if input_desired is not None else None
        self.mode = dev(mode, torch.int32) if mode is not None else None
        self.out = torch.zeros((B, 54), dtype=f64, device=DEVICE); self.status = torch.zeros(B, dtype=torch.int32, device=DEVICE)
        self.args = api.GpuSolver.wbc_args(B, self.rbd, self.period, self.time, self.il, self.out, self.status, self.xd, self.ud, self.mode, variant)

    def results(self):
        _sync()
        return dict(out=self.out.cpu().numpy(), status=self.status.cpu().numpy(), input_last=self.il.cpu().numpy())


def make_solver(interface, max_batch, max_nodes, dtype="f64"):
    if DEVICE != "cuda":
        return api.GpuSolver(interface, max_batch, max_nodes, device=0, dtype=dtype)
    s = api.GpuSolver(interface, max_batch, max_nodes, device=torch.cuda.current_device(), dtype=dtype)
    s.set_stream(torch.cuda.current_stream().cuda_stream)
    return s
