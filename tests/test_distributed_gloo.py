"""world_size-2 gloo test of the multi-GPU path's host logic: contiguous block sharding of independent
instances, one all-gather of the packed [X | U | wbc | mode] records, bit-identical to the single-process result.
The per-shard solves use the emulated kernels through the same C-ABI wrapper the GPU path uses."""
import os

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import support as S
from qm_door_amd import abi, api, sharding

B_TOTAL, N = 3, 3


def _solve(itf, orc, lo, hi):
    x_nom = itf.initial_state
    x0 = S.perturbed_states(x_nom, B_TOTAL, seed=9)[lo:hi]
    B = hi - lo
    tgt = S.nominal_target(orc, x_nom)
    tt = np.zeros((B, 1)); ts = np.tile(tgt, (B, 1, 1)).copy()
    nev, ev, md = S.trot_schedule(2.0, phase0=0.02)
    sol = api.GpuSolver(itf, max_batch=B, max_nodes=N)
    oT, oX, oU, oM, oS = np.zeros((B, N + 1)), np.zeros((B, N + 1, 30)), np.zeros((B, N, 30)), np.zeros((B, N + 1), dtype=np.int32), np.zeros((B, abi.NSTATS))
    a = sol.mpc_args(B, N, x0, tt, ts, np.full(B, nev, dtype=np.int32), np.tile(ev, (B, 1)).copy(), np.tile(md, (B, 1)).copy(), oT, oX, oU, oM, oS, t0=np.zeros(B))
    rbd = np.array([S.rbd_from_state(orc, x) for x in x0])
    out, st = np.zeros((B, 54)), np.zeros(B, dtype=np.int32)
    w = sol.wbc_args(B, rbd, np.full(B, 0.002), np.full(B, 20.0), np.zeros((B, 30)), out, st)
    sol.cycle(a, np.zeros(B), w)
    return sharding.pack(oX, oU, out, oM)


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lib = abi.load_library(S.EMU_LIB)
    itf = api.QMInterface(lib=lib)
    orc = S.Oracle(itf.problem)
    lo, hi = sharding.shard_bounds(B_TOTAL, world, rank)
    mine = torch.from_numpy(_solve(itf, orc, lo, hi))
    # ragged shards: pad to the largest block for the gather, then drop the padding
    sizes = [sharding.shard_bounds(B_TOTAL, world, r) for r in range(world)]
    width = max(h - l for l, h in sizes)
    padded = torch.zeros((width, mine.shape[1]), dtype=torch.float64); padded[:mine.shape[0]] = mine
    bucket = [torch.zeros_like(padded) for _ in range(world)]
    dist.all_gather(bucket, padded)
    full = torch.cat([bucket[r][:sizes[r][1] - sizes[r][0]] for r in range(world)]).numpy()
    if rank == 0:
        ret["full"] = full
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_gather_equals_single_process():
    assert [sharding.shard_bounds(5, 2, r) for r in range(2)] == [(0, 3), (3, 5)]
    assert [sharding.shard_bounds(2048, 8, r)[1] - sharding.shard_bounds(2048, 8, r)[0] for r in range(8)] == [256] * 8
    S.build_emu(); S.build_oracle()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, 29541, ret), nprocs=2, join=True)
    lib = abi.load_library(S.EMU_LIB)
    itf = api.QMInterface(lib=lib)
    single = _solve(itf, S.Oracle(itf.problem), 0, B_TOTAL)
    assert ret["full"].shape == (B_TOTAL, sharding.pack_len(N))
    assert np.array_equal(ret["full"], single)                 # instances are independent: sharding must not change a single bit
    X, U, wbc, modes = sharding.unpack(single, N)
    assert X.shape == (B_TOTAL, N + 1, 30) and U.shape == (B_TOTAL, N, 30) and wbc.shape == (B_TOTAL, 54) and modes.shape == (B_TOTAL, N + 1)
