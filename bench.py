#!/usr/bin/env python3
"""Benchmark of the MPC+WBC hot path (BASELINE.json metric) on N GPUs of one node.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = one pass of the hot path (qmgpu_cycle_batch: one SQP-iteration MPC solve over N=100 shooting nodes, policy
evaluation, three-level WBC) over a batch of 256 independent AlienGo+Z1 instances per GPU (weak scaling).
  --gpus 1 : BASELINE.json configs[1] (256 instances, trot, seeded perturbations of the nominal state).
  --gpus N : BASELINE.json configs[2] -- ONE global batch of 256 N instances (2048 at N = 8) with randomised base pose + EE target, seed 1,
             contiguous shards of it per rank (qm_door_amd/sharding.py), the solved trajectories + torques all-gathered over RCCL inside the
             timed region: the only exchange the path has.  Without a launcher environment `bench.py --gpus N` starts its own N ranks
             (torch.distributed.run on 127.0.0.1).
Inputs are resident in HBM before the timed region.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (HIP events on the kernels' own stream, recorded during
the timed steps); `cpu_baseline` times the CPU oracle (a restatement, NOT the reference OCS2 path, which cannot be built here)
on a bounded sample, rank 0 at N = 1 only.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def _test_support():
    """tests/support.py (the oracle wrapper and the emulation build): reached from the cpu_baseline leg and from --emulate only -- never from a timed region"""
    t = os.path.join(ROOT, "tests")
    if t not in sys.path:
        sys.path.insert(0, t)
    import support
    return support

BATCH_PER_GPU = 256
HORIZON_N = 100
OVERLAP = True     # --no-overlap: the WBC of a cycle on the compute stream itself (qmgpu_set_overlap off); see config.overlap in the line
CONFIG3_GLOBAL_BATCH = 2048           # BASELINE.json configs[2]
FP64_MFMA_PEAK_TFLOPS = 78.6  # MI355X public spec, v_mfma_f64_16x16x4_f64 (not listed in MI355X_MICROARCH.md)


def node_flops(nc):
    """Algorithmic dense-contraction FLOPs of one shooting node, split by kernel."""
    n, m = 30, 30
    mt = m - nc
    discretise = 2 * n**3 + 2 * n * n * m
    project = (2 * m * nc * nc + 2 * m * n * nc + (2 * n * n * m + 2 * n * mt * m) + (2 * m * m * mt + 2 * mt * mt * m + 2 * m * m * n + 2 * mt * n * m)
               + 3 * 2 * n * n * m)
    riccati = 2 * 2 * n**3 + 2 * n * n * mt + 2 * mt * mt * n + 2 * mt * n * n + mt**3 / 3.0 + 2 * mt * mt * n + 2 * n * n * mt
    forward = 2 * (2 * mt * n + n * n) + 2 * m * mt + 2 * m * n
    return {"ad_node_kernel": discretise, "lq_node_kernel": project, "riccati_kernel": riccati + forward}


def build_scenario(itf, batch, seed):
    """Config 2 of SURVEY.md 8(d): x0 = nominal + U(-1,1)*s, trot, hold-pose target (QMController.cpp:107-113 construction)."""
    from qm_door_amd import abi
    rng = np.random.default_rng(seed)
    x_nom = itf.initial_state
    s = np.r_[np.full(6, 0.1), np.full(3, 0.05), np.full(3, 0.05), np.full(18, 0.1)]
    x0 = x_nom[None, :] + rng.uniform(-1, 1, (batch, 30)) * s[None, :]
    ee = np.array([x_nom[6] + 0.6, x_nom[7], x_nom[8] + 0.036, 0.0, 0.0, 0.0, 1.0])  # StartingPosition.h:13-15, yaw 0
    target = np.r_[x_nom, ee]
    tt = np.zeros((batch, 1)); ts = np.tile(target, (batch, 1, 1)).copy()
    from qm_door_amd import api
    nev, ev, md = api.GaitSchedule(lib=itf.lib).mode_schedule("trot", 0.0, 0.0, HORIZON_N * itf.problem.settings.dt + 0.5)
    rbd = np.zeros((batch, 55))
    rbd[:, 0:3] = x0[:, 9:12]; rbd[:, 3:6] = x0[:, 6:9]; rbd[:, 6:24] = x0[:, 12:30]
    return dict(x0=x0, tt=tt, ts=ts, nev=nev, ev=ev, md=md, rbd=rbd)


def build_config3(itf, total=CONFIG3_GLOBAL_BATCH, seed=1):
    """BASELINE.json configs[2] / SURVEY.md 8(d) config 3: ONE global batch of 2048 instances, randomised base pose (xy in U(-0.5, 0.5) m,
    yaw in U(-0.5, 0.5) rad, z nominal) and EE target (base target + Rz(yaw) (0.6, 0, 0.036) + U(-0.1, 0.1)^3, yaw-only orientation; the constants of
    qm_controllers/include/qm_controllers/StartingPosition.h:13-15), trot, seed 1.  tests/test_gpu_configs.py solves exactly this batch on one GPU
    and compares every instance with the oracle."""
    from qm_door_amd import api
    rng = np.random.default_rng(seed)
    x_nom = itf.initial_state
    x0 = np.tile(x_nom, (total, 1))
    xy = rng.uniform(-0.5, 0.5, (total, 2)); yaw = rng.uniform(-0.5, 0.5, total)
    x0[:, 6:8] = xy; x0[:, 9] = yaw
    ts = np.zeros((total, 1, 37)); tt = np.zeros((total, 1))
    for i in range(total):
        c, s_ = np.cos(yaw[i]), np.sin(yaw[i])
        ee = np.r_[xy[i, 0] + c * 0.6, xy[i, 1] + s_ * 0.6, x_nom[8] + 0.036] + rng.uniform(-0.1, 0.1, 3)
        ts[i, 0] = np.r_[x0[i], ee, 0.0, 0.0, np.sin(yaw[i] / 2), np.cos(yaw[i] / 2)]
    nev, ev, md = api.GaitSchedule(lib=itf.lib).mode_schedule("trot", 0.0, 0.0, HORIZON_N * itf.problem.settings.dt + 0.5)
    rbd = np.zeros((total, 55)); rbd[:, 0:3] = x0[:, 9:12]; rbd[:, 3:6] = x0[:, 6:9]; rbd[:, 6:24] = x0[:, 12:30]
    return dict(x0=x0, tt=tt, ts=ts, nev=nev, ev=ev, md=md, rbd=rbd)


def _digest(*tensors):
    """one number per tensor from the device (sum of the raw 64-bit words, exact in int64 arithmetic modulo 2^64): a running fingerprint of a step's outputs that costs a
    reduction launch per tensor and no host synchronisation; two runs that agree bit for bit agree in it"""
    import torch
    return torch.stack([t.contiguous().view(torch.int64).sum() for t in tensors])


def steady_state(itf, sc, steps, warmup, emulate=False, wbc_state="cold"):
    """config.steady_state: what the controller does between the first tick and shutdown -- the SAME 256 instances in a receding
    horizon: every step shifts the horizon by one MPC period (10 ms, mpcDesiredFrequency task.info:147), resamples the previous solution on the shifted grid
    ON THE DEVICE as the initial guess (qmgpu_warm_start_batch; coldStart false, task.info:143), solves, evaluates the policy between two nodes and runs the
    WBC on a robot IN MOTION: the measured state follows the instance's own plan plus a seeded disturbance (qm_door_amd/harness.py: measurement), inputLast_
    is carried from step to step, the centroidal observation comes from qmgpu_frontend_batch.
    wbc_state: "carry" -- the WBC's solver state (qmgpu_wbc_args::working_set) travels from step to step like inputLast_; "cold" -- every tick starts cold (the record is
    zeroed before every step: it still collects the pass counts).
    The measurements depend on the plans, so the sequence is produced once, untimed (record pass: plan -> host -> measurement -> device), and then REPLAYED
    from device-resident inputs with nothing but qmgpu_warm_start_batch + qmgpu_cycle_batch per step inside the timed region; the replay must reproduce the
    recorded run bit for bit on EVERY step (a fingerprint of X, U, the WBC output and its status words per step, compared after the timed region)."""
    import torch
    from qm_door_amd import harness as CL, harness as G
    from qm_door_amd import abi, api
    f64 = torch.float64
    B, N = len(sc["x0"]), HORIZON_N
    dt = itf.problem.settings.dt
    total = warmup + steps
    dist_ = CL.Disturbance(B, np.random.default_rng(5))
    sol = G.make_solver(itf, B, N)
    sol.set_overlap(OVERLAP)
    z = lambda *shape, dtype=f64: torch.zeros(shape, dtype=dtype, device=G.DEVICE)  # noqa: E731
    sets = [dict(T=z(B, N + 1), X=z(B, N + 1, 30), U=z(B, N, 30), M=z(B, N + 1, dtype=torch.int32), S=z(B, abi.NSTATS)) for _ in range(2)]
    wx, wu = z(B, N + 1, 30), z(B, N, 30)
    tt, ts = G.dev(sc["tt"], f64), G.dev(sc["ts"], f64)
    sn, se, sm = G.dev(np.full(B, sc["nev"], dtype=np.int32), torch.int32), G.dev(np.tile(sc["ev"], (B, 1)), f64), G.dev(np.tile(sc["md"], (B, 1)), torch.int32)
    kind, cmd, lastee, ftt, fts = z(B, dtype=torch.int32), z(B, 7), z(B, 7), z(B, 2), z(B, 2, 37)
    il0 = np.zeros((B, 30)); il0[:, 12:] = 0.0
    # the WBC writes alternating output sets: with the overlap on, the fingerprint of step k is taken behind cycle(k + 1) (which has made the stream wait for WBC k)
    outs = [dict(out=z(B, 54), status=z(B, dtype=torch.int32)) for _ in range(2)]
    # WbcBase::update's `period` is the time since the previous update (ros_control hands the elapsed period on; the desired joint accelerations are
    # (v_des - inputLast_) / period, WbcBase.cpp:224-225).  This leg runs ONE tick per MPC cycle, 10 ms after the previous one: period = 10 ms.  (Until round 5 it passed the
    # controller's nominal 1 ms with ticks 10 ms apart: accelerations ten times too large, torque limits violated by thousands of N m on a few robots whose first level
    # then takes 40-46 working-set changes -- 4.4 ms launches in three of twenty steps, the whole of the "steady-state regression" of BENCH_r05; profiles/r06_notes.md.)
    period = G.dev(np.full(B, CL.MPC_PERIOD), f64)
    # (cold: two records used in turn and zeroed before their step -- the WBC of step k - 2 has been joined by then, the overlap stays as it is)
    wss = [z(B, abi.WBC_STATE_WORDS, dtype=torch.int64) for _ in range(2 if wbc_state == "cold" else 1)]
    rec = []          # per step: device-resident inputs of the replay
    sync = (lambda: None) if emulate else torch.cuda.synchronize

    def run_step(k, r, il):
        cur, prev = sets[k & 1], sets[(k & 1) ^ 1]
        if k > 0:
            sol.warm_start(B, N, prev["T"], prev["X"], prev["U"], N, r["grid"], r["x0"], wx, wu)
        a = api.GpuSolver.mpc_args(B, N, r["x0"], tt, ts, sn, se, sm, cur["T"], cur["X"], cur["U"], cur["M"], cur["S"], t0=r["t0"], time_grid=r["grid"],
                                   warm_x=wx if k > 0 else None, warm_u=wu if k > 0 else None)
        ws = wss[k % len(wss)]
        if wbc_state == "cold":
            ws.zero_()
        o = outs[k & 1]
        w = api.GpuSolver.wbc_args(B, r["rbd"], period, r["time"], il, o["out"], o["status"], working_set=ws)
        sol.cycle(a, r["t_eval"], w)
        return cur, o

    def fingerprints(k, prints, last=False):
        """the fingerprint of step k - 1 once cycle(k) has been issued (the stream has joined WBC k - 1 by then); of step k itself after the last one"""
        if k > 0:
            prints.append(_digest(sets[(k - 1) & 1]["X"], sets[(k - 1) & 1]["U"], outs[(k - 1) & 1]["out"], outs[(k - 1) & 1]["status"]))
        if last:
            sol.join_wbc()
            prints.append(_digest(sets[k & 1]["X"], sets[k & 1]["U"], outs[k & 1]["out"], outs[k & 1]["status"]))

    # ---- record pass (untimed)
    v0 = np.c_[np.random.default_rng(6).uniform(-0.1, 0.1, (B, 6)), np.random.default_rng(7).uniform(-0.2, 0.2, (B, 18))]
    q0 = sc["x0"][:, 6:30]
    rbd = CL.pack_rbd(q0 + dist_.dq(0.0), v0 + dist_.dv(0.0))
    il = G.dev(il0, f64)
    plan = None
    rec_prints, passes, refuted = [], [], []
    bad_steps = 0
    for k in range(total):
        t0 = k * CL.MPC_PERIOD
        if k > 0:
            rbd = CL.measurement(dist_, plan, t0)
        r = dict(rbd=G.dev(rbd, f64), x0=z(B, 30), grid=G.dev(np.tile(t0 + dt * np.arange(N + 1), (B, 1)), f64), t0=G.dev(np.full(B, t0), f64),
                 t_eval=G.dev(np.full(B, t0 + 0.3 * CL.WBC_PERIOD), f64), time=G.dev(np.full(B, 20.0 + t0), f64))
        sol.frontend(sol.frontend_args(B, r["rbd"], r["t0"], kind, cmd, lastee, r["x0"], ftt, fts))
        cur, o = run_step(k, r, il)
        fingerprints(k, rec_prints, last=(k == total - 1))
        sync()
        plan = dict(T=cur["T"].cpu().numpy(), X=cur["X"].cpu().numpy(), U=cur["U"].cpu().numpy())
        # interior-point + active-set passes of every instance in this step's WBC (words 13 / 14 of the record: a byte per solve, bit 7 = carried guess refuted)
        cb = np.ascontiguousarray(wss[k % len(wss)].cpu().numpy()[:, 13:15]).view(np.uint8).reshape(B, 16)
        passes.append((cb & 127).astype(np.int64).sum(axis=1)); refuted.append((cb >> 7).astype(np.int64).sum(axis=1))
        stats_k, status_k, out_k = cur["S"].cpu().numpy(), o["status"].cpu().numpy(), o["out"].cpu().numpy()
        bad_steps += int(not (np.isfinite(plan["X"]).all() and np.isfinite(out_k).all() and (stats_k[:, 7] == 0).all() and (status_k == 0).all()))
        rec.append(r)
    rec_last = dict(stats=stats_k)
    # ---- replay: the first `warmup` steps untimed (step 0 is the cold start), then `steps` steps timed
    il = G.dev(il0, f64)
    for w_ in wss:
        w_.zero_()
    # (the timed region holds the hot path and nothing else: until the end of round 6 the per-step fingerprints -- four torch reductions per step, 0.2-0.3 ms on the handle's
    #  stream -- sat inside it.  They now run in the synchronised pass below, on every step; the timed pass is tied to the record by the fingerprints of its last two steps,
    #  taken after its closing synchronisation: inputLast_ and the warm start chain every step to all the steps before it.)
    for k in range(warmup):
        run_step(k, rec[k], il)
    sol.enable_timing(True)
    sync()
    t_begin = time.perf_counter()
    for k in range(warmup, total):
        run_step(k, rec[k], il)
    sync()
    elapsed = time.perf_counter() - t_begin
    kms = sol.kernel_ms_mean(steps)
    hist = sol.kernel_ms_history(min(steps, 256))
    sol.enable_timing(False)
    timed_tail = [_digest(sets[j & 1]["X"], sets[j & 1]["U"], outs[j & 1]["out"], outs[j & 1]["status"]) for j in (total - 2, total - 1)]
    timed_tail_same = [bool(torch.equal(a_, b_)) for a_, b_ in zip(timed_tail, rec_prints[-2:])]
    speed = np.abs(np.concatenate([r["rbd"].cpu().numpy()[:, 24:48] for r in rec[warmup:]])).mean(axis=0)
    # a second, short replay of the same timed steps with a host synchronisation after every step: the wall-clock spread of single steps
    il = G.dev(il0, f64)
    for w_ in wss:
        w_.zero_()
    prints = []
    for k in range(warmup):
        run_step(k, rec[k], il)
        sol.synchronize()
        prints.append(_digest(sets[k & 1]["X"], sets[k & 1]["U"], outs[k & 1]["out"], outs[k & 1]["status"]))
    sync()
    per_step = []
    for k in range(warmup, total):
        t1 = time.perf_counter()
        run_step(k, rec[k], il)
        sol.synchronize()
        per_step.append(1e3 * (time.perf_counter() - t1))
        prints.append(_digest(sets[k & 1]["X"], sets[k & 1]["U"], outs[k & 1]["out"], outs[k & 1]["status"]))      # (outside the step's clock ...
        sync()                                                                                                        #  ... and finished before the next one starts)
    sync()
    same_steps = [bool(torch.equal(a_, b_)) for a_, b_ in zip(prints, rec_prints)]
    pt, rt = np.array(passes[warmup:]), np.array(refuted[warmup:])
    wbc_ms = hist[:, 4]
    res = {"value": B * steps / elapsed, "unit": "cycles/s", "ms_per_step": 1e3 * elapsed / steps, "steps": steps, "warmup": warmup,
           "what": "receding horizon: shifted grid (10 ms per step), device-resampled warm start of every solve (qmgpu_warm_start_batch inside the timed region), policy evaluated between "
                   "nodes, one WBC tick per step on robots in motion (plan-following measurement + seeded disturbance, inputLast_ carried, period = the 10 ms between two ticks); replay of a recorded input sequence, inputs resident"
                   + ("; the WBC of step k next to the node kernels of step k + 1 (config.overlap): kernel_ms are overlapping intervals" if OVERLAP else ""),
           "wbc_state": {"carry": "qmgpu_wbc_args::working_set travels from step to step like inputLast_ (each level of the hierarchical QP starts from the rows and the point the previous tick ended with; "
                                  "refuted guesses fall back to the cold path)", "cold": "every tick cold (record zeroed before every step), as the reference's qpOASES call"}[wbc_state],
           "kernel_ms": dict(zip(["ad", "lq", "riccati", "linesearch", "wbc", "whole"], kms)),
           "ms_per_step_synchronised": {"min": float(np.min(per_step)), "median": float(np.median(per_step)), "max": float(np.max(per_step)),
                                        "note": "the same timed steps once more with a host synchronisation after each (no overlap across steps): the spread a caller that waits for every tick sees"},
           "wbc_ms_per_step": {"min": float(wbc_ms.min()), "mean": float(wbc_ms.mean()), "max": float(wbc_ms.max()), "max_over_mean": float(wbc_ms.max() / wbc_ms.mean()),
                               "all": [round(float(v), 4) for v in wbc_ms]},
           "wbc_passes_per_instance": {"what": "interior-point + active-set iterations of all level QPs of one WBC tick (0 = every level ended at its first factorisation); a launch lasts as long as its slowest instance",
                                       "mean": float(pt.mean()), "p99": float(np.percentile(pt, 99)), "max": int(pt.max()), "max_per_step": [int(v) for v in pt.max(axis=1)],
                                       "mean_per_step": [round(float(v), 2) for v in pt.mean(axis=1)], "guesses_refuted_per_step": [int(v) for v in rt.sum(axis=1)]},
           "replay_reproduces_the_recorded_run_bit_for_bit": bool(all(same_steps) and len(same_steps) == total and all(timed_tail_same)),
           "replay_steps_compared": len(same_steps), "replay_steps_differing": int(len(same_steps) - sum(same_steps)),
           "replay_check": "every step of the synchronised replay against the recorded run (a device-side fingerprint of X, U, WBC output and status per step, outside the step's clock); "
                           "the TIMED replay, whose region holds nothing but the hot path, by the fingerprints of its last two steps after the closing synchronisation",
           "timed_replay_last_two_steps_match_the_record": bool(all(timed_tail_same)),
           "results_finite_and_converged": bool(bad_steps == 0), "steps_with_a_non_finite_or_unconverged_instance": bad_steps,
           "line_search_full_steps_last_solve": int((rec_last["stats"][:, 4] == 1.0).sum()),
           "mean_abs_measured_velocity": {"base_angular": float(speed[0:3].mean()), "base_linear": float(speed[3:6].mean()), "joints": float(speed[6:].mean())}}
    sol.close()
    return res


def device_for_rank(local_rank, device_count, visible=None):
    """LOCAL_RANK -> device index of THIS process.  torch.cuda.device_count() already honours HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES: a launcher that
    starts more ranks than the process can see devices (e.g. 8 ranks under HIP_VISIBLE_DEVICES=0,1) is a configuration error, reported here instead of as
    an 'invalid device ordinal' from the first allocation."""
    if not 0 <= local_rank < device_count:
        raise SystemExit(f"[bench] LOCAL_RANK {local_rank} has no GPU: this process sees {device_count} device(s)"
                         + (f" (HIP_VISIBLE_DEVICES={visible})" if visible else "") + "; start at most that many ranks per node")
    return local_rank


def shard_of(sc, lo, hi):
    return dict(x0=sc["x0"][lo:hi], tt=sc["tt"][lo:hi], ts=sc["ts"][lo:hi], nev=sc["nev"], ev=sc["ev"], md=sc["md"], rbd=sc["rbd"][lo:hi])


def profile_counters():
    """Counters of the committed rocprofv3 PMC passes of this same command (profiles/CURRENT names the set; tools/make_profile_summaries.py writes
    both): HBM bytes and matrix-core busy cycles per launch.  NOT measured in the bench run itself -- labelled so in the line."""
    try:
        tag = open(os.path.join(ROOT, "profiles", "CURRENT")).read().strip()
        return tag, json.load(open(os.path.join(ROOT, "profiles", f"{tag}_counters.json")))
    except (OSError, ValueError):
        return None, {}


def self_launch(args, argv):
    """`python bench.py --gpus N` without a launcher: start N ranks of this script on this node (one per GPU, RCCL over xGMI)."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    if args.emulate:   # build the emulation library once, before the ranks race for it
        S = _test_support()
        S.build_emu()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + argv
    return subprocess.call(cmd, env=env)


def usable_cpus():
    """Host threads this process can actually keep busy: the affinity mask capped by the cgroup CPU quota (the GPU boxes of this pool
    show 256 hardware threads and grant 16 CPUs' worth of time: cpu.max = 1600000 100000)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(np.ceil(int(txt[0]) / int(txt[1])))))
            else:
                q = int(txt[0]); p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, int(np.ceil(q / p))))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def cpu_baseline(itf, sc, budget_s=15.0):
    """The CPU restatement ("port": our own fp64 code, NOT the reference's OCS2 path, which cannot be built here) in its timing-grade
    build (oracle/Makefile target libqm_oracle_fast.so: -O3 -march=x86-64-v3, structured derivatives) on a bounded sample of the same
    workload, in SURVEY.md 8(d)'s three modes: one thread; three worker threads over the shooting nodes of one instance (the
    reference's own nThreads = 3, task.info:78); all hardware threads over instances (the reported value).  Plus the split of a
    one-thread cycle into LQ approximation + projection / Riccati / line search / WBC model / WBC QPs."""
    S = _test_support()
    orc = S.Oracle(itf.problem, fast=True)
    assert orc.lib.qmo_is_fast_build() == 1

    def args(n):
        have = len(sc["x0"])
        reps = (n + have - 1) // have     # more cycles than instances: the batch is walked through again
        x0 = np.tile(sc["x0"], (reps, 1))[:n].copy(); rbd = np.tile(sc["rbd"], (reps, 1))[:n].copy()
        return (n, HORIZON_N, x0, sc["tt"][0], sc["ts"][0], sc["nev"], sc["ev"], sc["md"], rbd)

    probe = orc.time_cycles(*args(2)) / 2.0
    n1 = int(max(4, 0.25 * budget_s / max(probe, 1e-3)))
    orc.time_split()
    sec1 = orc.time_cycles(*args(n1))
    split = orc.time_split()
    tot = sum(split.values()) or 1.0
    n3 = int(max(4, 0.25 * budget_s * 2.3 / max(probe, 1e-3)))
    sec3 = orc.time_cycles_node_threads(*args(n3), node_threads=3)
    threads = usable_cpus()
    nT = int(max(2 * threads, 0.4 * budget_s * 0.7 * threads / max(probe, 1e-3)))
    secT = orc.time_cycles(*args(nT), threads=threads)
    one, three, allt = n1 / sec1, n3 / sec3, nT / secT
    return {"value": allt, "unit": "cycles/s", "cores": threads, "kind": "port",
            "sample": f"{nT} cycles (the {BATCH_PER_GPU} instances of the bench, same x0/target/gait, N={HORIZON_N}, walked through repeatedly) over {threads} threads (= the CPUs this container may use: affinity mask capped by the cgroup quota) in "
                      f"{secT:.1f} s; one thread: {one:.2f} cycles/s ({n1} cycles, {sec1:.1f} s); three threads over the nodes of one instance (task.info nThreads 3): "
                      f"{three:.2f} cycles/s ({n3} cycles, {sec3:.1f} s); own CPU restatement, g++ -O3 -march=x86-64-v3, not OCS2",
            "hardware_threads_visible": os.cpu_count(), "one_thread": one, "three_threads_over_nodes": three, "all_threads_over_instances": allt, "scaling_per_thread": allt / (one * threads),
            "one_thread_split_percent": {k: round(100.0 * v / tot, 2) for k, v in split.items()},
            "reference_design_rate_note": "the reference is configured for 100 MPC solves/s at 67 nodes with 3 threads (task.info:79,141,147: a configuration "
                                          "value, not a measurement); its CppAD-generated sparse straight-line derivative code is not reproducible here"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=120, help="timed steps; the default makes the timed region >= 0.2 s at 1.9 ms per step")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-steady-state", action="store_true", help="skip config.steady_state (the receding-horizon leg, N = 1 only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-overlap", action="store_true", help="qmgpu_set_overlap off: every kernel of a step on one stream, one after the other (the kernel times then add up to the step)")
    ap.add_argument("--wbc-state", choices=["carry", "cold"], default="cold", help="steady-state leg: every WBC tick cold as the reference's qpOASES call (default), or the WBC's solver state travels from step to step (qmgpu_wbc_args::working_set)")
    ap.add_argument("--force-collective", action="store_true",
                    help="single process: run the N > 1 code path (pack -> RCCL all_gather_into_tensor -> unpack) on a 1-rank nccl group and verify it")
    ap.add_argument("--emulate", action="store_true",
                    help="CPU test of the multi-rank path only (tests/test_bench_contract.py): host-emulated kernels (tests/emu), gloo, tiny sizes; NOT a measurement")
    ap.add_argument("--batch-per-gpu", type=int, default=BATCH_PER_GPU, help="only with --emulate / --sweep")
    ap.add_argument("--nodes", type=int, default=HORIZON_N, help="only with --emulate")
    ap.add_argument("--global-batch", type=int, default=CONFIG3_GLOBAL_BATCH, help="only with --emulate: size of the ONE global batch of configs[2] (2048), so that a toy run reaches the 'all of it' branch")
    ap.add_argument("--sweep", action="store_true", help="N = 1 only: also time 512 / 1024 / 2048 instances on the one GPU (config.batch_sweep)")
    args = ap.parse_args()
    global OVERLAP
    OVERLAP = not args.no_overlap
    if not args.emulate and (args.batch_per_gpu != BATCH_PER_GPU or args.nodes != HORIZON_N):
        raise SystemExit("--batch-per-gpu / --nodes change the workload BASELINE.json names: allowed with --emulate only")

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args, sys.argv[1:]))

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and not (args.gpus == 1 and world == 1):
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} ranks")
    from qm_door_amd import abi, api, harness as G
    if args.emulate:
        S = _test_support()
        G.DEVICE = "cpu"
        lib = abi.load_library(S.build_emu())
    else:
        torch.cuda.set_device(device_for_rank(local_rank, torch.cuda.device_count(), os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("ROCR_VISIBLE_DEVICES")))
        lib = abi.load_library()
    collective = world > 1 or args.force_collective
    if collective:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if world == 1:   # --force-collective without a launcher: a 1-rank group on the loopback address
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29517")
            os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        import datetime
        # a rank that never arrives must not hold the others for the default 30 minutes (under torch.distributed.run the agent also tears the job down
        # as soon as one worker exits non-zero; the timeout covers launchers that do not)
        if args.emulate:
            dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=180))
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=datetime.timedelta(seconds=180))
    if os.environ.get("QM_BENCH_FAIL_RANK") == str(rank):     # tests/test_bench_contract.py: one rank fails before the first barrier
        raise RuntimeError("injected failure (QM_BENCH_FAIL_RANK)")

    itf = api.QMInterface(lib=lib)
    B, N = args.batch_per_gpu, args.nodes
    from qm_door_amd import sharding
    if world > 1:
        # configs[2]: the first 256 * world instances of the ONE global batch (all 2048 at world = 8), contiguous shards
        total = B * world
        whole = args.global_batch
        if whole != CONFIG3_GLOBAL_BATCH and not args.emulate:
            raise SystemExit("--global-batch changes the workload BASELINE.json names: allowed with --emulate only")
        glob = build_config3(itf, total=max(total, whole))
        lo, hi = sharding.shard_bounds(total, world, rank)
        sc = shard_of(glob, lo, hi)
        workload = (f"configs[2]: ONE global batch of {total} MPC instances ({'all' if total == whole else 'the first ' + str(total)} of the {whole} of seed 1), randomised base pose + EE "
                    f"target, horizon N={N}, dt=0.015, trot; contiguous shards of {B} per GPU; 1 SQP iteration + filter line search + 3-level WBC; all-gather of trajectories + torques")
    else:
        sc = build_scenario(itf, B, seed=0)
        workload = f"configs[1]: batch={B} MPC instances per GPU, horizon N={N}, dt=0.015, trot, 1 SQP iteration + filter line search + 3-level WBC"
    assert hi - lo == B if world > 1 else True

    def make_buffers(sc_, b):
        mb_ = G.MpcBatch(sc_["x0"], sc_["tt"], sc_["ts"], np.full(b, sc_["nev"], dtype=np.int32), np.tile(sc_["ev"], (b, 1)), np.tile(sc_["md"], (b, 1)), N)
        wb_ = G.WbcBatch(sc_["rbd"], np.full(b, 0.002), np.full(b, 20.0), np.zeros((b, 30)))
        return mb_, wb_

    def make(sc_, b):
        sol_ = G.make_solver(itf, b, N)
        sol_.set_overlap(OVERLAP)
        mb_ = G.MpcBatch(sc_["x0"], sc_["tt"], sc_["ts"], np.full(b, sc_["nev"], dtype=np.int32), np.tile(sc_["ev"], (b, 1)), np.tile(sc_["md"], (b, 1)), N)
        wb_ = G.WbcBatch(sc_["rbd"], np.full(b, 0.002), np.full(b, 20.0), np.zeros((b, 30)))
        return sol_, mb_, wb_, G.dev(np.zeros(b), torch.float64)

    sol, mb, wb, t_eval = make(sc, B)
    # packed result gathered across ranks: X | U | wbc out | modes (qm_door_amd/sharding.py)
    gathered = torch.zeros((world * B, sharding.pack_len(N)), dtype=torch.float64, device=G.DEVICE) if collective else None

    # The gather of step k runs on RCCL's stream while step k + 1 computes: its completion is only awaited (by the compute stream, not the
    # host) right before the next gather is enqueued, and once more before the closing synchronisation -- every gather is inside the timed
    # region.  The packed tensor is a fresh allocation per step, so the solver may overwrite its output buffers meanwhile.
    inflight = {"work": None, "buf": None}

    def sync():
        if not args.emulate:
            torch.cuda.synchronize()

    # With the overlap on, the WBC of step k is still running when qmgpu_cycle_batch returns.  The gather of step k is therefore enqueued one step LATE, behind
    # qmgpu_cycle_batch(k + 1) -- which has made the compute stream wait for WBC k before its own policy evaluation -- from a second set of output buffers (the solver
    # alternates between two), so nothing of step k is overwritten before it is packed; the last step's gather follows a qmgpu_join_wbc in drain().  Every gather is inside the
    # timed region, as before.
    from qm_door_amd import api as _api
    two_sets = collective and OVERLAP
    mbs, wbs = [mb], [wb]
    if two_sets:
        mb1, wb1 = make_buffers(sc, B)
        wb1.il = wb.il      # inputLast_ is carried from step to step: one buffer
        wb1.args = _api.GpuSolver.wbc_args(B, wb1.rbd, wb1.period, wb1.time, wb1.il, wb1.out, wb1.status)
        mbs.append(mb1); wbs.append(wb1)
    state = {"k": 0, "pending": None, "gathers": 0}

    # the record of one step, [B][pack_len(N)]: written by ONE launch on the handle's stream (qmgpu_pack_results) into a buffer that exists before the timed region -- until round 5
    # four strided copies into a fresh 12.5 MB torch.cat allocation per step
    packs = [torch.zeros((B, sharding.pack_len(N)), dtype=torch.float64, device=G.DEVICE) for _ in range(2)] if collective else []

    def gather_of(i):
        if inflight["work"] is not None:
            inflight["work"].wait()          # (the collective still in flight reads the other buffer; the stream waits, not the host)
        packed = packs[state["gathers"] & 1]
        state["gathers"] += 1
        sol.pack_results(B, N, mbs[i].oX, mbs[i].oU, wbs[i].out, mbs[i].oM, packed)
        inflight["work"] = dist.all_gather_into_tensor(gathered, packed, async_op=True)
        inflight["buf"] = packed

    def step():
        cur = state["k"] % len(mbs)
        sol.cycle(mbs[cur].args, t_eval, wbs[cur].args)
        state["k"] += 1
        state["last"] = cur
        if collective:
            if two_sets:
                if state["pending"] is not None:
                    gather_of(state["pending"])
                state["pending"] = cur
            else:
                gather_of(cur)

    def drain():
        if two_sets and state["pending"] is not None:
            sol.join_wbc()
            gather_of(state["pending"])
            state["pending"] = None
        if inflight["work"] is not None:
            inflight["work"].wait()
            inflight["work"] = None

    for _ in range(args.warmup):
        step()
    drain()
    sol.enable_timing(True)
    if collective:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    drain()
    sync()
    if collective:
        dist.barrier()
    elapsed_rank = time.perf_counter() - t0
    elapsed = elapsed_rank
    rank_values = [B * args.steps / elapsed_rank]
    if collective:
        tall = [torch.zeros(1, dtype=torch.float64, device=G.DEVICE) for _ in range(world)]
        dist.all_gather(tall, torch.tensor([elapsed_rank], dtype=torch.float64, device=G.DEVICE))
        elapsed = max(float(t.item()) for t in tall)
        rank_values = [B * args.steps / float(t.item()) for t in tall]
    kernel_ms = sol.kernel_ms_mean(args.steps)  # [ad, lq, riccati, linesearch, wbc, whole]
    sol.enable_timing(False)

    res = mbs[state.get("last", 0)].results(); wres = wbs[state.get("last", 0)].results()
    # Per-kernel launch durations with the device to themselves (roofline): with the overlap on, ad_node of step k + 1 is LAUNCHED while the WBC of step k still holds the CUs
    # and its HIP-event interval includes that wait (0.71 instead of 0.37 ms) -- not a property of the kernel.  A short calibration pass of the same cycle on one stream, after
    # the timed region and outside `value`, gives the unobstructed durations; the kernels the overlap does not touch (lq_node, riccati, line search) measure the same in both.
    kernel_ms_serial = kernel_ms
    calibration_same = None
    if OVERLAP and not args.emulate:
        cal = max(5, min(args.steps, 20))
        sol.set_overlap(False); sol.enable_timing(True)
        for _ in range(cal):
            sol.cycle(mbs[0].args, t_eval, wbs[0].args)
        sync()
        kernel_ms_serial = sol.kernel_ms_mean(cal)
        sol.enable_timing(False); sol.set_overlap(True)
        # the calibration pass solves the same cycle as the timed steps (same inputs; inputLast_ has been the cycle's own input since the second step): same results, bit for bit
        rc, wc = mbs[0].results(), wbs[0].results()
        calibration_same = bool(np.array_equal(rc["X"], res["X"]) and np.array_equal(rc["U"], res["U"]) and np.array_equal(wc["out"], wres["out"]))
    ok = bool(np.isfinite(res["X"]).all() and np.isfinite(wres["out"]).all() and (res["stats"][:, 7] == 0).all() and (wres["status"] == 0).all())
    gather_ok = None
    if collective:   # this rank's block of the gathered tensor is what it solved, bit for bit; every other block is finite and carries that rank's initial states
        full = gathered.cpu().numpy()
        gX, gU, gW, gM = sharding.unpack(full[rank * B:(rank + 1) * B], N)
        gather_ok = bool(np.array_equal(gX, res["X"]) and np.array_equal(gU, res["U"]) and np.array_equal(gW, wres["out"]) and np.array_equal(gM, res["mode"].astype(np.float64)))
        if world > 1:
            aX, _, aW, _ = sharding.unpack(full, N)
            gather_ok = gather_ok and bool(np.isfinite(full).all() and np.array_equal(aX[:, 0], glob["x0"][:world * B]))

    batch_sweep = None
    if args.sweep and world == 1 and rank == 0:
        batch_sweep = {}
        for b in (512, 1024, 2048):
            scb = build_scenario(itf, b, seed=0)
            s2, m2, w2, te2 = make(scb, b)
            for _ in range(2):
                s2.cycle(m2.args, te2, w2.args)
            s2.enable_timing(True)
            sync(); t1 = time.perf_counter()
            for _ in range(5):
                s2.cycle(m2.args, te2, w2.args)
            sync(); dt_ = time.perf_counter() - t1
            kms = s2.kernel_ms_mean(5)
            batch_sweep[str(b)] = {"cycles_per_s": b * 5 / dt_, "ms_per_step": 1e3 * dt_ / 5, "kernel_ms": dict(zip(["ad", "lq", "riccati", "linesearch", "wbc", "whole"], kms))}
            s2.close()

    if rank == 0:
        names = ["ad_node_kernel", "lq_node_kernel", "riccati_kernel", "linesearch_kernel", "wbc_kernel"]
        modes = res["mode"][:, :N]
        nst = np.array([[bin(int(m)).count("1") for m in row] for row in modes])
        nc = 3 * nst + 4 * (4 - nst)
        flops = {k: 0.0 for k in ("ad_node_kernel", "lq_node_kernel", "riccati_kernel")}
        for v in np.unique(nc):
            cnt = int((nc == v).sum())
            for k, f in node_flops(int(v)).items():
                flops[k] += cnt * f
        dom_name = max(flops, key=lambda k: kernel_ms_serial[names.index(k)])   # longest launch (device to itself) among the kernels that carry algorithmic FLOPs
        path_flops = flops["ad_node_kernel"] + flops["lq_node_kernel"] + flops["riccati_kernel"]
        roof_kernel = dom_name
        tf = lambda fl, ms: (fl / (ms * 1e-3) / 1e12) if ms > 0 else None   # noqa: E731  (the emulation has no clocks)
        # ONE source for the roofline block: the kernel, its duration, achieved / frac, kernel_frac and hbm_gbps all come from the one-stream durations (the calibration pass
        # when the overlap is on, the timed region itself otherwise); the timed region's own interval of the same kernel is reported next to it
        achieved = tf(flops[roof_kernel], kernel_ms_serial[names.index(roof_kernel)])
        achieved_timed = tf(flops[roof_kernel], kernel_ms[names.index(roof_kernel)])
        tag, ctr = profile_counters()
        traffic = ctr.get(roof_kernel, {}).get("bytes")
        src = (f"profiles/{tag}_counters.json (separate rocprofv3 --pmc passes of `bench.py --no-cpu-baseline --no-overlap --steps 5`, summarised by tools/make_profile_summaries.py; "
               "NOT measured in this run)") if tag else "no committed counter set"
        # HBM GB/s and matrix-core busy per kernel: counter values per launch (committed profile) over THIS run's HIP-event launch times
        hbm = {k: (ctr[k]["bytes"] / (kernel_ms_serial[names.index(k)] * 1e-3) / 1e9) for k in names if k in ctr and kernel_ms_serial[names.index(k)] > 0}
        busy = {k: ctr[k]["mfma_busy"] for k in names if k in ctr and "mfma_busy" in ctr[k]}
        out = {
            "metric": "MPC+WBC cycles/sec (AlienGo+Z1, N=100)",
            "value": world * B * args.steps / elapsed,
            "unit": "cycles/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic" if not args.emulate else "synthetic; HOST-EMULATED kernels (tests/emu) -- a functional test of the multi-rank path, not a measurement",
            "config": {"workload": workload,
                       "overlap": ("the WBC launch of step k runs on a second stream of the handle (qmgpu_set_overlap) next to the node kernels of step k + 1: a WBC launch lasts as long as its slowest instance, "
                                   "the CUs its fast instances leave are filled by ad_node / lq_node of the next step (the reference runs MPC and WBC in different threads); every step's outputs are produced, "
                                   "the timed region ends with a device-wide synchronisation; kernel_ms are per-launch HIP-event durations and now overlap (their sum exceeds ms_per_step); --no-overlap: one stream"
                                   + ("; the all-gather of step k is enqueued behind the launch of step k + 1, from alternating output buffers" if collective else "")) if OVERLAP else "off (--no-overlap): one stream, kernels back to back",
                       "wbc_inputs": "headline: robots at rest, t = 20 s, FIRST tick (inputLast_ = 0: the reference's spurious first-tick joint accelerations, torque limits active); the moving-robot, carried-inputLast_ regime is config.steady_state",
                       "batch_per_gpu": B, "global_batch": world * B, "horizon_nodes": N, "gait": "trot", "seed": 0 if world == 1 else 1, "results_finite_and_converged": ok,
                       "collective": ("all_gather(X,U,tau,mode) over " + ("gloo (emulation)" if args.emulate else "RCCL")) if collective else "none",
                       "per_rank_value": rank_values, "per_rank_value_min": min(rank_values), "per_rank_value_max": max(rank_values),
                       "slowest_rank_over_fastest": min(rank_values) / max(rank_values),
                       "gathered_bytes_per_step": int(world * B * sharding.pack_len(N) * 8) if collective else 0,
                       "gather_matches_local_results": gather_ok},
            "roofline": {"bound": "mfma", "kernel": roof_kernel, "achieved": achieved, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": (achieved / FP64_MFMA_PEAK_TFLOPS) if achieved is not None else None,
                         "achieved_from_the_timed_regions_interval": achieved_timed, "calibration_pass_reproduces_the_timed_results": calibration_same,
                         "traffic": traffic, "traffic_source": src,
                         "mfma_busy": busy.get(roof_kernel), "hbm_gbps": hbm.get(roof_kernel),
                         "mfma_busy_by_kernel": busy, "hbm_gbps_by_kernel": hbm,
                         "hbm_bytes_per_step": sum(v["bytes"] for k, v in ctr.items() if isinstance(v, dict) and "bytes" in v) or None,
                         "counters_note": "mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (launch duration x 2.4 GHz x 1024 SIMDs) of the committed PMC pass; hbm_gbps = committed "
                                          "(2 x FETCH_SIZE + WRITE_SIZE) bytes per launch / this run's HIP-event launch time; HBM peak 8000 GB/s spec, 6290 GB/s measured copy rate",
                         "note": "algorithmic dense-contraction FLOPs of SURVEY.md 8(d) per launch / HIP-event kernel time; fp64 matrix peak is the public spec; "
                                 "the path is latency bound, not MFMA bound (DESIGN.md)",
                         "kernel_ms": dict(zip(names + ["whole_call"], kernel_ms)),
                         "kernel_ms_one_stream": dict(zip(names + ["whole_call"], kernel_ms_serial)),
                         "kernel_ms_note": "kernel_ms: HIP events over the TIMED region (with the overlap on, ad_node's interval includes its wait for the CUs the previous step's WBC still holds, and whole_call spans two "
                                           "streams); kernel_ms_one_stream: the same cycle on one stream in a calibration pass after the timed region (outside `value`): what `kernel`, kernel_frac and hbm_gbps are computed "
                                           "from, `achieved` / `frac` included (achieved_from_the_timed_regions_interval: the same FLOPs over the timed region's own interval of that kernel)",
                         # launches within 5 % of the longest one: since round 3 the three FLOP-carrying kernels take 0.45-0.47 ms each, so which of them is
                         # "the dominant kernel" (the longest; `kernel`, `frac` above) changes from run to run -- their fractions are all in kernel_frac
                         "dominant_within_5pct": [k for k in flops if kernel_ms_serial[names.index(k)] >= 0.95 * kernel_ms_serial[names.index(roof_kernel)]],
                         "kernel_frac": {k: (tf(flops[k], kernel_ms_serial[names.index(k)]) or 0.0) / FP64_MFMA_PEAK_TFLOPS for k in flops},
                         # the path: FLOPs of one step over the step's share of the timed region (with the overlap on, the launches of one cycle span more than that)
                         "path_achieved": tf(path_flops, 1e3 * elapsed / args.steps) if not args.emulate else None,
                         "path_frac": (tf(path_flops, 1e3 * elapsed / args.steps) or 0.0) / FP64_MFMA_PEAK_TFLOPS if not args.emulate else 0.0},
        }
        if batch_sweep is not None:
            out["config"]["batch_sweep"] = batch_sweep
        if world == 1 and not args.no_steady_state and not args.emulate:
            from qm_door_amd import api as _api
            sc_ss = dict(sc)
            # one trot schedule covers the whole leg (QMGPU_MAX_EVENTS = 40 mode switches: ~14 s of trot); a longer --steps is cut to 1000 receding-horizon steps here
            ss_steps, ss_warm = min(args.steps, 1000), min(args.warmup, 50)
            t_end = (ss_steps + ss_warm) * 0.01 + N * itf.problem.settings.dt + 0.5
            sc_ss["nev"], sc_ss["ev"], sc_ss["md"] = _api.GaitSchedule(lib=itf.lib).mode_schedule("trot", 0.0, 0.0, t_end)
            out["config"]["steady_state"] = steady_state(itf, sc_ss, ss_steps, ss_warm, wbc_state=args.wbc_state)
        if world == 1 and not args.no_cpu_baseline and not args.emulate:
            out["cpu_baseline"] = cpu_baseline(itf, sc)
        print(json.dumps(out), flush=True)
    if collective:
        dist.destroy_process_group()


if __name__ == "__main__":
    try:
        main()
    except SystemExit:
        raise
    except KeyboardInterrupt:
        raise
    except Exception as e:      # noqa: BLE001 -- say WHICH rank failed, then exit non-zero at once: the launcher tears the other ranks down
        import traceback
        traceback.print_exc()
        print(f"[bench] rank {os.environ.get('RANK', '0')} of {os.environ.get('WORLD_SIZE', '1')} failed: {type(e).__name__}: {e}", file=sys.stderr, flush=True)
        sys.stdout.flush(); sys.stderr.flush()
        # (no destroy_process_group here: with its peers parked in a barrier the hand-shake would never return; the launcher tears them down on this exit code)
        os._exit(1)
