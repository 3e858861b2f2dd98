"""bench.py's output contract and its multi-rank code path on the one GPU a test box has: --force-collective takes the world > 1 branch
(pack -> RCCL all_gather_into_tensor on a 1-rank nccl group -> unpack) and the gathered block must equal the local results bit for bit."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_line_and_forced_collective():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--force-collective"], capture_output=True,
                       text=True, timeout=600, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in d
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["dtype"] == "f64" and d["scaling"] == "weak" and d["vs_baseline"] is None
    cfg = d["config"]
    assert cfg["collective"].startswith("all_gather") and cfg["gather_matches_local_results"] is True
    assert cfg["gathered_bytes_per_step"] == 256 * (101 * 30 + 100 * 30 + 54 + 101) * 8 and len(cfg["per_rank_value"]) == 1
    assert cfg["results_finite_and_converged"] is True
    assert cfg["workload"].startswith("configs[1]")
    ss = cfg["steady_state"]          # the receding-horizon leg (warm start on the device, robots in motion): replayed inputs reproduce the recorded run bit for bit
    assert ss["replay_reproduces_the_recorded_run_bit_for_bit"] is True and ss["results_finite_and_converged"] is True and ss["value"] > 0
    assert ss["mean_abs_measured_velocity"]["joints"] > 0.01 and set(ss["kernel_ms"]) >= {"ad", "lq", "riccati", "linesearch", "wbc"}
    assert "per_rank_value_min" in cfg and cfg["slowest_rank_over_fastest"] == 1.0
    r = d["roofline"]
    assert r["bound"] == "mfma" and 0.0 < r["frac"] < 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and "traffic_source" in r
    # north_star's evidence fields: matrix-core busy fraction and HBM GB/s of the roofline kernel (counters of the committed PMC passes, labelled)
    assert 0.0 < r["mfma_busy"] < 1.0 and 0.0 < r["hbm_gbps"] < 8000.0 and r["traffic"] > 0 and "profiles/" in r["traffic_source"]
    assert set(r["hbm_gbps_by_kernel"]) >= {"ad_node_kernel", "lq_node_kernel", "riccati_kernel", "wbc_kernel"}


@pytest.mark.gpu
def test_bench_scenario_every_instance_against_oracle(interface):
    """The bench's OWN scenario (bench.build_scenario: EE target from StartingPosition.h constants, schedule from qmgpu_tile_gait) -- all 256 instances of
    one cycle against the multi-threaded oracle, not only `results_finite_and_converged`."""
    import numpy as np
    import torch
    import bench
    import gpu_harness as G
    import support as S
    B, N = bench.BATCH_PER_GPU, bench.HORIZON_N
    sc = bench.build_scenario(interface, B, seed=0)
    sol = G.make_solver(interface, B, N)
    mb = G.MpcBatch(sc["x0"], sc["tt"], sc["ts"], np.full(B, sc["nev"], dtype=np.int32), np.tile(sc["ev"], (B, 1)), np.tile(sc["md"], (B, 1)), N)
    wb = G.WbcBatch(sc["rbd"], np.full(B, 0.002), np.full(B, 20.0), np.zeros((B, 30)))
    sol.cycle(mb.args, G.dev(np.zeros(B), torch.float64), wb.args)
    got = mb.results(); got.update(wb.results())
    ref = S.Oracle(interface.problem, fast=True).cycle_batch(N, sc["x0"], sc["tt"], sc["ts"], sc["nev"], sc["ev"], sc["md"], rbd=sc["rbd"])
    S.assert_parity(S.parity_report("bench_scenario_configs1_256xN100", got, ref))


def test_bench_two_ranks_on_the_emulation_path():
    """CPU: `python bench.py --gpus 2` without a launcher environment starts its own two ranks (torch.distributed.run on 127.0.0.1); with --emulate
    the kernels are the host-emulated ones and the collective runs over gloo.  End to end: configs[2] shards of ONE global batch, all-gather, the line."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--emulate", "--batch-per-gpu", "2", "--nodes", "4", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                       # rank 0 prints the one line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["steps"] == 1
    cfg = d["config"]
    assert cfg["workload"].startswith("configs[2]: ONE global batch of 4") and cfg["global_batch"] == 4 and cfg["batch_per_gpu"] == 2
    assert cfg["gather_matches_local_results"] is True and cfg["results_finite_and_converged"] is True
    assert len(cfg["per_rank_value"]) == 2 and cfg["collective"].startswith("all_gather")
    assert "EMULATED" in d["data"] and "cpu_baseline" not in d


def test_bench_eight_ranks_take_the_whole_global_batch_on_the_emulation_path():
    """CPU: the launch the driver makes on an 8-GPU node -- eight ranks, configs[2]'s ONE global batch taken WHOLE (the branch no 1- or 2-rank run reaches), contiguous
    shards, all-gather, the line -- at toy size on the host-emulated kernels over gloo (--global-batch 8: one instance per rank)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--emulate", "--batch-per-gpu", "1", "--global-batch", "8", "--nodes", "4", "--steps", "1",
                        "--warmup", "0"], capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    cfg = d["config"]
    assert d["n_gpus"] == 8 and cfg["global_batch"] == 8 and cfg["batch_per_gpu"] == 1 and len(cfg["per_rank_value"]) == 8
    assert cfg["workload"].startswith("configs[2]: ONE global batch of 8 MPC instances (all of the 8 of seed 1)")
    assert cfg["gather_matches_local_results"] is True and cfg["results_finite_and_converged"] is True


def test_bench_refuses_a_changed_workload_outside_the_emulation():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--batch-per-gpu", "8"], capture_output=True, text=True, timeout=120, cwd=ROOT)
    assert p.returncode != 0 and "BASELINE" in (p.stderr + p.stdout)


def test_a_rank_that_fails_before_the_first_barrier_ends_the_whole_job_promptly():
    """VERDICT r03 missing 1: N > 1 has never run on hardware here, so what can be checked without it is.  One of two ranks raises before the first barrier
    (QM_BENCH_FAIL_RANK): the job must exit non-zero within seconds -- not sit in dist.barrier() until a 30-minute collective timeout -- and say which rank failed."""
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["QM_BENCH_FAIL_RANK"] = "1"
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--emulate", "--batch-per-gpu", "2", "--nodes", "4", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    took = time.time() - t0
    assert p.returncode != 0
    assert "[bench] rank 1 of 2 failed: RuntimeError: injected failure" in p.stderr, p.stderr[-3000:]
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]          # no result line from a job that lost a rank
    assert took < 120, took


def test_rank_to_device_mapping_honours_the_visible_devices():
    """LOCAL_RANK -> device: torch.cuda.device_count() already reflects HIP_VISIBLE_DEVICES, so a launcher that starts more ranks than visible devices is
    reported as such (with the variable's value) instead of surfacing as an invalid device ordinal from the first allocation."""
    import bench
    assert [bench.device_for_rank(r, 8) for r in range(8)] == list(range(8))
    assert bench.device_for_rank(1, 2, visible="3,5") == 1                      # the SECOND visible device, whatever its physical index
    for bad in ((2, 2), (8, 8), (-1, 4), (0, 0)):
        with pytest.raises(SystemExit) as e:
            bench.device_for_rank(*bad, visible="0,1")
        assert "LOCAL_RANK" in str(e.value) and "HIP_VISIBLE_DEVICES=0,1" in str(e.value)


@pytest.mark.gpu
def test_bench_refuses_a_local_rank_without_a_device():
    """On a one-GPU box: LOCAL_RANK 1 must fail at once with the attributed message."""
    env = dict(os.environ, LOCAL_RANK="1", HIP_VISIBLE_DEVICES="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-steady-state"], capture_output=True, text=True,
                       timeout=300, env=env, cwd=ROOT)
    assert p.returncode != 0 and "LOCAL_RANK 1 has no GPU" in (p.stderr + p.stdout)
