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
    r = d["roofline"]
    assert r["bound"] == "mfma" and 0.0 < r["frac"] < 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and "traffic_source" in r
