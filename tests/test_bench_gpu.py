"""bench.py's N > 1 path rehearsed on ONE GPU: two ranks share cuda:0 and exchange over gloo
(OZK_BENCH_REHEARSAL=1).  Checks the contract line and that the 2-rank result is the sum of two
different per-rank MSMs (not a copy of one)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_two_ranks_on_one_gpu():
    env = dict(os.environ, OZK_BENCH_REHEARSAL="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29517", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4",
           "--warmup", "1", "--logn", "16"]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["scaling"] == "weak" and d["unit"] == "Mscalar-mul/s"
    assert d["value"] > 0 and d["cpu_baseline"] is None and d["roofline"]["bound"] == "hbm"
    assert "x2" in d["config"]["parallelism"]


@pytest.mark.parametrize("schedule", ["pipeline", "streams"])
def test_bench_single_gpu_contract_and_parity(schedule):
    # bench.py asserts GPU bytes == C oracle bytes on its own inputs (cpu_baseline leg) in both schedules
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "6", "--warmup", "1", "--logn", "16",
                          "--schedule", schedule], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d
    assert d["n_gpus"] == 1 and d["steps"] == 6 and d["vs_baseline"] is None and d["config"]["schedule"] == schedule
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(d["roofline"])
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] == 1 and d["cpu_baseline"]["value"] > 0
