"""bench.py on one GPU: contract line and parity of its own inputs.  (The two-rank rehearsal, with its result
checked against the global MSM, is tests/test_sharded_gpu.py.)"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("schedule", ["pipeline3", "pipeline", "streams"])
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
    # the level-1 kernel is timed twice: device clock inside the timed region, HIP events in a second pass
    k, e = d["roofline"]["kernel_ms"], d["roofline"]["kernel_ms_hip_events"]
    assert d["roofline"]["launches_timed"] == 6 and k["min"] <= k["median"] <= k["max"] and k["min"] > 0
    assert e["launches"] == 6 and e["min"] > 0


def test_bench_default_schedule_is_the_three_stage_pipeline():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "1", "--logn", "14",
                          "--no-cpu-baseline"], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert d["config"]["schedule"] == "pipeline3" and d["config"]["msms_in_flight"] == 3
