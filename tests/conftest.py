import ctypes
import os
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# ---- evidence for failures that come and go (VERDICT r3 "weak" 1) --------------------------------------------------
# Every GPU test leaves one line in gpurun_out/gpu_suite_events.log: what the calling thread's HIP error state was
# when the test started (hipPeekAtLastError: a non-zero value there was left by somebody else's HIP call — torch's
# probes, a previous test — and used to be reported by the library's next launch check as its own failure), and how
# the test ended.  Written as the suite runs, so a killed run keeps what it had.
_hip = None


def _hip_runtime():
    """the HIP runtime already mapped into this process (torch's), never a second copy"""
    global _hip
    if _hip is None:
        _hip = False
        try:
            import torch  # noqa: F401
            with open("/proc/self/maps") as f:
                paths = {l.split()[-1] for l in f if "libamdhip64" in l}
            for p in sorted(paths):
                _hip = ctypes.CDLL(p)
                break
        except Exception:
            _hip = False
    return _hip


def _event_log():
    d = os.path.join(ROOT, "gpurun_out")
    os.makedirs(d, exist_ok=True)
    return os.path.join(d, "gpu_suite_events.log")


@pytest.fixture(autouse=True)
def _gpu_test_record(request):
    if request.node.get_closest_marker("gpu") is None:
        yield
        return
    hip = _hip_runtime()
    stale = -1
    if hip:
        hip.hipPeekAtLastError.restype = ctypes.c_int
        stale = hip.hipPeekAtLastError()
    t0 = time.time()
    yield
    rep = getattr(request.node, "_ozk_rep", None)
    outcome = rep.outcome if rep is not None else "?"
    with open(_event_log(), "a") as f:
        f.write("%s\tstale_hip_error_at_start=%d\t%s\t%.2fs\n" % (request.node.nodeid, stale, outcome, time.time() - t0))


@pytest.hookimpl(hookwrapper=True)
def pytest_runtest_makereport(item, call):
    out = yield
    rep = out.get_result()
    if rep.when == "call":
        item._ozk_rep = rep
