"""No-GPU checks of the drop-in boundary: the C-ABI library loads and exports every symbol
include/ozk.h declares, the three JNI shim libraries export the six Java_* natives the
reference's Java binds, and the shims turn bad arguments / a missing device into a
java.lang.RuntimeException instead of exiting (no compute is attempted without a GPU)."""
import ctypes
import os
import re
import subprocess

import pytest

import jni_util as ju

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module", autouse=True)
def built():
    from octopuszk_amd import build
    build.build(verbose=False)


def test_c_abi_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "ozk.h")).read()
    declared = set(re.findall(r"\b(ozk_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 18
    lib = ctypes.CDLL(os.path.join(ROOT, "octopuszk_amd", "libozk_hip.so"))
    for name in sorted(declared):
        assert hasattr(lib, name), name
    from octopuszk_amd import lib as binding
    assert declared == set(binding.exported_symbols())
    binding.load()


def test_jni_shims_export_the_reference_natives():
    want = {
        ju.SHIM_VAR: ["Java_algebra_msm_VariableBaseMSM_variableBaseSerialMSMNativeHelper",
                      "Java_algebra_msm_VariableBaseMSM_variableBaseDoubleMSMNativeHelper"],
        ju.SHIM_FIXED: ["Java_algebra_msm_FixedBaseMSM_batchMSMNativeHelper",
                        "Java_algebra_msm_FixedBaseMSM_doubleBatchMSMNativeHelper",
                        "Java_algebra_msm_FixedBaseMSM_fieldBatchMSMNativeHelper"],
        ju.SHIM_FFT: ["Java_algebra_fft_FFTAuxiliary_serialRadix2FFTNativeHelper"],
    }
    for path, syms in want.items():
        out = subprocess.check_output(["nm", "-D", "--defined-only", path]).decode()
        for s in syms:
            assert re.search(r"\bT %s\b" % s, out), (path, s)


def test_shim_argument_validation_throws_runtime_exception():
    with pytest.raises(ju.JavaException, match="shorter than batch_size"):
        ju.var_msm(b"\x00" * 96, b"\x00" * 32, 2, 1)          # arrays too short for n = 2
    with pytest.raises(ju.JavaException, match="batch_size must be positive"):
        ju.var_msm(b"\x00" * 96, b"\x00" * 32, 0, 1)
    with pytest.raises(ju.JavaException, match="power of two"):
        ju.fft([b"\x01\x00\x00\x00"] * 3, b"\x01\x00\x00\x00")
    with pytest.raises(ju.JavaException, match="longer than 32"):
        ju.fft([b"\x01" * 36, b"\x01" * 4], b"\x01\x00\x00\x00")


def test_no_device_is_an_exception_not_a_cpu_fallback():
    from octopuszk_amd import lib
    if lib.load().ozk_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(ju.JavaException, match="no HIP device|no CPU path|failed"):
        ju.var_msm(b"\x00" * 96, b"\x00" * 32, 1, 1)
    from octopuszk_amd import variable_base_msm as vb
    with pytest.raises(lib.OzkError):
        vb.variable_base_serial_msm_native_helper(b"\x00" * 96, b"\x00" * 32, 1, 1, 0)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "octopuszk_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cuh", ".h", ".c", ".cpp")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle" not in text.lower() or f == "__init__.py" and "oracle" not in text, (dirpath, f)
