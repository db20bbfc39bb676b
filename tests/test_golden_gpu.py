"""HIP path (through the C ABI) against the committed golden fixtures: bit-exact."""
import ctypes

import pytest

import golden_util as gu

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", gu.load("var_msm_g1.json")["cases"], ids=lambda c: c["name"])
def test_hip_var_msm_g1_golden(case):
    from octopuszk_amd import variable_base_msm as vb
    bw, sw = gu.msm_case_wire("G1", case)
    assert vb.variable_base_serial_msm_native_helper(bw, sw, case["n"], 1, 0) == bytes.fromhex(case["expected_out"])


@pytest.mark.parametrize("case", gu.load("var_msm_g2.json")["cases"], ids=lambda c: c["name"])
def test_hip_var_msm_g2_golden(case):
    from octopuszk_amd import variable_base_msm as vb
    bw, sw = gu.msm_case_wire("G2", case)
    assert vb.variable_base_serial_msm_native_helper(bw, sw, case["n"], 2, 0) == bytes.fromhex(case["expected_out"])
