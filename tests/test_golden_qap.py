"""The committed QAP witness-map fixture (tests/golden/qap_witness.json): the oracle reproduces it on
CPU, the HIP path reproduces it on the GPU."""
import hashlib
import importlib.util
import os

import pytest

from oracle import bn254 as o
import golden_util as gu

_spec = importlib.util.spec_from_file_location(
    "make_golden_qap", os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "make_golden_qap.py"))
mk = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(mk)

CASES = gu.load("qap_witness.json")["cases"]
IDS = ["m%d%s" % (c["m"], "" if c["satisfied"] else "_unsat") for c in CASES]


def _inputs(case):
    a, b, c = mk.inputs(case["m"], case["satisfied"])
    assert hashlib.sha256(mk.enc(a) + mk.enc(b) + mk.enc(c)).hexdigest() == case["input_sha256"]
    if "a" in case:
        assert mk.enc(a).hex() == case["a"] and mk.enc(b).hex() == case["b"] and mk.enc(c).hex() == case["c"]
    return a, b, c


def _check(case, h):
    raw = mk.enc(h)
    assert hashlib.sha256(raw).hexdigest() == case["expected_sha256"]
    if "expected" in case:
        assert raw.hex() == case["expected"]


@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_oracle_reproduces_fixture(case):
    a, b, c = _inputs(case)
    _check(case, o.qap_witness_coefficients_h(a, b, c, case["g"]))


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_hip_reproduces_fixture(case):
    from octopuszk_amd import r1cs_to_qap as q
    a, b, c = _inputs(case)
    _check(case, q.coefficients_h(a, b, c, case["g"]))
