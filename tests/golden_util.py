import hashlib
import json
import os

from oracle import bn254 as o
import golden_inputs as gi

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def msm_case_wire(curve, case):
    """-> (bases_wire, scalars_wire) regenerated and digest-checked against the fixture"""
    C, wire = (o.G1, o.g1_to_wire) if curve == "G1" else (o.G2, o.g2_to_wire)
    if "bases" in case:
        bw, sw = bytes.fromhex(case["bases"]), bytes.fromhex(case["scalars"])
    else:
        scalars, bases = gi.msm_inputs(C, case["n"], case["kind"])
        bw = b"".join(wire(P) for P in bases)
        sw = b"".join(o.to_le32(s) for s in scalars)
    assert hashlib.sha256(bw + sw).hexdigest() == case["input_sha256"]
    return bw, sw


def fft_case_wire(case):
    if case.get("input"):
        data = bytes.fromhex(case["input"])
    else:
        data = b"".join(o.to_le32(x) for x in gi.fft_inputs(case["n"]))
    if "input_sha256" in case:
        assert hashlib.sha256(data).hexdigest() == case["input_sha256"]
    return data


def fft_check(case, out):
    if "expected_out" in case:
        assert out == bytes.fromhex(case["expected_out"])
    else:
        assert hashlib.sha256(out).hexdigest() == case["expected_out_sha256"]
        assert out[:64] == bytes.fromhex(case["expected_first"])
        assert out[-64:] == bytes.fromhex(case["expected_last"])
