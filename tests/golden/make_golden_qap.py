#!/usr/bin/env python3
"""Golden fixture for the QAP witness map (oracle/bn254.py qap_witness_coefficients_h ==
R1CStoQAP.java:163-230; the reference holds no vector for it, its own test only checks
QAPRelation.isSatisfied, which tests/test_qap_witness_cpu.py restates).  Run from the repo root:

    python tests/golden/make_golden_qap.py
"""
import hashlib
import json
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import bn254 as o  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def inputs(m, satisfied):
    rng = random.Random(7000 + m + (1 if satisfied else 0))
    a = [rng.randrange(o.R) for _ in range(m)]
    b = [rng.randrange(o.R) for _ in range(m)]
    c = [x * y % o.R for x, y in zip(a, b)] if satisfied else [rng.randrange(o.R) for _ in range(m)]
    return a, b, c


def enc(v):
    return b"".join(o.to_le32(x) for x in v)


def main():
    cases = []
    for m, sat in ((2, True), (8, True), (8, False), (256, True), (1024, True)):
        a, b, c = inputs(m, sat)
        h = o.qap_witness_coefficients_h(a, b, c)
        case = {"m": m, "satisfied": sat, "g": o.FR_MULT_GEN, "omega": o.to_le32(o.fr_root_of_unity(m)).hex(),
                "input_sha256": hashlib.sha256(enc(a) + enc(b) + enc(c)).hexdigest(),
                "expected_sha256": hashlib.sha256(enc(h)).hexdigest()}
        if m <= 8:
            case.update(a=enc(a).hex(), b=enc(b).hex(), c=enc(c).hex(), expected=enc(h).hex())
        cases.append(case)
    json.dump({"cases": cases}, open(os.path.join(HERE, "qap_witness.json"), "w"), indent=1)
    print("ok")


if __name__ == "__main__":
    main()
