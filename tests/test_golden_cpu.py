"""The C oracle (the CPU baseline / large-size checker) against the committed golden
fixtures (tests/golden/*.json, made by the exact-integer Python oracle)."""
import pytest

from oracle import coracle
import golden_util as gu


@pytest.mark.parametrize("case", gu.load("var_msm_g1.json")["cases"], ids=lambda c: c["name"])
def test_c_oracle_var_msm_g1(case):
    bw, sw = gu.msm_case_wire("G1", case)
    assert coracle.pippenger_g1(bw, sw, case["n"]) == bytes.fromhex(case["expected_out"])


@pytest.mark.parametrize("case", gu.load("fft_fr.json")["cases"], ids=lambda c: "n%d%s" % (c["n"], "_kat" if "kat" in c else ""))
def test_c_oracle_fft(case):
    data = gu.fft_case_wire(case)
    gu.fft_check(case, coracle.fft_fr(data, case["n"], bytes.fromhex(case["omega"])))


def test_c_oracle_fixed_base_and_field_mul():
    for case in gu.load("fixed_base.json")["cases"]:
        if case["curve"] == "G1":
            sc = bytes.fromhex(case["scalars"])
            got = coracle.fixed_base_g1(bytes.fromhex(case["base"]), sc, len(sc) // 32, case["outerc"], case["window"])
            assert got == bytes.fromhex(case["expected_out"])
        elif case["curve"] == "Fr":
            got = coracle.field_batch_mul(bytes.fromhex(case["field_mul_in"]), case["n"])
            assert got == bytes.fromhex(case["expected_out"])
