"""CPU oracle for the OctopusZK/DIZK Groth16 hot path (BN254 MSM + radix-2 FFT).

THIS PACKAGE IS TEST INFRASTRUCTURE.  It restates the reference's serial
Java/BigInteger semantics so the HIP path can be checked bit-for-bit.  Only
`tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py`
may import or execute anything under `oracle/`; the product package
(`octopuszk_amd/`) never does and fails loudly when its HIP library is missing.

Pinning: the reference holds no stored BN254 vectors for this path (SURVEY.md
§8c).  `java.math.BigInteger` is exact integer arithmetic, so Python `int`
reproduces it bit-for-bit; the oracle is pinned against every KAT/property the
reference's own tests hold for the path (toy MSM KAT 75 / 60, CurvesTest group
identities, the m=4 FFT KAT, FFT∘IFFT = id) in `tests/test_oracle.py`.
"""
