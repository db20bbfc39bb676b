"""java.util.Random restated (TEST INFRASTRUCTURE ONLY, see oracle/__init__.py).

The reference draws every "random" field element as
`new Random(seed).nextLong()` reduced mod p (algebra/fields/Fp.java:72-80) and
its micro-benchmark inputs the same way
(profiler/profiling/VariableBaseMSMProfiling.java:19-31).  java.util.Random is a
JDK class (not under /root/reference); its 48-bit LCG is specified in the JDK
API documentation: seed' = (seed * 0x5DEECE66D + 0xB) mod 2^48,
next(bits) = seed' >> (48 - bits); nextLong = (next(32) << 32) + next(32) with
signed 32-bit halves.
"""

_MULT = 0x5DEECE66D
_MASK = (1 << 48) - 1


class JavaRandom:
    def __init__(self, seed: int):
        self.seed = (seed ^ _MULT) & _MASK

    def _next(self, bits: int) -> int:
        self.seed = (self.seed * _MULT + 0xB) & _MASK
        v = self.seed >> (48 - bits)
        if v >= 1 << (bits - 1):  # signed int
            v -= 1 << bits
        return v

    def next_long(self) -> int:
        v = (self._next(32) << 32) + self._next(32)
        v &= (1 << 64) - 1
        if v >= 1 << 63:
            v -= 1 << 64
        return v


def fp_random(seed: int, modulus: int) -> int:
    """Fp.java:72-80: new Fp(new Random(seed).nextLong(), params) — the
    constructor reduces mod p (Fp.java:21-24), so negative longs wrap to p - |v|."""
    return JavaRandom(seed).next_long() % modulus
