#!/usr/bin/env python3
"""Derive and cross-check the GLV constants of octopuszk_amd/csrc/glv.cuh (prints them).
lambda / beta: primitive cube roots of unity mod r / mod q with (beta x, y) = lambda (x, y) on G1;
lattice basis: extended Euclid on (r, lambda) (Gallant-Lambert-Vanstone 2001, Algorithm 3.74 in
Hankerson-Menezes-Vanstone)."""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import bn254 as o  # noqa: E402

r, q = o.R, o.Q
LAM = 4407920970296243842393367215006156084916469457145843978461
BETA1 = 2203960485148121921418603742825762020974279258880205651966
BETA2 = BETA1 * BETA1 % q
assert (LAM * LAM + LAM + 1) % r == 0 and (BETA1 ** 2 + BETA1 + 1) % q == 0
P = o.G1.to_affine(o.G1.mul(o.G1.one, 123456789))
assert o.G1.equals((BETA1 * P[0] % q, P[1], 1), o.G1.mul(P, LAM))
Q2 = o.G2.to_affine(o.G2.mul(o.G2.one, 987654321))
assert o.G2.equals(((BETA2 * Q2[0][0] % q, BETA2 * Q2[0][1] % q), Q2[1], (1, 0)), o.G2.mul(Q2, LAM))
rs, ts = [r, LAM], [0, 1]
while rs[-1]:
    qq = rs[-2] // rs[-1]
    rs.append(rs[-2] - qq * rs[-1])
    ts.append(ts[-2] - qq * ts[-1])
s = math.isqrt(r)
l = max(i for i in range(len(rs)) if rs[i] >= s)
v1 = (rs[l + 1], -ts[l + 1])
c1, c2 = (rs[l], -ts[l]), (rs[l + 2], -ts[l + 2])
v2 = c1 if c1[0] ** 2 + c1[1] ** 2 <= c2[0] ** 2 + c2[1] ** 2 else c2
(a1, b1), (a2, b2) = v1, v2
assert (a1 + b1 * LAM) % r == 0 and (a2 + b2 * LAM) % r == 0 and a1 * b2 - a2 * b1 == r
assert b1 < 0 < b2 and a1 > 0 and a2 > 0
print("lambda", LAM)
print("beta_G1", BETA1, "beta_G2", BETA2)
print("a1", a1, "|b1|", -b1, "a2", a2, "b2", b2)
print("g1 = floor(2^256 b2 / r)", hex((b2 << 256) // r))
print("g2 = floor(2^256 |b1| / r)", hex(((-b1) << 256) // r))
