#!/usr/bin/env python3
"""Host check of tools/ubench_batch_affine: every `CHK xa ya xb yb x3 y3` line (Montgomery residues, R = 2^261, as the
kernel stored them) must satisfy the chord formula over exact integers mod the BN254 base-field prime.
    tools/ubench_batch_affine | tee out.txt; python tools/ubench_batch_affine_check.py out.txt"""
import sys
P = 21888242871839275222246405745257275088696311157297823662689037894645226208583
RINV = pow(1 << 261, -1, P)
n = bad = 0
for line in open(sys.argv[1]):
    if not line.startswith("CHK"):
        continue
    xa, ya, xb, yb, x3, y3 = [int(h, 16) * RINV % P for h in line.split()[1:]]
    lam = (yb - ya) * pow(xb - xa, -1, P) % P
    ex = (lam * lam - xa - xb) % P
    ey = (lam * (xa - ex) - ya) % P
    n += 1
    if (ex, ey) != (x3, y3):
        bad += 1
        print("MISMATCH", line.strip()[:80])
print(f"{n} checked, {bad} wrong")
sys.exit(1 if bad or n == 0 else 0)
