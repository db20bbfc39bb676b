"""CPU model of the device MSM pipeline's combinatorics (segmented reduction slot rules of
k_segreduce and the segmented running sums of k_wsum in msm_var.cuh), run with the oracle's
group law on small inputs.  It checks the ALGORITHM (every bucket written exactly once,
pieces stay adjacent, any digit distribution) independently of the GPU; the HIP kernels
follow this model line by line."""
import random

import pytest

from oracle import bn254 as o

NONE = 0xffffffff
G = o.G1


def segreduce(bids, pts, L, buckets, written):
    n_in = len(bids)
    lanes = (n_in + L - 1) // L
    out_b = [NONE] * (2 * lanes)
    out_p = [G.zero] * (2 * lanes)
    for t in range(lanes):
        s = t * L
        e = min(s + L, n_in)
        first, last = bids[s], bids[e - 1]
        cb = s > 0 and bids[s - 1] == first and first != NONE
        cf = e < n_in and bids[e] == last and last != NONE
        cur, cur_cb, acc = NONE, False, G.zero

        def flush_inside():
            if cur_cb:
                out_b[2 * t], out_p[2 * t] = cur, acc
            else:
                assert cur not in written
                written.add(cur)
                buckets[cur] = acc

        for p in range(s, e):
            b = bids[p]
            if b != cur:
                if cur != NONE:
                    flush_inside()
                cur, cur_cb = b, (p == s and cb)
                if b != NONE:
                    acc = pts[p]
            elif b != NONE:
                acc = G.add(acc, pts[p])
        if cur != NONE:
            if cur_cb:
                out_b[2 * t], out_p[2 * t] = cur, acc
                if cf:
                    out_b[2 * t + 1], out_p[2 * t + 1] = cur, G.zero
            elif cf:
                out_b[2 * t + 1], out_p[2 * t + 1] = cur, acc
            else:
                assert cur not in written
                written.add(cur)
                buckets[cur] = acc
    return out_b, out_p, lanes


RUN_MAX = 4  # small in the model so that both the short- and the long-run paths are exercised


def runmerge(bids, pts, buckets, written):
    """k_runmerge: every run of <= RUN_MAX adjacent slots is summed by the owner of its first slot."""
    n = len(bids)
    out = [NONE] * n
    for x in range(n):
        bx = bids[x]
        if bx == NONE:
            continue
        s = e = x
        while s > 0 and x - s < RUN_MAX and bids[s - 1] == bx:
            s -= 1
        while e + 1 < n and e - x < RUN_MAX and bids[e + 1] == bx:
            e += 1
        short = (x - s < RUN_MAX) and (e - x < RUN_MAX) and (e - s + 1 <= RUN_MAX)
        if not short:
            out[x] = bx
        elif s == x:
            acc = pts[s]
            for q in range(s + 1, e + 1):
                acc = G.add(acc, pts[q])
            assert bx not in written
            written.add(bx)
            buckets[bx] = acc
    return out


def bucket_sums(entries, L1, LK):
    """entries: sorted list of (bid, point)."""
    buckets, written = {}, set()
    bids = [b for b, _ in entries]
    pts = [p for _, p in entries]
    bids, pts, lanes = segreduce(bids, pts, L1, buckets, written)
    bids = runmerge(bids, pts, buckets, written)
    while True:
        bids, pts, lanes = segreduce(bids, pts, LK, buckets, written)
        if lanes == 1:
            break
    assert all(b == NONE for b in bids)
    return buckets


def wsum(B, S):
    """sum_d d*B[d] by the (A, R, g) recursion of k_wsum."""
    m = len(B)
    A = [G.zero] * m
    Rr = list(B)
    g = 0
    sg = S.bit_length() - 1
    while m > 1:
        mo = (m + S - 1) // S
        A2, R2 = [], []
        for j in range(mo):
            lo, hi = j * S, min(j * S + S, m)
            run, ws, asum = G.zero, G.zero, G.zero
            for e in range(hi - 1, lo - 1, -1):
                run = G.add(run, Rr[e])
                if e > lo:
                    ws = G.add(ws, run)
                asum = G.add(asum, A[e])
            for _ in range(g):
                ws = G.twice(ws)
            A2.append(G.add(ws, asum))
            R2.append(run)
        A, Rr, m, g = A2, R2, mo, g + sg
    return A[0]


def model_msm(scalars, bases, c, L1, LK, S):
    W = (256 + c - 1) // c
    entries = []
    for w in range(W):
        for i, s in enumerate(scalars):
            d = (s >> (w * c)) & ((1 << c) - 1)
            if d and not G.is_zero(bases[i]):
                entries.append(((w << c) | d, bases[i]))
    entries.sort(key=lambda t: t[0])
    buckets = bucket_sums(entries, L1, LK) if entries else {}
    res = G.zero
    for w in range(W - 1, -1, -1):
        for _ in range(c):
            res = G.twice(res)
        Bw = [buckets.get((w << c) | d, G.zero) for d in range(1 << c)]
        res = G.add(res, wsum(Bw, S))
    return res


@pytest.mark.parametrize("n,c,L1,LK,S,skew", [
    (37, 3, 4, 4, 2, False), (64, 4, 16, 16, 8, False), (50, 2, 3, 4, 4, True),
    (90, 5, 2, 4, 8, True), (33, 4, 5, 7, 4, False),
])
def test_model_matches_naive(n, c, L1, LK, S, skew):
    rng = random.Random(n * 131 + c)
    bases = [G.mul(G.one, rng.randrange(1, 1000)) for _ in range(n)]
    if skew:
        scalars = [rng.choice([5, 5, 5, o.R - 3, 1, 0, (1 << 200) + 5]) for _ in range(n)]
    else:
        scalars = [rng.randrange(1 << 40) for _ in range(n)]
    got = model_msm(scalars, bases, c, L1, LK, S)
    assert G.equals(got, o.naive_msm(G, scalars, bases))
