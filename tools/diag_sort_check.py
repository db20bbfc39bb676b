"""Diagnostic: full validity check of the SORT stage on the profiler-shaped (skewed) 2^20 input against a
host model of the GLV split + signed recoding: every (virtual point, window) with a non-zero digit must
appear exactly once, in the right bucket, with the right sign, and bucket ids must be non-decreasing."""
import ctypes, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from octopuszk_amd import lib as ozk
from oracle import bn254 as o
L = ozk.load()
def ptr(t): return ctypes.c_void_p(t.data_ptr())
n = 1 << int(os.environ.get("DIAG_LOGN", "20"))
rng = np.random.default_rng(10)
lows = rng.integers(0, 1 << 63, size=n, dtype=np.uint64)
neg = rng.integers(0, 2, size=n).astype(bool)
vals = [(o.R - int(v)) if ng else int(v) for v, ng in zip(lows, neg)]
sc = np.frombuffer(b"".join(v.to_bytes(32, "little") for v in vals), dtype=np.uint8).copy()
base = o.G1.to_affine(o.G1.mul(o.G1.one, 987654321))
bases = np.frombuffer(o.g1_to_wire(base) * n, dtype=np.uint8).copy()
d_bases, d_scalars = torch.from_numpy(bases).cuda(), torch.from_numpy(sc).cuda()
wb, wn = ctypes.c_int32(), ctypes.c_int32()
ozk.check(L.ozk_var_msm_plan(n, ctypes.byref(wb), ctypes.byref(wn)))
c, W = wb.value, wn.value
assert L.ozk_var_msm_glv(n) == 1
ne, cb = 2 * n, c - 1
# ---- host model (tests/test_glv.py model + signed_digit_codes)
A1 = 9931322734385697763; B1 = -147946756881789319000765030803803410728
A2 = 147946756881789319010696353538189108491; B2 = 9931322734385697763
g1 = (B2 << 256) // o.R; g2 = ((-B1) << 256) // o.R
exp_b = np.full((ne, W), -1, dtype=np.int32); exp_s = np.zeros((ne, W), dtype=np.int8)
half, mask = 1 << (c - 1), (1 << c) - 1
t0 = time.time()
for i, k in enumerate(vals):
    k %= o.R
    c1 = (k * g1) >> 256; c2 = (k * g2) >> 256
    k1 = k - c1 * A1 - c2 * A2; k2 = -c1 * B1 - c2 * B2
    for hidx, kk in ((i, k1), (n + i, k2)):
        ng = kk < 0; m = -kk if ng else kk
        thr = half - 1 if (c == 16 and ng) else half
        cy = 0
        for w in range(W):
            d = ((m >> (c * w)) & mask) + cy
            cy = 1 if d > thr else 0
            mag = (1 << c) - d if cy else d
            if mag:
                exp_b[hidx, w] = mag - 1
                exp_s[hidx, w] = cy ^ int(ng)
print("host model %.1f s; non-zero digits %d" % (time.time() - t0, int((exp_b >= 0).sum())), flush=True)
sb, swb, ab = ctypes.c_size_t(), ctypes.c_size_t(), ctypes.c_size_t()
ozk.check(L.ozk_var_msm_stage_bytes(n, 1, ctypes.byref(sb), ctypes.byref(swb), ctypes.byref(ab)))
d_sorted = torch.zeros(sb.value, dtype=torch.uint8, device="cuda")
d_sortws = torch.zeros(swb.value, dtype=torch.uint8, device="cuda")
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
NB = W << cb; cap = ne * W
a256 = lambda x: (x + 255) & ~255
off_hist = a256(ne * 64); off_total = a256(off_hist + NB * 4); off_sent = a256(off_total + 16)   # (round 4: the sorted entries are 8-byte pairs (index | sign << 31, bucket id))
want_total = int((exp_b >= 0).sum())
for rep in range(int(os.environ.get("DIAG_REPS", "12"))):
    d_sorted.zero_(); d_sortws.zero_()
    ozk.check(L.ozk_var_msm_sort_dev(ptr(d_bases), ptr(d_scalars), n, 1, ptr(d_sorted), sb.value, ptr(d_sortws), swb.value, st))
    torch.cuda.synchronize()
    total = int(d_sorted[off_total:off_total + 4].view(torch.int32)[0])
    msgs = []
    if total != want_total: msgs.append("total %d != %d" % (total, want_total))
    m = min(total, cap)
    sidx = d_sorted[off_sent:off_sent + m * 8].view(torch.int32).cpu().numpy().astype(np.int64).reshape(-1, 2)[:, 0] & 0xffffffff
    sbid = d_sorted[off_sent:off_sent + m * 8].view(torch.int32).cpu().numpy().astype(np.int64).reshape(-1, 2)[:, 1] & 0xffffffff
    hist = d_sorted[off_hist:off_hist + NB * 4].view(torch.int32).cpu().numpy()
    v = sidx & 0xffffff; s = (sidx >> 31) & 1; w = sbid >> cb; b = sbid & ((1 << cb) - 1)
    inr = (v < ne) & (w < W)
    if not inr.all(): msgs.append("%d entries out of range" % int((~inr).sum()))
    vv, ww, bb_, ss = v[inr], w[inr], b[inr], s[inr]
    okb = exp_b[vv, ww] == bb_; oks = exp_s[vv, ww] == ss
    if not okb.all(): msgs.append("%d entries in the wrong bucket" % int((~okb).sum()))
    if not oks.all(): msgs.append("%d entries with the wrong sign" % int((~oks).sum()))
    key = vv * W + ww
    uq = np.unique(key)
    if len(uq) != len(key): msgs.append("%d duplicate (point, window) entries" % (len(key) - len(uq)))
    if len(uq) != want_total: msgs.append("%d (point, window) pairs missing" % (want_total - len(uq)))
    if not (sbid[1:] >= sbid[:-1]).all():
        bad = np.nonzero(sbid[1:] < sbid[:-1])[0]
        msgs.append("bucket ids not sorted at %d places (first %d: %x -> %x)" % (len(bad), bad[0], sbid[bad[0]], sbid[bad[0] + 1]))
    cnt = np.bincount(sbid[inr], minlength=NB)[:NB]
    if not (cnt == hist).all(): msgs.append("bucket counts differ from hist in %d buckets" % int((cnt != hist).sum()))
    print("rep", rep, "OK" if not msgs else "; ".join(msgs), flush=True)
