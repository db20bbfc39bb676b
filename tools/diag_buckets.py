"""Diagnostic: on the profiler-shaped 2^20 input (ONE base), every bucket must hold net_count * P (first
GLV half) or net_count * phi(P) (second half).  Run sort + accumulate, read the bucket records back, and
list the buckets that are wrong."""
import ctypes, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from octopuszk_amd import lib as ozk
from oracle import bn254 as o
L = ozk.load()
def ptr(t): return ctypes.c_void_p(t.data_ptr())
n = 1 << 20
rng = np.random.default_rng(10)
lows = rng.integers(0, 1 << 63, size=n, dtype=np.uint64)
neg = rng.integers(0, 2, size=n).astype(bool)
vals = [(o.R - int(v)) if ng else int(v) for v, ng in zip(lows, neg)]
sc = np.frombuffer(b"".join(v.to_bytes(32, "little") for v in vals), dtype=np.uint8).copy()
base = o.G1.to_affine(o.G1.mul(o.G1.one, 987654321))
bases = np.frombuffer(o.g1_to_wire(base) * n, dtype=np.uint8).copy()
d_bases, d_scalars = torch.from_numpy(bases).cuda(), torch.from_numpy(sc).cuda()
want = o.g1_out_le(o.G1.to_affine(o.G1.mul(base, sum(vals) % o.R)))
wb, wn = ctypes.c_int32(), ctypes.c_int32()
ozk.check(L.ozk_var_msm_plan(n, ctypes.byref(wb), ctypes.byref(wn)))
c, W = wb.value, wn.value
ne, cb = 2 * n, c - 1
A1 = 9931322734385697763; B1 = -147946756881789319000765030803803410728
A2 = 147946756881789319010696353538189108491; B2 = 9931322734385697763
g1 = (B2 << 256) // o.R; g2 = ((-B1) << 256) // o.R
LAM = 4407920970296243842393367215006156084916469457145843978461
NB = W << cb
net = np.zeros((2, NB), dtype=np.int64)   # [half][bucket] signed count
cntb = np.zeros(NB, dtype=np.int64)
half, mask = 1 << (c - 1), (1 << c) - 1
for i, k in enumerate(vals):
    k %= o.R
    c1 = (k * g1) >> 256; c2 = (k * g2) >> 256
    k1 = k - c1 * A1 - c2 * A2; k2 = -c1 * B1 - c2 * B2
    for h, kk in ((0, k1), (1, k2)):
        ng = kk < 0; m = -kk if ng else kk
        thr = half - 1 if (c == 16 and ng) else half
        cy = 0
        for w in range(W):
            d = ((m >> (c * w)) & mask) + cy
            cy = 1 if d > thr else 0
            mag = (1 << c) - d if cy else d
            if mag:
                b = (w << cb) | (mag - 1)
                net[h, b] += -1 if (cy ^ int(ng)) else 1
                cntb[b] += 1
print("host model done", flush=True)
sb, swb, ab = ctypes.c_size_t(), ctypes.c_size_t(), ctypes.c_size_t()
ozk.check(L.ozk_var_msm_stage_bytes(n, 1, ctypes.byref(sb), ctypes.byref(swb), ctypes.byref(ab)))
tb = int(L.ozk_var_msm_tail_bytes(n, 1))
fillv = int(os.environ.get("DIAG_FILLV", "255"))
d_sorted = torch.zeros(sb.value, dtype=torch.uint8, device="cuda")
d_sortws = torch.zeros(swb.value, dtype=torch.uint8, device="cuda")
d_acc = torch.full((ab.value,), fillv, dtype=torch.uint8, device="cuda")
d_tail = torch.full((tb,), fillv, dtype=torch.uint8, device="cuda")
d_out = torch.zeros(192, dtype=torch.uint8, device="cuda")
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
a256 = lambda x: (x + 255) & ~255
cap = ne * W
off_hist = a256(ne * 64); off_total = a256(off_hist + NB * 4); off_sent = a256(off_total + 16)   # (round 4: the sorted entries are 8-byte pairs (index | sign << 31, bucket id))
REC = 40
for attempt in range(12):
    d_sorted.zero_(); d_sortws.zero_(); d_acc.fill_(fillv); d_tail.fill_(fillv)
    ozk.check(L.ozk_var_msm_sort_dev(ptr(d_bases), ptr(d_scalars), n, 1, ptr(d_sorted), sb.value, ptr(d_sortws), swb.value, st))
    torch.cuda.synchronize()
    total = int(d_sorted[off_total:off_total + 4].view(torch.int32)[0])
    hist_sort = d_sorted[off_hist:off_hist + NB * 4].view(torch.int32).cpu().numpy().copy()
    sbid0 = d_sorted[off_sent:off_sent + total * 8].view(torch.int32).cpu().numpy().astype(np.int64).reshape(-1, 2)[:, 1] & 0xffffffff
    sidx0 = d_sorted[off_sent:off_sent + total * 8].view(torch.int32).cpu().numpy().astype(np.int64).reshape(-1, 2)[:, 0] & 0xffffffff
    ozk.check(L.ozk_var_msm_accum_dev(n, 1, ptr(d_sorted), sb.value, ptr(d_acc), ab.value, ptr(d_tail), tb, st))
    torch.cuda.synchronize()
    recs = d_tail[:NB * REC * 4].view(torch.int32).cpu().numpy().astype(np.int64).reshape(NB, REC) & 0xffffffff
    hist_t = d_tail[NB * REC * 4:NB * REC * 4 + NB * 4].view(torch.int32).cpu().numpy().copy()
    hist_after = d_sorted[off_hist:off_hist + NB * 4].view(torch.int32).cpu().numpy().copy()
    sbid1 = d_sorted[off_sent:off_sent + total * 8].view(torch.int32).cpu().numpy().astype(np.int64).reshape(-1, 2)[:, 1] & 0xffffffff
    ozk.check(L.ozk_var_msm_tail_dev(n, 1, ptr(d_tail), tb, ptr(d_out), st))
    torch.cuda.synchronize()
    ok = bytes(d_out.cpu().numpy()) == want
    print("attempt", attempt, "result ok:", ok, "| total", total, "| hist(sort)==model", bool((hist_sort == cntb).all()),
          "| hist unchanged by accum", bool((hist_sort == hist_after).all()), "| hist_t==hist", bool((hist_t == hist_sort).all()),
          "| sbid unchanged by accum", bool((sbid0 == sbid1).all()), "| sbid sorted", bool((sbid0[1:] >= sbid0[:-1]).all()),
          "| bincount(sbid)==hist", bool((np.bincount(sbid0, minlength=NB)[:NB] == hist_sort).all()), flush=True)
    if not ok:
        for nm, arr in (("hist(sort)", hist_sort), ("hist_t", hist_t)):
            d = np.nonzero(arr != cntb)[0]
            print(" ", nm, "differs from the model in", len(d), "buckets:", [(hex(int(i)), int(arr[i]), int(cntb[i])) for i in d[:8]])
        bc = np.bincount(sbid0, minlength=NB)[:NB]
        d = np.nonzero(bc != cntb)[0]
        print("  bincount(sbid) differs from the model in", len(d), "buckets:", [(hex(int(i)), int(bc[i]), int(cntb[i])) for i in d[:8]])
        break
if ok:
    print("no failing attempt"); sys.exit(0)
RINV = pow(1 << 261, -1, o.Q)
def fe(words):
    return sum(int(x) << (29 * i) for i, x in enumerate(words)) * RINV % o.Q
P = base
phiP = (2203960485148121921418603742825762020974279258880205651966 * P[0] % o.Q, P[1], 1)
G = o.G1
bad = []
t0 = time.time()
suspects = set(int(i) for i in np.nonzero(cntb > 200)[0]) | set(range(0, 64)) | set(int(i) for i in np.nonzero(hist_t != cntb)[0])
for b in sorted(suspects):
    if cntb[b] == 0: continue
    r = recs[b]
    tag = r[36]
    if tag == 0:    # XYZZ
        X, Y, ZZ, ZZZ = fe(r[0:9]), fe(r[9:18]), fe(r[18:27]), fe(r[27:36])
        got = G.zero if ZZ == 0 else (X * pow(ZZ, -1, o.Q) % o.Q, Y * pow(ZZZ, -1, o.Q) % o.Q, 1)
    elif tag == 1:  # Jacobian
        X, Y, Z = fe(r[0:9]), fe(r[9:18]), fe(r[18:27])
        got = (X, Y, Z)
    else:
        bad.append((b, int(cntb[b]), "tag %x (never written)" % tag)); continue
    exp = G.add(G.mul(P, int(net[0, b]) % o.R), G.mul(phiP, int(net[1, b]) % o.R))
    if not G.equals(got, exp):
        bad.append((b, int(cntb[b]), "tag %d wrong value; net %d/%d" % (tag, net[0, b], net[1, b])))
print("checked %d suspect buckets in %.0f s; wrong: %d" % (len(suspects), time.time() - t0, len(bad)), flush=True)
for x in bad[:40]: print("  bucket %x (window %d) entries %d: %s" % (x[0], x[0] >> cb, x[1], x[2]))
