"""Diagnostic: per-item counts of the big-bin sort (k_sortbig_*) on the profiler-shaped 2^20 input,
reconstructed from the per-item bases the scan leaves in T, against numpy counts of the coarse array."""
import ctypes, os, sys
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
wb, wn = ctypes.c_int32(), ctypes.c_int32()
ozk.check(L.ozk_var_msm_plan(n, ctypes.byref(wb), ctypes.byref(wn)))
c, W = wb.value, wn.value
ne, cb = 2 * n, c - 1
lo_bits = 7; NH = 1 << (cb - lo_bits); nblk = (ne + 4095) // 4096
nC1 = W * NH * nblk; cap = ne * W; NB = W << cb
a256 = lambda x: (x + 255) & ~255
o_C1 = 0; o_P1 = a256(o_C1 + nC1 * 4); o_bs = a256(o_P1 + nC1 * 4); o_dig = a256(o_bs + (nC1 // 4096 + 2) * 4)
o_neg = a256(o_dig + cap * 2); o_coarse = a256(o_neg + ne); o_bb = a256(o_coarse + cap * 4)
BB_WORDS = 2 + 4 * 1024
o_T = a256(o_bb + BB_WORDS * 4)
off_hist = a256(ne * 64); off_total = a256(off_hist + NB * 4)
sb, swb, ab = ctypes.c_size_t(), ctypes.c_size_t(), ctypes.c_size_t()
ozk.check(L.ozk_var_msm_stage_bytes(n, 1, ctypes.byref(sb), ctypes.byref(swb), ctypes.byref(ab)))
d_sorted = torch.zeros(sb.value, dtype=torch.uint8, device="cuda")
d_sortws = torch.zeros(swb.value, dtype=torch.uint8, device="cuda")
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for attempt in range(int(os.environ.get("DIAG_REPS", "8"))):
    d_sorted.zero_(); d_sortws.zero_()
    ozk.check(L.ozk_var_msm_sort_dev(ptr(d_bases), ptr(d_scalars), n, 1, ptr(d_sorted), sb.value, ptr(d_sortws), swb.value, st))
    torch.cuda.synchronize()
    total = int(d_sorted[off_total:off_total + 4].view(torch.int32)[0])
    hist = d_sorted[off_hist:off_hist + NB * 4].view(torch.int32).cpu().numpy().astype(np.int64)
    coarse = d_sortws[o_coarse:o_coarse + total * 4].view(torch.int32).cpu().numpy().astype(np.int64) & 0xffffffff
    bb = d_sortws[o_bb:o_bb + BB_WORDS * 4].view(torch.int32).cpu().numpy().astype(np.int64) & 0xffffffff
    n_big, n_items = int(bb[0]), int(bb[1])
    bins, b0s, sizes, firsts = bb[2:2 + 1024], bb[2 + 1024:2 + 2048], bb[2 + 2048:2 + 3072], bb[2 + 3072:2 + 4096]
    T = d_sortws[o_T:o_T + n_items * 256 * 4].view(torch.int32).cpu().numpy().astype(np.int64).reshape(n_items, 256) & 0xffffffff
    problems = []
    cover = np.zeros(n_items, dtype=np.int64)
    for r in range(n_big):
        items = (int(sizes[r]) + 8191) // 8192
        f = int(firsts[r]); cover[f:f + items] += 1
        w, h = int(bins[r]) // NH, int(bins[r]) % NH
        bucket0 = (w << cb) | (h << lo_bits)
        # expected bucket bases inside the bin
        seg = coarse[int(b0s[r]):int(b0s[r]) + int(sizes[r])] & 0x7f
        tot = np.bincount(seg, minlength=128)[:128]
        start = int(b0s[r]) + np.concatenate(([0], np.cumsum(tot)[:-1]))
        run = start.copy()
        for k in range(items):
            chunk = seg[k * 8192:(k + 1) * 8192]
            cnt = np.bincount(chunk, minlength=128)[:128]
            if False and not (T[f + k, :128] == run).all():
                t = int(np.nonzero(T[f + k, :128] != run)[0][0])
                problems.append("bin %d (w %d h %x size %d) item %d/%d (global item %d): base for lo %x is %d, expected %d (delta %d)"
                                % (r, w, h, sizes[r], k, items, f + k, t, T[f + k, t], run[t], T[f + k, t] - run[t]))
            run += cnt
        # per-item counts reconstructed from consecutive bases, for every lo with > 1000 entries
        for t in np.nonzero(tot > 1000)[0]:
            basesT = T[f:f + items, t]
            end_t = int(b0s[r]) + int(np.cumsum(tot)[t])
            gpu_cnt = np.diff(np.concatenate((basesT, [int(b0s[r]) + int(np.cumsum(np.where(np.arange(128) <= t, hist[bucket0:bucket0 + 128], 0))[-1])])))
            exp_cnt = np.array([np.count_nonzero(seg[k * 8192:(k + 1) * 8192] == t) for k in range(items)])
            d = np.nonzero(gpu_cnt != exp_cnt)[0]
            if len(d):
                problems.append("bin %d lo %x: %d of %d items have a wrong count: %s" % (r, t, len(d), items,
                                [(int(k), int(gpu_cnt[k]), int(exp_cnt[k])) for k in d[:10]]))
        if not (hist[bucket0:bucket0 + 128] == tot).all():
            t = int(np.nonzero(hist[bucket0:bucket0 + 128] != tot)[0][0])
            problems.append("bin %d: hist[%x] = %d, coarse count %d" % (r, bucket0 + t, hist[bucket0 + t], tot[t]))
    if not (cover == 1).all(): problems.append("item ranges overlap or leave gaps: %s" % np.nonzero(cover != 1)[0][:10])
    print("attempt", attempt, "n_big", n_big, "n_items", n_items, "problems", len(problems), flush=True)
    for pmsg in problems[:12]: print("   ", pmsg)
    if problems: break
