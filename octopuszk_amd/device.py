"""Device-resident entry points on torch CUDA(HIP) tensors: torch is plumbing only (HBM
buffers, streams, torch.distributed); all arithmetic is in libozk_hip.so."""
import ctypes
import os
import sys

import torch

from . import lib as _lib


def _ptr(t):
    return int(t.data_ptr())


def _stream():
    return int(torch.cuda.current_stream().cuda_stream)


class VarMsmWorkspace:
    """Pre-allocated workspace + output for repeated device-resident MSMs of size n.

    The library receives raw device pointers, so — as with any stream-ordered foreign library —
    this object (its workspace) and the input tensors must stay alive until the stream the work
    was enqueued on has been synchronised; `run` keeps references to its last inputs to help."""

    def __init__(self, n, type_=1, device="cuda"):
        L = _lib.load()
        self.n, self.type = n, type_
        self.bytes = int(L.ozk_var_msm_workspace_bytes(n, type_))
        if self.bytes == 0:
            raise _lib.OzkError("workspace size query failed")
        self.ws = torch.empty(self.bytes, dtype=torch.uint8, device=device)
        self.out = torch.zeros(192 if type_ == 1 else 384, dtype=torch.uint8, device=device)

    def run(self, d_bases, d_scalars, prepared=False):
        """d_bases: uint8 [n*96|192] wire format (or the records of ozk_var_msm_prepare_dev with
        prepared=True), d_scalars: uint8 [n*32], in HBM.
        Asynchronous on the current stream; returns the output tensor (192|384 B)."""
        L = _lib.load()
        self._inputs = (d_bases, d_scalars)
        fn = L.ozk_var_msm_prepared_dev if prepared else L.ozk_var_msm_dev
        _lib.check(fn(_ptr(d_bases), _ptr(d_scalars), self.n, self.type, _ptr(self.out), _ptr(self.ws), self.bytes,
                      _stream()))
        return self.out


class VarMsmPipeline:
    """Several device-resident MSMs in flight: heads (throughput-bound) run back to back on the
    caller's stream and share ONE workspace; each tail (latency-bound: upper window-sum levels,
    Horner, normalisation) runs on a side stream out of its own small tail buffer, overlapping the
    next head.  This is how a prover issues its six independent MSMs.

        pipe = VarMsmPipeline(n, depth=2)
        t = pipe.submit(d_bases, d_scalars)      # returns a ticket
        out = pipe.result(t)                     # makes the current stream wait for that tail
    """

    def __init__(self, n, type_=1, depth=2, device="cuda"):
        L = _lib.load()
        self.n, self.type, self.depth = n, type_, depth
        self.ws_bytes = int(L.ozk_var_msm_head_workspace_bytes(n, type_))
        self.tail_bytes = int(L.ozk_var_msm_tail_bytes(n, type_))
        self.ws = torch.empty(self.ws_bytes, dtype=torch.uint8, device=device)
        self.tails = [torch.empty(self.tail_bytes, dtype=torch.uint8, device=device) for _ in range(depth)]
        self.outs = [torch.zeros(192 if type_ == 1 else 384, dtype=torch.uint8, device=device) for _ in range(depth)]
        self.side = torch.cuda.Stream(device=device)
        self.head_done = [torch.cuda.Event() for _ in range(depth)]
        self.tail_done = [torch.cuda.Event() for _ in range(depth)]
        # ordering hint (include/ozk.h): the next head's bucket accumulation is dispatched after the
        # previous tail's multi-wave levels, so that its single-wave Horner kernel is resident first
        self.levels_done = []
        for _ in range(depth):
            ev = ctypes.c_void_p()
            _lib.check(L.ozk_order_event_create(ctypes.byref(ev)))
            self.levels_done.append(ev)
        self.count = 0

    def close(self):
        """Destroy the ordering events (after the work that uses them has drained)."""
        if self.levels_done:
            L = _lib.load()
            torch.cuda.synchronize()
            for ev in self.levels_done:
                L.ozk_order_event_destroy(ev)
            self.levels_done = []

    def __del__(self):
        # never call into HIP while the interpreter (and possibly the HIP runtime) is being torn down
        # (`sys` is a module-level import: an import statement here fails during interpreter shutdown)
        if sys is None or sys.is_finalizing():
            return
        try:
            self.close()
        except Exception:
            pass

    def prepare(self, d_bases):
        """Affine Montgomery records of `d_bases` (wire format) for submit(..., prepared=True): done once
        per proving key, skips the conversion kernel in every MSM."""
        L = _lib.load()
        nbytes = int(L.ozk_var_msm_prepared_bytes(self.n, self.type))
        out = torch.empty(nbytes, dtype=torch.uint8, device=d_bases.device)
        _lib.check(L.ozk_var_msm_prepare_dev(_ptr(d_bases), self.n, self.type, _ptr(out), nbytes, _stream()))
        return out

    def submit(self, d_bases, d_scalars, prepared=False):
        L = _lib.load()
        slot = self.count % self.depth
        main = torch.cuda.current_stream()
        if self.count >= self.depth:
            main.wait_event(self.tail_done[slot])      # the tail that last used this slot's buffers
        prev = self.levels_done[(self.count - 1) % self.depth] if (self.count and self.depth > 1) else None
        head = L.ozk_var_msm_head_prepared_dev if prepared else L.ozk_var_msm_head_ordered_dev
        _lib.check(head(_ptr(d_bases), _ptr(d_scalars), self.n, self.type, _ptr(self.ws), self.ws_bytes,
                        _ptr(self.tails[slot]), self.tail_bytes, int(main.cuda_stream), prev))
        self.head_done[slot].record(main)
        self.side.wait_event(self.head_done[slot])
        _lib.check(L.ozk_var_msm_tail_ordered_dev(self.n, self.type, _ptr(self.tails[slot]), self.tail_bytes,
                                                  _ptr(self.outs[slot]), int(self.side.cuda_stream),
                                                  self.levels_done[slot]))
        self.tail_done[slot].record(self.side)
        self.count += 1
        return self.count - 1

    def result(self, ticket):
        """Output tensor of `ticket` (valid until `depth` more submissions); the current stream
        waits for its tail."""
        assert self.count - ticket <= self.depth, "result buffer already reused"
        slot = ticket % self.depth
        torch.cuda.current_stream().wait_event(self.tail_done[slot])
        return self.outs[slot]


class VarMsmPipeline3:
    """Three-stage schedule of consecutive device-resident MSMs: SORT of MSM k+1 (base conversion, digits, counting
    sort: HBM / LDS-bound) on the caller's stream | ACCUMULATE of MSM k (bucket accumulation, run merge: vector-ALU
    bound) on a second stream | TAIL of MSM k-1 (window sums, Horner, normalisation: latency-bound) on one or two more.
    The sort writes a double-buffered "sorted set"; each stage has private scratch (include/ozk.h,
    ozk_var_msm_sort_dev / _accum_dev / _tail_dev).  The sort kernels are sized to be resident beside three
    accumulation blocks per CU (csrc/msm_var.cuh, k_sort2), which is what lets the multiplier run back to back.
    Same interface as VarMsmPipeline (submit -> ticket, result(ticket))."""

    def __init__(self, n, type_=1, depth=3, tail_streams=2, device="cuda", split_accum=None, tail_cus=None):
        L = _lib.load()
        ts = max(1, tail_streams)
        # tail_cus = N > 0: the tail streams are confined to N compute units and the accumulate stream to the others
        # (ozk_stream_create_cu_range): the latency-bound tail waves no longer sit on the accumulation's SIMDs.  Such
        # streams are BLOCKING streams — they synchronise with the null stream — so submit() refuses to run on it.
        self.tail_cus = int(os.environ.get("OZK_P3_TAIL_CUS", "0")) if tail_cus is None else int(tail_cus)
        self._owned = []
        # split_accum: level 1 alone on the accumulate stream, the rest of the stage (run merge, generic levels) at the
        # head of the tail stream (ozk_var_msm_accum_part_dev); the accumulate scratch is then double-buffered too
        self.split = bool(int(os.environ.get("OZK_P3_SPLIT_ACCUM", "0"))) if split_accum is None else bool(split_accum)
        # a result slot is always served by the same tail stream (slot = k mod depth, stream = k mod ts), so whatever a
        # caller enqueues on stream_of(ticket) after result(ticket) is ordered before the slot's next tail
        self.n, self.type, self.depth = n, type_, (max(2, depth) + ts - 1) // ts * ts
        sb, swb, awb = ctypes.c_size_t(), ctypes.c_size_t(), ctypes.c_size_t()
        _lib.check(L.ozk_var_msm_stage_bytes(n, type_, ctypes.byref(sb), ctypes.byref(swb), ctypes.byref(awb)))
        self.sorted_bytes, self.sort_ws_bytes, self.accum_ws_bytes = sb.value, swb.value, awb.value
        self.tail_bytes = int(L.ozk_var_msm_tail_bytes(n, type_))
        buf = lambda b: torch.empty(b, dtype=torch.uint8, device=device)
        self.sorted = [buf(self.sorted_bytes) for _ in range(2)]
        self.sort_ws, self.accum_ws = buf(self.sort_ws_bytes), buf(self.accum_ws_bytes)
        self.accum_ws2 = [self.accum_ws, buf(self.accum_ws_bytes)] if self.split else None
        self.rest_st = None
        self.tails = [buf(self.tail_bytes) for _ in range(self.depth)]
        self.outs = [torch.zeros(192 if type_ == 1 else 384, dtype=torch.uint8, device=device) for _ in range(self.depth)]
        if self.tail_cus > 0:
            total = int(L.ozk_device_cu_count())
            if not 0 < self.tail_cus < total:
                raise ValueError(f"tail_cus={self.tail_cus}: the device has {total} compute units")

            def confined(first, count):
                h = ctypes.c_void_p()
                _lib.check(L.ozk_stream_create_cu_range(first, count, ctypes.byref(h)))
                self._owned.append(h.value)
                return torch.cuda.ExternalStream(h.value)
            self.acc = confined(self.tail_cus, total - self.tail_cus)
            self.tail_st = [confined(0, self.tail_cus) for _ in range(ts)]
        else:
            self.acc = torch.cuda.Stream(device=device)
            self.tail_st = [torch.cuda.Stream(device=device) for _ in range(ts)]
        self.side = self.tail_st[0]
        ev = lambda k: [torch.cuda.Event() for _ in range(k)]
        self.sort_done, self.accum_done = ev(2), ev(2)
        self.l1_done = ev(2)
        self.tail_done = ev(self.depth)
        self.count = 0
        self._inputs = None

    def close(self):
        if self._owned:
            torch.cuda.synchronize()
            L = _lib.load()
            for h in self._owned:
                L.ozk_stream_destroy(h)
            self._owned = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check_stream(self, main):
        if self.tail_cus > 0 and int(main.cuda_stream) == 0:
            raise RuntimeError("VarMsmPipeline3(tail_cus > 0) on the null stream: the confined streams are blocking "
                               "streams and would serialise against it; submit under torch.cuda.stream(<a stream>)")

    def prepare(self, d_bases):
        L = _lib.load()
        nbytes = int(L.ozk_var_msm_prepared_bytes(self.n, self.type))
        out = torch.empty(nbytes, dtype=torch.uint8, device=d_bases.device)
        _lib.check(L.ozk_var_msm_prepare_dev(_ptr(d_bases), self.n, self.type, _ptr(out), nbytes, _stream()))
        return out

    def _submit_split(self, d_bases, d_scalars, prepared, last):
        """submit() with the accumulate stage in two parts: sorted set s and accumulate scratch s are free again once
        the REST of MSM k has run — on the tail stream, which is where accum_done[s] is recorded."""
        L = _lib.load()
        k = self.count
        s, slot = k % 2, k % self.depth
        main = torch.cuda.current_stream()
        self._check_stream(main)
        self._inputs = (d_bases, d_scalars)
        if k >= 2:
            main.wait_event(self.accum_done[s])
        sort = L.ozk_var_msm_sort_prepared_dev if prepared else L.ozk_var_msm_sort_dev
        _lib.check(sort(_ptr(d_bases), _ptr(d_scalars), self.n, self.type, _ptr(self.sorted[s]), self.sorted_bytes,
                        _ptr(self.sort_ws), self.sort_ws_bytes, int(main.cuda_stream)))
        self.sort_done[s].record(main)
        self.acc.wait_event(self.sort_done[s])   # (ordered after rest(k - 2) through the sort's wait above)
        if k >= self.depth:
            self.acc.wait_event(self.tail_done[slot])
        args = (_ptr(d_bases) if prepared else None, self.n, self.type, _ptr(self.sorted[s]), self.sorted_bytes,
                _ptr(self.accum_ws2[s]), self.accum_ws_bytes, _ptr(self.tails[slot]), self.tail_bytes)
        _lib.check(L.ozk_var_msm_accum_part_dev(*args, int(self.acc.cuda_stream), 1))
        self.l1_done[s].record(self.acc)
        T = self.tail_st[k % len(self.tail_st)]
        R = self.rest_st or T     # (the rest on a stream of its own when the tail streams are confined to a few CUs)
        R.wait_event(self.l1_done[s])
        _lib.check(L.ozk_var_msm_accum_part_dev(*args, int(R.cuda_stream), 2))
        self.accum_done[s].record(R)
        if R is not T:
            T.wait_event(self.accum_done[s])
        if last:
            _lib.check(L.ozk_var_msm_tail_mode_dev(self.n, self.type, _ptr(self.tails[slot]), self.tail_bytes,
                                                   _ptr(self.outs[slot]), int(T.cuda_stream), None, 0))
        else:
            _lib.check(L.ozk_var_msm_tail_dev(self.n, self.type, _ptr(self.tails[slot]), self.tail_bytes,
                                              _ptr(self.outs[slot]), int(T.cuda_stream)))
        self.tail_done[slot].record(T)
        self.count += 1
        return k

    def submit(self, d_bases, d_scalars, prepared=False, last=False):
        """last=True: the caller knows that no MSM follows this one (the end of a burst, a prover's final MSM): its
        tail then runs with the chip to itself and takes the LATENCY shape of the window sums (fused first level +
        wave levels: ~40 dependent additions shorter) instead of the throughput shape the overlapped tails use."""
        if self.split:
            return self._submit_split(d_bases, d_scalars, prepared, last)
        L = _lib.load()
        k = self.count
        s, slot = k % 2, k % self.depth
        main = torch.cuda.current_stream()
        self._check_stream(main)
        self._inputs = (d_bases, d_scalars)
        if k >= 2:
            main.wait_event(self.accum_done[s])        # sorted set s is free again
        sort = L.ozk_var_msm_sort_prepared_dev if prepared else L.ozk_var_msm_sort_dev
        _lib.check(sort(_ptr(d_bases), _ptr(d_scalars), self.n, self.type, _ptr(self.sorted[s]), self.sorted_bytes,
                        _ptr(self.sort_ws), self.sort_ws_bytes, int(main.cuda_stream)))
        self.sort_done[s].record(main)
        self.acc.wait_event(self.sort_done[s])
        if k >= self.depth:
            self.acc.wait_event(self.tail_done[slot])  # the tail that last used this slot's buffers
        if prepared:
            _lib.check(L.ozk_var_msm_accum_prepared_dev(_ptr(d_bases), self.n, self.type, _ptr(self.sorted[s]),
                                                        self.sorted_bytes, _ptr(self.accum_ws), self.accum_ws_bytes,
                                                        _ptr(self.tails[slot]), self.tail_bytes,
                                                        int(self.acc.cuda_stream)))
        else:
            _lib.check(L.ozk_var_msm_accum_dev(self.n, self.type, _ptr(self.sorted[s]), self.sorted_bytes,
                                               _ptr(self.accum_ws), self.accum_ws_bytes, _ptr(self.tails[slot]),
                                               self.tail_bytes, int(self.acc.cuda_stream)))
        self.accum_done[s].record(self.acc)
        T = self.tail_st[k % len(self.tail_st)]
        T.wait_event(self.accum_done[s])
        if last:
            _lib.check(L.ozk_var_msm_tail_mode_dev(self.n, self.type, _ptr(self.tails[slot]), self.tail_bytes,
                                                   _ptr(self.outs[slot]), int(T.cuda_stream), None, 0))
        else:
            _lib.check(L.ozk_var_msm_tail_dev(self.n, self.type, _ptr(self.tails[slot]), self.tail_bytes,
                                              _ptr(self.outs[slot]), int(T.cuda_stream)))
        self.tail_done[slot].record(T)
        self.count += 1
        return k

    def stream_of(self, ticket):
        """the stream ticket's tail ran on: consumers of result(ticket) that must not stall the caller's (sort)
        stream enqueue there"""
        return self.tail_st[ticket % len(self.tail_st)]

    def result(self, ticket):
        assert self.count - ticket <= self.depth, "result buffer already reused"
        slot = ticket % self.depth
        torch.cuda.current_stream().wait_event(self.tail_done[slot])
        return self.outs[slot]


def gen_g1_bases(n, seed, device="cuda"):
    L = _lib.load()
    out = torch.empty(n * 96, dtype=torch.uint8, device=device)
    _lib.check(L.ozk_gen_bases_dev(seed, n, 1, _ptr(out), _stream()))
    return out


def points_sum(d_points, k, type_=1):
    L = _lib.load()
    out = torch.zeros(192 if type_ == 1 else 384, dtype=torch.uint8, device=d_points.device)
    _lib.check(L.ozk_points_sum_dev(_ptr(d_points), k, type_, _ptr(out), _stream()))
    return out


SPLITMIX_MASK = (1 << 64) - 1


def splitmix64(x):
    x = (x + 0x9E3779B97F4A7C15) & SPLITMIX_MASK
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & SPLITMIX_MASK
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & SPLITMIX_MASK
    return x ^ (x >> 31)


def gen_base_logs(n, seed):
    """the k_i of gen_g1_bases, for CPU-side checking"""
    out = []
    for i in range(n):
        k = splitmix64((seed + i) & SPLITMIX_MASK)
        out.append(k if k else 1)
    return out
