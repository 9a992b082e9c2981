"""Tile-sharded multi-GPU rendering: one process per GPU, RCCL over xGMI through torch.distributed.

Pixels are independent inside an iteration (the sampling SD-tree is frozen; the building tree only
accumulates — guided_path.cpp:610-613, 308, 332, 397), so every rank renders the 32x32 tiles t with
t % world == rank for all passes and the only exchange per iteration is
    one all_reduce(SUM) of the building tree's fixed-point leaf sums + the per-D-tree statistical weights (uint64 as int64: exact, order independent)
    one all_reduce(SUM) of image / squared image / weights of the iteration (disjoint supports → exact)
plus, when the BSDF sampling fraction is learned, per ROUND of the optimiser (include/ppg.h "Sharded optimiser") one all-to-all of
its records — every D-tree has ONE owner, rank = S-tree node / ceil(nodes / world), which sorts and applies that D-tree's records of
all ranks — and one all-gather of the 24-byte optimiser state per S-tree node (`gather_all=True` restores round 2's scheme: every
rank gathers and applies everything); after which refine/reset/build are deterministic functions of identical data on every rank: the SD-tree
topology stays bit-identical across ranks and equal to a single-GPU render.  torch is plumbing here
(device-pointer views + the collective); all compute is in libppg_hip.so.

`TorchReducer` works on device pointers (HIP engine, backend nccl = RCCL).
`HostReducer` is the same exchange over host arrays (gloo) and exists for the CPU tests that drive the
oracle through this exact control flow.
"""
import ctypes as C

import numpy as np


class _DevArray:
    """Exposes a raw device pointer to torch through __cuda_array_interface__ (zero copy)."""

    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": typestr, "data": (int(ptr), False), "version": 3}


def _view(torch, ptr, n, typestr, device):
    return torch.as_tensor(_DevArray(ptr, n, typestr), device=device)


class RenderAborted(RuntimeError):
    """Some rank reported a failure or a cancellation in the status word of an exchange: every rank leaves the render at the same point."""


class _Status:
    """Every exchange carries the ranks' status words: a rank that was cancelled or failed sets `status = 1` and still enters the exchange
    the others are in; the word travels WITH the data (one more element of a fused all-reduce, a column of the count exchange, or a
    one-element all-reduce enqueued right behind a large in-place one — the protocol of host/rccl_reducer.h), all ranks see the same sum,
    raise RenderAborted when it is not zero, and none is left waiting in a collective.  One host synchronisation per exchange."""
    status = 0

    def begin_render(self):
        self.status = 0

    def _small(self, values, dtype):
        """A few numbers for a collective, where the collective wants them (device memory for RCCL): through a cached pinned staging buffer —
        torch.tensor(list, device=cuda) is a pageable, synchronous copy of ~0.3 ms, paid two or three times per round of the optimiser."""
        t = self.torch.tensor(values, dtype=dtype)
        kw = self._tensor_kw()
        if not kw:
            return t
        cache = self.__dict__.setdefault("_smalls", {})
        key = (dtype, len(values))
        if key not in cache:
            cache[key] = (self.torch.empty(len(values), dtype=dtype).pin_memory(), self.torch.empty(len(values), dtype=dtype, **kw))
        pin, dev = cache[key]
        pin.copy_(t)
        dev.copy_(pin, non_blocking=True)
        return dev

    def _abort_if(self, total):
        if total != 0:
            raise RenderAborted("render aborted: a rank reported a failure or a cancellation")

    # ---- the collectives themselves: torch.distributed on the tensors as they are (RCCL on device tensors, gloo on host tensors);
    # `StagedReducer` overrides these four to carry DEVICE tensors through host memory over gloo --------------------------------------
    def _all_reduce(self, t):
        self.dist.all_reduce(t)

    def _all_gather_into(self, out, inp):
        self.dist.all_gather_into_tensor(out, inp)

    def _all_to_all(self, out, inp, recv_splits, send_splits):
        self.dist.all_to_all_single(out, inp, recv_splits, send_splits)

    def _broadcast(self, t):
        self.dist.broadcast(t, src=0)

    def broadcast(self, value):
        """rank 0's `value` (a float) on every rank: the clock readings of a sharded budgetType = seconds render"""
        t = self._small([float(value)], self.torch.float64)
        self._broadcast(t)
        return float(t.item())

    def stop_decision(self, local_stop):
        """The stop decision of a time budget (include/ppg.h ppg_set_stop_hook): rank 0's, or "stop" as soon as any rank's status word is
        set — ONE all-reduce of (decision, status).  A cancelled rank meets the others here; they all leave the batch loop for the image
        exchange, where the status aborts the render."""
        t = self._small([int(bool(local_stop)) if self.dist.get_rank() == 0 else 0, int(self.status)], self.torch.int64)
        self._all_reduce(t)
        stop, bad = (int(v) for v in t.tolist())
        return 1 if (stop or bad) else 0

    @property
    def world(self):
        return self.dist.get_world_size()


class TorchReducer(_Status):
    """The exchanges of a sharded render over torch.distributed.  ONE class holds the protocol — what is exchanged when, the status word, the two
    phases of the round hook, the final iteration's groups, the stop decision; where the arrays live is four small accessors: device memory of
    the HIP engine here (backend nccl = RCCL), host memory of the oracle in `HostReducer` (gloo), so that the world-2 / 3 / 4 / 8 CPU tests run the
    protocol code `bench.py --gpus N` runs."""

    def __init__(self, dist, device, gather_all=False):
        import torch
        self.torch, self.dist, self.device, self.gather_all = torch, dist, device, gather_all

    def _tensor_kw(self):
        return dict(device=self.device)

    # ---- where the arrays live -------------------------------------------------------------------------------------------------------
    def _sync(self):
        self.torch.cuda.synchronize()

    def _view(self, ptr, n, typestr):
        return _view(self.torch, ptr, n, typestr, self.device)

    def _sdtree_views(self, e):
        """([views of the building tree's sums and weights as int64], what to call once they are reduced)"""
        (ps, ns), (pw, nw) = e.stat_buffers()
        return [self._view(ptr, n, "<i8") for ptr, n in ((ps, ns), (pw, nw)) if n], None

    def _image_views(self, e):
        n = e.width * e.height
        a, b = e.image_buffers()
        w = e.image_weight_buffer()
        return [self._view(a, 3 * n, "<f4"), self._view(b, 3 * n, "<f4"), self._view(w, n, "<f4")]

    def _film_views(self, e):
        n = e.width * e.height
        a, w = e.film_buffers()
        return [self._view(a, 3 * n, "<f4"), self._view(w, n, "<f4")]

    # ---- the protocol ------------------------------------------------------------------------------------------------------------------
    def _exchange(self, views):
        """ONE exchange: the arrays all-reduced (sum) together with the status word.  Several arrays are separate allocations of the context,
        so they are packed into a staging tensor with the word as its last element, reduced in ONE collective and copied back — the xGMI
        ring is latency bound at these sizes (a few MB to ~100 MB), fewer and larger calls win.  A single array is reduced in place (no
        staging copy of what can be gigabytes: a final iteration's slots) and the word follows in a one-element collective enqueued right
        behind it.  One host synchronisation, at the end."""
        torch = self.torch
        views = [v for v in views if v.numel()]
        dtype = views[0].dtype if views else torch.int64
        st = self._small([self.status], dtype)
        if len(views) <= 1:
            for v in views:
                self._all_reduce(v)
            self._all_reduce(st)
            bad = st
        else:
            flat = torch.cat(views + [st])
            self._all_reduce(flat)
            off = 0
            for v in views:
                v.copy_(flat[off:off + v.numel()])
                off += v.numel()
            bad = flat[-1:]
        self._abort_if(float(bad.item()))  # (.item() is the exchange's one host synchronisation)

    def reduce_sdtree(self, e):
        views, done = self._sdtree_views(e)
        self._exchange(views)
        if done:
            done()

    def reduce_images(self, e):
        n = e.width * e.height
        try:
            views = self._image_views(e)
        except Exception:  # this rank cannot produce its buffers: zeros of the same sizes and its status word
            self.status = 1
            views = [self.torch.zeros(k, dtype=self.torch.float32, **self._tensor_kw()) for k in (3 * n, 3 * n, n)]
        self._exchange(views)

    def reduce_final_partials(self, e, ptr, count):
        """A final iteration's groups of passes (include/ppg.h "Final iteration: groups of passes"): every rank rendered its groups — whole
        ones over the whole film, or all of them on its tiles —; ONE all-reduce of the film head + all group slots (each pixel of a slot is
        non-zero on one rank: exact), in place, then the library adds the slots in group order.  ptr = None: this rank has nothing to give
        (it failed or was cancelled) and joins with zeros and its status word."""
        buf = self._view(ptr, count, "<f4") if ptr else self.torch.zeros(count, dtype=self.torch.float32, **self._tensor_kw())
        self._exchange([buf])
        e.final_partials_commit()

    def reduce_adam(self, e):
        """Round hook of the sampling-fraction optimiser (called twice per round, include/ppg.h "Sharded optimiser").
        Phase 0: this rank's records, in key order, are split by OWNER of their D-tree and exchanged with ONE all-to-all (32-byte records
        as int64 quadruples): a rank receives — and then sorts and applies — only the records of the D-trees it owns, 1 / world of the
        total instead of all of them.  The counts travel first, in one all-gather of (world + 1) numbers per rank: what it sends to every
        owner, and its status word.  Phase 1: the owners' results, 24 bytes of optimiser state per S-tree node, are all-gathered in place.
        Same records in the same key order at the owner ⇒ the fractions stay bit-identical to a single-GPU render."""
        torch, dist = self.torch, self.dist
        world, rank = dist.get_world_size(), dist.get_rank()
        if self.gather_all:
            return self._reduce_adam_gather_all(e)
        if e.hook_phase() == 0:
            ptr, send = None, [0] * world
            if not self.status:
                try:
                    ptr, send = e.adam_records_by_owner(world)
                except Exception:  # cannot take part: still joins the count exchange, with no records and its status word set
                    self.status = 1
            if self.status:
                send = [0] * world
            row = self._small(send + [int(self.status)], torch.int64)
            rows = torch.empty(world * (world + 1), dtype=torch.int64, **self._tensor_kw())
            self._all_gather_into(rows, row)
            table = rows.view(world, world + 1).tolist()             # (the host sizes the messages by the counts: the one synchronisation)
            self._abort_if(sum(r[world] for r in table))
            recv = [int(table[r][rank]) for r in range(world)]
            n_send, n_recv = sum(send), sum(recv)
            src = self._view(ptr, 4 * n_send, "<i8") if n_send else torch.empty(0, dtype=torch.int64, **self._tensor_kw())
            self._adam_recv = torch.empty(4 * max(n_recv, 1), dtype=torch.int64, **self._tensor_kw())
            self._all_to_all(self._adam_recv[:4 * n_recv], src.contiguous(), [4 * c for c in recv], [4 * c for c in send])
            self._sync()
            e.adam_records_replace(self._adam_recv.data_ptr() if n_recv else 0, n_recv)
        else:
            ptr, seg = e.adam_state(world)
            state = self._view(ptr, 3 * seg * world, "<i8")      # 24 bytes per node = three int64
            mine = state[3 * seg * rank:3 * seg * (rank + 1)].clone()
            self._all_gather_into(state, mine)
            self._sync()
            e.adam_state_commit()

    def _reduce_adam_gather_all(self, e):
        """Round 2's scheme, kept for comparison: every rank gathers the records of ALL ranks and applies the union."""
        torch, dist = self.torch, self.dist
        ptr, n = (None, 0) if self.status else e.adam_records()
        world = dist.get_world_size()
        counts = torch.zeros(world + 1, dtype=torch.int64, **self._tensor_kw())
        counts[dist.get_rank()] = n
        counts[world] = int(self.status)
        self._all_reduce(counts)
        counts = [int(c) for c in counts.tolist()]
        self._abort_if(counts.pop())
        most = max(counts)
        if most == 0:
            return
        mine = torch.zeros(4 * most, dtype=torch.int64, **self._tensor_kw())
        if n:
            mine[:4 * n] = self._view(ptr, 4 * n, "<i8")
        gathered = torch.empty(world * mine.numel(), dtype=mine.dtype, **self._tensor_kw())
        self._all_gather_into(gathered, mine)
        parts = gathered.view(world, -1)
        self._union = torch.cat([p[:4 * c] for p, c in zip(parts, counts)]).contiguous()
        self._sync()
        e.adam_records_replace(self._union.data_ptr(), sum(counts))

    def reduce_film(self, e, inverse_variance=False):
        if inverse_variance:
            return  # the retained iteration images were already reduced by reduce_images
        self._exchange(self._film_views(e))


class StagedReducer(TorchReducer):
    """The HIP engine under the reducer WITHOUT RCCL: the arrays stay in device memory (the accessors of TorchReducer), every collective is
    carried through host memory over a gloo process group — copy out, exchange, copy back.  Two ranks can then share ONE GPU
    (tests/test_two_ranks_one_gpu.py): the stream ordering between the library's kernels and the exchanges — the round hook on the context's
    stream, the splats beside it, the stragglers' side stream — is exercised by real, concurrent processes, which neither the in-process
    two-context tests nor the oracle-over-gloo tests can.  Not a production path: RCCL over xGMI is (TorchReducer with backend nccl)."""

    def _all_reduce(self, t):
        h = t.cpu()  # (synchronises with the device work that produced t)
        self.dist.all_reduce(h)
        t.copy_(h)

    def _all_gather_into(self, out, inp):
        ho = self.torch.empty(out.shape, dtype=out.dtype)
        self.dist.all_gather_into_tensor(ho, inp.cpu())
        out.copy_(ho)

    def _all_to_all(self, out, inp, recv_splits, send_splits):
        ho = self.torch.empty(out.shape, dtype=out.dtype)
        self.dist.all_to_all_single(ho, inp.cpu(), recv_splits, send_splits)
        out.copy_(ho)

    def _broadcast(self, t):
        h = t.cpu()
        self.dist.broadcast(h, src=0)
        t.copy_(h)


class HostReducer(TorchReducer):
    """The same protocol for an oracle engine: host memory, gloo.  Only the accessors differ — the oracle's statistics are per-node objects
    (exported into arrays, reduced, imported), its images and film plain host arrays."""

    def __init__(self, dist, gather_all=False):
        import torch
        super().__init__(dist, torch.device("cpu"), gather_all)

    def _tensor_kw(self):
        return {}

    def _sync(self):
        pass

    def _view(self, ptr, n, typestr):
        ct = {"<i8": C.c_int64, "<f4": C.c_float}[typestr]
        return self.torch.from_numpy(np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ct)), shape=(int(n),)))  # (shares the memory)

    def _sdtree_views(self, e):
        ns, nw = C.c_uint64(), C.c_uint64()
        e._call("stat_sizes", C.byref(ns), C.byref(nw))
        sums, wts = np.zeros(ns.value, np.int64), np.zeros(nw.value, np.int64)
        u64 = C.POINTER(C.c_uint64)
        e._call("stat_export", sums.ctypes.data_as(u64), ns, wts.ctypes.data_as(u64), nw)
        return [self.torch.from_numpy(sums), self.torch.from_numpy(wts)], lambda: e._call("stat_import", sums.ctypes.data_as(u64), ns, wts.ctypes.data_as(u64), nw)

    def _two(self, e, fn):
        a, b = C.POINTER(C.c_float)(), C.POINTER(C.c_float)()
        e._call(fn, C.byref(a), C.byref(b))
        return C.cast(a, C.c_void_p).value, C.cast(b, C.c_void_p).value

    def _image_views(self, e):
        n = e.width * e.height
        a, b = self._two(e, "image_ptrs")
        w = C.POINTER(C.c_float)()
        e._call("image_weight_ptr", C.byref(w))
        return [self._view(a, 3 * n, "<f4"), self._view(b, 3 * n, "<f4"), self._view(C.cast(w, C.c_void_p).value, n, "<f4")]

    def _film_views(self, e):
        n = e.width * e.height
        a, w = self._two(e, "film_ptrs")
        return [self._view(a, 3 * n, "<f4"), self._view(w, n, "<f4")]
