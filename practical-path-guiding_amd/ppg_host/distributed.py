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

    def _abort_if(self, total):
        if total != 0:
            raise RenderAborted("render aborted: a rank reported a failure or a cancellation")

    def broadcast(self, value):
        """rank 0's `value` (a float) on every rank: the clock readings of a sharded budgetType = seconds render"""
        t = self.torch.tensor([float(value)], dtype=self.torch.float64, **self._tensor_kw())
        self.dist.broadcast(t, src=0)
        return float(t.item())

    def stop_decision(self, local_stop):
        """The stop decision of a time budget (include/ppg.h ppg_set_stop_hook): rank 0's, or "stop" as soon as any rank's status word is
        set — ONE all-reduce of (decision, status).  A cancelled rank meets the others here; they all leave the batch loop for the image
        exchange, where the status aborts the render."""
        t = self.torch.tensor([int(bool(local_stop)) if self.dist.get_rank() == 0 else 0, int(self.status)], dtype=self.torch.int64, **self._tensor_kw())
        self.dist.all_reduce(t)
        stop, bad = (int(v) for v in t.tolist())
        return 1 if (stop or bad) else 0

    @property
    def world(self):
        return self.dist.get_world_size()


class TorchReducer(_Status):
    def __init__(self, dist, device, gather_all=False):
        import torch
        self.torch, self.dist, self.device, self.gather_all = torch, dist, device, gather_all

    def _tensor_kw(self):
        return dict(device=self.device)

    def _exchange(self, views):
        """ONE exchange: the arrays all-reduced (sum) together with the status word.  Several arrays are separate allocations of the context,
        so they are packed into a staging tensor with the word as its last element, reduced in ONE collective and copied back — the xGMI
        ring is latency bound at these sizes (a few MB to ~100 MB), fewer and larger calls win.  A single array is reduced in place (no
        staging copy of what can be gigabytes: a final iteration's slots) and the word follows in a one-element collective enqueued right
        behind it.  One host synchronisation, at the end."""
        torch = self.torch
        views = [v for v in views if v.numel()]
        dtype = views[0].dtype if views else torch.int64
        st = torch.tensor([self.status], dtype=dtype, device=self.device)
        if len(views) <= 1:
            for v in views:
                self.dist.all_reduce(v)
            self.dist.all_reduce(st)
            bad = st
        else:
            flat = torch.cat(views + [st])
            self.dist.all_reduce(flat)
            off = 0
            for v in views:
                v.copy_(flat[off:off + v.numel()])
                off += v.numel()
            bad = flat[-1:]
        self._abort_if(float(bad.item()))  # (.item() is the exchange's one host synchronisation)

    def reduce_sdtree(self, e):
        (ps, ns), (pw, nw) = e.stat_buffers()
        self._exchange([_view(self.torch, ptr, n, "<i8", self.device) for ptr, n in ((ps, ns), (pw, nw)) if n])

    def reduce_images(self, e):
        n = e.width * e.height
        try:
            a, b = e.image_buffers()
            w = e.image_weight_buffer()
            views = [_view(self.torch, a, 3 * n, "<f4", self.device), _view(self.torch, b, 3 * n, "<f4", self.device), _view(self.torch, w, n, "<f4", self.device)]
        except Exception:  # this rank cannot produce its buffers: zeros of the same sizes and its status word
            self.status = 1
            views = [self.torch.zeros(k, dtype=self.torch.float32, device=self.device) for k in (3 * n, 3 * n, n)]
        self._exchange(views)

    def reduce_final_partials(self, e, ptr, count):
        """A final iteration's groups of passes (include/ppg.h "Final iteration: groups of passes"): every rank rendered its groups — whole
        ones over the whole film, or all of them on its tiles —; ONE all-reduce of the film head + all group slots (each pixel of a slot is
        non-zero on one rank: exact), in place, then the library adds the slots in group order.  ptr = None: this rank has nothing to give
        (it failed or was cancelled) and joins with zeros and its status word."""
        buf = _view(self.torch, ptr, count, "<f4", self.device) if ptr else self.torch.zeros(count, dtype=self.torch.float32, device=self.device)
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
            row = torch.tensor(send + [int(self.status)], dtype=torch.int64, device=self.device)
            rows = torch.empty(world * (world + 1), dtype=torch.int64, device=self.device)
            dist.all_gather_into_tensor(rows, row)
            table = rows.view(world, world + 1).tolist()             # (the host sizes the messages by the counts: the one synchronisation)
            self._abort_if(sum(r[world] for r in table))
            recv = [int(table[r][rank]) for r in range(world)]
            n_send, n_recv = sum(send), sum(recv)
            src = _view(torch, ptr, 4 * n_send, "<i8", self.device) if n_send else torch.empty(0, dtype=torch.int64, device=self.device)
            self._adam_recv = torch.empty(4 * max(n_recv, 1), dtype=torch.int64, device=self.device)
            dist.all_to_all_single(self._adam_recv[:4 * n_recv], src, [4 * c for c in recv], [4 * c for c in send])
            torch.cuda.synchronize()
            e.adam_records_replace(self._adam_recv.data_ptr(), n_recv)
        else:
            ptr, seg = e.adam_state(world)
            state = _view(torch, ptr, 3 * seg * world, "<i8", self.device)      # 24 bytes per node = three int64
            mine = state[3 * seg * rank:3 * seg * (rank + 1)].clone()
            dist.all_gather_into_tensor(state, mine)
            torch.cuda.synchronize()
            e.adam_state_commit()

    def _reduce_adam_gather_all(self, e):
        """Round 2's scheme, kept for comparison: every rank gathers the records of ALL ranks and applies the union."""
        torch, dist = self.torch, self.dist
        ptr, n = (None, 0) if self.status else e.adam_records()
        world = dist.get_world_size()
        counts = torch.zeros(world + 1, dtype=torch.int64, device=self.device)
        counts[dist.get_rank()] = n
        counts[world] = int(self.status)
        dist.all_reduce(counts)
        counts = [int(c) for c in counts.tolist()]
        self._abort_if(counts.pop())
        most = max(counts)
        if most == 0:
            return
        mine = torch.zeros(4 * most, dtype=torch.int64, device=self.device)
        if n:
            mine[:4 * n] = _view(torch, ptr, 4 * n, "<i8", self.device)
        parts = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine)
        union = torch.cat([p[:4 * c] for p, c in zip(parts, counts)])
        torch.cuda.synchronize()
        e.adam_records_replace(union.data_ptr(), sum(counts))

    def reduce_film(self, e, inverse_variance=False):
        if inverse_variance:
            return  # the retained iteration images were already reduced by reduce_images
        n = e.width * e.height
        a, w = e.film_buffers()
        self._exchange([_view(self.torch, a, 3 * n, "<f4", self.device), _view(self.torch, w, n, "<f4", self.device)])


class HostReducer(_Status):
    """Same exchange for an oracle engine (host memory, gloo)."""

    def __init__(self, dist, gather_all=False):
        import torch
        self.torch, self.dist, self.gather_all = torch, dist, gather_all

    def _tensor_kw(self):
        return {}

    def _exchange(self, arrays):
        """the arrays (numpy, reduced in place) and the status word in ONE all-reduce"""
        arrays = [a for a in arrays if a.size]
        dtype = arrays[0].dtype if arrays else np.dtype(np.int64)
        flat = np.concatenate([a.reshape(-1) for a in arrays] + [np.array([self.status], dtype)])
        self.dist.all_reduce(self.torch.from_numpy(flat))
        off = 0
        for a in arrays:
            a.reshape(-1)[:] = flat[off:off + a.size]
            off += a.size
        self._abort_if(float(flat[-1]))

    def reduce_sdtree(self, e):
        ns, nw = C.c_uint64(), C.c_uint64()
        e._call("stat_sizes", C.byref(ns), C.byref(nw))
        sums = np.zeros(ns.value, np.int64)
        wts = np.zeros(nw.value, np.int64)
        u64 = C.POINTER(C.c_uint64)
        e._call("stat_export", sums.ctypes.data_as(u64), ns, wts.ctypes.data_as(u64), nw)
        self._exchange([sums, wts])
        e._call("stat_import", sums.ctypes.data_as(u64), ns, wts.ctypes.data_as(u64), nw)

    def _ptr_arrays(self, e, fn, sizes):
        a, b = C.POINTER(C.c_float)(), C.POINTER(C.c_float)()
        e._call(fn, C.byref(a), C.byref(b))
        return [np.ctypeslib.as_array(p, shape=(s,)) for p, s in zip((a, b), sizes)]

    def reduce_images(self, e):
        n = e.width * e.height
        img, sq = self._ptr_arrays(e, "image_ptrs", (3 * n, 3 * n))
        w = C.POINTER(C.c_float)()
        e._call("image_weight_ptr", C.byref(w))
        self._exchange([img, sq, np.ctypeslib.as_array(w, shape=(n,))])

    def reduce_final_partials(self, e, ptr, count):
        arr = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_float)), shape=(count,)) if ptr else np.zeros(count, np.float32)
        self._exchange([arr])
        e.final_partials_commit()

    def reduce_adam(self, e):
        """The same two-phase exchange as TorchReducer.reduce_adam, on host arrays."""
        torch, dist = self.torch, self.dist
        world, rank = dist.get_world_size(), dist.get_rank()
        if self.gather_all:
            return self._reduce_adam_gather_all(e)
        if e.hook_phase() == 0:
            ptr, send = None, [0] * world
            if not self.status:
                try:
                    ptr, send = e.adam_records_by_owner(world)
                except Exception:
                    self.status = 1
            if self.status:
                send = [0] * world
            rows = [torch.empty(world + 1, dtype=torch.int64) for _ in range(world)]
            dist.all_gather(rows, torch.tensor(send + [int(self.status)], dtype=torch.int64))
            table = [r.tolist() for r in rows]
            self._abort_if(sum(r[world] for r in table))
            recv = [int(table[r][rank]) for r in range(world)]
            n_send, n_recv = sum(send), sum(recv)
            src = torch.from_numpy(np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_int64)), shape=(4 * n_send,)).copy()) if n_send else torch.empty(0, dtype=torch.int64)
            got = torch.empty(4 * n_recv, dtype=torch.int64)
            dist.all_to_all_single(got, src, [4 * c for c in recv], [4 * c for c in send])
            self._adam_recv = np.ascontiguousarray(got.numpy())
            e.adam_records_replace(self._adam_recv.ctypes.data if n_recv else 0, n_recv)
        else:
            ptr, seg = e.adam_state(world)
            state = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_int64)), shape=(3 * seg * world,))
            parts = [torch.empty(3 * seg, dtype=torch.int64) for _ in range(world)]
            dist.all_gather(parts, torch.from_numpy(state[3 * seg * rank:3 * seg * (rank + 1)].copy()))
            for r, part in enumerate(parts):
                state[3 * seg * r:3 * seg * (r + 1)] = part.numpy()
            e.adam_state_commit()

    def _reduce_adam_gather_all(self, e):
        torch, dist = self.torch, self.dist
        ptr, n = (None, 0) if self.status else e.adam_records()
        world = dist.get_world_size()
        counts = torch.zeros(world + 1, dtype=torch.int64)
        counts[dist.get_rank()] = n
        counts[world] = int(self.status)
        dist.all_reduce(counts)
        counts = [int(c) for c in counts.tolist()]
        self._abort_if(counts.pop())
        most = max(counts)
        if most == 0:
            return
        mine = np.zeros(4 * most, np.int64)
        if n:
            mine[:4 * n] = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_int64)), shape=(4 * n,))
        parts = [torch.empty(4 * most, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(parts, torch.from_numpy(mine))
        union = np.ascontiguousarray(np.concatenate([p.numpy()[:4 * c] for p, c in zip(parts, counts)]))
        e.adam_records_replace(union.ctypes.data, sum(counts))

    def reduce_film(self, e, inverse_variance=False):
        if inverse_variance:
            return
        n = e.width * e.height
        film, w = self._ptr_arrays(e, "film_ptrs", (3 * n, n))
        self._exchange([film, w])
