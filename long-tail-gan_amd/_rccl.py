"""The collectives ltg_g_step_sharded issues in-stream (include/ltg.h: ltg_comm).

`RcclComm`   RCCL itself, bound with ctypes to the librccl.so torch has loaded: its own communicator (ncclCommInitRank; the unique
             id travels over the torch.distributed group that already exists), and the ADDRESSES of ncclAllReduce / ncclAllGather go
             into ltg_comm -- the library then calls RCCL directly on the step's stream, between its own kernels: no host round
             trip per exchange (a torch.distributed call costs ~30 us of host time, three of them per G step).
`HostComm`   test rigs whose ranks share one GPU (RCCL refuses two ranks on a device): the same two entry points as host
             callbacks that wait for the stream and reduce through the torch.distributed group (gloo).  Same C code path,
             same results; only the transport differs.
Nothing here computes: transports only.
"""
from __future__ import annotations

import ctypes as C
import os

import torch
import torch.distributed as dist

from . import _cabi as cabi


class _UniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]          # NCCL_UNIQUE_ID_BYTES (rccl.h:40)


def _librccl():
    path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    lib = C.CDLL(path if os.path.exists(path) else "librccl.so")
    lib.ncclGetUniqueId.argtypes = [C.POINTER(_UniqueId)]
    lib.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _UniqueId, C.c_int]
    lib.ncclCommDestroy.argtypes = [C.c_void_p]
    lib.ncclCommAbort.argtypes = [C.c_void_p]
    lib.ncclGetErrorString.restype = C.c_char_p
    lib.ncclGetErrorString.argtypes = [C.c_int]
    lib.ncclCommCount.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    lib.ncclCommCuDevice.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    return lib


class RcclComm:
    """A communicator of this process group's ranks created directly on RCCL (one rank per GPU)."""

    def __init__(self, group=None, device=None, warm_counts=(128 * 600, 640)):
        """warm_counts: floats of the step's all-reduce message and of one rank's all-gather block (ltg_pipe.h1pre, rowpart_all / R).
        ncclCommInitRank is a COLLECTIVE: every step before it that can fail on one rank alone (loading librccl, the unique id, its
        broadcast) is followed by an agreement over the existing group, so a rank that failed BEFORE it never leaves the others blocked inside
        it; a rank whose ncclCommInitRank itself returns an error is caught by one more agreement behind it (the others abort their
        communicator instead of entering the warm-up collectives)."""
        self.n_ranks, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.comm = C.c_void_p()
        self.group = group
        err = None
        try:
            self.lib = _librccl()
        except Exception as e:
            err = "librccl: %r" % (e,)
        self._agree(err, device)
        uid = _UniqueId()
        box = [None]
        try:
            if self.rank == 0:
                self._check(self.lib.ncclGetUniqueId(C.byref(uid)), "ncclGetUniqueId")
                box = [C.string_at(C.byref(uid), 128)]    # (the raw 128 bytes: a c_char array reads as a C string)
        except Exception as e:
            err = "ncclGetUniqueId: %r" % (e,)
        self._agree(err, device)
        if self.n_ranks > 1:
            dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        self._agree(None if (box[0] is not None and len(box[0]) == 128) else "unique id not received", device)
        C.memmove(C.byref(uid), box[0], 128)
        if device is not None:
            torch.cuda.set_device(device)
        err = None
        try:
            self._check(self.lib.ncclCommInitRank(C.byref(self.comm), self.n_ranks, uid, self.rank), "ncclCommInitRank")
        except Exception as e:
            err = "ncclCommInitRank: %r" % (e,)
        # (a rank whose init FAILED returns from it; the others have returned too -- the call is collective -- and would otherwise wait in
        # _warm_up's collectives for ever.  A rank that never returns from ncclCommInitRank is beyond this agreement: RCCL's own timeout.)
        try:
            self._agree(err, device)
        except Exception:
            if self.comm:
                self.lib.ncclCommAbort(self.comm)
                self.comm = C.c_void_p()
            raise
        self._warm_counts = (int(warm_counts[0]), int(warm_counts[1]))
        n = C.c_int()
        self._check(self.lib.ncclCommCount(self.comm, C.byref(n)), "ncclCommCount")
        self.count = int(n.value)                       # world size as RCCL reports it
        self.c = cabi.ltg_comm(self.comm, self.n_ranks, self.rank, C.cast(self.lib.ncclAllReduce, C.c_void_p), C.cast(self.lib.ncclAllGather, C.c_void_p))
        self.kind = "rccl-direct"
        self._warm_up()

    def _agree(self, err, device):
        """every rank reports whether it got this far; all raise together if one did not"""
        ok = torch.tensor([0 if err else 1], dtype=torch.int32, device=device if (device is not None and dist.get_backend(self.group) == "nccl") else "cpu")
        if self.n_ranks > 1:
            dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=self.group)
        if int(ok.item()) == 0:
            raise cabi.LtgError("RCCL communicator not created: %s" % (err or "another rank failed before ncclCommInitRank"))

    def _warm_up(self):
        """One all-reduce and one all-gather of the step's message sizes, then a stream sync: RCCL sets its channels up at the first
        collective (that can take seconds on a node), and the first training step has kernels polling device words with a bound."""
        vp, sz = C.c_void_p, C.c_size_t
        self.lib.ncclAllReduce.argtypes = [vp, vp, sz, C.c_int, C.c_int, vp, vp]
        self.lib.ncclAllGather.argtypes = [vp, vp, sz, C.c_int, vp, vp]
        st = torch.cuda.current_stream().cuda_stream
        n_ar, n_ag = self._warm_counts
        a = torch.zeros(n_ar, dtype=torch.float32, device="cuda")
        g = torch.zeros(self.n_ranks * n_ag, dtype=torch.float32, device="cuda")
        for _ in range(2):
            self._check(self.lib.ncclAllReduce(a.data_ptr(), a.data_ptr(), a.numel(), cabi.LTG_NCCL_FLOAT32, cabi.LTG_NCCL_SUM, self.comm, st), "ncclAllReduce")
            self._check(self.lib.ncclAllGather(g.data_ptr() + 4 * n_ag * self.rank, g.data_ptr(), n_ag, cabi.LTG_NCCL_FLOAT32, self.comm, st), "ncclAllGather")
        torch.cuda.current_stream().synchronize()

    def _check(self, rc, what):
        if rc != 0:
            raise cabi.LtgError("%s failed: %s" % (what, self.lib.ncclGetErrorString(rc).decode()))

    def close(self):
        """ncclCommDestroy: blocks on outstanding operations, so it is called explicitly (ShardedTrainer.close) while the process group and
        the HIP runtime are alive -- never from a destructor (interpreter teardown, a one-rank exception path)"""
        if self.comm:
            self.lib.ncclCommDestroy(self.comm)
            self.comm = C.c_void_p()


    def abort(self):
        """ncclCommAbort: does NOT wait for outstanding operations -- the exception path of one rank (ShardedTrainer.__exit__ with an error), so
        that its peers' in-stream collectives fail promptly and do not sit in RCCL's timeout"""
        if self.comm:
            self.lib.ncclCommAbort(self.comm)
            self.comm = C.c_void_p()


class HostComm:
    """ltg_comm whose entry points are host functions over the torch.distributed group (any backend).  `buffers`: the device
    tensors the library will hand in (looked up by address; the functions receive raw pointers)."""

    def __init__(self, group, buffers, ordered=False):
        """ordered: the all-reduce adds the ranks' contributions IN RANK ORDER (all-gather + a left-to-right sum) instead of in the backend's own
        order -- the reference the one-shot transport (OneShotComm, same fixed order) is compared with bit for bit"""
        self.group, self.ordered = group, bool(ordered)
        self.n_ranks, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.bufs = [(t.data_ptr(), t.numel() * t.element_size(), t.view(-1)) for t in buffers]
        self._ar = cabi.ALL_REDUCE_FN(self._all_reduce)              # (kept alive with the object)
        self._ag = cabi.ALL_GATHER_FN(self._all_gather)
        self.c = cabi.ltg_comm(None, self.n_ranks, self.rank, C.cast(self._ar, C.c_void_p), C.cast(self._ag, C.c_void_p))
        self.kind = "host%s (%s)" % ("-ordered" if self.ordered else "", dist.get_backend(group))
        self.count = self.n_ranks

    def _view(self, ptr, count):
        """the registered device buffer that holds [ptr, ptr + 4 count) as a flat float32 view"""
        for base, nbytes, t in self.bufs:
            if base <= ptr and ptr + 4 * count <= base + nbytes:
                off = (ptr - base) // 4
                return t[off:off + count]
        raise KeyError("pointer %x is not inside a registered exchange buffer" % ptr)

    def _all_reduce(self, send, recv, count, dtype, op, comm, stream):
        try:
            if dtype != cabi.LTG_NCCL_FLOAT32 or op != cabi.LTG_NCCL_SUM or send != recv:
                return 1
            t = self._view(recv, count)
            if self.ordered:
                # every rank's vector, one all-reduce of the zero-padded [R, count] buffer (x + 0 exactly), then acc = ((0 + x_0) + x_1) + ...
                import torch
                allv = torch.zeros(self.n_ranks, count, dtype=t.dtype, device=t.device)
                allv[self.rank] = t
                dist.all_reduce(allv, op=dist.ReduceOp.SUM, group=self.group)
                acc = torch.zeros_like(t)
                for q in range(self.n_ranks):
                    acc += allv[q]
                t.copy_(acc)
                return 0
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
            return 0
        except Exception:                                             # an exception must not unwind through the C frame
            import traceback
            traceback.print_exc()
            return 2

    def _all_gather(self, send, recv, sendcount, dtype, comm, stream):
        try:
            if dtype != cabi.LTG_NCCL_FLOAT32:
                return 1
            src, out = self._view(send, sendcount).clone(), self._view(recv, self.n_ranks * sendcount)    # (in place: the source is a block of `out`)
            if dist.get_backend(self.group) == "nccl":
                dist.all_gather_into_tensor(out, src, group=self.group)
            else:
                # (gloo's list all-gather of device tensors takes milliseconds; an all-reduce of the zero-padded buffer gathers the same
                # values exactly -- x + 0 -- in a fraction of that; test rigs only)
                out.zero_()
                out[self.rank * sendcount:(self.rank + 1) * sendcount] = src
                dist.all_reduce(out, op=dist.ReduceOp.SUM, group=self.group)
            return 0
        except Exception:
            import traceback
            traceback.print_exc()
            return 2

    def close(self):
        pass

    def abort(self):
        pass


class OneShotComm:
    """ltg_comm over the library's own one-shot exchange (csrc/ltg_oneshot.h; include/ltg.h: ltg_oneshot): every rank's staging buffer
    mapped into every other rank's address space through HIP IPC, one kernel per exchange, contributions added in rank order.  A second
    transport for A/B on boxes with more than one GPU and for test rigs; correctness only -- RCCL (RcclComm) is the default."""

    def __init__(self, group, device, max_floats, limit_ms=0):
        import torch
        from ._hip import IpcBuffer
        self.group = group
        self.n_ranks, self.rank = dist.get_world_size(group), dist.get_rank(group)
        if not 1 <= self.n_ranks <= cabi.LTG_ONESHOT_MAX_RANKS:
            raise ValueError("one-shot exchange: 1..%d ranks" % cabi.LTG_ONESHOT_MAX_RANKS)
        lib = cabi.load()
        self.lib = lib
        torch.cuda.set_device(device)
        self.max_floats = int(max_floats)
        self.buf, err = None, None
        try:
            self.buf = IpcBuffer(lib.ltg_oneshot_stage_bytes(self.n_ranks, self.max_floats))
        except Exception as e:      # (every rank must get past the handle exchange: agree on failure below)
            err = repr(e)
        handles = [None] * self.n_ranks
        dist.all_gather_object(handles, None if self.buf is None else self.buf.handle, group=group)
        if any(h is None for h in handles):
            if self.buf is not None:
                self.buf.close()
            raise RuntimeError("one-shot exchange: a rank could not create its staging buffer (%s)" % err)
        self.os = cabi.ltg_oneshot(self.n_ranks, self.rank, 0, int(limit_ms), self.max_floats)
        ok = True
        try:
            for q in range(self.n_ranks):
                self.os.stage[q] = self.buf.ptr.value if q == self.rank else self.buf.open_peer(handles[q])
        except Exception as e:
            ok, err = False, repr(e)
        flags = [None] * self.n_ranks
        dist.all_gather_object(flags, ok, group=group)      # (also the barrier behind which every stage is zeroed and mapped)
        if not all(flags):
            self.buf.close()
            raise RuntimeError("one-shot exchange: a rank could not map a peer's staging buffer (%s)" % err)
        self.c = cabi.ltg_comm(C.addressof(self.os), self.n_ranks, self.rank, C.cast(lib.ltg_oneshot_all_reduce, C.c_void_p),
                               C.cast(lib.ltg_oneshot_all_gather, C.c_void_p))
        self.kind = "oneshot-ipc"
        self.count = self.n_ranks
        self._expired_off = int(lib.ltg_oneshot_expired_offset(self.n_ranks))

    def expired_waits(self):
        """device-side waits for a peer's message that gave up (0 in a healthy run); synchronises the device"""
        import torch
        torch.cuda.synchronize()
        return self.buf.read_u32(self._expired_off)

    def close(self):
        import torch
        if self.buf is not None:
            torch.cuda.synchronize()
            dist.barrier(group=self.group)       # no peer may still be writing into this rank's stage
            self.buf.close()
            self.buf = None

    def abort(self):
        self.buf = None                          # (leak on the error path: a peer may still hold the mapping)
