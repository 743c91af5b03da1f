"""Minimal HIP runtime access (events only) for timing kernels on the stream the library launches
on -- torch.cuda.Event only sees torch's own record calls.  Binds the libamdhip64 that torch has
already loaded into the process."""
from __future__ import annotations

import ctypes as C

_hip = None


def _lib():
    global _hip
    if _hip is None:
        import torch  # noqa: F401  (loads libamdhip64 into the process)
        path = None
        with open("/proc/self/maps") as f:
            for line in f:
                if "libamdhip64" in line:
                    path = line.split()[-1]
                    break
        _hip = C.CDLL(path or "libamdhip64.so")
        _hip.hipEventCreate.argtypes = [C.POINTER(C.c_void_p)]
        _hip.hipEventCreateWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
        _hip.hipEventDestroy.argtypes = [C.c_void_p]
        _hip.hipEventSynchronize.argtypes = [C.c_void_p]
        _hip.hipEventElapsedTime.argtypes = [C.POINTER(C.c_float), C.c_void_p, C.c_void_p]
        _hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        _hip.hipFree.argtypes = [C.c_void_p]
        _hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
        _hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        _hip.hipDeviceSynchronize.argtypes = []
        _hip.hipIpcGetMemHandle.argtypes = [C.c_void_p, C.c_void_p]            # (hipIpcMemHandle_t*, void*): 64 opaque bytes
        _hip.hipIpcOpenMemHandle.argtypes = [C.POINTER(C.c_void_p), IpcHandle, C.c_uint]
        _hip.hipIpcCloseMemHandle.argtypes = [C.c_void_p]
    return _hip


class IpcHandle(C.Structure):
    """hipIpcMemHandle_t (passed BY VALUE to hipIpcOpenMemHandle)"""
    _fields_ = [("reserved", C.c_char * 64)]


class IpcBuffer:
    """Zeroed device memory of this process that other processes of the node can map (hipMalloc + hipIpcGetMemHandle); `open_peer` maps a
    peer's.  The one-shot exchange's staging buffers (ltgan._rccl.OneShotComm).  Needs HSA_ENABLE_IPC_MODE_LEGACY=0 on this pool."""

    def __init__(self, nbytes):
        h = _lib()
        self.ptr = C.c_void_p()
        if h.hipMalloc(C.byref(self.ptr), nbytes) != 0:
            raise RuntimeError("hipMalloc(%d) failed" % nbytes)
        self.nbytes = int(nbytes)
        assert h.hipMemset(self.ptr, 0, nbytes) == 0 and h.hipDeviceSynchronize() == 0
        hd = IpcHandle()
        if h.hipIpcGetMemHandle(C.byref(hd), self.ptr) != 0:
            h.hipGetLastError()
            h.hipFree(self.ptr)
            raise RuntimeError("hipIpcGetMemHandle failed")
        self.handle = bytes(hd.reserved if isinstance(hd.reserved, bytes) and len(hd.reserved) == 64 else C.string_at(C.addressof(hd), 64))
        self._peers = []

    def open_peer(self, handle_bytes):
        h = _lib()
        hd = IpcHandle()
        C.memmove(C.addressof(hd), handle_bytes, 64)
        p = C.c_void_p()
        if h.hipIpcOpenMemHandle(C.byref(p), hd, 1) != 0:      # hipIpcMemLazyEnablePeerAccess
            h.hipGetLastError()
            raise RuntimeError("hipIpcOpenMemHandle failed")
        self._peers.append(p)
        return p.value

    def read_u32(self, offset):
        h = _lib()
        out = C.c_uint32()
        assert h.hipMemcpy(C.byref(out), C.c_void_p(self.ptr.value + offset), 4, 2) == 0     # hipMemcpyDeviceToHost
        return int(out.value)

    def close(self):
        h = _lib()
        for p in self._peers:
            h.hipIpcCloseMemHandle(p)
        self._peers = []
        if self.ptr:
            h.hipFree(self.ptr)
            self.ptr = C.c_void_p()


class EventPair:
    def __init__(self, timing=True):
        h = _lib()
        self.start, self.stop = C.c_void_p(), C.c_void_p()
        if timing:
            assert h.hipEventCreate(C.byref(self.start)) == 0 and h.hipEventCreate(C.byref(self.stop)) == 0
        else:  # hipEventDisableTiming = 0x2: cheap ordering-only events (fork / join of two streams)
            assert h.hipEventCreateWithFlags(C.byref(self.start), 2) == 0 and h.hipEventCreateWithFlags(C.byref(self.stop), 2) == 0

    def elapsed_ms(self):
        h = _lib()
        ms = C.c_float()
        if h.hipEventSynchronize(self.stop) != 0 or h.hipEventElapsedTime(C.byref(ms), self.start, self.stop) != 0:
            h.hipGetLastError()      # an event that was never recorded: clear the (sticky) error, report "no sample"
            return None
        return float(ms.value)

    def __del__(self):
        try:
            h = _lib()
            h.hipEventDestroy(self.start)
            h.hipEventDestroy(self.stop)
        except Exception:
            pass
