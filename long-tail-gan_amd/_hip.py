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
    return _hip


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
