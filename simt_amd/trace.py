"""roctx ranges around the phases of an iteration (SURVEY 5 tracing row; the reference has only `time.asctime` prints,
tools/trainV2_simt.py:234,453-456).  Off by default; SIMT_ROCTX=1 loads librocprofiler-sdk-roctx (falls back to libroctx64) and
`with trace.range("forward"):` pushes / pops a named range on the calling thread, which `rocprofv3 --marker-trace --kernel-trace` shows
above the kernels enqueued inside it (frozen-forward / forward / head / backward / exchange / optimiser).  With the switch off
`range()` returns a shared no-op context manager: no ctypes call on the hot path."""
import contextlib
import ctypes
import os

_lib = None
_on = os.environ.get("SIMT_ROCTX", "0") not in ("", "0")
_null = contextlib.nullcontext()


def _load():
    global _lib, _on
    if _lib is None and _on:
        for name in ("librocprofiler-sdk-roctx.so", "libroctx64.so"):
            try:
                _lib = ctypes.CDLL(name)
                _lib.roctxRangePushA.argtypes = [ctypes.c_char_p]
                break
            except OSError:
                _lib = None
        if _lib is None:
            _on = False
    return _lib


class _Range:
    __slots__ = ("name",)

    def __init__(self, name):
        self.name = name.encode()

    def __enter__(self):
        _lib.roctxRangePushA(self.name)

    def __exit__(self, *exc):
        _lib.roctxRangePop()
        return False


def enabled():
    return _on and _load() is not None


def range(name):
    """Context manager: a roctx range named `name` (SIMT_ROCTX=1), else a no-op."""
    return _Range(name) if enabled() else _null
