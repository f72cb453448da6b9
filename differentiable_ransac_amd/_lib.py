"""ctypes binding of libdransac.so (C ABI declared in include/dransac.h).

The library is the product: there is no Python/CPU fallback.  `lib()` raises if the shared
object is missing or cannot be loaded; every wrapper raises `DransacError` on a non-zero status.
torch is imported first so that the HIP runtime torch ships (same SONAME, libamdhip64.so.7) is the
one the library binds to -- one runtime per process, shared streams and allocations.
"""
from __future__ import annotations

import ctypes
import os
import threading
from ctypes import c_char_p, c_double, c_float, c_int, c_uint64, c_void_p
from typing import Optional

import torch  # noqa: F401  (must precede the CDLL load, see module docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DRANSAC_LIB") or os.path.join(_HERE, "libdransac.so")   # DRANSAC_LIB: A/B runs against another build
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "dransac.h")

_lib: Optional[ctypes.CDLL] = None


class DransacError(RuntimeError):
    pass


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DransacError(
                f"{LIB_PATH} not found: build it with `python -m differentiable_ransac_amd.build` "
                "(there is deliberately no CPU fallback)")
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.dr_last_error.restype = c_char_p
        _lib.dr_version.restype = c_int
    return _lib


def check(status: int, what: str) -> None:
    if status != 0:
        msg = lib().dr_last_error().decode("utf-8", "replace")
        raise DransacError(f"{what} failed with status {status}: {msg}")


# Devices of the tensors whose pointers were taken since the last call(): every wrapper builds its argument list with
# ptr(...) ... stream() and hands it to call(), so call() can check that one launch never mixes GPUs and can make that
# GPU current for the launch (HIP launches on the CURRENT device, whatever device the pointers belong to).
_ctx = threading.local()


def _reset() -> None:
    _ctx.devs = set()
    _ctx.keep = []


def _devices() -> set:
    d = getattr(_ctx, "devs", None)
    if d is None:
        d = _ctx.devs = set()
    return d


def ptr(t: Optional[torch.Tensor]) -> c_void_p:
    """Device pointer of a contiguous CUDA tensor (None -> NULL).  The tensor is kept alive until the call() it is an
    argument of has enqueued its launch: `ptr(x.contiguous())` / `ptr(x.to(dtype))` create temporaries that would
    otherwise be freed as soon as ptr() returns, and the caching allocator would hand their block to the next temporary of
    the same argument list."""
    if t is None:
        return c_void_p(0)
    if not t.is_cuda or not t.is_contiguous():
        # the argument list this pointer belongs to will never reach call(): drop what its earlier ptr() calls recorded, or
        # the next call on this thread would see stale devices ("tensors of one call live on different GPUs") and pin tensors
        _reset()
        raise DransacError("libdransac operates on GPU tensors only (got a CPU tensor)" if not t.is_cuda
                           else "tensor must be contiguous")
    _devices().add(t.device.index)
    keep = getattr(_ctx, "keep", None)
    if keep is None:
        keep = _ctx.keep = []
    keep.append(t)
    return c_void_p(t.data_ptr())


def _launch_device() -> Optional[int]:
    devs = _devices()
    if len(devs) > 1:
        _reset()
        raise DransacError(f"tensors of one call live on different GPUs: {sorted(devs)}")
    return next(iter(devs)) if devs else None


def stream() -> c_void_p:
    """The current torch stream OF THE DEVICE THE CALL'S TENSORS LIVE ON (not of the current device)."""
    dev = _launch_device()
    return c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def suffix(dtype: torch.dtype) -> str:
    if dtype == torch.float32:
        return "f32"
    if dtype == torch.float64:
        return "f64"
    raise DransacError(f"unsupported dtype {dtype}: f32 and f64 only (SURVEY Q15)")


def scalar(dtype: torch.dtype, v: float):
    return c_float(v) if dtype == torch.float32 else c_double(v)


def call(name: str, *args) -> None:
    fn = getattr(lib(), name)
    fn.restype = c_int
    dev = _launch_device()
    _ctx.devs = set()
    try:
        if dev is not None and dev != torch.cuda.current_device():
            with torch.cuda.device(dev):       # tensors on another GPU than the current one: launch there
                status = fn(*args)
        else:
            status = fn(*args)
    finally:
        _ctx.keep = []                         # the launch is enqueued on the tensors' stream: stream order protects them now
    check(status, name)


__all__ = ["lib", "check", "ptr", "stream", "suffix", "scalar", "call", "DransacError", "LIB_PATH", "HEADER_PATH",
           "c_int", "c_uint64", "c_float", "c_double", "c_void_p"]
