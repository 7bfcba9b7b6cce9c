"""Test-only backend: binds the C ABI to tests/emu/libeco_emu.so (the CPU fiber-emulated
build of csrc/*.hip) with NumPy host arrays standing in for device memory.

This is how the CPU test-suite exercises the kernels' index math and the engine's fused
plan without a GPU.  The product package never imports this module.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(os.path.dirname(_HERE))
EMU_LIB = os.path.join(_HERE, "libeco_emu.so")
CSRC = os.path.join(_ROOT, "eco-efficient-video-understanding_amd", "csrc")


def build_emu(force: bool = False) -> str:
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))]
    srcs += [os.path.join(_HERE, "hip_emu.cpp"), os.path.join(_HERE, "hip_emu.h"),
             os.path.join(_ROOT, "include", "eco_hip.h")]
    stale = force or not os.path.exists(EMU_LIB) or any(
        os.path.getmtime(s) > os.path.getmtime(EMU_LIB) for s in srcs)
    if stale:
        subprocess.run(["make", "-j8", "-C", CSRC, "emu"], check=True, stdout=subprocess.DEVNULL)
    return EMU_LIB


class NumpyAllocator:
    """Host arrays as "device" buffers, registered with the emulator's bounds checker."""

    def __init__(self, dll: ctypes.CDLL) -> None:
        self._dll = dll
        dll.emu_register_buffer.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
        dll.emu_violation_count.restype = ctypes.c_int
        dll.emu_set_strict(1)

    def empty(self, nelems: int, dtype=np.float32) -> np.ndarray:
        dt = np.dtype(dtype)   # poison: NaN (fp32 and, as 0xffff, bf16 bits) or -1
        a = np.full(max(int(nelems), 1), np.nan if dt == np.float32 else (0xFFFF if dt == np.uint16 else -1), dtype=dt)
        self._dll.emu_register_buffer(a.ctypes.data, a.nbytes)
        return a

    @staticmethod
    def ptr(h: np.ndarray) -> int:
        return h.ctypes.data

    @staticmethod
    def upload(h: np.ndarray, arr: np.ndarray) -> None:
        a = np.ascontiguousarray(arr).reshape(-1)
        h[: a.size] = a

    @staticmethod
    def download(h: np.ndarray, nelems: int) -> np.ndarray:
        return h[:nelems].copy()

    def violations(self) -> int:
        return int(self._dll.emu_violation_count())


def emu_backend():
    """(EcoLib bound to the emulator build, NumpyAllocator)."""
    sys.path.insert(0, _ROOT)
    from eco_amd import hip
    path = build_emu()
    lib = hip.EcoLib(path)
    assert not lib.is_device_build
    dll = ctypes.CDLL(path)
    dll.emu_clear_buffers()
    return lib, NumpyAllocator(dll)
