"""pytest configuration: `gpu` marker, import paths, and the two kernel backends.

* backend "emu": tests/emu/libeco_emu.so -- csrc/*.hip compiled against the CPU fiber
  emulator; runs everywhere (CPU suite, `-m "not gpu"`).
* backend "hip": the product library libeco_hip.so on a real MI355X (`-m gpu`).
Both are driven through the same C ABI (include/eco_hip.h), so the parity tests are
literally the same code on CPU and GPU.
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

REFERENCE = "/root/reference"
HAVE_REFERENCE = os.path.isdir(os.path.join(REFERENCE, "caffe_3d"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "reference: needs /root/reference (authoring container only)")


class Backend:
    def __init__(self, kind, lib, alloc):
        self.kind, self.lib, self.alloc = kind, lib, alloc
        self._keep = []  # buffers stay alive for the whole test (kernels hold raw pointers)

    def dev(self, arr):
        arr = np.ascontiguousarray(arr)
        h = self.alloc.empty(arr.size, arr.dtype)
        self.alloc.upload(h, arr)
        self._keep.append(h)
        return h

    def empty(self, shape, dtype=np.float32):
        h = self.alloc.empty(int(np.prod(shape)), dtype)
        self._keep.append(h)
        return h

    def ptr(self, h, offset_elems=0):
        return self.alloc.ptr(h) + 4 * offset_elems

    def host(self, h, shape):
        return np.asarray(self.alloc.download(h, int(np.prod(shape)))).reshape(shape).copy()

    def check_clean(self):
        if self.kind == "emu":
            assert self.alloc.violations() == 0, "emulator saw out-of-bounds global accesses"


def _make_backend(kind):
    if kind == "emu":
        from tests.emu.backend import emu_backend
        lib, alloc = emu_backend()
        return Backend("emu", lib, alloc)
    import torch
    from eco_amd import hip
    from eco_amd.engine import TorchAllocator
    assert torch.cuda.is_available(), "the gpu-marked tests need a GPU"
    lib = hip.load()  # fails loudly if the HIP library is missing
    assert lib.is_device_build
    return Backend("hip", lib, TorchAllocator(0))


@pytest.fixture(params=["emu", pytest.param("hip", marks=pytest.mark.gpu)])
def backend(request):
    be = _make_backend(request.param)
    yield be
    be.check_clean()


@pytest.fixture
def hip_backend():
    return _make_backend("hip")
