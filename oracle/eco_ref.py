"""ctypes binding of oracle/_ref/libeco_ref.so -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

``libeco_ref.so`` is built by oracle/Makefile from the reference's own, unmodified sources
(caffe_3d/src/caffe/util/im2col.cpp, layers/pooling_layer.cpp) behind the stand-in headers of
oracle/ref_shim/, plus ``ref_conv_forward`` = the reference's ConvolutionLayer::Forward_cpu call sequence
(per image: reference im2col -> cblas_sgemm -> bias sgemm; conv_layer.cpp:28-43, base_conv_layer.cpp:264-287).
The GEMM is SciPy's bundled OpenBLAS (``scipy_cblas_sgemm``), the class of library the reference links.

Only tests/ and bench.py's cpu_baseline leg may import this; it pins oracle/eco_oracle.py against compiled
reference code (tests/test_oracle_ref.py) and is the "caffe-cost" CPU baseline."""
from __future__ import annotations

import ctypes as C
import glob
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_ref", "libeco_ref.so")
REFERENCE = "/root/reference/caffe_3d"

_lib = None
_blas = None


def available() -> bool:
    return os.path.exists(LIB_PATH) or os.path.isdir(REFERENCE)


def build() -> str:
    """(Re)build from the reference tree when it is present (authoring container); else use the shipped .so."""
    if os.path.isdir(REFERENCE):
        subprocess.run(["make", "-C", _HERE, "all"], check=True, stdout=subprocess.DEVNULL)
    if not os.path.exists(LIB_PATH):
        raise FileNotFoundError(f"{LIB_PATH} missing and {REFERENCE} not present to build it from")
    return LIB_PATH


def _openblas():
    """SciPy's bundled OpenBLAS: (CDLL, sgemm symbol, set_num_threads symbol)."""
    global _blas
    if _blas is None:
        import scipy
        cands = glob.glob(os.path.join(os.path.dirname(scipy.__file__), "..", "scipy.libs", "libscipy_openblas*.so*"))
        if not cands:
            raise ImportError("SciPy's bundled OpenBLAS (scipy.libs/libscipy_openblas*.so) not found")
        dll = C.CDLL(cands[0])
        pre = "scipy_" if hasattr(dll, "scipy_cblas_sgemm") else ""
        _blas = (dll, getattr(dll, pre + "cblas_sgemm"), getattr(dll, pre + "openblas_set_num_threads"),
                 getattr(dll, pre + "openblas_get_num_threads"))
    return _blas


def set_blas_threads(n: int) -> int:
    _, _, setter, getter = _openblas()
    getter.restype = C.c_int
    old = getter()
    setter(C.c_int(int(n)))
    return old


def lib():
    global _lib
    if _lib is None:
        d = C.CDLL(build())
        ip = C.POINTER(C.c_int)
        d.ref_im2col.argtypes = [C.c_void_p, C.c_int, ip, ip, ip, ip, ip, C.c_void_p]
        d.ref_im2col.restype = None
        d.ref_conv_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                       ip, ip, ip, ip, C.c_void_p, C.c_int]
        d.ref_conv_forward.restype = C.c_int
        d.ref_pool_forward.argtypes = [C.c_void_p, ip, C.c_int, C.c_int, ip, ip, ip, C.c_void_p, ip]
        d.ref_pool_forward.restype = C.c_int
        _lib = d
    return _lib


def _ia(v):
    return (C.c_int * len(v))(*[int(x) for x in v])


def im2col(x: np.ndarray, kernel, stride, pad) -> np.ndarray:
    """Reference im2col_cpu / im2col_nd_cpu on one image [C, *spatial] -> [C*prod(kernel), prod(out)]."""
    x = np.ascontiguousarray(x, np.float32)
    nsp = x.ndim - 1
    out = [(x.shape[1 + i] + 2 * pad[i] - kernel[i]) // stride[i] + 1 for i in range(nsp)]
    kdim = x.shape[0] * int(np.prod(kernel))
    col = np.empty((kdim, int(np.prod(out))), np.float32)
    lib().ref_im2col(x.ctypes.data, nsp, _ia(x.shape), _ia([kdim] + out), _ia(kernel), _ia(pad), _ia(stride),
                     col.ctypes.data)
    return col


def convolution(x, w, b, kernel, stride, pad, image_threads: int = 1) -> np.ndarray:
    """ConvolutionLayer::Forward_cpu's call sequence over the reference im2col and OpenBLAS sgemm.
    image_threads=1: images in sequence, BLAS threaded (the reference's structure); >1: that many images at a
    time, one BLAS thread each."""
    x = np.ascontiguousarray(x, np.float32)
    w = np.ascontiguousarray(w, np.float32)
    n, cin = x.shape[:2]
    nsp = x.ndim - 2
    cout = w.shape[0]
    out = [(x.shape[2 + i] + 2 * pad[i] - kernel[i]) // stride[i] + 1 for i in range(nsp)]
    y = np.empty((n, cout) + tuple(out), np.float32)
    bb = None if b is None else np.ascontiguousarray(b, np.float32)
    _, sgemm, _, _ = _openblas()
    old = set_blas_threads(1) if image_threads > 1 else None
    try:
        rc = lib().ref_conv_forward(x.ctypes.data, w.ctypes.data, None if bb is None else bb.ctypes.data, y.ctypes.data,
                                    n, cin, cout, nsp, _ia(x.shape[2:]), _ia(kernel), _ia(stride), _ia(pad),
                                    C.cast(sgemm, C.c_void_p), int(image_threads))
    finally:
        if old is not None:
            set_blas_threads(old)
    if rc != 0:
        raise ValueError("ref_conv_forward: unsupported geometry")
    return y


def pooled_shape(shape, method, kernel, stride, pad):
    """PoolingLayer::LayerSetUp + Reshape (the ceil rule, pooling_layer.cpp:117-147) for a 2-D or 3-D blob shape."""
    out = (C.c_int * len(shape))()
    lib().ref_pool_forward(None, _ia(shape), len(shape), 0 if method == "MAX" else 1,
                           None if kernel is None else _ia(kernel), _ia(stride), _ia(pad), None, out)
    return tuple(out)


def pooling(x, method, kernel, stride, pad) -> np.ndarray:
    """PoolingLayer::Forward_cpu (2-D blobs [n,c,h,w] only: the reference CPU path has no N-D pooling)."""
    x = np.ascontiguousarray(x, np.float32)
    assert x.ndim == 4
    shp = pooled_shape(x.shape, method, kernel, stride, pad)
    y = np.empty(shp, np.float32)
    out = (C.c_int * 4)()
    rc = lib().ref_pool_forward(x.ctypes.data, _ia(x.shape), 4, 0 if method == "MAX" else 1,
                                None if kernel is None else _ia(kernel), _ia(stride), _ia(pad), y.ctypes.data, out)
    assert rc == 0
    return y
