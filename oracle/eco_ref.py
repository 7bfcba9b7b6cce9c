"""ctypes binding of oracle/_ref/libeco_ref.so -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

``libeco_ref.so`` is built by oracle/Makefile from the reference's own, unmodified sources
(caffe_3d/src/caffe/util/im2col.cpp and layers/{pooling,bn,permute,eltwise,concat,inner_product,reshape,relu,
base_conv,conv}_layer.cpp) behind the stand-in headers of oracle/ref_shim/: ``convolution_layer`` runs the compiled
ConvolutionLayer class itself; ``convolution`` (``ref_conv_forward``) is the same call sequence (per image: reference
im2col -> cblas_sgemm -> bias sgemm; conv_layer.cpp:28-43, base_conv_layer.cpp:264-287) with an option to spread a
batch's images over host threads, pinned bit-identically to the class by tests/test_oracle_ref.py.
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
        if hasattr(d, "ref_bn_forward"):   # (a .so of round 2 shipped to an old box lacks the layer entry points)
            fp, pp = C.c_void_p, C.POINTER(C.c_void_p)
            d.ref_set_sgemm.argtypes = [C.c_void_p]
            d.ref_set_sgemm.restype = None
            d.ref_bn_forward.argtypes = [fp, ip, C.c_int, fp, fp, fp, fp, C.c_float, C.c_int, fp]
            d.ref_permute_forward.argtypes = [fp, ip, C.c_int, ip, C.c_int, fp, ip]
            d.ref_eltwise_forward.argtypes = [pp, C.c_int, ip, C.c_int, C.c_int, fp, fp]
            d.ref_concat_forward.argtypes = [pp, C.c_int, ip, C.c_int, C.c_int, fp, ip]
            d.ref_inner_product_forward.argtypes = [fp, ip, C.c_int, fp, fp, C.c_int, C.c_int, fp]
            d.ref_reshape_shape.argtypes = [ip, C.c_int, C.POINTER(C.c_longlong), C.c_int, C.c_int, C.c_int, ip]
            d.ref_relu_forward.argtypes = [fp, C.c_long, C.c_float, fp]
            if hasattr(d, "ref_convolution_layer_forward"):   # (round 4: the compiled ConvolutionLayer class)
                d.ref_convolution_layer_forward.argtypes = [fp, ip, C.c_int, fp, fp, C.c_int, ip, C.c_int, ip, C.c_int,
                                                            ip, C.c_int, C.c_int, fp, ip]
                d.ref_convolution_layer_forward.restype = C.c_int
            for f in ("ref_bn_forward", "ref_permute_forward", "ref_eltwise_forward", "ref_concat_forward",
                      "ref_inner_product_forward", "ref_reshape_shape", "ref_relu_forward"):
                getattr(d, f).restype = C.c_int
            d.ref_set_sgemm(C.cast(_openblas()[1], C.c_void_p))   # every caffe_cpu_gemm<float> of the compiled layers
        _lib = d
    return _lib


def has_layers() -> bool:
    """True when the loaded library carries the compiled layer files (BN, Permute, Eltwise, Concat, InnerProduct, ...)."""
    return hasattr(lib(), "ref_bn_forward")


def has_conv_layer() -> bool:
    """True when the loaded library carries the compiled BaseConvolutionLayer / ConvolutionLayer."""
    return hasattr(lib(), "ref_convolution_layer_forward")


def _ia(v):
    return (C.c_int * len(v))(*[int(x) for x in v])


def _f32(a):
    return np.ascontiguousarray(a, np.float32)


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


def convolution_layer(x, w, b, kernel, stride=(), pad=(), force_nd_im2col: bool = False) -> np.ndarray:
    """The compiled ConvolutionLayer<float> itself: LayerSetUp / Reshape (base_conv_layer.cpp:13-262) and Forward_cpu
    (conv_layer.cpp:28-43) over OpenBLAS sgemm.  kernel / stride / pad as the prototxt's repeated fields (one value, or one
    per spatial axis; stride / pad may be empty = the schema defaults 1 / 0)."""
    x, w = _f32(x), _f32(w)
    bb = None if b is None else _f32(b)
    out = (C.c_int * x.ndim)()
    args = (_ia(x.shape), x.ndim)
    geo = (_ia(kernel), len(kernel), _ia(stride) if len(stride) else None, len(stride), _ia(pad) if len(pad) else None,
           len(pad), int(force_nd_im2col))
    lib().ref_convolution_layer_forward(None, *args, None, None if bb is None else bb.ctypes.data, w.shape[0], *geo, None, out)
    y = np.empty(tuple(out), np.float32)
    rc = lib().ref_convolution_layer_forward(x.ctypes.data, *args, w.ctypes.data, None if bb is None else bb.ctypes.data,
                                             w.shape[0], *geo, y.ctypes.data, out)
    assert rc == 0
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


# ---- the other layer types of the deploy graphs, through their compiled reference Forward_cpu ------------------------

def bn_inference(x, slope, bias, mean, var, eps, frozen: bool = False) -> np.ndarray:
    """BNLayer::Forward_cpu, TEST phase (bn_layer.cpp:93-207).  Blobs of at most 4 axes -- the reference's CPU code
    CHECK-fails beyond (LegacyShape); a 5-D blob [n,c,d,h,w] is evaluated as [n,c,d*h,w] (the arithmetic is per
    channel; the eps rule of the cuDNN path the reference takes for 5-D blobs is the CALLER's business)."""
    x = _f32(x)
    shp = x.shape if x.ndim <= 4 else (x.shape[0], x.shape[1], int(np.prod(x.shape[2:-1])), x.shape[-1])
    y = np.empty(shp, np.float32)
    ps = [_f32(p).reshape(-1) for p in (slope, bias, mean, var)]
    rc = lib().ref_bn_forward(x.ctypes.data, _ia(shp), len(shp), ps[0].ctypes.data, ps[1].ctypes.data, ps[2].ctypes.data,
                              ps[3].ctypes.data, float(eps), int(frozen), y.ctypes.data)
    assert rc == 0
    return y.reshape(x.shape)


def permute(x, order) -> np.ndarray:
    """PermuteLayer::Forward_cpu (permute_layer.cpp:98-114)."""
    x = _f32(x)
    out = (C.c_int * x.ndim)()
    lib().ref_permute_forward(None, _ia(x.shape), x.ndim, _ia(order), len(order), None, out)
    y = np.empty(tuple(out), np.float32)
    assert lib().ref_permute_forward(x.ctypes.data, _ia(x.shape), x.ndim, _ia(order), len(order), y.ctypes.data, out) == 0
    return y


def _ptrs(arrs):
    return (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])


def eltwise(xs, op="SUM", coeffs=None) -> np.ndarray:
    """EltwiseLayer::Forward_cpu (eltwise_layer.cpp:49-119)."""
    xs = [_f32(x) for x in xs]
    y = np.empty(xs[0].shape, np.float32)
    cf = None if coeffs is None else _f32(coeffs)
    rc = lib().ref_eltwise_forward(_ptrs(xs), len(xs), _ia(xs[0].shape), xs[0].ndim, {"PROD": 0, "SUM": 1, "MAX": 2}[op],
                                   None if cf is None else cf.ctypes.data, y.ctypes.data)
    assert rc == 0
    return y


def concat(xs, axis=1) -> np.ndarray:
    """ConcatLayer::Reshape + Forward_cpu (concat_layer.cpp:17-70)."""
    xs = [_f32(x) for x in xs]
    nd = xs[0].ndim
    shapes = _ia([d for x in xs for d in x.shape])
    out = (C.c_int * nd)()
    lib().ref_concat_forward(_ptrs(xs), len(xs), shapes, nd, int(axis), None, out)
    y = np.empty(tuple(out), np.float32)
    assert lib().ref_concat_forward(_ptrs(xs), len(xs), shapes, nd, int(axis), y.ctypes.data, out) == 0
    return y


def inner_product(x, w, b, axis=1) -> np.ndarray:
    """InnerProductLayer::Forward_cpu (inner_product_layer.cpp:81-93) over OpenBLAS sgemm."""
    x, w = _f32(x), _f32(w)
    n_out = w.shape[0]
    y = np.empty(x.shape[:axis] + (n_out,), np.float32)
    bb = None if b is None else _f32(b)
    rc = lib().ref_inner_product_forward(x.ctypes.data, _ia(x.shape), x.ndim, w.reshape(n_out, -1).ctypes.data,
                                         None if bb is None else bb.ctypes.data, n_out, int(axis), y.ctypes.data)
    assert rc == 0
    return y


def reshape_shape(shape, dims, axis=0, num_axes=-1):
    """ReshapeLayer::LayerSetUp + Reshape (reshape_layer.cpp:9-90): the top shape for bottom `shape`."""
    out = (C.c_int * 16)()
    d = (C.c_longlong * len(dims))(*[int(v) for v in dims])
    n = lib().ref_reshape_shape(_ia(shape), len(shape), d, len(dims), int(axis), int(num_axes), out)
    return tuple(out[:n])


def relu(x, negative_slope=0.0) -> np.ndarray:
    """ReLULayer::Forward_cpu (relu_layer.cpp:10-20)."""
    x = _f32(x)
    y = np.empty(x.shape, np.float32)
    assert lib().ref_relu_forward(x.ctypes.data, x.size, float(negative_slope), y.ctypes.data) == 0
    return y
