"""CPU ORACLE for the ECO forward path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.  The product package
(``eco-efficient-video-understanding_amd``) never imports it and has no CPU
fallback: it fails loudly when the HIP library is missing.

What this is: a NumPy fp32 restatement of the reference's (caffe_3d) layer
arithmetic for the layers on the ECO inference path, written from the
reference sources cited per function.  The reference itself cannot be built in
this image as a whole (needs protobuf+protoc, glog, gflags, boost, HDF5,
LMDB/LevelDB, OpenCV, a CBLAS; and its CPU path cannot execute 5-D BN / 3-D
pooling at all: layers/bn_layer.cpp:70-73, layers/pooling_layer.cpp:177-201).
PINNED, in three ways:
  * against the golden vectors of the reference's own unit tests
    (tests/test_oracle_golden.py; fixtures in tests/golden/) -- GEMM, naive-loop
    convolution (2-D and 3-D), 2-D/3-D pooling known answers, BN inference
    formula, concat/eltwise/reshape/inner-product properties;
  * layer by layer against the reference's OWN object code: eleven reference
    .cpp files compiled unmodified into oracle/_ref/libeco_ref.so
    (oracle/Makefile; tests/test_oracle_ref.py: bit-identical on im2col, the
    ConvolutionLayer forward, MAX pooling, BN, Permute, Eltwise, Concat, ReLU,
    Reshape; fp32 rounding on AVE pooling and InnerProduct);
  * whole-net logits against EXECUTED reference code at full width:
    tests/golden/reference_logits.json (generator make_reference_logits.py: the
    reference's deploy prototxt files walked layer by layer through those
    compiled classes) holds ECO-Lite N=4 (BASELINE configs[0]), N=8 x 2 clips,
    N=16 (configs[1]'s clip geometry), N=32 (configs[4]'s) and ECO-Full N=4;
    tests/test_reference_logits.py compares this module with it (2-3e-7 of
    the largest logit, every blob's fingerprint at 2e-5) and, on the GPU, the
    HIP path directly.
An independent torch-CPU implementation cross-checks every op as well.

Third-party arithmetic the reference delegates to (not under /root/reference):
CBLAS ``cblas_sgemm`` (ATLAS/OpenBLAS/MKL, unpinned; call sites
util/math_functions.cpp:12-21) and, for the 5-D layers, cuDNN >= 5
(``cudnnBatchNormalizationForwardInference`` SPATIAL mode,
``cudnnPoolingForward`` with AVERAGE_COUNT_INCLUDE_PADDING;
layers/cudnn_bn_layer.cu:24-37, util/cudnn.hpp:234-262).  GEMM here is
``numpy.matmul`` in fp32 (OpenBLAS sgemm: the same routine the reference
links); only the summation order differs.

Layout: everything is fp32, row-major N,C,[D,]H,W exactly like ``Blob``.
"""
from __future__ import annotations

import numpy as np

F32 = np.float32
FLT_MAX = np.finfo(np.float32).max


# --------------------------------------------------------------------------
# Convolution: im2col + GEMM per image
# --------------------------------------------------------------------------
def im2col_nd(x: np.ndarray, kernel, stride, pad) -> np.ndarray:
    """One image ``[C, *spatial]`` -> col ``[C*prod(kernel), prod(out)]``.

    Restates ``im2col_cpu`` (util/im2col.cpp:27-64, 2-D) and
    ``im2col_nd_core_cpu`` (util/im2col.cpp:91-158, N-D): row index
    ``(c*kd + i)*kh*kw + ...`` i.e. channel-major then kernel offsets in
    row-major order; column index = output position in row-major order;
    out-of-image taps read 0.
    """
    C = x.shape[0]
    sp = x.shape[1:]
    n = len(sp)
    out = [(sp[i] + 2 * pad[i] - kernel[i]) // stride[i] + 1 for i in range(n)]
    xp = np.zeros((C,) + tuple(sp[i] + 2 * pad[i] for i in range(n)), dtype=F32)
    xp[(slice(None),) + tuple(slice(pad[i], pad[i] + sp[i]) for i in range(n))] = x
    col = np.empty((C,) + tuple(kernel) + tuple(out), dtype=F32)
    for tap in np.ndindex(*kernel):
        sl = tuple(slice(tap[i], tap[i] + stride[i] * (out[i] - 1) + 1, stride[i]) for i in range(n))
        col[(slice(None),) + tap] = xp[(slice(None),) + sl]
    return col.reshape(C * int(np.prod(kernel)), int(np.prod(out)))


def convolution(x: np.ndarray, w: np.ndarray, b, kernel, stride, pad) -> np.ndarray:
    """``ConvolutionLayer::Forward_cpu`` (layers/conv_layer.cpp:28-43): for each
    image, ``forward_cpu_gemm`` = im2col + ``W[Cout x K] . col[K x N]``
    (layers/base_conv_layer.cpp:264-279; 1x1/s1/p0 skips im2col, :110-117,267)
    then ``forward_cpu_bias`` = rank-1 update with an all-ones multiplier
    (:282-287).  Cross-correlation, zero padding, group 1, dilation 1."""
    x = np.ascontiguousarray(x, dtype=F32)
    N = x.shape[0]
    cout = w.shape[0]
    sp = x.shape[2:]
    n = len(sp)
    out = [(sp[i] + 2 * pad[i] - kernel[i]) // stride[i] + 1 for i in range(n)]
    wm = np.ascontiguousarray(w, dtype=F32).reshape(cout, -1)
    is_1x1 = all(k == 1 for k in kernel) and all(s == 1 for s in stride) and all(p == 0 for p in pad)
    y = np.empty((N, cout) + tuple(out), dtype=F32)
    ones = np.ones((1, int(np.prod(out))), dtype=F32)
    for i in range(N):
        col = x[i].reshape(x.shape[1], -1) if is_1x1 else im2col_nd(x[i], kernel, stride, pad)
        yi = np.matmul(wm, col)  # sgemm
        if b is not None:
            yi += np.matmul(np.asarray(b, dtype=F32).reshape(cout, 1), ones)
        y[i] = yi.reshape((cout,) + tuple(out))
    return y


def convolution_naive(x, w, b, kernel, stride, pad) -> np.ndarray:
    """Direct 7-deep loop, restating the reference TEST-side ``caffe_conv``
    (src/caffe/test/test_convolution_layer.cpp:18-134).  Tiny shapes only."""
    x = np.asarray(x, dtype=F32)
    n = x.ndim - 2
    k3 = [1] * (3 - n) + list(kernel)
    s3 = [1] * (3 - n) + list(stride)
    p3 = [0] * (3 - n) + list(pad)
    x5 = x.reshape(x.shape[:2] + (1,) * (3 - n) + x.shape[2:])
    w5 = np.asarray(w, dtype=F32).reshape(w.shape[:2] + tuple(k3))
    N, C, D, H, W = x5.shape
    O = w5.shape[0]
    od = [(x5.shape[2 + i] + 2 * p3[i] - k3[i]) // s3[i] + 1 for i in range(3)]
    y = np.zeros((N, O) + tuple(od), dtype=F32)
    for nn in range(N):
        for o in range(O):
            for c in range(C):
                for z in range(od[0]):
                    for yy in range(od[1]):
                        for xx in range(od[2]):
                            acc = y[nn, o, z, yy, xx]
                            for r in range(k3[0]):
                                for p in range(k3[1]):
                                    for q in range(k3[2]):
                                        iz = z * s3[0] - p3[0] + r
                                        iy = yy * s3[1] - p3[1] + p
                                        ix = xx * s3[2] - p3[2] + q
                                        if 0 <= iz < D and 0 <= iy < H and 0 <= ix < W:
                                            acc = F32(acc + x5[nn, c, iz, iy, ix] * w5[o, c, r, p, q])
                            y[nn, o, z, yy, xx] = acc
    if b is not None:
        y += np.asarray(b, dtype=F32).reshape((1, O, 1, 1, 1))
    return y.reshape((N, O) + tuple(od[3 - n:]))


# --------------------------------------------------------------------------
# BN (inference / frozen branch)
# --------------------------------------------------------------------------
def _powf(a: np.ndarray, b: float) -> np.ndarray:
    """Elementwise C ``powf(a[i], b)``: what ``caffe_powx`` is (mkl_alternate.hpp:56, ``y[i] = pow(a[i], b)`` on floats
    under <math.h>).  numpy's float32 power loop is a SIMD approximation that differs from libm in the last bit for a
    few percent of inputs, and the correctly rounded double power still differs for ~2 % -- so libm itself is called
    (tests/test_oracle_ref.py pins bn_inference bit for bit against the compiled bn_layer.cpp)."""
    import ctypes
    import ctypes.util
    global _LIBM
    if "_LIBM" not in globals():
        _LIBM = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
        _LIBM.powf.restype = ctypes.c_float
        _LIBM.powf.argtypes = [ctypes.c_float, ctypes.c_float]
    a = np.asarray(a, F32)
    out = np.empty(a.shape, F32)
    flat, of = a.reshape(-1), out.reshape(-1)
    for i in range(flat.size):
        of[i] = _LIBM.powf(float(flat[i]), float(b))
    return out


def bn_inference(x, scale, shift, mean, var, eps) -> np.ndarray:
    """``BNLayer::Forward_cpu`` TEST branch (layers/bn_layer.cpp:93-207): per
    channel (axis 1) ``top = x - mean; top *= (var+eps)^-0.5; top *= scale;
    top += shift`` in that order (lines :103-115, :137-170, :181-206).  N-D
    blobs follow the cuDNN SPATIAL-mode semantics the reference uses for them
    (layers/cudnn_bn_layer.cu:24-37), eps = max(eps, CUDNN_BN_MIN_EPSILON=1e-5)."""
    x = np.asarray(x, dtype=F32)
    C = x.shape[1]
    bshape = (1, C) + (1,) * (x.ndim - 2)
    inv_std = _powf(np.asarray(var, F32).reshape(C) + F32(eps), -0.5)
    y = x - np.asarray(mean, F32).reshape(bshape)
    y *= inv_std.reshape(bshape)
    y *= np.asarray(scale, F32).reshape(bshape)
    y += np.asarray(shift, F32).reshape(bshape)
    return y


def relu(x, negative_slope=0.0) -> np.ndarray:
    """``ReLULayer::Forward_cpu`` (layers/relu_layer.cpp:10-20)."""
    x = np.asarray(x, dtype=F32)
    return (np.maximum(x, F32(0)) + F32(negative_slope) * np.minimum(x, F32(0))).astype(F32)


# --------------------------------------------------------------------------
# Pooling
# --------------------------------------------------------------------------
def pooled_dim(in_dim, k, s, p) -> int:
    """layers/pooling_layer.cpp:131-147."""
    o = int(np.ceil(np.float32(in_dim + 2 * p - k) / s)) + 1
    if p and (o - 1) * s >= in_dim + p:
        o -= 1
    return o


def pooling(x, method, kernel, stride, pad) -> np.ndarray:
    """``PoolingLayer::Forward_cpu`` (layers/pooling_layer.cpp:168-277),
    generalised from (H, W) to any number of spatial axes.

    MAX: window clipped to the image, running max from -FLT_MAX (:199-225).
    AVE: sum over the clipped window divided by the window size *including
    padding*, the window end clipped to ``dim + pad`` (:238-262).  For 3-D
    blobs the reference runs cuDNN AVERAGE_COUNT_INCLUDE_PADDING
    (util/cudnn.hpp:247-249), which agrees with this rule whenever no window
    overhangs ``dim + pad`` (true for every ECO pooling layer)."""
    x = np.asarray(x, dtype=F32)
    sp = x.shape[2:]
    n = len(sp)
    out = [pooled_dim(sp[i], kernel[i], stride[i], pad[i]) for i in range(n)]
    y = np.empty(x.shape[:2] + tuple(out), dtype=F32)
    for o in np.ndindex(*out):
        start = [o[i] * stride[i] - pad[i] for i in range(n)]
        if method == "MAX":
            end = [min(start[i] + kernel[i], sp[i]) for i in range(n)]
            lo = [max(s, 0) for s in start]
            win = x[(slice(None), slice(None)) + tuple(slice(lo[i], end[i]) for i in range(n))]
            y[(slice(None), slice(None)) + o] = np.maximum(
                win.reshape(x.shape[0], x.shape[1], -1).max(axis=2), F32(-FLT_MAX))
        elif method == "AVE":
            end = [min(start[i] + kernel[i], sp[i] + pad[i]) for i in range(n)]
            size = 1
            for i in range(n):
                size *= end[i] - start[i]
            lo = [max(s, 0) for s in start]
            hi = [min(end[i], sp[i]) for i in range(n)]
            win = x[(slice(None), slice(None)) + tuple(slice(lo[i], hi[i]) for i in range(n))]
            # sequential fp32 accumulation in window row-major order, as the reference loop does
            flat = win.reshape(x.shape[0], x.shape[1], -1)
            acc = np.zeros(x.shape[:2], dtype=F32)
            for j in range(flat.shape[2]):
                acc += flat[:, :, j]
            y[(slice(None), slice(None)) + o] = acc / F32(size)
        else:
            raise ValueError(method)
    return y


def pooling_fast(x, method, kernel, stride, pad) -> np.ndarray:
    """Same semantics as :func:`pooling`, vectorised over output positions (used for
    full-size nets where the per-position Python loop is too slow).  AVE sums taps
    in window row-major order, like the reference."""
    x = np.asarray(x, dtype=F32)
    sp = x.shape[2:]
    n = len(sp)
    out = [pooled_dim(sp[i], kernel[i], stride[i], pad[i]) for i in range(n)]
    # pad so every window is in range; MAX pads with -FLT_MAX, AVE with 0
    hi_pad = [max(0, (out[i] - 1) * stride[i] + kernel[i] - pad[i] - sp[i]) for i in range(n)]
    fill = F32(-FLT_MAX) if method == "MAX" else F32(0)
    xp = np.full(x.shape[:2] + tuple(sp[i] + pad[i] + hi_pad[i] for i in range(n)), fill, dtype=F32)
    xp[(slice(None), slice(None)) + tuple(slice(pad[i], pad[i] + sp[i]) for i in range(n))] = x
    acc = None
    for tap in np.ndindex(*kernel):
        sl = tuple(slice(tap[i], tap[i] + stride[i] * (out[i] - 1) + 1, stride[i]) for i in range(n))
        v = xp[(slice(None), slice(None)) + sl]
        if acc is None:
            acc = v.copy()
        elif method == "MAX":
            np.maximum(acc, v, out=acc)
        else:
            acc += v
    if method == "AVE":
        size = np.ones(out, dtype=F32)
        for i in range(n):
            start = np.arange(out[i]) * stride[i] - pad[i]
            end = np.minimum(start + kernel[i], sp[i] + pad[i])
            shape = [1] * n
            shape[i] = out[i]
            size = size * (end - start).astype(F32).reshape(shape)
        acc /= size
    return acc


# --------------------------------------------------------------------------
# glue layers
# --------------------------------------------------------------------------
def concat(xs, axis=1) -> np.ndarray:
    """``ConcatLayer::Forward_cpu`` (layers/concat_layer.cpp:54-70)."""
    return np.concatenate([np.asarray(x, F32) for x in xs], axis=axis)


def eltwise_sum(xs, coeffs=None) -> np.ndarray:
    """``EltwiseLayer::Forward_cpu`` SUM (layers/eltwise_layer.cpp:66-72):
    ``top = 0; for i: top += coeff[i]*bottom[i]`` (caffe_set + caffe_axpy)."""
    y = np.zeros_like(np.asarray(xs[0], F32))
    coeffs = coeffs or [1.0] * len(xs)
    for c, x in zip(coeffs, xs):
        y += F32(c) * np.asarray(x, F32)
    return y


def permute(x, order) -> np.ndarray:
    """``PermuteLayer`` / ``Permute()`` (layers/permute_layer.cpp:9-26,98-114):
    ``top[..., i_k, ...] = bottom`` with top axis k = bottom axis order[k]."""
    return np.ascontiguousarray(np.transpose(np.asarray(x, F32), order))


def inner_product(x, w, b, axis=1) -> np.ndarray:
    """``InnerProductLayer::Forward_cpu`` (layers/inner_product_layer.cpp:81-93):
    ``Y[M,N] = X[M,K] . W[N,K]^T (+ 1 . b^T)``."""
    x = np.asarray(x, F32)
    M = int(np.prod(x.shape[:axis]))
    y = np.matmul(x.reshape(M, -1), np.asarray(w, F32).reshape(w.shape[0], -1).T)
    if b is not None:
        y += np.matmul(np.ones((M, 1), F32), np.asarray(b, F32).reshape(1, -1))
    return y.reshape(x.shape[:axis] + (w.shape[0],))


def softmax(x, axis=1) -> np.ndarray:
    """``SoftmaxLayer::Forward_cpu`` (layers/softmax_layer.cpp:37-80): subtract the
    per-position max over ``axis``, exp, divide by the sum."""
    x = np.asarray(x, F32)
    e = np.exp(x - x.max(axis=axis, keepdims=True)).astype(F32)
    return (e / e.sum(axis=axis, keepdims=True)).astype(F32)


def accuracy(x, label, top_k=1, axis=1, ignore_label=None) -> np.ndarray:
    """``AccuracyLayer::Forward_cpu`` (layers/accuracy_layer.cpp:46-92): for every (outer, inner) sample
    build (score, class) pairs, ``partial_sort`` them with ``std::greater`` and test whether the label is
    among the first ``top_k``; result = hits / counted samples (0-axis blob)."""
    x = np.asarray(x, F32)
    outer = int(np.prod(x.shape[:axis]))
    c = x.shape[axis]
    inner = int(np.prod(x.shape[axis + 1:]))
    xs = x.reshape(outer, c, inner)
    lab = np.asarray(label).reshape(outer, inner)
    hits = count = 0
    for i in range(outer):
        for j in range(inner):
            lv = int(lab[i, j])
            if ignore_label is not None and lv == ignore_label:
                continue
            pairs = sorted(((float(xs[i, k, j]), k) for k in range(c)), reverse=True)  # greater<pair<Dtype,int>>
            hits += any(k == lv for _, k in pairs[:top_k])
            count += 1
    return np.asarray(np.float32(hits) / np.float32(count), F32)


def softmax_loss(x, label, axis=1, normalize=True, ignore_label=None) -> np.ndarray:
    """``SoftmaxWithLossLayer::Forward_cpu`` (layers/softmax_loss_layer.cpp:52-84): softmax over ``axis``,
    ``loss -= log(max(prob[label], FLT_MIN))`` over the counted samples, divided by the count
    (``normalize``, the default) or by ``outer_num_``."""
    prob = softmax(x, axis)
    outer = int(np.prod(prob.shape[:axis]))
    c = prob.shape[axis]
    inner = int(np.prod(prob.shape[axis + 1:]))
    ps = prob.reshape(outer, c, inner)
    lab = np.asarray(label).reshape(outer, inner)
    loss = np.float32(0)
    count = 0
    for i in range(outer):
        for j in range(inner):
            lv = int(lab[i, j])
            if ignore_label is not None and lv == ignore_label:
                continue
            loss -= np.log(np.maximum(ps[i, lv, j], np.finfo(np.float32).tiny)).astype(F32)
            count += 1
    return np.asarray(loss / np.float32(count if normalize else outer), F32)


# --------------------------------------------------------------------------
# VideoData output contract (TEST phase)
# --------------------------------------------------------------------------
def video_transform(frames_hwc_u8, crop_h, crop_w, h_off, w_off, mean, scale=1.0, mirror=False) -> np.ndarray:
    """``ReadSegmentRGBToDatum`` + ``DataTransformer::Transform`` for uint8 colour frames
    (util/io.cpp:398-408: planar copy ``datum[c][h][w] = img(h,w)[c]``; data_transformer.cpp:258-316:
    ``top[c][h][w or W-1-w] = (datum[c][h_off+h][w_off+w] - mean_values[c]) * scale``)."""
    f = np.asarray(frames_hwc_u8)
    assert f.dtype == np.uint8 and f.ndim == 4 and f.shape[3] == 3
    F = f.shape[0]
    out = np.empty((F, 3, crop_h, crop_w), np.float32)
    for c in range(3):
        plane = f[:, h_off:h_off + crop_h, w_off:w_off + crop_w, c].astype(np.float32)
        if mirror:
            plane = plane[:, :, ::-1]
        out[:, c] = (plane - np.float32(mean[c])) * np.float32(scale)
    return out


# --------------------------------------------------------------------------
# whole-net forward (Net::ForwardFromTo, net.cpp:566-583)
# --------------------------------------------------------------------------
def forward(spec, params, inputs, keep=None, fast_pool=True, timings=None, conv_impl=None, store_hook=None,
            input_hook=None, layer_impl=None):
    """Run ``spec`` (an ``eco_amd.netspec.NetSpec``) layer by layer in file order.

    ``params``: {layer name: [ndarray, ...]} in the reference's blob order
    (conv/fc: weight, bias; BN: scale, shift, running mean, running variance).
    ``inputs``: {input blob name: ndarray}.  Returns {blob name: ndarray} for the
    net outputs plus every name in ``keep`` (``keep='all'`` keeps everything).
    In-place layers overwrite their blob exactly as the reference does.
    ``conv_impl(x, w, b, kernel, stride, pad)`` replaces the NumPy convolution (bench.py's CPU baselines plug in
    the compiled reference im2col + OpenBLAS path of oracle/eco_ref.py, or torch-CPU).
    ``layer_impl``: {layer type: fn(L, bottoms, params_of_layer) -> list of tops or None} replaces the restatement
    of a layer type (None = fall through to it); tests/test_oracle_ref.py runs a whole net through the COMPILED
    reference layer code that way.
    ``store_hook(blob name, array)`` / ``input_hook`` transform a top / an input before it is stored: the
    bf16-storage comparisons round there exactly where the blocked path rounds (tests/test_blocked.py)."""
    import time
    conv_fn = conv_impl or convolution
    blobs = {k: np.ascontiguousarray(v, dtype=F32) for k, v in inputs.items()}
    if input_hook is not None:
        blobs = {k: np.ascontiguousarray(input_hook(k, v), dtype=F32) for k, v in blobs.items()}
    pool_fn = pooling_fast if fast_pool else pooling
    last_use = {}
    for i, L in enumerate(spec.layers):
        for b in L.bottoms:
            last_use[b] = i
    wanted = set(spec.outputs)
    if keep == "all":
        wanted = None
    elif keep:
        wanted |= set(keep)
    for i, L in enumerate(spec.layers):
        t0 = time.perf_counter()
        bt = [blobs[b] for b in L.bottoms]
        g = L.geom
        top = None
        if layer_impl is not None and L.type in layer_impl:
            top = layer_impl[L.type](L, bt, params.get(L.name))
        if top is not None:
            pass
        elif L.type == "Convolution":
            p = params[L.name]
            top = [conv_fn(bt[0], p[0], p[1] if g["bias_term"] else None, g["kernel"], g["stride"], g["pad"])]
        elif L.type == "BN":
            p = params[L.name]
            top = [bn_inference(bt[0], p[0], p[1], p[2], p[3], max(g["eps"], 1e-5) if bt[0].ndim > 4 else g["eps"])]
        elif L.type == "ReLU":
            top = [relu(bt[0], g["negative_slope"])]
        elif L.type == "Pooling":
            top = [pool_fn(bt[0], g["method"], g["kernel"], g["stride"], g["pad"])]
        elif L.type == "Concat":
            top = [concat(bt, g["axis"])]
        elif L.type == "Eltwise":
            top = [eltwise_sum(bt, g["coeff"])]
        elif L.type == "Reshape":
            top = [bt[0].reshape(L.top_shapes[0])]
        elif L.type == "Permute":
            top = [permute(bt[0], g["order"])]
        elif L.type == "Dropout":  # TEST phase: identity copy (layers/dropout_layer.cpp:46-48)
            top = [bt[0]]
        elif L.type == "Split":  # layers/split_layer.cpp:26-32: ShareData
            top = [bt[0] for _ in L.tops]
        elif L.type == "InnerProduct":
            p = params[L.name]
            top = [inner_product(bt[0], p[0], p[1] if g["bias_term"] else None, g["axis"])]
        elif L.type == "Softmax":
            top = [softmax(bt[0], g["axis"])]
        elif L.type == "Accuracy":
            top = [accuracy(bt[0], bt[1], g["top_k"], g["axis"], g["ignore_label"])]
        elif L.type == "SoftmaxWithLoss":
            top = [softmax_loss(bt[0], bt[1], g["axis"], g["normalize"], g["ignore_label"])]
            if len(L.tops) == 2:
                top.append(softmax(bt[0], g["axis"]))
        else:
            raise NotImplementedError(L.type)
        for name, v, shp in zip(L.tops, top, L.top_shapes):
            assert tuple(v.shape) == tuple(shp), (L.name, v.shape, shp)
            blobs[name] = v if store_hook is None else np.ascontiguousarray(store_hook(name, v), dtype=F32)
        if wanted is not None:
            for b in L.bottoms:
                if last_use.get(b) == i and b not in wanted and b not in L.tops:
                    blobs.pop(b, None)
        if timings is not None:
            timings.append((L.name, L.type, time.perf_counter() - t0))
    if wanted is None:
        return blobs
    return {k: v for k, v in blobs.items() if k in wanted}
