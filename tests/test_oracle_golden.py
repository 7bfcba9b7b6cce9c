"""Pins the CPU oracle (and, for pooling, the HIP kernels on both backends) to the reference's
own known-answer tests.  Vectors: tests/golden/reference_vectors.json, extracted verbatim from
caffe_3d/src/caffe/test/*.cpp by tests/golden/make_golden.py; property tests restate the
reference tests that have no literal vectors (conv vs naive loops, Sobel separability, BN
frozen formula, inner-product lower bound, concat/eltwise/reshape/dropout/relu)."""
import json
import os

import numpy as np
import pytest

import eco_oracle as orc
from eco_amd import hip
from eco_amd.netspec import pooled_dim

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "reference_vectors.json")) as f:
    GOLDEN = {c["name"]: c for c in json.load(f)["cases"]}
POOL_CASES = [c for c in GOLDEN.values() if c["op"] == "pool"]


def test_gemm_known_answer():
    c = GOLDEN["gemm_2x3_3x4"]  # test_util_blas.cpp:22-40
    a = np.array(c["a"], np.float32).reshape(c["a_shape"])
    b = np.array(c["b"], np.float32).reshape(c["b_shape"])
    exp = np.array(c["expected"], np.float32).reshape(2, 4)
    assert np.array_equal(np.matmul(a, b), exp)
    # the oracle's inner_product is X.W^T: feed W = B^T
    assert np.array_equal(orc.inner_product(a, np.ascontiguousarray(b.T), None), exp)
    # and as a 1x1 convolution over a 1x4 "image" with 3 channels
    x = b.reshape(1, 3, 1, 4)
    w = a.reshape(2, 3, 1, 1)
    assert np.array_equal(orc.convolution(x, w, None, (1, 1), (1, 1), (0, 0)).reshape(2, 4), exp)


@pytest.mark.parametrize("case", POOL_CASES, ids=[c["name"] for c in POOL_CASES])
def test_pool_known_answers_oracle(case):
    x = np.array(case["x"], np.float32).reshape(case["x_shape"])
    exp = np.array(case["expected"], np.float32).reshape(case["y_shape"])
    for fn in (orc.pooling, orc.pooling_fast):
        y = fn(x, case["method"], case["kernel"], case["stride"], case["pad"])
        assert y.shape == exp.shape
        assert np.abs(y - exp).max() <= max(case["tol"], 1e-6), (case["name"], case["source"])


@pytest.mark.parametrize("case", POOL_CASES, ids=[c["name"] for c in POOL_CASES])
def test_pool_known_answers_kernels(backend, case):
    """The same reference vectors through the HIP pooling kernel (emulated on CPU, real on GPU)."""
    x = np.tile(np.array(case["x"], np.float32).reshape(case["x_shape"]), (2, 2) + (1,) * (len(case["x_shape"]) - 2))
    exp = np.tile(np.array(case["expected"], np.float32).reshape(case["y_shape"]), (2, 2) + (1,) * (len(case["y_shape"]) - 2))
    g = hip.pool_geom(2, 2, case["x_shape"][2:], case["kernel"], case["stride"], case["pad"], case["y_shape"][2:], case["method"])
    y = backend.empty(exp.shape)
    backend.lib.pool_forward(g, backend.ptr(backend.dev(x)), backend.ptr(y))
    assert np.abs(backend.host(y, exp.shape) - exp).max() <= max(case["tol"], 1e-6)


def test_pooled_shape_rules():
    c = GOLDEN["pooled_shape_rules"]  # test_pooling_layer.cpp:373-403
    for r in c["rules"]:
        got = [pooled_dim(d, r["kernel"], r["stride"], r["pad"]) for d in c["in_hw"]]
        assert got == r["out_hw"]
        assert [orc.pooled_dim(d, r["kernel"], r["stride"], r["pad"]) for d in c["in_hw"]] == r["out_hw"]
    # ECO layers: 112 -> 56 -> 28 (MAX 3x3 s2, ceil), 28 -> 28 (AVE 3x3 s1 p1), ECO-Full 28 -> 14 -> 7
    assert pooled_dim(112, 3, 2, 0) == 56 and pooled_dim(56, 3, 2, 0) == 28 and pooled_dim(28, 3, 1, 1) == 28
    assert pooled_dim(28, 3, 2, 0) == 14 and pooled_dim(14, 3, 2, 0) == 7 and pooled_dim(7, 3, 1, 1) == 7


# reference shapes: TestSimpleConvolution [2,3,6,4] k3 s2 (:226-260), TestSimple3DConvolution
# [2,3,5,6,4] k3 s2 (:300-345), Test1x1Convolution (:348-373); all num_output 4, bias constant 0.1
@pytest.mark.parametrize("insp,k,s,p", [((6, 4), (3, 3), (2, 2), (0, 0)), ((5, 6, 4), (3, 3, 3), (2, 2, 2), (0, 0, 0)),
                                         ((6, 4), (1, 1), (1, 1), (0, 0)), ((6, 4), (3, 3), (1, 1), (1, 1)),
                                         ((4, 5, 4), (3, 3, 3), (1, 1, 1), (1, 1, 1))])
def test_conv_vs_reference_naive_loops(insp, k, s, p):
    rng = np.random.default_rng(0)
    x = rng.standard_normal((2, 3) + insp).astype(np.float32)
    w = rng.standard_normal((4, 3) + k).astype(np.float32)
    b = np.full(4, 0.1, np.float32)
    a = orc.convolution(x, w, b, k, s, p)
    n = orc.convolution_naive(x, w, b, k, s, p)
    assert a.shape == n.shape
    assert np.abs(a - n).max() < 1e-4  # the reference's own tolerance


def test_conv_sobel_separability():
    """test_convolution_layer.cpp:403-494: a 3x3 Sobel filter equals the (3x1) then (1x3) pair."""
    rng = np.random.default_rng(1)
    x = rng.standard_normal((2, 1, 7, 6)).astype(np.float32)
    sob = np.array([[-1, 0, 1], [-2, 0, 2], [-1, 0, 1]], np.float32).reshape(1, 1, 3, 3)
    full = orc.convolution(x, sob, None, (3, 3), (1, 1), (0, 0))
    col = np.array([1, 2, 1], np.float32).reshape(1, 1, 3, 1)
    row = np.array([-1, 0, 1], np.float32).reshape(1, 1, 1, 3)
    sep = orc.convolution(orc.convolution(x, col, None, (3, 1), (1, 1), (0, 0)), row, None, (1, 3), (1, 1), (0, 0))
    assert np.abs(full - sep).max() < 1e-4


def test_conv_nd_matches_2d():
    """TestNDAgainst2D (:496-613): the N-D lowering on a [*,*,1,H,W] blob equals the 2-D one."""
    rng = np.random.default_rng(2)
    x = rng.standard_normal((2, 3, 6, 5)).astype(np.float32)
    w = rng.standard_normal((4, 3, 3, 3)).astype(np.float32)
    b = rng.standard_normal(4).astype(np.float32)
    y2 = orc.convolution(x, w, b, (3, 3), (2, 2), (1, 1))
    y3 = orc.convolution(x[:, :, None], w[:, :, None], b, (1, 3, 3), (1, 2, 2), (0, 1, 1))
    assert np.array_equal(y2, y3[:, :, 0])


def test_bn_frozen_formula():
    """test_bn_layer.cpp:107-154: mean c, var c+1, eps 0, slope 1, bias 0 -> (x - c)/sqrt(c+1)."""
    rng = np.random.default_rng(3)
    for shape in [(5, 2, 3, 4), (2, 3, 2, 3, 4)]:
        x = rng.standard_normal(shape).astype(np.float32)
        C = shape[1]
        y = orc.bn_inference(x, np.ones(C), np.zeros(C), np.arange(C), np.arange(C) + 1.0, 0.0)
        for c in range(C):
            assert np.abs(y[:, c] - (x[:, c] - c) / np.sqrt(c + 1)).max() < 1e-3


def test_glue_layers():
    rng = np.random.default_rng(4)
    a = rng.standard_normal((2, 3, 6, 5)).astype(np.float32)
    b = rng.standard_normal((2, 5, 6, 5)).astype(np.float32)
    c = orc.concat([a, b], 1)  # test_concat_layer.cpp:102-155
    assert c.shape == (2, 8, 6, 5) and np.array_equal(c[:, :3], a) and np.array_equal(c[:, 3:], b)
    e = orc.eltwise_sum([a, a, a])  # test_eltwise_layer.cpp:87-105
    assert np.allclose(e, 3 * a, rtol=1e-6)
    e = orc.eltwise_sum([a, a], [1.0, -0.5])  # :107-127 (coefficients)
    assert np.allclose(e, 0.5 * a, rtol=1e-6)
    r = orc.relu(a)  # test_neuron_layer.cpp:190-203
    assert (r >= 0).all() and np.array_equal(r[a > 0], a[a > 0]) and (r[a <= 0] == 0).all()
    r = orc.relu(a, 0.01)  # :205-220
    assert np.allclose(r[a <= 0], 0.01 * a[a <= 0])
    # inner product with bias in [1,2] and non-negative inputs/weights >= 1 (test_inner_product_layer.cpp:58-86)
    x = rng.uniform(0, 1, (2, 60)).astype(np.float32)
    w = rng.uniform(0, 1, (10, 60)).astype(np.float32)
    bias = rng.uniform(1, 2, 10).astype(np.float32)
    assert (orc.inner_product(x, w, bias) >= 1.0).all()
    # permute = numpy transpose semantics; permuting back is the identity
    z = rng.standard_normal((2, 4, 3, 2, 5)).astype(np.float32)
    assert np.array_equal(orc.permute(orc.permute(z, [0, 2, 1, 3, 4]), [0, 2, 1, 3, 4]), z)
    assert orc.permute(z, [0, 2, 1, 3, 4])[1, 2, 3, 1, 4] == z[1, 3, 2, 1, 4]
    s = orc.softmax(rng.standard_normal((3, 7, 2)).astype(np.float32), 1)
    assert np.allclose(s.sum(1), 1.0, atol=1e-6)
