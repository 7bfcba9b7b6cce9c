"""Pins oracle/eco_oracle.py against COMPILED REFERENCE CODE: oracle/_ref/libeco_ref.so holds the reference's own
util/im2col.cpp and layers/pooling_layer.cpp, built unmodified behind stand-in headers (oracle/Makefile), plus the
reference's conv forward call sequence over them and OpenBLAS sgemm.  CPU-only; the .so is built in the
authoring container and shipped (git-ignored) to the GPU box."""
import json
import os

import numpy as np
import pytest

import eco_oracle as orc
import eco_ref

pytestmark = pytest.mark.skipif(not eco_ref.available(), reason="oracle/_ref not built and no /root/reference")
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_vectors.json")

IM2COL_CASES = [  # (C, spatial, kernel, stride, pad)
    (3, (6, 4), (3, 3), (2, 2), (0, 0)), (4, (7, 9), (3, 3), (1, 1), (1, 1)), (3, (11, 11), (7, 7), (2, 2), (3, 3)),
    (5, (8, 8), (1, 1), (1, 1), (0, 0)), (3, (5, 6, 4), (3, 3, 3), (2, 2, 2), (0, 0, 0)),
    (4, (4, 7, 7), (3, 3, 3), (1, 1, 1), (1, 1, 1)), (2, (6, 5, 5), (3, 1, 1), (1, 1, 1), (1, 0, 0)),
    (2, (4, 6, 6), (3, 3, 3), (2, 2, 2), (1, 1, 1)),
]


@pytest.mark.parametrize("C,sp,k,s,p", IM2COL_CASES)
def test_im2col_bit_exact(C, sp, k, s, p):
    x = np.random.default_rng(1).normal(size=(C,) + sp).astype(np.float32)
    assert np.array_equal(eco_ref.im2col(x, k, s, p), orc.im2col_nd(x, k, s, p))


@pytest.mark.parametrize("C,sp,k,s,p", IM2COL_CASES)
@pytest.mark.parametrize("image_threads", [1, 3])
def test_conv_forward_matches_oracle(C, sp, k, s, p, image_threads):
    rng = np.random.default_rng(2)
    x = rng.normal(size=(3, C) + sp).astype(np.float32)
    w = rng.normal(size=(6, C) + k).astype(np.float32)
    b = rng.normal(size=6).astype(np.float32)
    got = eco_ref.convolution(x, w, b, k, s, p, image_threads=image_threads)
    ref = orc.convolution(x, w, b, k, s, p)
    assert got.shape == ref.shape and np.abs(got - ref).max() <= 2e-5 * np.abs(ref).max()
    if x.size * w.size < 4e6:
        naive = orc.convolution_naive(x, w, b, k, s, p)       # test_convolution_layer.cpp:18-134 loops
        assert np.abs(got - naive).max() <= 1e-4 * np.abs(naive).max()
    nob = eco_ref.convolution(x, w, None, k, s, p)
    assert np.abs(nob - orc.convolution(x, w, None, k, s, p)).max() <= 2e-5 * np.abs(ref).max()


POOL_CASES = [((2, 3, 6, 5), "MAX", (2, 2), (1, 1), (0, 0)), ((2, 3, 7, 7), "MAX", (3, 3), (2, 2), (0, 0)),
              ((1, 2, 6, 6), "MAX", (3, 3), (2, 2), (1, 1)), ((2, 3, 7, 7), "AVE", (3, 3), (1, 1), (1, 1)),
              ((1, 4, 8, 8), "AVE", (3, 3), (2, 2), (1, 1)), ((2, 3, 7, 7), "AVE", (7, 7), (1, 1), (0, 0)),
              ((1, 3, 28, 28), "MAX", (3, 3), (2, 2), (0, 0)), ((1, 2, 56, 56), "MAX", (3, 3), (2, 2), (0, 0))]


@pytest.mark.parametrize("shape,method,k,s,p", POOL_CASES)
def test_pooling_forward_bit_exact(shape, method, k, s, p):
    x = np.random.default_rng(3).normal(size=shape).astype(np.float32)
    ref = eco_ref.pooling(x, method, k, s, p)
    for fn in (orc.pooling, orc.pooling_fast):
        got = fn(x, method, k, s, p)
        assert got.shape == ref.shape
        if method == "MAX":
            assert np.array_equal(got, ref)
        else:  # same window sums, accumulated in the same (h, w) order by orc.pooling
            assert np.abs(got - ref).max() <= (0 if fn is orc.pooling else 1e-6)


def test_pooled_shape_rule_2d_and_3d():
    """pooling_layer.cpp:117-147 through the compiled LayerSetUp/Reshape, vs the host rule the product uses."""
    from eco_amd.netspec import pooled_dim
    for shape, k, s, p in [((1, 2, 112, 112), (3, 3), (2, 2), (0, 0)), ((1, 2, 56, 56), (3, 3), (2, 2), (0, 0)),
                           ((1, 2, 28, 28), (3, 3), (1, 1), (1, 1)), ((1, 2, 4, 7, 7), (4, 7, 7), (1, 1, 1), (0, 0, 0)),
                           ((1, 2, 8, 7, 7), (8, 7, 7), (1, 1, 1), (0, 0, 0)), ((1, 2, 5, 9, 9), (3, 3, 3), (2, 2, 2), (1, 1, 1)),
                           ((1, 2, 6, 6), (3, 3), (2, 2), (1, 1)), ((1, 2, 7, 7), (2, 2), (2, 2), (0, 0))]:
        got = eco_ref.pooled_shape(shape, "AVE", k, s, p)
        want = tuple(shape[:2]) + tuple(pooled_dim(shape[2 + i], k[i], s[i], p[i]) for i in range(len(k)))
        assert got == want, (shape, k, s, p)
    assert eco_ref.pooled_shape((3, 5, 4, 7, 7), "AVE", None, (1, 1, 1), (0, 0, 0)) == (3, 5, 1, 1, 1)   # global_pooling


def test_reference_golden_vectors_through_compiled_reference():
    """The known answers of the reference's own unit tests (tests/golden/reference_vectors.json), evaluated by the
    reference's own compiled Forward_cpu: the fixture, the oracle and the compiled code agree."""
    cases = json.load(open(GOLD))["cases"]
    seen = 0
    for c in cases:
        if c.get("op") == "pool" and len(c["x_shape"]) == 4:
            x = np.asarray(c["x"], np.float32).reshape(c["x_shape"])
            y = eco_ref.pooling(x, c["method"], c["kernel"], c["stride"], c["pad"])
            assert list(y.shape) == c["y_shape"], c["name"]
            assert np.abs(y.reshape(-1) - np.asarray(c["expected"], np.float32)).max() <= max(c["tol"], 1e-6), c["name"]
            seen += 1
        elif c.get("op") == "pool":  # 3-D: the CPU Forward cannot run it (pooling_layer.cpp:177-201), the shape rule can
            assert list(eco_ref.pooled_shape(c["x_shape"], c["method"], c["kernel"], c["stride"], c["pad"])) == c["y_shape"]
            seen += 1
        elif c.get("op") == "pooled_shape":
            for r in c["rules"]:
                got = eco_ref.pooled_shape([1, 1] + c["in_hw"], "MAX", [r["kernel"]] * 2, [r["stride"]] * 2, [r["pad"]] * 2)
                assert list(got[2:]) == r["out_hw"]
            seen += 1
    assert seen >= 6
