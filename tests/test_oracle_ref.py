"""Pins oracle/eco_oracle.py against COMPILED REFERENCE CODE: oracle/_ref/libeco_ref.so holds the reference's own
util/im2col.cpp and layers/{pooling,bn,permute,eltwise,concat,inner_product,reshape,relu}_layer.cpp, built unmodified
behind stand-in headers (oracle/Makefile), plus the reference's conv forward call sequence over them and OpenBLAS
sgemm.  Every layer type of the deploy graphs is covered except the two the reference's own CPU code cannot run
(5-D BN: bn_layer.cpp reads num/channels/height/width through LegacyShape, which CHECK-fails beyond 4 axes; N-D
pooling: pooling_layer.cpp:177-201 is 2-D only) -- those keep the cuDNN-rule restatement and the torch cross-check
(tests/test_oracle_torch.py).  CPU-only; the .so is built in the authoring container and shipped (git-ignored) to
the GPU box."""
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


# ---- the remaining layer types, through their compiled Forward_cpu ---------------------------------------------------
layers_built = pytest.mark.skipif(not (eco_ref.available() and eco_ref.has_layers()),
                                  reason="oracle/_ref built without the layer files")


@layers_built
@pytest.mark.parametrize("shape", [(2, 5, 6, 7), (3, 64, 1, 1), (1, 8, 14, 14), (4, 3), (2, 6, 9)])
@pytest.mark.parametrize("eps", [1e-5, 1e-3])
def test_bn_test_phase_bit_exact(shape, eps):
    """bn_layer.cpp:93-207, TEST branch: x + (-mean) broadcast by sgemm, * powf(var + eps, -0.5), * slope, + bias."""
    rng = np.random.default_rng(len(shape) * 10 + shape[1])
    c = shape[1]
    x = (rng.normal(size=shape) * 10).astype(np.float32)
    slope, bias, mean = (rng.normal(size=c).astype(np.float32) for _ in range(3))
    var = rng.uniform(1e-4, 3, c).astype(np.float32)
    ref = eco_ref.bn_inference(x, slope, bias, mean, var, eps)
    assert np.array_equal(orc.bn_inference(x, slope, bias, mean, var, eps), ref)
    # frozen (bn_param.frozen) takes the same branch in TRAIN and TEST
    assert np.array_equal(eco_ref.bn_inference(x, slope, bias, mean, var, eps, frozen=True), ref)


@layers_built
def test_bn_5d_is_the_4d_arithmetic_per_channel():
    """The reference cannot run 5-D BN on the CPU (cuDNN only, eps clamped to CUDNN_BN_MIN_EPSILON); per channel it is
    the same map, so the 5-D blob folded to [n, c, d*h, w] through the compiled 4-D code equals the oracle's N-D
    restatement at the same eps."""
    rng = np.random.default_rng(5)
    x = rng.normal(size=(2, 6, 4, 7, 7)).astype(np.float32)
    slope, bias, mean = (rng.normal(size=6).astype(np.float32) for _ in range(3))
    var = rng.uniform(0.1, 2, 6).astype(np.float32)
    assert np.array_equal(eco_ref.bn_inference(x, slope, bias, mean, var, 1e-5), orc.bn_inference(x, slope, bias, mean, var, 1e-5))


@layers_built
@pytest.mark.parametrize("shape,order", [((2, 4, 3, 5, 6), (0, 2, 1, 3, 4)), ((2, 3, 4, 5), (0, 2, 3, 1)), ((3, 4, 5), (2, 0, 1)),
                                         ((2, 3, 4, 5), (0, 1, 2, 3)), ((2, 3, 4, 5, 6), (0, 2))])
def test_permute_bit_exact(shape, order):
    """permute_layer.cpp:9-26,98-114 (an order shorter than the blob is completed with the remaining axes, :44-48)."""
    x = np.random.default_rng(7).normal(size=shape).astype(np.float32)
    full = list(order) + [a for a in range(len(shape)) if a not in order]
    assert np.array_equal(eco_ref.permute(x, order), orc.permute(x, full))


@layers_built
@pytest.mark.parametrize("n", [2, 3])
def test_eltwise_sum_bit_exact(n):
    rng = np.random.default_rng(n)
    xs = [rng.normal(size=(2, 8, 4, 7, 7)).astype(np.float32) for _ in range(n)]
    assert np.array_equal(eco_ref.eltwise(xs), orc.eltwise_sum(xs))
    cf = [1.0, -0.5, 2.0][:n]
    assert np.array_equal(eco_ref.eltwise(xs, coeffs=cf), orc.eltwise_sum(xs, cf))


@layers_built
@pytest.mark.parametrize("axis", [0, 1, 2])
def test_concat_bit_exact(axis):
    rng = np.random.default_rng(axis)
    shapes = [[2, 3, 4, 5], [2, 3, 4, 5], [2, 3, 4, 5]]
    shapes[1][axis] += 2
    shapes[2][axis] = 1
    xs = [rng.normal(size=s).astype(np.float32) for s in shapes]
    assert np.array_equal(eco_ref.concat(xs, axis), orc.concat(xs, axis))


@layers_built
@pytest.mark.parametrize("shape,n_out,bias", [((4, 512), 400, True), ((2, 32, 1, 1, 1), 10, True), ((3, 6, 5, 5), 7, False)])
def test_inner_product_matches(shape, n_out, bias):
    """inner_product_layer.cpp:81-93: sgemm(NoTrans, Trans) + the bias as a rank-1 sgemm; the oracle's np.matmul is the
    same OpenBLAS family, summation order may differ."""
    rng = np.random.default_rng(n_out)
    x = rng.normal(size=shape).astype(np.float32)
    w = rng.normal(size=(n_out, int(np.prod(shape[1:])))).astype(np.float32)
    b = rng.normal(size=n_out).astype(np.float32) if bias else None
    got, ref = eco_ref.inner_product(x, w, b), orc.inner_product(x, w, b)
    assert got.shape == ref.shape and np.abs(got - ref).max() <= 2e-6 * np.abs(ref).max()


@layers_built
def test_relu_bit_exact():
    x = np.random.default_rng(1).normal(size=(3, 5, 7)).astype(np.float32)
    for slope in (0.0, 0.1):
        assert np.array_equal(eco_ref.relu(x, slope), orc.relu(x, slope))


@layers_built
def test_reshape_rules_through_compiled_layer():
    """reshape_layer.cpp:9-90 (0 = copy, -1 = infer, axis / num_axes windows) against the host rule the product uses
    (eco_amd.netspec), on the ECO reshapes and a few corner cases."""
    from eco_amd.netspec import NetSpec
    from eco_amd import models
    spec = NetSpec.from_prototxt(models.eco_full_deploy(num_segments=8, num_clips=3))
    seen = 0
    for L in spec.layers:
        if L.type != "Reshape":
            continue
        p = L.param.msg("reshape_param")
        dims = [int(d) for d in p.msg("shape").getall("dim")]
        got = eco_ref.reshape_shape(L.bottom_shapes[0], dims, int(p.get("axis", 0)), int(p.get("num_axes", -1)))
        assert got == tuple(L.top_shapes[0]), L.name
        seen += 1
    assert seen >= 3
    assert eco_ref.reshape_shape((2, 3, 4, 5), [0, -1]) == (2, 60)
    assert eco_ref.reshape_shape((2, 3, 4, 5), [6, -1], axis=1, num_axes=2) == (2, 6, 2, 5)
    assert eco_ref.reshape_shape((2, 3, 4, 5), [1, 1], axis=-1, num_axes=0) == (2, 3, 4, 5, 1, 1)


@layers_built
def test_reduced_eco_lite_layer_by_layer_through_compiled_reference_code():
    """One reduced ECO-Lite net (same graph, 32x32 frames, channels / 8) run layer by layer through COMPILED reference
    code only -- conv (the compiled ConvolutionLayer class over OpenBLAS sgemm), BN, ReLU, 2-D pooling, Concat, Eltwise, Permute,
    InnerProduct, Reshape's shape rule -- against the NumPy restatement.  Two layer kinds cannot be executed by the
    reference's CPU code and are evaluated as documented: 5-D BN through the 4-D code on the folded blob with the
    cuDNN eps rule, the 3-D global AVE pool by the oracle (its 2-D golden vectors are pinned above)."""
    from eco_amd import fillers, models
    from eco_amd.netspec import NetSpec
    spec = NetSpec.from_prototxt(models.eco_lite_deploy(num_segments=4, num_clips=2, num_classes=10, input_size=32, width_div=8))
    params = fillers.synthetic_params(spec, seed=7)
    x = fillers.synthetic_frames(8, 32, 32, seed=3)
    used = {}

    def count(t):
        used[t] = used.get(t, 0) + 1

    def conv(L, bt, p):
        count("Convolution")
        g = L.geom                                                                  # the compiled ConvolutionLayer class
        return [eco_ref.convolution_layer(bt[0], p[0], p[1] if g["bias_term"] else None, g["kernel"], g["stride"], g["pad"])]

    def bn(L, bt, p):
        count("BN")
        eps = max(L.geom["eps"], 1e-5) if bt[0].ndim > 4 else L.geom["eps"]      # cudnn_bn_layer.cu:24 for 5-D blobs
        return [eco_ref.bn_inference(bt[0], p[0], p[1], p[2], p[3], eps)]

    def pool(L, bt, p):
        if bt[0].ndim != 4:
            return None                                                             # N-D: not in the reference's CPU code
        count("Pooling")
        g = L.geom
        return [eco_ref.pooling(bt[0], g["method"], g["kernel"], g["stride"], g["pad"])]

    def reshape(L, bt, p):
        count("Reshape")
        rp = L.param.msg("reshape_param")
        dims = [int(d) for d in rp.msg("shape").getall("dim")]
        return [bt[0].reshape(eco_ref.reshape_shape(bt[0].shape, dims, int(rp.get("axis", 0)), int(rp.get("num_axes", -1))))]

    impl = {"Convolution": conv, "BN": bn, "Pooling": pool, "Reshape": reshape,
            "ReLU": lambda L, bt, p: (count("ReLU"), [eco_ref.relu(bt[0], L.geom["negative_slope"])])[1],
            "Concat": lambda L, bt, p: (count("Concat"), [eco_ref.concat(bt, L.geom["axis"])])[1],
            "Eltwise": lambda L, bt, p: (count("Eltwise"), [eco_ref.eltwise(bt, coeffs=L.geom["coeff"])])[1],
            "Permute": lambda L, bt, p: (count("Permute"), [eco_ref.permute(bt[0], L.geom["order"])])[1],
            "InnerProduct": lambda L, bt, p: (count("InnerProduct"), [eco_ref.inner_product(
                bt[0], p[0], p[1] if L.geom["bias_term"] else None, L.geom["axis"])])[1]}
    got = orc.forward(spec, params, {"data": x}, keep="all", layer_impl=impl)
    ref = orc.forward(spec, params, {"data": x}, keep="all")
    assert set(got) == set(ref)
    for name in ref:
        assert got[name].shape == ref[name].shape
        assert np.abs(got[name] - ref[name]).max() <= 2e-5 * max(np.abs(ref[name]).max(), 1e-6), name
    for t in ("Convolution", "BN", "ReLU", "Pooling", "Concat", "Eltwise", "Permute", "InnerProduct", "Reshape"):
        assert used.get(t, 0) >= 1, t
    assert used["Convolution"] == sum(L.type == "Convolution" for L in spec.layers)
    assert used["BN"] == sum(L.type == "BN" for L in spec.layers)


@layers_built
def test_compiled_convolution_layer_class():
    """base_conv_layer.cpp + conv_layer.cpp compiled unmodified (round 4): the class's LayerSetUp / Reshape / Forward_cpu
    against (a) ref_conv_forward, the restated call sequence bench.py's image-parallel CPU baseline uses -- bit-identical,
    same sgemm calls -- and (b) the NumPy oracle; prototxt-style repeated fields (one value for every axis, empty stride /
    pad = schema defaults), bias_term false, force_nd_im2col on a 2-D blob, the ECO geometries incl. the stem and a
    strided 3-D conv."""
    if not eco_ref.has_conv_layer():
        pytest.skip("oracle/_ref built before round 4")
    rng = np.random.default_rng(5)
    cases = [((2, 5, 9, 11), (3, 3), (1, 1), (1, 1), 7), ((2, 3, 20, 20), (7,), (2,), (3,), 8),
             ((2, 6, 4, 8, 8), (3, 3, 3), (2, 2, 2), (1, 1, 1), 8), ((2, 6, 4, 8, 8), (1,), (), (), 5),
             ((1, 4, 3, 6, 6), (3,), (1,), (1,), 4), ((3, 8, 7, 7), (1, 1), (), (), 16)]
    for shp, k, s, p, cout in cases:
        nsp = len(shp) - 2
        kk = tuple(k) * nsp if len(k) == 1 else tuple(k)
        ss = (tuple(s) * nsp if len(s) == 1 else tuple(s)) or (1,) * nsp
        pp = (tuple(p) * nsp if len(p) == 1 else tuple(p)) or (0,) * nsp
        x = rng.normal(size=shp).astype(np.float32)
        w = rng.normal(size=(cout, shp[1]) + kk).astype(np.float32)
        b = rng.normal(size=cout).astype(np.float32)
        for bias in (b, None):
            got = eco_ref.convolution_layer(x, w, bias, k, s, p)
            assert np.array_equal(got, eco_ref.convolution(x, w, bias, kk, ss, pp)), (shp, k)
            ref = orc.convolution(x, w, bias, kk, ss, pp)
            assert got.shape == ref.shape and np.abs(got - ref).max() <= 2e-6 * np.abs(ref).max()
        if nsp == 2:   # force_nd_im2col: the N-D im2col on a 2-D blob gives the same col buffer
            assert np.array_equal(eco_ref.convolution_layer(x, w, b, k, s, p, force_nd_im2col=True),
                                  eco_ref.convolution_layer(x, w, b, k, s, p))
