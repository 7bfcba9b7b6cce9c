"""Whole-graph parity: geometrically-reduced ECO-Lite / ECO-Full nets (same topology and layer
names as the reference prototxts, 32x32 frames, channels / 8) run through the pycaffe-style
``Net`` on both backends and compared blob-by-blob with the CPU oracle; plus the drop-in API
behaviours of ``Net`` / ``Blob`` (caffe_3d/python/caffe/pycaffe.py, _caffe.cpp)."""
import numpy as np
import pytest

import eco_oracle as orc
from eco_amd import fillers, models
from eco_amd.net import Net
from eco_amd.netspec import NetSpec, NetSpecError

TOL = 2e-5


def mini(variant, num_segments=4, num_clips=2, **kw):
    gen = models.eco_lite_deploy if variant == "lite" else models.eco_full_deploy
    return gen(num_segments=num_segments, num_clips=num_clips, num_classes=10, input_size=32, width_div=8, **kw)


def make_net(backend, proto, params, fuse, **kw):
    if backend.kind == "emu":
        return Net(proto, params=params, fuse=fuse, _backend=(backend.lib, backend.alloc), **kw)
    return Net(proto, params=params, fuse=fuse, **kw)


def relerr(got, ref):
    return float(np.abs(got - ref).max() / (np.abs(ref).max() + 1e-30))


@pytest.mark.parametrize("variant", ["lite", "full"])
@pytest.mark.parametrize("fuse", [False, True])
def test_mini_eco_matches_oracle(backend, variant, fuse):
    proto = mini(variant)
    spec = NetSpec.from_prototxt(proto)
    params = fillers.synthetic_params(spec, seed=7)
    x = fillers.synthetic_frames(8, 32, 32, seed=3)
    ref = orc.forward(spec, params, {"data": x}, keep="all", fast_pool=False)
    # winograd=4 forces the F(4x4,3x3) route on the res5 convs (64 channels here); the default size rule would
    # evaluate a two-clip 32x32 net directly
    net = make_net(backend, proto, params, fuse, winograd=4)
    net.blobs["data"].data[...] = x
    out = net.forward()
    assert list(out) == ["fc8"] and out["fc8"].shape == (2, 10)
    assert relerr(out["fc8"], ref["fc8"]) < TOL
    seen = 0
    for name in net.blobs:
        if name in net._engine.tensors:
            got = net.blobs[name].data
            assert relerr(got, ref[name].reshape(got.shape)) < TOL, name
            seen += 1
        else:
            assert fuse, name
            with pytest.raises(KeyError, match="fuse=False"):
                net.blobs[name].data
    if fuse:
        assert seen < len(net.blobs)
        n_conv = sum(L.type == "Convolution" for L in spec.layers)
        # every BN/ReLU/Eltwise and the Lite Concat/Permute copies are gone
        assert len([l for l in net.op_labels() if "winograd input" not in l and "transformed" not in l]) < n_conv + 25
    else:
        assert seen == len(net.blobs)


@pytest.mark.parametrize("fuse", [False, True])
def test_test_phase_evaluator_net(backend, fuse):
    """The TEST phase of the train/val prototxts (ECO_Lite.prototxt:66-179,1883-1923): VideoData tops become
    the inputs ``data`` [B,3N,H,W] and ``label`` [B,1,1,1]; outputs ``loss`` / ``top1`` / ``top5``."""
    B, N, C = 6, 4, 10
    proto = models.test_phase_net(mini("lite", num_segments=N, num_clips=B), N, batch_size=B, input_size=32)
    spec = NetSpec.from_prototxt(proto)
    assert spec.inputs == ["data", "label"] and sorted(spec.outputs) == ["loss", "top1", "top5"]
    assert spec.input_shapes == {"data": (B, 3 * N, 32, 32), "label": (B, 1, 1, 1)}
    params = fillers.synthetic_params(spec, seed=11)
    x = fillers.synthetic_frames(B * N, 32, 32, seed=5).reshape(B, 3 * N, 32, 32)
    # labels: take the oracle's own ranking so that top1 / top5 are neither 0 nor 1
    logits = orc.forward(spec, params, {"data": x, "label": np.zeros((B, 1, 1, 1), np.float32)}, keep="all",
                         fast_pool=False)["fc8"]
    order = np.argsort(-logits, axis=1)
    label = np.array([order[0, 0], order[1, 0], order[2, 3], order[3, 4], order[4, 6], order[5, 9]],
                     np.float32).reshape(B, 1, 1, 1)
    ref = orc.forward(spec, params, {"data": x, "label": label}, keep="all", fast_pool=False)
    assert abs(float(ref["top1"]) - 2 / 6) < 1e-6 and abs(float(ref["top5"]) - 4 / 6) < 1e-6
    net = make_net(backend, proto, params, fuse)
    net.blobs["data"].data[...] = x
    net.blobs["label"].data[...] = label
    out = net.forward()
    assert sorted(out) == ["loss", "top1", "top5"] and out["top1"].shape == ()
    assert float(out["top1"]) == float(ref["top1"]) and float(out["top5"]) == float(ref["top5"])
    assert abs(float(out["loss"]) - float(ref["loss"])) < 1e-4 * abs(float(ref["loss"]))


def test_fused_plan_structure(backend):
    """The MI355X plan for ECO-Lite: 32 conv launches carry every BN/ReLU/Eltwise/Concat/Permute."""
    proto = mini("lite")
    spec = NetSpec.from_prototxt(proto)
    net = make_net(backend, proto, fillers.synthetic_params(spec), True, winograd=4)
    labels = net.op_labels()
    # 32 convs (30 launches: since round 6 a residual block's strided first conv and its projection shortcut are one launch by
    # default, engine.sibling_blocks) + 4 pools + fused tail; the three res5 stride-1 convs (64 channels in this reduced net) take
    # the Winograd route: input transform + 16 batched (3,1,1) convs + output transform with the fused epilogue
    assert len(labels) == 35 + 2 * 3
    wino = [l for l in labels if "winograd" in l or "transformed" in l]
    assert len(wino) == 9 and "res5b_2+res5b+res5b_bn+res5b_relu [winograd F(4x4,3x3) output transform]" in wino
    for wg in (False, True):   # True = size rule: two clips of a 32x32 net are too small for the Winograd route
        direct = make_net(backend, proto, fillers.synthetic_params(spec), True, winograd=wg)
        assert len(direct.op_labels()) == 35 and "res5b_2+res5b+res5b_bn+res5b_relu" in direct.op_labels()
    assert "res3b_2+res3b+res3b_bn+res3b_relu" in labels
    # res4a_1 and res4a_down read res3b's output with one geometry: one launch, the shortcut keeps its raw value and the Eltwise
    # rides on the later operand, res4a_2
    assert "res4a_1+res4a_1_bn+res4a_1_relu | res4a_down" in labels and "res5a_1+res5a_1_bn+res5a_1_relu | res5a_down" in labels
    assert any(l.startswith("res4a_2+res4a+res4a_bn+res4a_relu") for l in labels)
    assert "inception_3c_double_3x3_1+inception_3c_double_3x3_1_bn+inception_3c_relu_double_3x3_1_inp" in labels
    assert labels[-1] == "global_pool+fc8"
    # opt-out: two launches per block, the Eltwise on the shortcut (the later producer in layer order)
    two = make_net(backend, proto, fillers.synthetic_params(spec), True, winograd=False)
    two._engine.sibling_blocks = False
    two._engine.build()
    assert len(two.op_labels()) == 37 and "res4a_down+res4a+res4a_bn+res4a_relu" in two.op_labels()
    assert "res4a_2" in two.op_labels()
    fa = net._engine.fused_away
    assert "res2b_bn_pre" in fa and "inception_3a_1x1_bn" in fa and "res3b_2" in fa
    for keep in ("res3a", "res4a", "res5a", "res2b_bn", "inception_3a_output", "fc8"):
        assert keep in net._engine.tensors, keep  # dual outputs / concat tops / permuted volume exist


def test_winograd_route_for_wide_2d_convs(backend):
    """2-D 3x3 stride-1 convs with cin >= 128 (ECO-Full's inception 4x/5x stream) take the Winograd route too:
    input transform, T*T batched 1x1 convs, output transform carrying BN + ReLU and a Concat-slice store."""
    proto = """name: "wide2d"
input: "data" input_dim: 3 input_dim: 128 input_dim: 9 input_dim: 10
layer { name: "c1" type: "Convolution" bottom: "data" top: "c1" convolution_param { num_output: 144 kernel_size: 3 pad: 1 } }
layer { name: "c1_bn" type: "BN" bottom: "c1" top: "c1_bn" bn_param { frozen: true } }
layer { name: "c1_relu" type: "ReLU" bottom: "c1_bn" top: "c1_bn" }
layer { name: "c2" type: "Convolution" bottom: "data" top: "c2" convolution_param { num_output: 32 kernel_size: 1 } }
layer { name: "cat" type: "Concat" bottom: "c1_bn" bottom: "c2" top: "cat" }
"""
    spec = NetSpec.from_prototxt(proto)
    params = fillers.synthetic_params(spec, seed=5)
    x = np.random.default_rng(1).standard_normal((3, 128, 9, 10)).astype(np.float32)
    ref = orc.forward(spec, params, {"data": x}, keep="all", fast_pool=False)
    for wg, tol in ((4, 2e-4), (2, TOL), (False, TOL)):
        net = make_net(backend, proto, params, True, winograd=wg)
        labels = net.op_labels()
        assert any("winograd" in l for l in labels) == bool(wg), labels
        net.blobs["data"].data[...] = x
        out = net.forward()["cat"]
        assert out.shape == (3, 176, 9, 10) and relerr(out, ref["cat"]) < tol


def test_fused_winograd_route_for_short_reduction_2d_convs(backend):
    """2-D 3x3 stride-1 convs with 64..128 input channels and cout % 32 == 0 (conv2_3x3, the inception 3x3 convs) run as
    two launches: input transform into the V4 layout, then GEMM + output transform fused (the transformed products
    stay in LDS).  BN + ReLU and the Concat-slice store ride in its epilogue; a weight update repacks its operand;
    wfused=False falls back to the three-launch form with the same result."""
    proto = """name: "short2d"
input: "data" input_dim: 2 input_dim: 96 input_dim: 10 input_dim: 9
layer { name: "c1" type: "Convolution" bottom: "data" top: "c1" convolution_param { num_output: 64 kernel_size: 3 pad: 1 } }
layer { name: "c1_bn" type: "BN" bottom: "c1" top: "c1_bn" bn_param { frozen: true } }
layer { name: "c1_relu" type: "ReLU" bottom: "c1_bn" top: "c1_bn" }
layer { name: "c2" type: "Convolution" bottom: "data" top: "c2" convolution_param { num_output: 32 kernel_size: 1 } }
layer { name: "cat" type: "Concat" bottom: "c1_bn" bottom: "c2" top: "cat" }
"""
    spec = NetSpec.from_prototxt(proto)
    params = fillers.synthetic_params(spec, seed=5)
    x = np.random.default_rng(1).standard_normal((2, 96, 10, 9)).astype(np.float32)
    ref = orc.forward(spec, params, {"data": x}, keep="all", fast_pool=False)
    net = make_net(backend, proto, params, True, winograd=4)
    labels = net.op_labels()
    assert sum("winograd" in l for l in labels) == 2 and any("+ winograd F(4x4,3x3) output transform]" in l for l in labels), labels
    net.blobs["data"].data[...] = x
    out = net.forward()["cat"]
    assert out.shape == (2, 96, 10, 9) and relerr(out, ref["cat"]) < 2e-4
    net.params["c1"][0].data[...] *= -0.5
    params["c1"] = [np.array(b.data) for b in net.params["c1"]]
    ref2 = orc.forward(spec, params, {"data": x}, keep="all", fast_pool=False)
    assert relerr(net.forward()["cat"], ref2["cat"]) < 2e-4 and relerr(ref2["cat"], ref["cat"]) > 1e-3
    net._engine.wfused = False
    net._engine.build()
    assert sum("winograd" in l for l in net.op_labels()) == 2 and sum("transformed-domain" in l for l in net.op_labels()) == 1
    net.blobs["data"].data[...] = x
    assert relerr(net.forward()["cat"], ref2["cat"]) < 2e-4


def test_fused_winograd_conv_absorbs_the_max_pooling_behind_it(backend):
    """conv2_3x3 -> BN -> ReLU -> pool2 (MAX 3x3 / 2, unpadded) on planes that tile by 4: the fused Winograd kernel stores
    partial window maxima and a second launch finishes the pooling -- two launches behind the input transform, the conv's
    own blob is gone; planes that do NOT tile by 4, a second consumer of the conv blob, or wpool=False keep the pooling
    layer as its own launch.  All against the oracle."""
    def proto_for(H, W, extra=""):
        return f"""name: "convpool"
input: "data" input_dim: 2 input_dim: 64 input_dim: {H} input_dim: {W}
layer {{ name: "c1" type: "Convolution" bottom: "data" top: "c1" convolution_param {{ num_output: 96 kernel_size: 3 pad: 1 }} }}
layer {{ name: "c1_bn" type: "BN" bottom: "c1" top: "c1_bn" bn_param {{ frozen: true }} }}
layer {{ name: "c1_relu" type: "ReLU" bottom: "c1_bn" top: "c1_bn" }}
layer {{ name: "p1" type: "Pooling" bottom: "c1_bn" top: "p1" pooling_param {{ pool: MAX kernel_size: 3 stride: 2 }} }}
{extra}"""
    rng = np.random.default_rng(3)
    for H, W, extra, fused in ((12, 16, "", True), (10, 14, "", False),
                               (12, 16, 'layer { name: "r" type: "ReLU" bottom: "c1_bn" top: "r" }', False)):
        proto = proto_for(H, W, extra)
        spec = NetSpec.from_prototxt(proto)
        params = fillers.synthetic_params(spec, seed=5)
        x = rng.standard_normal((2, 64, H, W)).astype(np.float32)
        ref = orc.forward(spec, params, {"data": x}, keep="all", fast_pool=False)
        net = make_net(backend, proto, params, True, winograd=4)
        labels = net.op_labels()
        assert any("partial window maxima" in l for l in labels) == fused, labels
        assert len(labels) == (2 if fused else 3 + bool(extra)), labels
        net.blobs["data"].data[...] = x
        out = net.forward()
        assert relerr(out["p1"], ref["p1"]) < 2e-4
        if fused:
            assert "c1_bn" in net._engine.fused_away and "c1_bn" not in net._engine.tensors
            net._engine.wpool = False
            net._engine.build()
            assert len(net.op_labels()) == 3 and not any("partial window maxima" in l for l in net.op_labels())
            net.blobs["data"].data[...] = x
            assert relerr(net.forward()["p1"], ref["p1"]) < 2e-4


def test_trunk_takes_the_3d_winograd_route_where_depth_tiles_by_four(backend):
    """num_segments = 16: the res5 stage has 4 planes -> F(4x4x4,3x3x3) (csrc/eco_wino3.hip) on its three stride-1 convs
    (the only stage wide enough in this reduced net), residual + BN + ReLU + raw epilogue included; wino3 = False and
    num_segments = 4 (one plane) keep the F(4x4,3x3) + direct-depth-taps route.  All against the oracle."""
    proto = mini("lite", num_segments=16, num_clips=2)
    spec = NetSpec.from_prototxt(proto)
    params = fillers.synthetic_params(spec, seed=11)
    x = fillers.synthetic_frames(32, 32, 32, seed=5)
    ref = orc.forward(spec, params, {"data": x}, keep="all", fast_pool=False)
    net = make_net(backend, proto, params, True, winograd=4)
    labels = net.op_labels()
    w3 = [l for l in labels if "F(4x4x4,3x3x3)" in l]
    assert len(w3) == 6 and sum("216 transformed-domain GEMMs, K = 64" in l for l in labels) == 3, labels
    assert "res5b_2+res5b+res5b_bn+res5b_relu [winograd F(4x4x4,3x3x3) output transform]" in w3
    net.blobs["data"].data[...] = x
    out = net.forward()["fc8"].copy()
    assert relerr(out, ref["fc8"]) < TOL
    for name in ("res5a", "res5b_bn", "res5b_1_bn"):
        if name in net._engine.tensors:
            got = net.blobs[name].data
            assert relerr(got, ref[name].reshape(got.shape)) < TOL, name
    net._engine.wino3 = False
    net._engine.build()
    assert not any("F(4x4x4" in l for l in net.op_labels()) and sum("F(4x4,3x3)" in l for l in net.op_labels()) == 6
    net.blobs["data"].data[...] = x
    out2 = net.forward()["fc8"]
    assert relerr(out2, ref["fc8"]) < TOL and relerr(out2, out) < TOL
    one_plane = make_net(backend, mini("lite"), fillers.synthetic_params(NetSpec.from_prototxt(mini("lite"))), True, winograd=4)
    assert not any("F(4x4x4" in l for l in one_plane.op_labels())


def test_reshape_grows_winograd_buffers(backend):
    """net.reshape() to a larger clip batch re-plans the Winograd route (transformed-volume scratch, batched
    plans, per-point weights) and still matches the oracle; shrinking back reuses the storage."""
    params = None
    outs = {}
    net = None
    for clips in (1, 3, 1):
        proto = mini("lite", num_clips=clips)
        spec = NetSpec.from_prototxt(proto)
        if params is None:
            params = fillers.synthetic_params(spec, seed=2)
            net = make_net(backend, proto, params, True, winograd=4)
        else:
            net.blobs["data"].reshape(4 * clips, 3, 32, 32)
            net.reshape()
        x = fillers.synthetic_frames(4 * clips, 32, 32, seed=clips)
        ref = orc.forward(spec, params, {"data": x}, fast_pool=False)["fc8"]
        net.blobs["data"].data[...] = x
        got = net.forward()["fc8"]
        assert got.shape == (clips, 10) and relerr(got, ref) < TOL
        outs[clips] = got.copy()


def test_pycaffe_surface(backend):
    proto = mini("lite")
    spec = NetSpec.from_prototxt(proto)
    params = fillers.synthetic_params(spec, seed=11)
    net = make_net(backend, proto, params, False)
    assert net.inputs == ["data"] and net.outputs == ["fc8"]
    assert list(net.blobs)[0] == "data" and list(net.blobs)[-1] == "fc8"
    assert net.blobs["data"].shape == (8, 3, 32, 32)
    assert (net.blobs["data"].num, net.blobs["data"].channels, net.blobs["data"].height) == (8, 3, 32)
    with pytest.raises(NetSpecError):
        net.blobs["res3a"].num  # legacy accessors CHECK-fail on 5-D blobs (blob.hpp:140-142)
    assert list(net.params["conv1_7x7_s2"][0].data.shape) == [8, 3, 7, 7]
    assert [b.data.shape for b in net.params["conv1_7x7_s2_bn"]] == [(1, 8)] * 4
    assert net.params["fc8"][0].data.shape == (10, 64)
    assert net._layer_names[0] == "conv1_7x7_s2" and len(net.layers) == len(spec.layers)
    x = fillers.synthetic_frames(8, 32, 32, seed=5)
    out1 = net.forward(data=x)["fc8"].copy()
    # kwargs form must name exactly the inputs; batch must match (pycaffe.py:83-90)
    with pytest.raises(Exception):
        net.forward(wrong=x)
    with pytest.raises(Exception):
        net.forward(data=x[:4])
    # extra blobs on request, and forward(end=...) returns that layer's top
    o = net.forward(blobs=["res5b_bn"])
    assert set(o) == {"fc8", "res5b_bn"}
    o = net.forward(end="pool1_3x3_s2")
    assert list(o) == ["pool1_3x3_s2"]
    # editing a parameter through .data is seen by the next forward (mutable_cpu_data semantics)
    net.params["fc8"][1].data[...] += 1.0
    out2 = net.forward()["fc8"]
    assert np.allclose(out2, out1 + 1.0, rtol=1e-5, atol=1e-4 * np.abs(out1).max())
    # reshape to 1 clip: blob.reshape + net.reshape (_caffe.cpp:193-205,224)
    net.blobs["data"].reshape(4, 3, 32, 32)
    net.reshape()
    assert net.blobs["fc8"].shape == (1, 10) and net.blobs["res2b_bn"].shape == (1, 12, 4, 4, 4)
    out3 = net.forward(data=x[:4])["fc8"]
    assert np.allclose(out3, out2[:1], rtol=1e-4, atol=1e-4 * np.abs(out2).max())
    # B*N not a multiple of N: r2Dto3D CHECK-fails (reshape_layer.cpp:79-81)
    net.blobs["data"].reshape(6, 3, 32, 32)
    with pytest.raises(NetSpecError, match="divisible"):
        net.reshape()


def test_filler_init_and_bad_params(backend):
    proto = mini("lite")
    spec = NetSpec.from_prototxt(proto)
    p = fillers.filler_params(spec, seed=0)
    w = p["conv1_7x7_s2"][0]
    assert abs(w).max() <= np.sqrt(3.0 / (3 * 49)) + 1e-6 and (p["conv1_7x7_s2"][1] == 0).all()  # xavier / constant 0
    assert (p["res3a_bn"][0] == 1).all() and (p["res3a_bn"][3] == 0).all()  # bn_layer.cpp:24-41 defaults
    bad = dict(p)
    bad["fc8"] = [p["fc8"][0]]
    with pytest.raises(ValueError, match="Incompatible number of blobs"):
        make_net(backend, proto, bad, True)
    bad = dict(p)
    bad["fc8"] = [p["fc8"][0][:, :5], p["fc8"][1]]
    with pytest.raises(ValueError, match="does not match"):
        make_net(backend, proto, bad, True)


def test_eco_full_plan_has_no_concat_copies(backend):
    """ECO-Full's stride-2 blocks (inception_3c / 4e) concatenate two strided convs and a MAX pool: the convs write their
    Concat slices from their epilogues, and -- round 5 -- the pool writes its slice itself (eco_pool_forward_strided), so the
    fused plan has no concat_copy launch left; pool_into_concat=False brings the two copies back.  Both match the oracle
    (test_mini_eco_matches_oracle runs the default plan blob by blob)."""
    proto = mini("full")
    spec = NetSpec.from_prototxt(proto)
    params = fillers.synthetic_params(spec, seed=7)
    x = fillers.synthetic_frames(8, 32, 32, seed=3)
    ref = orc.forward(spec, params, {"data": x}, fast_pool=False)["fc8"]
    net = make_net(backend, proto, params, True)
    labels = net.op_labels()
    assert not any(l.startswith("inception_3c_output[") or l.startswith("inception_4e_output[") for l in labels), labels
    assert "inception_3c_pool [into inception_3c_output]" in labels and "inception_4e_pool [into inception_4e_output]" in labels
    assert "inception_3c_pool" in net._engine.fused_away
    net.blobs["data"].data[...] = x
    assert relerr(net.forward()["fc8"], ref) < TOL
    net._engine.pool_into_concat = False
    net._engine.build()
    labels = net.op_labels()
    assert sum(l.startswith("inception_3c_output[") or l.startswith("inception_4e_output[") for l in labels) == 2
    net.blobs["data"].data[...] = x
    assert relerr(net.forward()["fc8"], ref) < TOL
