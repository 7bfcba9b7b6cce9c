"""The stride-2 polyphase Winograd route (csrc/eco_wino_s2.hip) inside the engine: a residual stage of the 3-D trunk
(res4a_1 | res4a_down -> res4a_2 -> Eltwise -> res4b_*; models_ECO_Lite/kinetics/deploy.prototxt:1262-1460) at a size where
the route applies (output volume 4k x 7 x 7), against the CPU oracle, fused and layer by layer."""
import numpy as np
import pytest

import eco_oracle as orc
from eco_amd import fillers, models
from eco_amd.netspec import NetSpec
from tests.test_net import make_net, relerr


def stage_proto(n=2, cin=16, cout=32, D=8, H=14, W=14, classes=5):
    b = models._Builder()
    b.bn("res3b_bn", "data", "res3b_bn", frozen_field=True)
    b.relu("res3b_relu", "res3b_bn")
    t = "res3b_bn"
    b.res_conv("res4a_1", t, "res4a_1", cout, 2)
    u = b.bn_relu_3d("res4a_1", "res4a_1")
    b.res_conv("res4a_2", u, "res4a_2", cout, 1)
    b.res_conv("res4a_down", t, "res4a_down", cout, 2)
    b.eltwise("res4a", "res4a_2", "res4a_down", "res4a")
    u = b.bn_relu_3d("res4a", "res4a")
    b.res_conv("res4b_1", u, "res4b_1", cout, 1)
    u = b.bn_relu_3d("res4b_1", "res4b_1")
    b.res_conv("res4b_2", u, "res4b_2", cout, 1)
    b.eltwise("res4b", "res4b_2", "res4a", "res4b")
    t = b.bn_relu_3d("res4b", "res4b")
    b.pool("global_pool", t, "global_pool", "AVE", f"kernel_size: [{D // 2}, {H // 2}, {W // 2}] stride: [1, 1, 1]")
    b.reshape("global_pool_reshape", "global_pool", "global_pool_reshape", [-1, cout])
    models._fc(b, "fc8", "global_pool_reshape", classes)
    hdr = 'name: "stage"\ninput: "data"\ninput_shape { ' + " ".join(f"dim: {d}" for d in (n, cin, D, H, W)) + " }\n"
    return hdr + "\n".join(b.out) + "\n"


S2_TOL = 5e-4     # two nested eight-point transforms: ~1e-4 of a layer's largest output (csrc/eco_wino_s2.hip)


@pytest.mark.parametrize("fuse", [True, False])
def test_stage_takes_the_stride2_route(backend, fuse):
    proto = stage_proto()
    spec = NetSpec.from_prototxt(proto)
    params = fillers.synthetic_params(spec, seed=3)
    x = np.random.default_rng(2).standard_normal((2, 16, 8, 14, 14)).astype(np.float32)
    ref = orc.forward(spec, params, {"data": x}, keep="all", fast_pool=False)
    net = make_net(backend, proto, params, fuse, winograd=4)
    labels = net.op_labels()
    s2 = [l for l in labels if "stride-2 winograd" in l or "transformed-domain GEMMs, K = 128" in l]
    if fuse:   # one input transform and one GEMM for the pair, one output transform per member
        assert len(s2) == 4 and sum("input transform" in l for l in s2) == 1, labels
        assert "res4a_1+res4a_1_bn+res4a_1_relu | res4a_down [320 transformed-domain GEMMs, K = 128]" in s2
        # the shortcut runs ahead of res4a_2, which now carries the Eltwise, BN and ReLU
        assert any(l.startswith("res4a_2+res4a+res4a_bn+res4a_relu") for l in labels), labels
    else:
        assert len(s2) == 6 and sum("input transform" in l for l in s2) == 2, labels
    net.blobs["data"].data[...] = x
    out = net.forward()["fc8"].copy()
    assert relerr(out, ref["fc8"]) < S2_TOL
    for name in ("res4a_down", "res4a_1_bn", "res4a", "res4a_bn", "res4b_bn"):
        if name in net._engine.tensors:
            got = net.blobs[name].data
            assert relerr(got, ref[name].reshape(got.shape)) < S2_TOL, name
    # the switch: the same net on the direct strided kernel
    net._engine.wino_s2 = False
    net._engine.build()
    assert not any("stride-2 winograd" in l or "stride-2" in l for l in net.op_labels())
    net.blobs["data"].data[...] = x
    out2 = net.forward()["fc8"]
    assert relerr(out2, ref["fc8"]) < 2e-5 and relerr(out2, out) < S2_TOL
    # ... and back: the group's weight image is rebuilt
    net._engine.wino_s2 = True
    net._engine.build()
    net.blobs["data"].data[...] = x
    assert np.array_equal(net.forward()["fc8"], out)


def test_size_rule_and_parameter_updates(backend):
    """Below wino_s2_min_positions the default plan keeps the direct kernel; an updated member weight rebuilds the group's image."""
    proto = stage_proto()
    spec = NetSpec.from_prototxt(proto)
    params = fillers.synthetic_params(spec, seed=3)
    x = np.random.default_rng(2).standard_normal((2, 16, 8, 14, 14)).astype(np.float32)
    net = make_net(backend, proto, params, True)            # winograd=True: 2 clips x 2 x 1 x 1 positions < 128
    assert not any("stride-2 winograd" in l for l in net.op_labels())
    net = make_net(backend, proto, params, True, winograd=4)
    net.blobs["data"].data[...] = x
    y0 = net.forward()["fc8"].copy()
    params2 = {k: [a.copy() for a in v] for k, v in params.items()}     # (the net may hold `params`' own arrays)
    params2["res4a_down"][0] = params2["res4a_down"][0] * 0.5
    net.params["res4a_down"][0].data[...] = params2["res4a_down"][0]
    ref = orc.forward(spec, params2, {"data": x}, keep="all", fast_pool=False)
    net.blobs["data"].data[...] = x
    y1 = net.forward()["fc8"]
    assert relerr(y1, ref["fc8"]) < S2_TOL and relerr(y1, y0) > 1e-3


def test_stage_with_few_depth_tiles_takes_the_2d_form(backend):
    """Output depth 2 (no 4-plane tiles): the pair runs on F(7,2) x F(7,2) with the depth taps in the reduction (K = 12 cin), every
    output plane a position -- the res5a form (4 x 7 x 7 outputs at num_segments 16)."""
    proto = stage_proto(D=4)
    spec = NetSpec.from_prototxt(proto)
    params = fillers.synthetic_params(spec, seed=4)
    x = np.random.default_rng(5).standard_normal((2, 16, 4, 14, 14)).astype(np.float32)
    ref = orc.forward(spec, params, {"data": x}, keep="all", fast_pool=False)
    net = make_net(backend, proto, params, True, winograd=4)
    labels = net.op_labels()
    assert "res4a_1+res4a_1_bn+res4a_1_relu | res4a_down [64 transformed-domain GEMMs, K = 192]" in labels, labels
    assert sum("F(7,2)xF(7,2), depth taps direct" in l for l in labels) == 3
    net.blobs["data"].data[...] = x
    out = net.forward()["fc8"].copy()
    assert relerr(out, ref["fc8"]) < S2_TOL
    for name in ("res4a_down", "res4a_1_bn", "res4a_bn"):
        got = net.blobs[name].data
        assert relerr(got, ref[name].reshape(got.shape)) < S2_TOL, name


def test_strided_2d_conv_takes_the_2d_form(backend):
    """A strided 3x3 2-D conv (+ BN + ReLU) writing its channel slice of a Concat top next to a MAX pool branch: the stride-2 block of
    ECO-Full's inception_3c / 4e (models_ECO_Full/kinetics/deploy.prototxt:1854-1990) in miniature."""
    b = models._Builder()
    t = b.conv_bn_relu_2d("inception_3c", "3x3_reduce", "data", 16, 1)
    c3 = b.conv_bn_relu_2d("inception_3c", "3x3", t, 32, 3, 2, 1)
    b.pool("inception_3c_pool", "data", "inception_3c_pool", "MAX", "kernel_size: 3 stride: 2")
    b.concat("inception_3c_output", [c3, "inception_3c_pool"], "inception_3c_output")
    b.pool("gp", "inception_3c_output", "gp", "AVE", "kernel_size: 7 stride: 1")
    b.reshape("gp_reshape", "gp", "gp_reshape", [-1, 40])
    models._fc(b, "fc8", "gp_reshape", 5)
    proto = 'name: "s2d"\ninput: "data"\ninput_shape { dim: 3 dim: 8 dim: 14 dim: 14 }\n' + "\n".join(b.out) + "\n"
    spec = NetSpec.from_prototxt(proto)
    params = fillers.synthetic_params(spec, seed=6)
    x = np.random.default_rng(7).standard_normal((3, 8, 14, 14)).astype(np.float32)
    ref = orc.forward(spec, params, {"data": x}, keep="all", fast_pool=False)
    net = make_net(backend, proto, params, True, winograd=4)
    labels = net.op_labels()
    assert any("[64 transformed-domain GEMMs, K = 64]" in l for l in labels), labels
    net.blobs["data"].data[...] = x
    out = net.forward()["fc8"].copy()
    assert relerr(out, ref["fc8"]) < S2_TOL
    got = net.blobs["inception_3c_output"].data
    assert relerr(got, ref["inception_3c_output"]) < S2_TOL
    # default plan (size and cost rules): three images of 7 x 7 outputs are far too few positions
    net = make_net(backend, proto, params, True)
    assert not any("stride-2 winograd" in l for l in net.op_labels())
