"""ABI v10 additions for ECO-Full (models_ECO_Full/kinetics/deploy.prototxt:1835-1870, 4607-4690):
the second destination of a fused conv's activated output (act2: the blob feeds a 2-D consumer and, through
r2Dto3D + Permute, the 3-D trunk) on every conv route, and the segment-consensus form of the pool + fc tail."""
import numpy as np
import pytest

import eco_oracle as orc
from eco_amd import fillers, hip, models
from eco_amd.net import Net
from eco_amd.netspec import NetSpec
from tests.test_net import make_net, mini, relerr


def _conv_dual(backend, cin, num_cu, winograd, H=8):
    """conv 3x3 s1 p1 + BN + ReLU on [B*T, cin, H, W]; act -> plain tensor, act2 -> permuted [B, C, T, H, W]."""
    rng = np.random.default_rng(11)
    B, T, cout = 2, 3, 32
    n, S = B * T, H * H
    x = rng.normal(size=(n, cin, H, H)).astype(np.float32)
    w = (rng.normal(size=(cout, cin, 3, 3)) / np.sqrt(9 * cin)).astype(np.float32)
    b = rng.normal(size=cout).astype(np.float32)
    sc, sh = rng.uniform(0.5, 1.5, cout).astype(np.float32), rng.normal(size=cout).astype(np.float32)
    ref = np.maximum(orc.convolution(x, w, b, (3, 3), (1, 1), (1, 1)) * sc[None, :, None, None] + sh[None, :, None, None], 0)
    lib = backend.lib
    y1, y2 = backend.empty(ref.shape), backend.empty((B, cout, T, H, H))
    ep = hip.ConvEpilogue()
    ep.bias = backend.ptr(backend.dev(b))
    ep.residual, ep.raw = hip.null_view(), hip.null_view()
    ep.bn_scale, ep.bn_shift, ep.relu = backend.ptr(backend.dev(sc)), backend.ptr(backend.dev(sh)), 1
    ep.act = hip.plain_view(backend.ptr(y1), cout, S)
    ep.act2 = hip.View(backend.ptr(y2), cout * T * S, S, T * S, T)
    g = hip.conv_geom(n, cin, cout, (H, H), (3, 3), (1, 1), (1, 1), (H, H))
    if winograd:
        M, P = 4, 36
        TH = TW = -(-H // M)
        u = np.empty((P, cout, cin, 1), np.float32)
        lib.wino_weight_transform(w.ctypes.data, cout, cin, 1, M, u.ctypes.data)
        gw = hip.conv_geom(n, cin, cout, (1, TH, TW), (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, TH, TW))
        plan = lib.conv_plan(gw, num_cu, batch=P)
        wps, kt = np.empty((P, plan.wp_elems), np.float32), np.empty(plan.ktab_elems, np.int32)
        for pt in range(P):
            lib.conv_pack_weights(gw, plan, u[pt].ctypes.data, wps[pt].ctypes.data, kt.ctypes.data)
        tin, tout = n * cin * TH * TW, n * cout * TH * TW
        v, m = backend.empty((P * tin,)), backend.empty((P * tout,))
        epg = hip.ConvEpilogue()
        epg.bias = None
        epg.residual, epg.act = hip.null_view(), hip.null_view()
        epg.bn_scale = epg.bn_shift = None
        epg.relu = 0
        epg.raw = hip.plain_view(backend.ptr(m), cout, TH * TW)
        ws = backend.empty((max(P * plan.ws_bytes, 4) // 4,))
        lib.wino_input_forward(backend.ptr(backend.dev(x)), backend.ptr(v), n * cin, H, H, M)
        lib.conv_forward_batched(gw, plan, backend.ptr(v), backend.ptr(backend.dev(wps)), backend.ptr(backend.dev(kt)), epg,
                                 backend.ptr(ws), P, tin, plan.wp_elems, tout)
        lib.wino_output_forward(backend.ptr(m), n, cout, 1, H, H, M, ep)
    else:
        plan = lib.conv_plan(g, num_cu)
        wp, kt = np.empty(plan.wp_elems, np.float32), np.empty(plan.ktab_elems, np.int32)
        lib.conv_pack_weights(g, plan, w.ctypes.data, wp.ctypes.data, kt.ctypes.data)
        ws = backend.empty((max(plan.ws_bytes, 4) // 4,))
        lib.conv_forward(g, plan, backend.ptr(backend.dev(x)), backend.ptr(backend.dev(wp)), backend.ptr(backend.dev(kt)), ep,
                         backend.ptr(ws))
    tol = 3e-5 * np.abs(ref).max()
    assert np.abs(backend.host(y1, ref.shape) - ref).max() <= tol
    assert np.abs(backend.host(y2, (B, cout, T, H, H)) - ref.reshape(B, T, cout, H, H).transpose(0, 2, 1, 3, 4)).max() <= tol
    return plan


@pytest.mark.parametrize("route", ["table", "span", "span-splitk", "winograd"])
def test_conv_second_destination(backend, route):
    if route == "table":
        assert _conv_dual(backend, 8, None, False).mode == 0
    elif route == "span":
        assert _conv_dual(backend, 16, None, False).mode == 2
    elif route == "span-splitk":
        assert _conv_dual(backend, 64, 1, False, H=14).ksplit > 1   # 5 tiles of 32x256 on 4 slots: tail split
    else:
        _conv_dual(backend, 16, None, True)


def test_act2_requires_act(backend):
    g = hip.conv_geom(1, 16, 16, (4, 4), (1, 1), (1, 1), (0, 0), (4, 4))
    plan = backend.lib.conv_plan(g)
    y = backend.empty((1, 16, 4, 4))
    ep = hip.ConvEpilogue()
    ep.raw = hip.plain_view(backend.ptr(y), 16, 16)
    ep.act2 = hip.plain_view(backend.ptr(y), 16, 16)
    with pytest.raises(hip.EcoError, match="act2"):
        backend.lib.conv_forward(g, plan, backend.ptr(y), backend.ptr(y), backend.ptr(y), ep, None)


def test_global_avgpool_fc_segment_consensus(backend):
    """x[b*T + f][c][s]: mean over the T*s values of a channel, then its columns of the fc; accumulate adds to y."""
    rng = np.random.default_rng(6)
    B, T, C2, S2, C3, S3, n_out = 3, 4, 40, 9, 24, 2 * 3 * 3, 7
    x2 = rng.normal(size=(B * T, C2, 3, 3)).astype(np.float32)
    x3 = rng.normal(size=(B, C3, 2, 3, 3)).astype(np.float32)
    w = rng.normal(size=(n_out, C2 + C3)).astype(np.float32)
    b = rng.normal(size=n_out).astype(np.float32)
    y = backend.empty((B, n_out))
    lib, wd = backend.lib, backend.ptr(backend.dev(w))
    lib.global_avgpool_fc_seg_forward(backend.ptr(backend.dev(x3)), wd, backend.ptr(backend.dev(b)), backend.ptr(y), B, 1, C3, S3,
                                      n_out, C2 + C3, C2, False)
    lib.global_avgpool_fc_seg_forward(backend.ptr(backend.dev(x2)), wd, None, backend.ptr(y), B, T, C2, S2, n_out, C2 + C3, 0, True)
    # the reference's layer sequence: 2-D global pool, consensus over T, concat with the 3-D global pool, fc
    f2 = orc.pooling(x2, "AVE", (3, 3), (1, 1), (0, 0)).reshape(B, 1, T, C2)
    f2 = orc.pooling(f2, "AVE", (T, 1), (1, 1), (0, 0)).reshape(B, C2)
    f3 = orc.pooling(x3, "AVE", (2, 3, 3), (1, 1, 1), (0, 0, 0)).reshape(B, C3)
    ref = orc.inner_product(np.concatenate([f2, f3], 1), w, b, 1)
    assert np.abs(backend.host(y, ref.shape) - ref).max() <= 1e-5 * np.abs(ref).max()


def test_eco_full_plan_fuses_permute_and_two_stream_tail(backend):
    """ECO-Full fused plan: inception_3c_double_3x3_1_bn is written to its 2-D blob and, as act2, through
    r2Dto3D + Transpose1 into the 3-D volume (no Permute launch); the ten tail layers are two pool+fc launches."""
    proto = mini("full")
    spec = NetSpec.from_prototxt(proto)
    params = fillers.synthetic_params(spec, seed=7)
    x = fillers.synthetic_frames(8, 32, 32, seed=3)
    net = make_net(backend, proto, params, True, winograd=4)
    labels = net.op_labels()
    assert not any(l == "Transpose1" for l in labels) and sum("+r2Dto3D+Transpose1" in l for l in labels) >= 1
    assert sum("stream" in l for l in labels) == 2 and not any(l.startswith("gn02_concat") or l == "fc8N" for l in labels)
    assert not any(l in ("global_pool2D", "segment_consensus_st2", "global_pool") for l in labels)
    out = net.forward(data=x)["fc8"]
    ref = orc.forward(spec, params, {"data": x}, keep="all")
    assert relerr(out, ref["fc8"]) < 2e-5
    for name in ("inception_3c_double_3x3_1_bn", "res2b_bn"):   # both destinations of the dual store are observable
        got = net.blobs[name].data
        assert relerr(got, ref[name].reshape(got.shape)) < 2e-5, name
    with pytest.raises(KeyError, match="two-stream tail"):
        net.blobs["global_pool2D"].data
