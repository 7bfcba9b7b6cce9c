"""Sibling convolutions as one launch (eco_conv_epilogue::nseg): the 1x1 / 3x3_reduce / double_3x3_reduce convs of an
Inception block (models_ECO_Lite/kinetics/deploy.prototxt:130-330) share their bottom; concatenated along the output
channel they are one GEMM whose 32-row tiles write to each member's own destination."""
import numpy as np
import pytest

import eco_oracle as orc
from eco_amd import fillers, hip
from tests.test_kernels import TOL, relerr


def _segmented(be, n, cin, couts, insp, kernel, pad, num_cu, concat_first=True, stride=None, raw_last=False,
               want_split=False):
    rng = np.random.default_rng(sum(couts) + cin)
    nd = len(insp)
    x = rng.standard_normal((n, cin) + tuple(insp)).astype(np.float32)
    ws = [(rng.standard_normal((c, cin) + tuple(kernel)) / np.sqrt(cin)).astype(np.float32) for c in couts]
    bs = [rng.standard_normal(c).astype(np.float32) for c in couts]
    scs = [rng.uniform(0.5, 1.5, c).astype(np.float32) for c in couts]
    shs = [rng.standard_normal(c).astype(np.float32) for c in couts]
    one = stride or (1,) * nd
    bshape = lambda c: (1, c) + (1,) * nd
    if raw_last:        # the last member wants its raw value: scale 1, shift 0, no ReLU
        scs[-1][:] = 1.0
        shs[-1][:] = 0.0
    refs = [orc.convolution(x, w, b, kernel, one, pad) * sc.reshape(bshape(len(b))) + sh.reshape(bshape(len(b)))
            for w, b, sc, sh in zip(ws, bs, scs, shs)]
    refs = [r if (raw_last and k == len(refs) - 1) else np.maximum(r, 0) for k, r in enumerate(refs)]
    outsp = refs[0].shape[2:]
    S = int(np.prod(outsp))
    ctot = sum(couts)
    lib = be.lib
    g = hip.conv_geom(n, cin, ctot, insp, kernel, one, pad, outsp)
    plan = lib.conv_plan(g, num_cu)
    assert (plan.ksplit > 1) == want_split
    wcat = np.ascontiguousarray(np.concatenate(ws, 0))
    wp = np.zeros(plan.wp_elems, np.float32)
    kt = np.zeros(plan.ktab_elems, np.int32)
    lib.conv_pack_weights(g, plan, wcat.ctypes.data, wp.ctypes.data, kt.ctypes.data)
    dx, dwp, dkt = be.dev(x), be.dev(wp), be.dev(kt)
    db, dsc, dsh = be.dev(np.concatenate(bs)), be.dev(np.concatenate(scs)), be.dev(np.concatenate(shs))
    ep = hip.ConvEpilogue()
    ep.bias, ep.bn_scale, ep.bn_shift, ep.relu = be.ptr(db), be.ptr(dsc), be.ptr(dsh), 1
    ep.residual, ep.raw, ep.act2 = hip.null_view(), hip.null_view(), hip.null_view()
    outs = []
    # member 0 writes into channels [5, 5 + c0) of a wider (Concat) tensor, the others into their own tensors
    c0off, wide = 5, couts[0] + 9
    if concat_first:
        big = be.dev(np.full((n, wide) + tuple(outsp), 7.0, np.float32))
        ep.act = hip.View(be.ptr(big, c0off * S), wide * S, 0, S, 1)
    else:
        big = be.empty(refs[0].shape)
        ep.act = hip.plain_view(be.ptr(big), couts[0], S)
    ep.nseg = len(couts) - 1
    begin = 0
    for s, c in enumerate(couts[1:]):
        begin += couts[s]
        t = be.dev(np.full(refs[s + 1].shape, -3.0, np.float32))
        outs.append(t)
        ep.seg_begin[s] = begin
        ep.seg_relu[s] = 0 if (raw_last and s == len(couts) - 2) else 1
        ep.seg_act[s] = hip.plain_view(be.ptr(t), c, S)
    wsb = be.ptr(be.empty((plan.ws_bytes // 4,))) if plan.ws_bytes else None
    lib.conv_forward(g, plan, be.ptr(dx), be.ptr(dwp), be.ptr(dkt), ep, wsb)
    if concat_first:
        got = be.host(big, (n, wide) + tuple(outsp))
        assert relerr(got[:, c0off:c0off + couts[0]], refs[0]) < TOL
        assert (got[:, :c0off] == 7.0).all() and (got[:, c0off + couts[0]:] == 7.0).all()
    else:
        assert relerr(be.host(big, refs[0].shape), refs[0]) < TOL
    for t, r in zip(outs, refs[1:]):
        assert relerr(be.host(t, r.shape), r) < TOL
    return plan


@pytest.mark.parametrize("n,cin,couts,insp,num_cu", [
    (3, 32, (64, 64, 64), (20, 20), 1),     # inception_3a's three 1x1 siblings: point kernel, bm 96 x 2
    (2, 48, (32, 96), (24, 24), 1),         # two members of different widths; tiles straddle the boundary block
    (9, 16, (64, 32, 64), (2, 8, 8), 1),    # 3-D blob
    (1, 16, (32, 32), (7, 7), None),        # plane size 49: the gather kernel's epilogue
])
def test_segmented_point_conv(backend, n, cin, couts, insp, num_cu):
    nd = len(insp)
    plan = _segmented(backend, n, cin, couts, insp, (1,) * nd, (0,) * nd, num_cu)
    if num_cu == 1:
        assert plan.mode == 3


def test_segmented_3x3_direct(backend):
    # the epilogue is shared by every direct kernel: a 3x3 conv through the span / gather kernels
    _segmented(backend, 2, 16, (32, 64), (9, 9), (3, 3), (1, 1), None, concat_first=False)


def test_segmented_strided_block_pair(backend):
    """res4a_1 | res4a_down: two stride-2 3x3x3 convs of one geometry; the shortcut keeps its raw value.  The default
    device plan for this size is split-K: the reduce launch applies the members' destinations too."""
    _segmented(backend, 2, 16, (64, 64), (4, 10, 10), (3, 3, 3), (1, 1, 1), None, concat_first=False, stride=(2, 2, 2),
               raw_last=True, want_split=True)
    _segmented(backend, 4, 16, (32, 64), (4, 12, 12), (3, 3, 3), (1, 1, 1), 1, concat_first=False, stride=(2, 2, 2),
               raw_last=True)


def test_segment_validation(backend):
    lib = backend.lib
    g = hip.conv_geom(1, 16, 64, (8, 8), (1, 1), (1, 1), (0, 0), (8, 8))
    plan = lib.conv_plan(g, 1)
    buf = backend.empty((1, 64, 8, 8))
    ep = hip.ConvEpilogue()
    ep.residual, ep.raw, ep.act2 = hip.null_view(), hip.null_view(), hip.null_view()
    ep.act = hip.plain_view(backend.ptr(buf), 32, 64)
    ep.nseg = 1
    ep.seg_begin[0] = 24                      # not a multiple of 32
    ep.seg_act[0] = hip.plain_view(backend.ptr(buf), 32, 64)
    p = backend.ptr(buf)
    with pytest.raises(hip.EcoError, match="multiple of 32"):
        lib.conv_forward(g, plan, p, p, p, ep, None)
    ep.seg_begin[0] = 32
    ep.raw = hip.plain_view(backend.ptr(buf), 64, 64)
    with pytest.raises(hip.EcoError, match="plain act destinations only"):
        lib.conv_forward(g, plan, p, p, p, ep, None)


@pytest.mark.parametrize("variant", ["lite", "full"])
def test_engine_fuses_inception_siblings(backend, variant):
    """width_div=2 keeps every 1x1 width a multiple of 32: the engine runs each block's sibling 1x1 convs as one
    launch, logits and every surviving blob still match the oracle, and a parameter update reaches the group."""
    from eco_amd import models
    from eco_amd.netspec import NetSpec
    from tests.test_net import make_net
    if variant == "full" and backend.kind == "emu":
        pytest.skip("the wider mini ECO-Full is GPU-only (emulator time)")
    gen = models.eco_lite_deploy if variant == "lite" else models.eco_full_deploy
    proto = gen(num_segments=4, num_clips=1, num_classes=10, input_size=32, width_div=2)
    spec = NetSpec.from_prototxt(proto)
    params = fillers.synthetic_params(spec, seed=11)
    x = fillers.synthetic_frames(4, 32, 32, seed=5)
    ref = orc.forward(spec, params, {"data": x}, keep="all", fast_pool=False)
    net = make_net(backend, proto, params, True, _num_cu=1)
    net.blobs["data"].data[...] = x
    out = net.forward()
    groups = [l for l in net.op_labels() if " | " in l]
    assert any(l.startswith("inception_3a_1x1+") and l.count(" | ") == 2 for l in groups), net.op_labels()
    assert len(groups) >= (2 if variant == "lite" else 5)
    assert relerr(out["fc8"], ref["fc8"]) < TOL
    for name in net.blobs:
        if name in net._engine.tensors:
            got = net.blobs[name].data
            assert relerr(got, ref[name].reshape(got.shape)) < TOL, name
    if backend.kind != "emu":
        # the same graph with the fusion switched off gives the same logits (different launches)
        net2 = make_net(backend, proto, params, True, _num_cu=1)
        net2._engine.siblings = False
        net2._engine.build()
        net2.blobs["data"].data[...] = x
        assert not any(" | " in l for l in net2.op_labels())
        assert relerr(net2.forward()["fc8"], ref["fc8"]) < TOL
    # a member's weights change: the group's concatenated weights are repacked
    name = "inception_3a_3x3_reduce"
    net.params[name][0].data[...] *= 0.5
    params[name] = [np.array(b.data) for b in net.params[name]]   # (the engine may share the caller's arrays)
    ref2 = orc.forward(spec, params, {"data": x}, keep="all", fast_pool=False)
    assert relerr(net.forward()["fc8"], ref2["fc8"]) < TOL
    assert relerr(ref2["fc8"], ref["fc8"]) > 1e-4


def test_engine_block_pairs_opt_in(backend):
    """engine.sibling_blocks: res4a_1 | res4a_down and res5a_1 | res5a_down as one launch each (the shortcut keeps its
    raw value, the Eltwise moves to the block's second conv): same logits and blobs as the oracle."""
    from eco_amd.netspec import NetSpec
    from tests.test_net import make_net, mini
    proto = mini("lite")
    spec = NetSpec.from_prototxt(proto)
    params = fillers.synthetic_params(spec, seed=7)
    x = fillers.synthetic_frames(8, 32, 32, seed=3)
    ref = orc.forward(spec, params, {"data": x}, keep="all", fast_pool=False)
    net = make_net(backend, proto, params, True, winograd=4)
    net._engine.sibling_blocks = True
    net._engine.build()
    net.blobs["data"].data[...] = x
    out = net.forward()
    labels = net.op_labels()
    assert "res4a_1+res4a_1_bn+res4a_1_relu | res4a_down" in labels and "res5a_1+res5a_1_bn+res5a_1_relu | res5a_down" in labels
    assert relerr(out["fc8"], ref["fc8"]) < TOL
    for name in ("res4a_down", "res4a", "res4a_bn", "res5a", "res5a_bn"):
        got = net.blobs[name].data
        assert relerr(got, ref[name].reshape(got.shape)) < TOL, name
