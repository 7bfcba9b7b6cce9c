"""The channel-blocked bf16-MFMA kernels (csrc/eco_blocked.hip) against the CPU oracle, through the C ABI, on the
emulator (CPU suite) and on the GPU (-m gpu).

Tolerances.  ECO_DT_BF16: operands are bf16 (the oracle is fed the same bf16-rounded inputs and weights, so the
products agree exactly), accumulation is fp32, and each stored output is rounded once to bf16: |err| <=
2^-8 |y| (half a bf16 ulp is 2^-9) + accumulation-order noise."""
import numpy as np
import pytest

import eco_oracle as orc
from eco_amd import blocked, hip

BF16 = hip.DT_BF16
DTS = [pytest.param(BF16, id="bf16")]


def dev_blocked(backend, x, dt):
    raw = blocked.to_blocked(x, dt)
    return backend.dev(raw)


def host_blocked(backend, h, shape, dt):
    n = int(np.prod(shape))
    raw = np.asarray(backend.alloc.download(h, n))
    return blocked.from_blocked(raw, shape, dt)


def empty_blocked(backend, shape, dt):
    return backend.empty(shape, blocked.STORAGE[dt])


def bptr(backend, h, dt, offset_blocks=0):
    return backend.alloc.ptr(h) + offset_blocks * 8 * (2 if dt == BF16 else 4)


def check(got, ref, dt, what=""):
    scale = np.abs(ref).max() + 1e-30
    if dt == BF16:
        err = np.abs(got - ref) - 2.0 ** -8 * np.abs(ref)
        assert err.max() <= 2e-5 * scale, (what, float(err.max() / scale))
    else:
        assert np.abs(got - ref).max() <= 2e-5 * scale, (what, float(np.abs(got - ref).max() / scale))


def quant(x, dt):
    return blocked.bf16_round(x) if dt == BF16 else np.asarray(x, np.float32)


CONVS = [  # n, cin, cout, in_sp, kernel, stride, pad
    (2, 32, 64, (9, 9), (3, 3), (1, 1), (1, 1)),          # 2-D 3x3 same
    (3, 64, 96, (7, 7), (1, 1), (1, 1), (0, 0)),          # 1x1, bm = 96
    (2, 32, 32, (4, 6, 6), (3, 3, 3), (2, 2, 2), (1, 1, 1)),   # 3-D strided, bm = 32
    (1, 64, 160, (3, 5, 5), (3, 3, 3), (1, 1, 1), (1, 1, 1)),  # 3-D same, cout 160 -> bm 96 (2 M-blocks)
    (2, 96, 128, (6, 6), (3, 3), (2, 2), (1, 1)),         # bm = 128, three stages per tap group
    (2, 24, 40, (5, 5), (3, 3), (1, 1), (1, 1)),          # cin = 24: zero-padded last channel group; cout 40
    (1, 32, 256, (66, 66), (1, 1), (1, 1), (0, 0)),       # bf16: 256 x 128 LDS-DMA tile (cout % 256 == 0, >= 4096 positions)
    (1, 64, 128, (2, 65, 65), (3, 1, 1), (1, 1, 1), (1, 0, 0)),   # bf16: 128 x 256 LDS-DMA tile (>= 8192 positions)
]


def run_convb(backend, dt, n, cin, cout, in_sp, kernel, stride, pad, num_cu=None, seed=0, residual=False,
              bn=True, raw=True):
    rng = np.random.default_rng(seed)
    out_sp = tuple((in_sp[i] + 2 * pad[i] - kernel[i]) // stride[i] + 1 for i in range(len(in_sp)))
    x = quant(rng.normal(size=(n, cin) + in_sp).astype(np.float32), dt)
    w = quant((rng.normal(size=(cout, cin) + kernel) / np.sqrt(cin * np.prod(kernel))).astype(np.float32), dt)
    b = rng.normal(size=cout).astype(np.float32)
    sc = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    sh = rng.normal(size=cout).astype(np.float32)
    res = quant(rng.normal(size=(n, cout) + out_sp).astype(np.float32), dt) if residual else None
    g = hip.conv_geom(n, cin, cout, in_sp, kernel, stride, pad, out_sp)
    plan = backend.lib.convb_plan(g, dt, num_cu)
    wp = np.zeros(plan.wp_vecs * 8, np.uint16)
    backend.lib.convb_pack_weights(g, plan, w.ctypes.data, wp.ctypes.data)
    S = int(np.prod(out_sp))
    y_raw, y_act = empty_blocked(backend, (n, cout) + out_sp, dt), empty_blocked(backend, (n, cout) + out_sp, dt)
    ep = hip.ConvEpilogue()
    ep.bias = backend.ptr(backend.dev(b))
    ep.residual = hip.View(bptr(backend, dev_blocked(backend, res, dt), dt), (cout // 8) * S, 0, S, 1) if residual else hip.null_view()
    ep.raw = hip.View(bptr(backend, y_raw, dt), (cout // 8) * S, 0, S, 1) if raw else hip.null_view()
    ep.bn_scale = backend.ptr(backend.dev(sc)) if bn else None
    ep.bn_shift = backend.ptr(backend.dev(sh)) if bn else None
    ep.relu = 1 if bn else 0
    ep.act = hip.View(bptr(backend, y_act, dt), (cout // 8) * S, 0, S, 1)
    ws = backend.empty((max(plan.ws_bytes, 4) // 4,)) if plan.ws_bytes else None
    backend.lib.convb_forward(g, plan, bptr(backend, dev_blocked(backend, x, dt), dt), backend.ptr(backend.dev(wp)), ep,
                              backend.ptr(ws) if ws is not None else None)
    v = orc.convolution(x, w, b, kernel, stride, pad)
    if residual:
        v = v + res
    a = v * sc.reshape((1, -1) + (1,) * len(out_sp)) + sh.reshape((1, -1) + (1,) * len(out_sp)) if bn else v
    if bn:
        a = np.maximum(a, 0)
    if raw:
        check(host_blocked(backend, y_raw, v.shape, dt), v, dt, "raw")
    check(host_blocked(backend, y_act, a.shape, dt), a, dt, "act")
    return plan


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("case", CONVS, ids=[f"c{i}" for i in range(len(CONVS))])
def test_convb_matches_oracle(backend, dt, case):
    run_convb(backend, dt, *case)


@pytest.mark.parametrize("dt", DTS)
def test_convb_residual_and_split_k(backend, dt):
    plan = run_convb(backend, dt, 1, 64, 64, (4, 5, 5), (3, 3, 3), (1, 1, 1), (1, 1, 1), num_cu=8, residual=True)
    assert plan.ksplit > 1 and plan.ws_bytes > 0
    run_convb(backend, dt, 2, 32, 64, (6, 6), (3, 3), (1, 1), (1, 1), residual=True, bn=False, raw=False)


@pytest.mark.parametrize("dt", DTS)
def test_convb_stem_7x7(backend, dt):
    """conv1_7x7_s2 form: fp32 N,3,H,W frames -> eco_stem_pack_forward -> the stem plan."""
    rng = np.random.default_rng(5)
    n, H, W, cout = 2, 20, 24, 32
    x = rng.uniform(-120, 130, size=(n, 3, H, W)).astype(np.float32)
    w = (rng.normal(size=(cout, 3, 7, 7)) / 12).astype(np.float32)
    b = rng.normal(size=cout).astype(np.float32)
    out_sp = ((H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1)
    g = hip.conv_geom(n, 3, cout, (H, W), (7, 7), (2, 2), (3, 3), out_sp)
    plan = backend.lib.convb_plan(g, dt)
    assert plan.stem == 1 and plan.nstages == 7 and plan.cblocks == 4
    wp = np.zeros(plan.wp_vecs * 8, np.uint16)
    backend.lib.convb_pack_weights(g, plan, w.ctypes.data, wp.ctypes.data)
    packed = backend.empty((n * (H + 6) * (W + 8) * 4,), blocked.STORAGE[dt])
    backend.lib.stem_pack_forward(backend.ptr(backend.dev(x)), backend.alloc.ptr(packed), n, H, W, dt)
    S = out_sp[0] * out_sp[1]
    y = empty_blocked(backend, (n, cout) + out_sp, dt)
    ep = hip.ConvEpilogue()
    ep.bias = backend.ptr(backend.dev(b))
    ep.residual, ep.act = hip.null_view(), hip.null_view()
    ep.bn_scale = ep.bn_shift = None
    ep.relu = 0
    ep.raw = hip.View(bptr(backend, y, dt), (cout // 8) * S, 0, S, 1)
    backend.lib.convb_forward(g, plan, backend.alloc.ptr(packed), backend.ptr(backend.dev(wp)), ep, None)
    ref = orc.convolution(quant(x, dt), quant(w, dt), b, (7, 7), (2, 2), (3, 3))
    check(host_blocked(backend, y, ref.shape, dt), ref, dt)


@pytest.mark.parametrize("dt", DTS)
def test_convb_concat_slice_and_permuted_store(backend, dt):
    """Blocked views: a channel slice of a Concat top, and r2Dto3D + Permute [B*T,C,H,W] -> [B,C,T,H,W]."""
    rng = np.random.default_rng(9)
    B, T, cin, cout, H = 2, 3, 32, 32, 5
    n, S = B * T, H * H
    x = quant(rng.normal(size=(n, cin, H, H)).astype(np.float32), dt)
    w = quant((rng.normal(size=(cout, cin, 1, 1)) / 6).astype(np.float32), dt)
    g = hip.conv_geom(n, cin, cout, (H, H), (1, 1), (1, 1), (0, 0), (H, H))
    plan = backend.lib.convb_plan(g, dt)
    wp = np.zeros(plan.wp_vecs * 8, np.uint16)
    backend.lib.convb_pack_weights(g, plan, w.ctypes.data, wp.ctypes.data)
    ctot, c0 = 96, 40
    cat = empty_blocked(backend, (n, ctot, H, H), dt)
    vol = empty_blocked(backend, (B, cout, T, H, H), dt)
    ep = hip.ConvEpilogue()
    ep.bias = None
    ep.residual = hip.null_view()
    ep.bn_scale = ep.bn_shift = None
    ep.relu = 0
    ep.raw = hip.View(bptr(backend, cat, dt, (c0 // 8) * S), (ctot // 8) * S, 0, S, 1)
    ep.act = hip.View(bptr(backend, vol, dt), (cout // 8) * T * S, S, T * S, T)
    backend.lib.convb_forward(g, plan, bptr(backend, dev_blocked(backend, x, dt), dt), backend.ptr(backend.dev(wp)), ep, None)
    ref = orc.convolution(x, w, None, (1, 1), (1, 1), (0, 0))
    got_cat = host_blocked(backend, cat, (n, ctot, H, H), dt)
    check(got_cat[:, c0:c0 + cout], ref, dt, "concat slice")
    got_vol = host_blocked(backend, vol, (B, cout, T, H, H), dt)
    check(got_vol, ref.reshape(B, T, cout, H, H).transpose(0, 2, 1, 3, 4), dt, "permuted volume")


POOLS = [((2, 16, 9, 9), "MAX", (3, 3), (2, 2), (0, 0)), ((1, 8, 7, 7), "AVE", (3, 3), (1, 1), (1, 1)),
         ((1, 8, 8, 10), "MAX", (3, 3), (2, 2), (0, 0)),    # ceil rule: the last windows overhang the image
         ((1, 8, 6, 6), "AVE", (3, 3), (2, 2), (1, 1)),     # divisor counts the padding the window covers

         ((2, 8, 4, 5, 5), "AVE", (4, 5, 5), (1, 1, 1), (0, 0, 0)), ((1, 16, 3, 6, 6), "MAX", (2, 3, 3), (1, 2, 2), (0, 1, 1))]


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("shape,method,k,s,p", POOLS)
def test_poolb_matches_oracle(backend, dt, shape, method, k, s, p):
    from eco_amd.netspec import pooled_dim
    x = quant(np.random.default_rng(2).normal(size=shape).astype(np.float32), dt)
    out_sp = tuple(pooled_dim(shape[2 + i], k[i], s[i], p[i]) for i in range(len(k)))
    g = hip.pool_geom(shape[0], shape[1], shape[2:], k, s, p, out_sp, method)
    y = empty_blocked(backend, shape[:2] + out_sp, dt)
    backend.lib.poolb_forward(g, dt, bptr(backend, dev_blocked(backend, x, dt), dt), bptr(backend, y, dt))
    ref = orc.pooling(x, method, k, s, p)
    check(host_blocked(backend, y, ref.shape, dt), ref, dt)


@pytest.mark.parametrize("dt", DTS)
def test_global_avgpool_fc_b(backend, dt):
    rng = np.random.default_rng(4)
    B, Cc, S, n_out = 3, 64, 2 * 3 * 3, 10
    x = quant(rng.normal(size=(B, Cc, 2, 3, 3)).astype(np.float32), dt)
    w = rng.normal(size=(n_out, Cc)).astype(np.float32)
    b = rng.normal(size=n_out).astype(np.float32)
    y = backend.empty((B, n_out))
    backend.lib.global_avgpool_fc_b_forward(bptr(backend, dev_blocked(backend, x, dt), dt), dt, backend.ptr(backend.dev(w)),
                                            backend.ptr(backend.dev(b)), backend.ptr(y), B, Cc, S, n_out, Cc)
    ref = x.reshape(B, Cc, S).mean(2) @ w.T + b
    assert np.abs(backend.host(y, ref.shape) - ref).max() <= 1e-5 * np.abs(ref).max()


SPANS = [  # n, cin, cout, in_sp, num_cu  (stride-1 same-size (kd)x3x3 with >= 2048 positions: bf16 takes the span kernel)
    (2, 32, 64, (34, 34), 1),            # 2-D, one channel group, no split-K; last tile ragged
    (1, 64, 128, (3, 26, 27), None),     # 3-D, two channel groups x three depth taps, split over groups
    (1, 32, 96, (2, 33, 32), 1),         # bm = 96 (3x2 wave tiles), depth 2: both depth borders in one tile
    (3, 40, 32, (28, 28), 1),            # cin = 40: zero-padded second channel group; three images per tile row
    (4, 64, 192, (30, 30), 1),           # two M-blocks of 96 per position tile; 30 items on 16 persistent workgroups
    (2, 32, 32, (4, 20, 20), 1),         # 3-D, 13 tiles on 8 persistent workgroups: items change under a running pipeline
    (1, 384, 64, (50, 50), 1),           # 10 tiles on 4 CUs, twelve channel groups: a K-split tail of 2 tiles x 2 slices
    (2, 128, 192, (3, 27, 26), 2),       # two M-blocks, 16 + 1 position tiles on 8 CUs, 12 groups: a tail of one position tile (2 tiles) x 4
]
TAILS = {"s6": (2, 2), "s7": (2, 4)}   # case id -> (tail_tiles, tail_ksplit) the plan must choose (long reductions only)


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("n,cin,cout,in_sp,num_cu", SPANS, ids=[f"s{i}" for i in range(len(SPANS))])
def test_convb_span_kernel(backend, dt, n, cin, cout, in_sp, num_cu):
    k = (3,) * len(in_sp)
    plan = run_convb(backend, dt, n, cin, cout, in_sp, k, (1,) * len(in_sp), (1,) * len(in_sp), num_cu=num_cu, seed=3,
                     residual=(cout == 128))
    if dt == BF16:
        assert plan.span_pieces == -(-(256 + 2 * (in_sp[-1] + 1)) // 64) and plan.bn == 256
        assert (plan.ksplit > 1) == (num_cu is None)
        case = f"s{SPANS.index((n, cin, cout, in_sp, num_cu))}"
        assert (plan.tail_tiles, plan.tail_ksplit) == TAILS.get(case, (0, 1)), (case, plan.tail_tiles, plan.tail_ksplit)
    else:
        assert plan.span_pieces == 0


@pytest.mark.parametrize("case", [6, 7, 0, 2])
@pytest.mark.parametrize("bn", [True, False], ids=["bn_relu", "bias_only"])
def test_convb_single_destination(backend, case, bn):
    """One destination and nothing else: the lean epilogue of the LDS-DMA kernel's 64-position wave tiles (cases 6, 7), with
    ReLU (max(y, 0)) and without (max(y, -inf)); cases 0 and 2 take the general epilogue on the same arguments."""
    run_convb(backend, BF16, *CONVS[case], seed=9, bn=bn, raw=False)


@pytest.mark.parametrize("n,cin,cout,in_sp,num_cu", SPANS, ids=[f"s{i}" for i in range(len(SPANS))])
def test_convb_span_kernel_single_destination(backend, n, cin, cout, in_sp, num_cu):
    """One destination, bias + BN + ReLU, no raw copy and no residual: the persistent kernel's lean epilogue
    (convb_epilogue_lean: bias folded into the BN shift once per workgroup); the tail / split cases finish in the
    reduce kernel's general epilogue, on the same data."""
    k = (3,) * len(in_sp)
    run_convb(backend, BF16, n, cin, cout, in_sp, k, (1,) * len(in_sp), (1,) * len(in_sp), num_cu=num_cu, seed=5, raw=False)


def test_convb_dynamic_items_counter_slots_are_reusable(backend):
    """The persistent kernel draws its items from a per-launch counter slot that its last workgroup clears (256 slots,
    taken in turn): 260 launches of a two-items-per-workgroup case walk every slot once and the first four twice -- a slot
    left dirty would make a later launch skip items."""
    case = (4, 64, 32, (4, 16, 16), 1)       # 16 tiles on 8 workgroups, 6 groups of 9 taps: dynamic shares
    n, cin, cout, in_sp, num_cu = case
    k = (3, 3, 3)
    plan = run_convb(backend, BF16, n, cin, cout, in_sp, k, (1, 1, 1), (1, 1, 1), num_cu=num_cu, seed=11, raw=False)
    assert plan.pgrid == 8 and plan.ksplit == 1
    rng = np.random.default_rng(12)
    x = quant(rng.normal(size=(n, cin) + in_sp).astype(np.float32), BF16)
    w = quant((rng.normal(size=(cout, cin) + k) / np.sqrt(cin * 27)).astype(np.float32), BF16)
    g = hip.conv_geom(n, cin, cout, in_sp, k, (1, 1, 1), (1, 1, 1), in_sp)
    wp = np.zeros(plan.wp_vecs * 8, np.uint16)
    backend.lib.convb_pack_weights(g, plan, w.ctypes.data, wp.ctypes.data)
    S = int(np.prod(in_sp))
    xd, wd = dev_blocked(backend, x, BF16), backend.dev(wp)
    y = empty_blocked(backend, (n, cout) + in_sp, BF16)
    ep = hip.ConvEpilogue()
    ep.bias = None; ep.residual = hip.null_view(); ep.raw = hip.null_view(); ep.bn_scale = None; ep.bn_shift = None; ep.relu = 0
    ep.act = hip.View(bptr(backend, y, BF16), (cout // 8) * S, 0, S, 1)
    for _ in range(260):
        backend.lib.convb_forward(g, plan, bptr(backend, xd, BF16), backend.ptr(wd), ep, None)
    check(host_blocked(backend, y, (n, cout) + in_sp, BF16), orc.convolution(x, w, None, k, (1, 1, 1), (1, 1, 1)), BF16, "after 260 launches")


def test_convb_rejects_unblocked_geometries(backend):
    g = hip.conv_geom(1, 20, 32, (4, 4), (1, 1), (1, 1), (0, 0), (4, 4))
    with pytest.raises(hip.EcoError, match="multiple of the 8-channel block"):
        backend.lib.convb_plan(g, BF16)
    g = hip.conv_geom(1, 32, 12, (4, 4), (1, 1), (1, 1), (0, 0), (4, 4))
    with pytest.raises(hip.EcoError, match="multiple of the 8-channel block"):
        backend.lib.convb_plan(g, BF16)
    g = hip.conv_geom(1, 32, 32, (4, 4), (1, 1), (1, 1), (0, 0), (4, 4))
    with pytest.raises(hip.EcoError, match="storage type"):
        backend.lib.convb_plan(g, 2)


# ---- whole nets on the blocked path --------------------------------------------------------------------------
def _mini_lite(num_segments=4, num_clips=2):
    from eco_amd import models
    # width_div=4: every channel count stays a multiple of the 8-channel block (96/4 = 24, a partial last stage)
    return models.eco_lite_deploy(num_segments=num_segments, num_clips=num_clips, num_classes=10, input_size=32, width_div=4)


def oracle_blocked(spec, params, x, dt, stored):
    """The oracle with the storage rounding of the blocked path: weights rounded to the storage type, and every
    blob the engine materialises rounded when it is stored (`stored` = its blob names); everything else fp32."""
    if dt != BF16:
        return orc.forward(spec, params, {"data": x}, keep="all", fast_pool=False)
    qp = {k: [blocked.bf16_round(b) if (i == 0 and spec.layer(k).type == "Convolution") else b for i, b in enumerate(v)]
          for k, v in params.items()}
    return orc.forward(spec, qp, {"data": x}, keep="all", fast_pool=False,
                       store_hook=lambda name, v: blocked.bf16_round(v) if name in stored else v,
                       input_hook=lambda name, v: blocked.bf16_round(v))


@pytest.mark.parametrize("dtype,dt", [("bf16", BF16)])
def test_mini_eco_lite_blocked(backend, dtype, dt):
    from eco_amd import fillers
    from eco_amd.net import Net
    from eco_amd.netspec import NetSpec
    proto = _mini_lite()
    spec = NetSpec.from_prototxt(proto)
    params = fillers.synthetic_params(spec, seed=7)
    x = fillers.synthetic_frames(8, 32, 32, seed=3)
    kw = {"_backend": (backend.lib, backend.alloc)} if backend.kind == "emu" else {}
    net = Net(proto, params=params, dtype=dtype, **kw)
    net.blobs["data"].data[...] = x
    out = net.forward()["fc8"].copy()
    stored = {n for n, t in net._engine.tensors.items() if t.dt}
    ref = oracle_blocked(spec, params, x, dt, stored)
    scale = np.abs(ref["fc8"]).max()
    # against the oracle with the same storage rounding (accumulation order and double rounding remain)
    assert np.abs(out - ref["fc8"]).max() <= (2e-2 if dt == BF16 else 2e-5) * scale
    if dt == BF16:  # and against the plain fp32 oracle within the stated bf16 tolerance
        full = orc.forward(spec, params, {"data": x})["fc8"]
        assert np.abs(out - full).max() <= 3e-2 * np.abs(full).max()
    seen = 0
    for name, t in net._engine.tensors.items():
        got = net.blobs[name].data
        r = ref[name].reshape(got.shape)
        tol = (2e-2 if dt == BF16 else 2e-5) * (np.abs(r).max() + 1e-30)
        assert np.abs(got - r).max() <= tol, name
        seen += 1
    assert seen >= 10
    with pytest.raises(AttributeError, match="channel-blocked"):
        net.blobs["res2b_bn"].tensor
    assert any("stem pack" in l for l in net.op_labels())


def test_blocked_path_refuses_unfused_and_foreign_graphs(backend):
    from eco_amd import fillers
    from eco_amd.net import Net
    from eco_amd.netspec import NetSpec, NetSpecError
    kw = {"_backend": (backend.lib, backend.alloc)} if backend.kind == "emu" else {}
    proto = _mini_lite()
    with pytest.raises(NetSpecError, match="fused plan only"):
        Net(proto, dtype="bf16", fuse=False, **kw)
    with pytest.raises(ValueError, match="dtype"):
        Net(proto, dtype="fp16", **kw)
    lone_relu = """
    input: "data" input_dim: 1 input_dim: 3 input_dim: 32 input_dim: 32
    layer { name: "c" type: "Convolution" bottom: "data" top: "c" convolution_param { num_output: 32 kernel_size: 7 stride: 2 pad: 3 } }
    layer { name: "r" type: "ReLU" bottom: "c" top: "c" }
    """
    with pytest.raises(NetSpecError, match="no stand-alone kernel on the blocked"):
        Net(lone_relu, dtype="bf16", **kw)


def test_blocked_sibling_groups(backend):
    """bf16 path: the sibling 1x1 convs of an Inception block run as one segmented launch (eco_convb_forward with
    eco_conv_epilogue::nseg); width_div=2 keeps their widths multiples of 32."""
    from eco_amd import fillers, models
    from eco_amd.net import Net
    from eco_amd.netspec import NetSpec
    proto = models.eco_lite_deploy(num_segments=4, num_clips=1, num_classes=10, input_size=32, width_div=2)
    spec = NetSpec.from_prototxt(proto)
    params = fillers.synthetic_params(spec, seed=11)
    x = fillers.synthetic_frames(4, 32, 32, seed=5)
    kw = {"_backend": (backend.lib, backend.alloc)} if backend.kind == "emu" else {}
    net = Net(proto, params=params, dtype="bf16", _num_cu=1, **kw)
    net.blobs["data"].data[...] = x
    out = net.forward()["fc8"].copy()
    groups = [l for l in net.op_labels() if " | " in l]
    assert len(groups) >= 2 and any(l.startswith("inception_3a_1x1+") and l.count(" | ") == 2 for l in groups)
    stored = {n for n, t in net._engine.tensors.items() if t.dt}
    ref = oracle_blocked(spec, params, x, BF16, stored)
    assert np.abs(out - ref["fc8"]).max() <= 2e-2 * np.abs(ref["fc8"]).max()
    for name in ("inception_3a_output", "inception_3b_output", "inception_3a_3x3_reduce_bn", "inception_3b_double_3x3_reduce_bn"):
        if name in net._engine.tensors:
            got = net.blobs[name].data
            r = ref[name].reshape(got.shape)
            assert np.abs(got - r).max() <= 2e-2 * (np.abs(r).max() + 1e-30), name
    # the pool_proj convs run ahead of their AVE pools as fourth members (pool_commute), the pools behind them
    assert any("[ahead of inception_3b_pool]" in l for l in groups)   # (3a's projection has 16 channels at this width: not a 32-row tile)
    assert any(l.startswith("inception_3b_pool+") and "average" in l for l in net.op_labels())
    # switched off: the same logits up to bf16 rounding (the exchanged form stores the projection's raw products where the
    # layer order stores the pooled input: different intermediates, each rounded once)
    net._engine.siblings = False
    net._engine.build()
    net.blobs["data"].data[...] = x
    assert not any(" | " in l for l in net.op_labels())
    out2 = net.forward()["fc8"]
    assert np.abs(out2 - out).max() <= 1e-2 * np.abs(out).max()


# ---- the AVE pool that runs behind its 1x1 projection (eco_poolb_avg_affine_forward) -----------------------------------
@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("n,c,H,W,relu,bn,slice_", [(2, 32, 7, 7, 1, True, True), (1, 64, 28, 28, 1, True, False),
                                                    (3, 8, 5, 9, 0, False, True)])
def test_poolb_avg_affine_matches_layer_sequence(backend, dt, n, c, H, W, relu, bn, slice_):
    rng = np.random.default_rng(n * 100 + c * 10 + W)
    x = quant(rng.standard_normal((n, c, H, W)).astype(np.float32), dt)
    b = rng.standard_normal(c).astype(np.float32)
    sc, sh = rng.uniform(0.5, 1.5, c).astype(np.float32), rng.standard_normal(c).astype(np.float32)
    ref = orc.pooling(x, "AVE", (3, 3), (1, 1), (1, 1)) + b[None, :, None, None]
    if bn:
        ref = ref * sc[None, :, None, None] + sh[None, :, None, None]
    if relu:
        ref = np.maximum(ref, 0)
    S = H * W
    c0, wide = (16, c + 24) if slice_ else (0, c)        # channels [c0, c0 + c) of a wider (Concat) tensor, in blocks of 8
    big = backend.dev(blocked.to_blocked(np.full((n, wide, H, W), 7.0, np.float32), dt))
    esz = 2 if dt == BF16 else 4
    # blocked view: strides in 8-channel vectors; the base pointer is moved to block c0 / 8
    dst = hip.View(backend.alloc.ptr(big) + (c0 // 8) * S * 8 * esz, (wide // 8) * S, 0, S, 1)
    backend.lib.poolb_avg_affine_forward(dt, backend.ptr(backend.dev(blocked.to_blocked(x, dt))), backend.ptr(backend.dev(b)),
                                         backend.ptr(backend.dev(sc)) if bn else None,
                                         backend.ptr(backend.dev(sh)) if bn else None, relu, dst, n, c, H, W)
    got = blocked.from_blocked(backend.host(big, (n * wide * S,)), (n, wide, H, W), dt)
    err = np.abs(got[:, c0:c0 + c] - ref)
    tol = 2.0 ** -8 * np.abs(ref) + 2e-6 * np.abs(ref).max() if dt == BF16 else 2e-6 * np.abs(ref).max()
    assert (err <= tol).all(), float(err.max())
    assert (got[:, :c0] == 7.0).all() and (got[:, c0 + c:] == 7.0).all()


def test_poolb_avg_affine_rejects_bad_arguments(backend):
    with pytest.raises(hip.EcoError, match="poolb affine"):
        backend.lib.poolb_avg_affine_forward(BF16, 0, None, None, None, 1, hip.null_view(), 1, 12, 4, 4)


# ---- the real ECO layer geometries on the bf16 path (GPU only: too slow for the emulator) --------------------------------
# Round-4 verdict: the fp32 kernels had a per-layer test at the ECO sizes (tests/test_kernels.py ECO_CONVS), the bf16
# persistent span kernel / LDS-DMA kernel were exercised at res3 / res4 / res5 size only through whole nets.  Same exact
# bound as every other test of this file: |err| <= 2^-8 |y| (one bf16 rounding of the stored value) + 2e-5 of the largest
# output (fp32 accumulation order); operands are rounded to bf16 first, so the oracle sees what the kernel sees.
ECO_CONVS_B = [  # id, (n, cin, cout, in_sp, kernel, stride, pad), residual, raw
    ("conv2_3x3_reduce", (8, 64, 64, (56, 56), (1, 1), (1, 1), (0, 0)), False, False),          # LDS-DMA kernel, 1x1
    ("conv2_3x3", (8, 64, 192, (56, 56), (3, 3), (1, 1), (1, 1)), False, False),                # persistent span kernel, 2 groups x 9 taps
    ("inception_3a_1x1", (8, 192, 64, (28, 28), (1, 1), (1, 1), (0, 0)), False, False),
    ("inception_3a_double_3x3_1", (8, 64, 96, (28, 28), (3, 3), (1, 1), (1, 1)), False, False),
    ("inception_3b_double_3x3_2", (8, 96, 96, (28, 28), (3, 3), (1, 1), (1, 1)), False, False),
    ("res3a_2n", (1, 96, 128, (16, 28, 28), (3, 3, 3), (1, 1, 1), (1, 1, 1)), False, True),      # raw + activated: two destinations
    ("res3b_2", (2, 128, 128, (16, 28, 28), (3, 3, 3), (1, 1, 1), (1, 1, 1)), True, True),       # Eltwise residual, raw sum + BN/ReLU
    ("res4a_1", (2, 128, 256, (16, 28, 28), (3, 3, 3), (2, 2, 2), (1, 1, 1)), False, False),     # strided: descriptor-addressed DMA kernel
    ("res4b_2", (2, 256, 256, (8, 14, 14), (3, 3, 3), (1, 1, 1), (1, 1, 1)), True, True),        # K-split tail + residual
    ("res5a_down", (2, 256, 512, (8, 14, 14), (3, 3, 3), (2, 2, 2), (1, 1, 1)), False, True),
    ("res5b_2", (12, 512, 512, (4, 7, 7), (3, 3, 3), (1, 1, 1), (1, 1, 1)), True, True),         # split-K partial sums + residual
    ("res3_n32", (1, 128, 128, (32, 28, 28), (3, 3, 3), (1, 1, 1), (1, 1, 1)), False, False),    # configs[4] depth (32 planes)
]


@pytest.mark.gpu
@pytest.mark.parametrize("name,cfg,residual,raw", ECO_CONVS_B, ids=[c[0] for c in ECO_CONVS_B])
def test_convb_eco_geometries(hip_backend, name, cfg, residual, raw):
    plan = run_convb(hip_backend, BF16, *cfg, seed=4, residual=residual, raw=raw)
    n, cin, cout, in_sp, kernel, stride, pad = cfg
    if kernel[-1] == 3 and stride[-1] == 1:      # (every case above has the >= 2048 positions the span route asks for)
        assert plan.span_pieces > 0, "stride-1 3x3 layers of ECO run on the persistent span kernel"


@pytest.mark.gpu
def test_convb_eco_permuted_volume_at_full_size(hip_backend):
    """inception_3c_double_3x3_1 (64 -> 96, 28 x 28) of one 16-frame clip, written straight into the [B, C, T, H, W] volume
    the 3-D trunk reads (r2Dto3D + Permute folded into the store: reshape_layer.cpp:88, permute_layer.cpp:9-26)."""
    be, dt = hip_backend, BF16
    rng = np.random.default_rng(21)
    B, T, cin, cout, H = 2, 16, 64, 96, 28
    n, S = B * T, H * H
    x = quant(rng.normal(size=(n, cin, H, H)).astype(np.float32), dt)
    w = quant((rng.normal(size=(cout, cin, 3, 3)) / 24).astype(np.float32), dt)
    b = rng.normal(size=cout).astype(np.float32)
    sc, sh = rng.uniform(0.5, 1.5, cout).astype(np.float32), rng.normal(size=cout).astype(np.float32)
    g = hip.conv_geom(n, cin, cout, (H, H), (3, 3), (1, 1), (1, 1), (H, H))
    plan = be.lib.convb_plan(g, dt)
    wp = np.zeros(plan.wp_vecs * 8, np.uint16)
    be.lib.convb_pack_weights(g, plan, w.ctypes.data, wp.ctypes.data)
    vol = empty_blocked(be, (B, cout, T, H, H), dt)
    ep = hip.ConvEpilogue()
    ep.bias = be.ptr(be.dev(b))
    ep.residual, ep.raw = hip.null_view(), hip.null_view()
    ep.bn_scale, ep.bn_shift, ep.relu = be.ptr(be.dev(sc)), be.ptr(be.dev(sh)), 1
    ep.act = hip.View(bptr(be, vol, dt), (cout // 8) * T * S, S, T * S, T)
    ws = be.empty((max(plan.ws_bytes, 4) // 4,)) if plan.ws_bytes else None
    be.lib.convb_forward(g, plan, bptr(be, dev_blocked(be, x, dt), dt), be.ptr(be.dev(wp)), ep, be.ptr(ws) if ws is not None else None)
    ref = orc.convolution(x, w, b, (3, 3), (1, 1), (1, 1))
    ref = np.maximum(ref * sc.reshape(1, -1, 1, 1) + sh.reshape(1, -1, 1, 1), 0)
    got = host_blocked(be, vol, (B, cout, T, H, H), dt)
    check(got, ref.reshape(B, T, cout, H, H).transpose(0, 2, 1, 3, 4), dt, "permuted volume")


def test_counters_reset_and_streams(backend):
    """eco_counters_reset clears the work counters of the dynamic-share launches (include/eco_hip.h, "Process-level state");
    dynamic launches issued under different stream handles take different counter slots and all match the oracle (the emulator
    ignores the stream for execution -- the slot bookkeeping is what runs here; on the GPU the handles are real streams)."""
    lib = backend.lib
    lib.counters_reset(None)
    case = (4, 64, 32, (4, 16, 16), 1)       # 16 tiles on 8 workgroups, 6 groups of 9 taps: dynamic shares
    n, cin, cout, in_sp, num_cu = case
    for seed in (1, 2, 3):
        plan = run_convb(backend, BF16, n, cin, cout, in_sp, (3, 3, 3), (1, 1, 1), (1, 1, 1), num_cu=num_cu, seed=seed, raw=False)
        assert plan.pgrid == 8
    lib.counters_reset(None)
    run_convb(backend, BF16, n, cin, cout, in_sp, (3, 3, 3), (1, 1, 1), (1, 1, 1), num_cu=num_cu, seed=4, raw=False)
