"""Winograd F(4x4,3x3) on the dedicated transformed-domain GEMM (csrc/eco_wgemm.hip) against the CPU oracle's
direct convolution, through the C ABI: eco_wino_weight_transform -> eco_wgemm_pack_weights (host),
eco_wino_input_pk_forward -> eco_wgemm_forward -> eco_wino_output_dm_forward (device).  Tolerance as for the
round-1 Winograd route: 3e-5 of the largest output (F(4x4) transform constants amplify fp32 rounding ~10x)."""
import numpy as np
import pytest

import eco_oracle as orc
from eco_amd import hip

CASES = [  # n, cin, cout, (D,H,W), kd, num_cu
    (2, 16, 32, (1, 8, 8), 1, None),        # 2-D, one stage
    (3, 32, 96, (1, 14, 14), 1, None),      # bm = 96 (zero-padded weight rows)
    (1, 64, 64, (1, 10, 7), 1, None),       # ragged tiles (H, W not multiples of 4)
    (2, 16, 128, (4, 8, 8), 3, None),       # 3-D: depth taps direct, zero planes at both ends
    (1, 32, 160, (3, 7, 7), 3, None),       # cout 160 -> two M-blocks of 96 (second one ragged)
    (2, 64, 32, (2, 12, 12), 3, 1),         # tiny "device": no split, several rounds
    (1, 128, 64, (4, 7, 7), 3, None),       # K = 384: split-K slices summed by the output transform
]


@pytest.mark.parametrize("n,cin,cout,dims,kd,num_cu", CASES, ids=[f"w{i}" for i in range(len(CASES))])
@pytest.mark.parametrize("mode", ["plain", "fused"])
def test_wgemm_route_matches_direct_conv(backend, n, cin, cout, dims, kd, num_cu, mode):
    D, H, W = dims
    rng = np.random.default_rng(cin + cout + H)
    nd = 3 if kd == 3 else 2
    xs = (n, cin) + ((D, H, W) if nd == 3 else (H, W))
    x = rng.normal(size=xs).astype(np.float32)
    w = (rng.normal(size=(cout, cin) + (3,) * nd) / np.sqrt(cin * 3 ** nd)).astype(np.float32)
    b = rng.normal(size=cout).astype(np.float32)
    v = orc.convolution(x, w, b, (3,) * nd, (1,) * nd, (1,) * nd)
    lib = backend.lib
    TH, TW = -(-H // 4), -(-W // 4)
    plan = lib.wgemm_plan(n, cin, cout, D, TH, TW, kd, num_cu)
    u = np.empty((36, cout, cin, kd), np.float32)
    lib.wino_weight_transform(w.ctypes.data, cout, cin, kd, 4, u.ctypes.data)
    up = np.empty(plan.u_elems, np.float32)
    lib.wgemm_pack_weights(plan, u.ctypes.data, up.ctypes.data)
    vbuf, mbuf = backend.empty((plan.v_elems,)), backend.empty((plan.m_elems,))
    S = D * H * W
    y_raw, y_act = backend.empty(v.shape), backend.empty(v.shape)
    ep = hip.ConvEpilogue()
    ep.bias = backend.ptr(backend.dev(b))
    ep.residual = hip.null_view()
    ep.raw = hip.plain_view(backend.ptr(y_raw), cout, S)
    ep.bn_scale = ep.bn_shift = None
    ep.relu = 0
    ref_act = None
    if mode == "fused":
        res = rng.normal(size=v.shape).astype(np.float32)
        sc, sh = rng.uniform(0.5, 1.5, cout).astype(np.float32), rng.normal(size=cout).astype(np.float32)
        ep.residual = hip.plain_view(backend.ptr(backend.dev(res)), cout, S)
        ep.bn_scale, ep.bn_shift, ep.relu = backend.ptr(backend.dev(sc)), backend.ptr(backend.dev(sh)), 1
        ep.act = hip.plain_view(backend.ptr(y_act), cout, S)
        v = v + res
        shp = (1, -1) + (1,) * nd
        ref_act = np.maximum(v * sc.reshape(shp) + sh.reshape(shp), 0)
    lib.wino_input_pk_forward(plan, backend.ptr(backend.dev(x)), backend.ptr(vbuf), H, W)
    lib.wgemm_forward(plan, backend.ptr(vbuf), backend.ptr(backend.dev(up)), backend.ptr(mbuf))
    lib.wino_output_dm_forward(plan, backend.ptr(mbuf), H, W, ep)
    tol = 3e-5 * np.abs(v).max()
    assert np.abs(backend.host(y_raw, v.shape) - v).max() <= tol
    if ref_act is not None:
        assert np.abs(backend.host(y_act, v.shape) - ref_act).max() <= tol
    if cin * kd >= 384:
        assert plan.ksplit > 1
    assert plan.nstages == cin // 16 * kd and plan.mblocks == -(-cout // plan.bm)


def test_wgemm_plan_rejects_bad_problems(backend):
    with pytest.raises(hip.EcoError, match="multiple of 16"):
        backend.lib.wgemm_plan(1, 24, 32, 1, 2, 2, 1)
    with pytest.raises(hip.EcoError, match="kd"):
        backend.lib.wgemm_plan(1, 32, 32, 1, 2, 2, 2)


def test_wgemm_plans_for_eco_layers(backend):
    """Tile / split choices at the benchmark size (32 clips): every trunk stage fills the 512 workgroup slots."""
    lib = backend.lib
    p3 = lib.wgemm_plan(32, 128, 128, 16, 7, 7, 3)      # res3: 98 tiles of 256 x 36 points
    p4 = lib.wgemm_plan(32, 256, 256, 8, 4, 4, 3)       # res4
    p5 = lib.wgemm_plan(32, 512, 512, 4, 2, 2, 3)       # res5: 512 positions per point
    pc = lib.wgemm_plan(512, 64, 192, 1, 14, 14, 1)     # conv2_3x3: K = 64, four stages
    assert (p3.bm, p3.nstages, p3.ksplit) == (128, 24, 1) and p3.q == 18 * 32 * 49
    assert p4.bm == 128 and p4.mblocks == 2 and p5.mblocks == 4 and p5.ksplit >= 1
    assert (pc.bm, pc.mblocks, pc.nstages, pc.ksplit) == (96, 2, 4, 1) and pc.q == 512 * 196


# ---- fused transformed-domain GEMM + output transform (wfused_kernel): the short-reduction 2-D layers ---------------
WTOL = 3e-5


def relerr(got, ref):
    return float(np.abs(got - ref).max() / (np.abs(ref).max() + 1e-30))


def run_wfused(be, n, cin, cout, H, W, mode="bn", num_cu=None):
    lib = be.lib
    rng = np.random.default_rng(n * 7 + cin + cout + H)
    x = rng.standard_normal((n, cin, H, W)).astype(np.float32)
    w = (rng.standard_normal((cout, cin, 3, 3)) / np.sqrt(cin * 9)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    ref = orc.convolution(x, w, b, (3, 3), (1, 1), (1, 1))
    TH, TW = (H + 3) // 4, (W + 3) // 4
    plan = lib.wgemm_plan(n, cin, cout, 1, TH, TW, 1, num_cu)
    u = np.empty((36, cout, cin, 1), np.float32)
    lib.wino_weight_transform(w.ctypes.data, cout, cin, 1, 4, u.ctypes.data)
    up = np.zeros(lib.wfused_weight_elems(plan), np.float32)
    lib.wfused_pack_weights(plan, u.ctypes.data, up.ctypes.data)
    dx, dup, db = be.dev(x), be.dev(up), be.dev(b)
    v = be.dev(np.full(plan.v_elems, np.nan, np.float32))
    lib.wino_input_q4_forward(plan, be.ptr(dx), be.ptr(v), H, W)
    ep = hip.ConvEpilogue()
    ep.bias = be.ptr(db)
    ep.residual, ep.raw, ep.act, ep.act2 = hip.null_view(), hip.null_view(), hip.null_view(), hip.null_view()
    S = H * W
    if mode == "plain":
        y = be.empty(ref.shape)
        ep.raw = hip.plain_view(be.ptr(y), cout, S)
        lib.wfused_forward(plan, be.ptr(v), be.ptr(dup), H, W, ep)
        assert relerr(be.host(y, ref.shape), ref) < WTOL
        return plan
    # bias + residual + raw + BN + ReLU -> a channel slice of a wider (Concat) tensor
    res = rng.standard_normal(ref.shape).astype(np.float32)
    sc = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    sh = rng.standard_normal(cout).astype(np.float32)
    dres, dsc, dsh = be.dev(res), be.dev(sc), be.dev(sh)
    raw = be.empty(ref.shape)
    c0, ctot = 8, cout + 16
    big = be.dev(np.full((n, ctot, H, W), 7.0, np.float32))
    ep.residual = hip.plain_view(be.ptr(dres), cout, S)
    ep.raw = hip.plain_view(be.ptr(raw), cout, S)
    ep.act = hip.View(be.ptr(big, c0 * S), ctot * S, 0, S, 1)
    ep.bn_scale, ep.bn_shift, ep.relu = be.ptr(dsc), be.ptr(dsh), 1
    lib.wfused_forward(plan, be.ptr(v), be.ptr(dup), H, W, ep)
    exp_raw = ref + res
    exp_act = np.maximum(exp_raw * sc.reshape(1, -1, 1, 1) + sh.reshape(1, -1, 1, 1), 0)
    assert relerr(be.host(raw, ref.shape), exp_raw) < WTOL
    got = be.host(big, (n, ctot, H, W))
    assert relerr(got[:, c0:c0 + cout], exp_act) < WTOL
    assert (got[:, :c0] == 7.0).all() and (got[:, c0 + cout:] == 7.0).all()
    return plan


@pytest.mark.parametrize("n,cin,cout,H,W,mode", [
    (2, 64, 64, 28, 28, "bn"),       # inception_3a_3x3: 98 tile columns, last block of 32 ragged
    (2, 96, 96, 12, 12, "bn"),       # 96 -> 96 (double_3x3_2): three 16-k-pair chunks per point
    (1, 64, 192, 16, 16, "plain"),   # conv2_3x3's widths: six channel blocks
    (2, 64, 96, 10, 14, "bn"),       # planes that do not tile by 4 (ragged tiles), W % 4 == 2: 8-byte stores
    (2, 96, 64, 7, 9, "plain"),      # odd width: scalar stores
    (1, 160, 64, 8, 8, "bn"),        # ECO-Full inception_4c/4d widths: five chunks per point
    (1, 224, 32, 7, 7, "plain"),     # inception_5a/5b double_3x3_2: seven chunks per point, 7x7 planes
])
def test_wfused_conv(backend, n, cin, cout, H, W, mode):
    run_wfused(backend, n, cin, cout, H, W, mode)


def test_wfused_rejects_other_shapes(backend):
    lib = backend.lib
    plan = lib.wgemm_plan(1, 128, 128, 4, 4, 4, 3, None)     # a 3-D trunk layer
    with pytest.raises(hip.EcoError, match="wfused"):
        lib.wfused_pack_weights(plan, 0, 0)
    plan = lib.wgemm_plan(1, 64, 48, 1, 4, 4, 1, None)        # cout not a multiple of 32
    with pytest.raises(hip.EcoError, match="multiples of 32"):
        lib.wfused_forward(plan, 0, 0, 16, 16, hip.ConvEpilogue())


# ---- Winograd F(4x4x4,3x3x3): the 3-D trunk's route (csrc/eco_wino3.hip) around the same GEMM kernel ---------------------
W3_CASES = [  # n, cin, cout, (D,H,W), num_cu
    (2, 16, 32, (4, 8, 8), None),        # one depth tile, planes tile by 4: 16-byte accesses throughout
    (1, 32, 96, (8, 7, 7), None),        # res5-like 7x7 planes (scalar accesses), two depth tiles, bm = 96
    (3, 16, 64, (6, 14, 14), None),      # ragged depth (6 -> two tiles, second half empty), W % 4 == 2: 8-byte accesses
    (1, 64, 32, (4, 28, 28), None),      # res3-like 28x28 planes: 49 tiles per image, one image per workgroup
    (5, 16, 160, (2, 5, 9), 1),          # D < 4, odd planes, image groups with a ragged last group, two M-blocks, tiny "device"
    (9, 32, 32, (4, 4, 4), None),        # one tile per image: eight images per workgroup, last group ragged
]


@pytest.mark.parametrize("n,cin,cout,dims,num_cu", W3_CASES, ids=[f"w3_{i}" for i in range(len(W3_CASES))])
@pytest.mark.parametrize("mode", ["plain", "fused", "views"])
def test_wino3_route_matches_direct_conv(backend, n, cin, cout, dims, num_cu, mode):
    D, H, W = dims
    rng = np.random.default_rng(cin + cout + H + D)
    x = rng.normal(size=(n, cin, D, H, W)).astype(np.float32)
    w = (rng.normal(size=(cout, cin, 3, 3, 3)) / np.sqrt(cin * 27)).astype(np.float32)
    b = rng.normal(size=cout).astype(np.float32)
    v = orc.convolution(x, w, b, (3, 3, 3), (1, 1, 1), (1, 1, 1))
    lib = backend.lib
    TD, TH, TW = -(-D // 4), -(-H // 4), -(-W // 4)
    plan = lib.wgemm_plan(n, cin, cout, TD, TH, TW, 1, num_cu, points=216)
    assert plan.points == 216 and plan.q == TD * n * TH * TW and plan.nstages == cin // 16
    u = np.empty((216, cout, cin, 1), np.float32)
    lib.wino3_weight_transform(w.ctypes.data, cout, cin, u.ctypes.data)
    up = np.empty(plan.u_elems, np.float32)
    lib.wgemm_pack_weights(plan, u.ctypes.data, up.ctypes.data)
    vbuf, mbuf = backend.empty((plan.v_elems,)), backend.empty((plan.m_elems,))
    S = D * H * W
    ep = hip.ConvEpilogue()
    ep.bias = backend.ptr(backend.dev(b))
    ep.residual, ep.raw, ep.act, ep.act2 = hip.null_view(), hip.null_view(), hip.null_view(), hip.null_view()
    ep.bn_scale = ep.bn_shift = None
    ep.relu = 0
    tol = 1e-4 * np.abs(v).max()     # three nested F(4,3) transforms: measured 1-2e-5 of the largest output
    lib.wino3_input_forward(plan, backend.ptr(backend.dev(x)), backend.ptr(vbuf), D, H, W)
    lib.wgemm_forward(plan, backend.ptr(vbuf), backend.ptr(backend.dev(up)), backend.ptr(mbuf))
    shp = (1, -1, 1, 1, 1)
    if mode == "plain":
        y_raw = backend.empty(v.shape)
        ep.raw = hip.plain_view(backend.ptr(y_raw), cout, S)
        lib.wino3_output_forward(plan, backend.ptr(mbuf), D, H, W, ep)
        assert np.abs(backend.host(y_raw, v.shape) - v).max() <= tol
        return
    res = rng.normal(size=v.shape).astype(np.float32)
    sc, sh = rng.uniform(0.5, 1.5, cout).astype(np.float32), rng.normal(size=cout).astype(np.float32)
    ep.residual = hip.plain_view(backend.ptr(backend.dev(res)), cout, S)
    ep.bn_scale, ep.bn_shift = backend.ptr(backend.dev(sc)), backend.ptr(backend.dev(sh))
    ref_raw = v + res
    if mode == "fused":        # bias + Eltwise residual + raw store + BN + ReLU
        y_raw, y_act = backend.empty(v.shape), backend.empty(v.shape)
        ep.relu = 1
        ep.raw = hip.plain_view(backend.ptr(y_raw), cout, S)
        ep.act = hip.plain_view(backend.ptr(y_act), cout, S)
        lib.wino3_output_forward(plan, backend.ptr(mbuf), D, H, W, ep)
        assert np.abs(backend.host(y_raw, v.shape) - ref_raw).max() <= tol
        assert np.abs(backend.host(y_act, v.shape) - np.maximum(ref_raw * sc.reshape(shp) + sh.reshape(shp), 0)).max() <= tol
        return
    # views: no ReLU (negative values survive), the activated output into a channel slice of a wider tensor and, a second
    # time, into another one; no raw output
    c0, ctot = 3, cout + 5
    big = backend.dev(np.full((n, ctot, D, H, W), 7.0, np.float32))
    big2 = backend.dev(np.full((n, ctot, D, H, W), -3.0, np.float32))
    ep.act = hip.View(backend.ptr(big, c0 * S), ctot * S, 0, S, 1)
    ep.act2 = hip.View(backend.ptr(big2, 1 * S), ctot * S, 0, S, 1)
    lib.wino3_output_forward(plan, backend.ptr(mbuf), D, H, W, ep)
    ref_act = ref_raw * sc.reshape(shp) + sh.reshape(shp)
    assert (ref_act < 0).any()
    got, got2 = backend.host(big, (n, ctot, D, H, W)), backend.host(big2, (n, ctot, D, H, W))
    assert np.abs(got[:, c0:c0 + cout] - ref_act).max() <= tol and np.abs(got2[:, 1:1 + cout] - ref_act).max() <= tol
    assert (got[:, :c0] == 7.0).all() and (got[:, c0 + cout:] == 7.0).all()
    assert (got2[:, :1] == -3.0).all() and (got2[:, 1 + cout:] == -3.0).all()


def test_wino3_weight_transform_is_the_kronecker_cube_of_g(backend):
    """u[(pz, py, px)] = sum G[pz][kz] G[py][ky] G[px][kx] w[kz][ky][kx] (Lavin & Gray 2015, F(4,3)), in float64."""
    G = np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6],
                  [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]])
    rng = np.random.default_rng(5)
    w = rng.normal(size=(3, 2, 3, 3, 3)).astype(np.float32)
    u = np.empty((216, 3, 2), np.float32)
    backend.lib.wino3_weight_transform(w.ctypes.data, 3, 2, u.ctypes.data)
    ref = np.einsum("az,by,cx,oizyx->abcoi", G, G, G, w.astype(np.float64)).reshape(216, 3, 2)
    assert np.abs(u - ref).max() < 1e-6


def test_wino3_rejects_wrong_plans(backend):
    lib = backend.lib
    p36 = lib.wgemm_plan(1, 32, 32, 4, 2, 2, 3)
    with pytest.raises(hip.EcoError, match="216"):
        lib.wino3_input_forward(p36, 0, 0, 4, 8, 8)
    p = lib.wgemm_plan(1, 32, 32, 1, 2, 2, 1, None, points=216)
    with pytest.raises(hip.EcoError, match="tiles"):
        lib.wino3_input_forward(p, 0, 0, 8, 8, 8)       # two depth tiles, plan has one
    with pytest.raises(hip.EcoError, match="kd"):
        lib.wgemm_plan(1, 32, 32, 1, 2, 2, 3, None, points=216)
    assert lib.wino3_lds_bytes(32, 7, 7) == 6 * 2 * 1 * 30 * 36 * 4


# ---- fused 2-D Winograd conv + the MAX 3x3 / 2 pooling that follows it (conv2_3x3 -> pool2) --------------------------------
@pytest.mark.parametrize("n,cin,cout,H,W,relu", [
    (2, 64, 64, 8, 8, True),         # 2 x 2 tiles per image: every pooled cell kind (alone / two tiles / four tiles / at an edge)
    (3, 64, 96, 12, 16, True),       # 3 x 4 tiles, 36 tile columns: two column blocks, the second ragged
    (1, 96, 32, 16, 12, False),      # no ReLU: negative maxima survive (nothing assumes values >= 0)
    (2, 64, 192, 56, 56, True),      # conv2_3x3 itself (two frames)
])
def test_wfused_conv_with_pooling(backend, n, cin, cout, H, W, relu):
    if backend.kind == "emu" and H >= 56:
        pytest.skip("full-size planes: GPU only")
    lib = backend.lib
    rng = np.random.default_rng(n + cin + cout + H)
    x = rng.standard_normal((n, cin, H, W)).astype(np.float32)
    w = (rng.standard_normal((cout, cin, 3, 3)) / np.sqrt(cin * 9)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    sc = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    sh = rng.standard_normal(cout).astype(np.float32)
    conv = orc.convolution(x, w, b, (3, 3), (1, 1), (1, 1)) * sc.reshape(1, -1, 1, 1) + sh.reshape(1, -1, 1, 1)
    if relu:
        conv = np.maximum(conv, 0)
    ref = orc.pooling(conv.astype(np.float32), "MAX", (3, 3), (2, 2), (0, 0))
    assert ref.shape == (n, cout, H // 2, W // 2)
    TH, TW = H // 4, W // 4
    plan = lib.wgemm_plan(n, cin, cout, 1, TH, TW, 1, None)
    u = np.empty((36, cout, cin, 1), np.float32)
    lib.wino_weight_transform(w.ctypes.data, cout, cin, 1, 4, u.ctypes.data)
    up = np.zeros(lib.wfused_weight_elems(plan), np.float32)
    lib.wfused_pack_weights(plan, u.ctypes.data, up.ctypes.data)
    v = backend.dev(np.full(plan.v_elems, np.nan, np.float32))
    lib.wino_input_q4_forward(plan, backend.ptr(backend.dev(x)), backend.ptr(v), H, W)
    ep = hip.ConvEpilogue()
    ep.bias = backend.ptr(backend.dev(b))
    ep.residual, ep.raw, ep.act, ep.act2 = hip.null_view(), hip.null_view(), hip.null_view(), hip.null_view()
    ep.bn_scale, ep.bn_shift, ep.relu = backend.ptr(backend.dev(sc)), backend.ptr(backend.dev(sh)), int(relu)
    assert lib.wfused_pool_scratch_elems(plan) == n * cout * 9 * TH * TW
    scratch = backend.empty((lib.wfused_pool_scratch_elems(plan),))
    y = backend.empty(ref.shape)
    lib.wfused_pool_forward(plan, backend.ptr(v), backend.ptr(backend.dev(up)), H, W, ep, backend.ptr(scratch), backend.ptr(y))
    got = backend.host(y, ref.shape)
    assert np.isfinite(got).all() and relerr(got, ref) < WTOL
    if not relu:
        assert (ref < 0).any()


def test_wfused_pooling_rejects_ragged_planes_and_views(backend):
    lib = backend.lib
    plan = lib.wgemm_plan(1, 64, 32, 1, 3, 3, 1, None)        # 10 x 10 planes: 3 x 3 tiles, the last ones ragged
    ep = hip.ConvEpilogue()
    ep.residual, ep.raw, ep.act, ep.act2 = hip.null_view(), hip.null_view(), hip.null_view(), hip.null_view()
    with pytest.raises(hip.EcoError, match="tile by 4"):
        lib.wfused_pool_forward(plan, 16, 16, 10, 10, ep, 16, 16)
    plan = lib.wgemm_plan(1, 64, 32, 1, 2, 2, 1, None)
    ep.act = hip.plain_view(64, 32, 64)
    with pytest.raises(hip.EcoError, match="only the pooled activation"):
        lib.wfused_pool_forward(plan, 16, 16, 8, 8, ep, 16, 16)


# ---- the trunk's layers at their real sizes on the F(4x4x4,3x3x3) route (GPU only) -------------------------------------------
ECO_TRUNK = [  # id, n, cin, cout, (D,H,W): models_ECO_Lite/kinetics/deploy.prototxt:1162-1660 at num_segments 16 / 32
    ("res3a_2n", 2, 96, 128, (16, 28, 28)),
    ("res3b", 2, 128, 128, (16, 28, 28)),
    ("res4b", 3, 256, 256, (8, 14, 14)),        # two images per workgroup, the last group ragged
    ("res5b", 9, 512, 512, (4, 7, 7)),          # eight images per workgroup, 7 x 7 planes: scalar accesses
    ("res3_n32", 1, 128, 128, (32, 28, 28)),    # configs[4]'s depth: eight depth tiles
]


@pytest.mark.gpu
@pytest.mark.parametrize("name,n,cin,cout,dims", ECO_TRUNK, ids=[c[0] for c in ECO_TRUNK])
def test_wino3_eco_trunk_layers(hip_backend, name, n, cin, cout, dims):
    """Input transform -> 216 GEMMs -> output transform with the res*b_2 epilogue (Eltwise residual, raw sum, BN + ReLU) against the
    oracle's direct convolution.  Tolerance 1e-4 of the largest output (measured 1-2e-5: three nested F(4,3) transforms)."""
    be = hip_backend
    D, H, W = dims
    rng = np.random.default_rng(len(name) + cin)
    x = np.maximum(rng.normal(size=(n, cin, D, H, W)), 0).astype(np.float32)     # a ReLU'd activation, as in the net
    w = (rng.normal(size=(cout, cin, 3, 3, 3)) / np.sqrt(cin * 27)).astype(np.float32)
    b = rng.normal(size=cout).astype(np.float32)
    res = rng.normal(size=(n, cout, D, H, W)).astype(np.float32)
    sc, sh = rng.uniform(0.5, 1.5, cout).astype(np.float32), rng.normal(size=cout).astype(np.float32)
    v = orc.convolution(x, w, b, (3, 3, 3), (1, 1, 1), (1, 1, 1)) + res
    act = np.maximum(v * sc.reshape(1, -1, 1, 1, 1) + sh.reshape(1, -1, 1, 1, 1), 0)
    lib = be.lib
    TD, TH, TW = -(-D // 4), -(-H // 4), -(-W // 4)
    plan = lib.wgemm_plan(n, cin, cout, TD, TH, TW, 1, None, points=216)
    u = np.empty((216, cout, cin, 1), np.float32)
    lib.wino3_weight_transform(w.ctypes.data, cout, cin, u.ctypes.data)
    up = np.empty(plan.u_elems, np.float32)
    lib.wgemm_pack_weights(plan, u.ctypes.data, up.ctypes.data)
    vbuf, mbuf = be.empty((plan.v_elems,)), be.empty((plan.m_elems,))
    S = D * H * W
    y_raw, y_act = be.empty(v.shape), be.empty(v.shape)
    ep = hip.ConvEpilogue()
    ep.bias = be.ptr(be.dev(b))
    ep.residual = hip.plain_view(be.ptr(be.dev(res)), cout, S)
    ep.raw = hip.plain_view(be.ptr(y_raw), cout, S)
    ep.act = hip.plain_view(be.ptr(y_act), cout, S)
    ep.act2 = hip.null_view()
    ep.bn_scale, ep.bn_shift, ep.relu = be.ptr(be.dev(sc)), be.ptr(be.dev(sh)), 1
    lib.wino3_input_forward(plan, be.ptr(be.dev(x)), be.ptr(vbuf), D, H, W)
    lib.wgemm_forward(plan, be.ptr(vbuf), be.ptr(be.dev(up)), be.ptr(mbuf))
    lib.wino3_output_forward(plan, be.ptr(mbuf), D, H, W, ep)
    tol = 1e-4 * np.abs(v).max()
    e_raw, e_act = np.abs(be.host(y_raw, v.shape) - v).max(), np.abs(be.host(y_act, v.shape) - act).max()
    print(f"{name}: raw {e_raw / np.abs(v).max():.2e}, act {e_act / np.abs(v).max():.2e} of the largest output")
    assert e_raw <= tol and e_act <= tol


# ---- stride-2 3x3x3 convolutions as polyphase F(4,2) x F(7,2) x F(7,2) problems (csrc/eco_wino_s2.hip) ------------------------
S2_CASES = [  # n, cin, couts (members sharing v and the GEMM), OUTPUT volume (Do, Ho, Wo), num_cu
    (1, 2, (32,), (4, 7, 7), None),          # one tile per image, K = 16: one stage, one depth tile per workgroup
    (2, 4, (32, 64), (8, 7, 14), None),      # two members (first conv + projection shortcut), two depth tiles, two column tiles
    (3, 2, (96,), (12, 14, 7), 1),           # three depth tiles (groups of two: the second ragged), bm = 96, tiny "device"
]


def run_wino_s2(backend, n, cin, couts, odims, num_cu, mode, relu_x=False, tol=5e-4):
    Do, Ho, Wo = odims
    D, H, W = 2 * Do, 2 * Ho, 2 * Wo
    rng = np.random.default_rng(cin + sum(couts) + Ho + Do)
    x = rng.normal(size=(n, cin, D, H, W)).astype(np.float32)
    if relu_x:
        x = np.maximum(x, 0)
    ctot = sum(couts)
    w = (rng.normal(size=(ctot, cin, 3, 3, 3)) / np.sqrt(cin * 27)).astype(np.float32)
    b = rng.normal(size=ctot).astype(np.float32)
    ref = orc.convolution(x, w, b, (3, 3, 3), (2, 2, 2), (1, 1, 1))
    assert ref.shape == (n, ctot, Do, Ho, Wo)
    lib = backend.lib
    TD, TH, TW = Do // 4, Ho // 7, Wo // 7
    plan = lib.wgemm_plan(n, 8 * cin, ctot, TD, TH, TW, 1, num_cu, points=320)
    assert plan.points == 320 and plan.q == n * TD * TH * TW and plan.nstages == cin // 2
    u = np.empty((320, ctot, 8 * cin, 1), np.float32)
    lib.wino_s2_weight_transform(w.ctypes.data, ctot, cin, u.ctypes.data)
    up = np.empty(plan.u_elems, np.float32)
    lib.wgemm_pack_weights(plan, u.ctypes.data, up.ctypes.data)
    vbuf = backend.dev(np.full(plan.v_elems, np.nan, np.float32))
    mbuf = backend.empty((plan.m_elems,))
    lib.wino_s2_input_forward(plan, backend.ptr(backend.dev(x)), backend.ptr(vbuf), D, H, W)
    lib.wgemm_forward(plan, backend.ptr(vbuf), backend.ptr(backend.dev(up)), backend.ptr(mbuf))
    S = Do * Ho * Wo
    shp = (1, -1, 1, 1, 1)
    worst = 0.0
    c0 = 0
    for k, cout in enumerate(couts):
        v = ref[:, c0:c0 + cout]
        scale = np.abs(v).max()
        ep = hip.ConvEpilogue()
        ep.bias = backend.ptr(backend.dev(b[c0:c0 + cout]))
        ep.residual, ep.raw, ep.act, ep.act2 = hip.null_view(), hip.null_view(), hip.null_view(), hip.null_view()
        ep.bn_scale = ep.bn_shift = None
        ep.relu = 0
        if mode == "plain" or (mode == "fused" and k % 2 == 1):     # (a shortcut keeps its raw value)
            y_raw = backend.empty(v.shape)
            ep.raw = hip.plain_view(backend.ptr(y_raw), cout, S)
            lib.wino_s2_output_forward(plan, backend.ptr(mbuf), c0, cout, Do, Ho, Wo, ep)
            worst = max(worst, np.abs(backend.host(y_raw, v.shape) - v).max() / scale)
        else:
            res = rng.normal(size=v.shape).astype(np.float32)
            sc, sh = rng.uniform(0.5, 1.5, cout).astype(np.float32), rng.normal(size=cout).astype(np.float32)
            ep.residual = hip.plain_view(backend.ptr(backend.dev(res)), cout, S)
            ep.bn_scale, ep.bn_shift = backend.ptr(backend.dev(sc)), backend.ptr(backend.dev(sh))
            ref_raw = v + res
            if mode == "fused":        # bias + Eltwise residual + raw store + BN + ReLU
                y_raw, y_act = backend.empty(v.shape), backend.empty(v.shape)
                ep.relu = 1
                ep.raw = hip.plain_view(backend.ptr(y_raw), cout, S)
                ep.act = hip.plain_view(backend.ptr(y_act), cout, S)
                lib.wino_s2_output_forward(plan, backend.ptr(mbuf), c0, cout, Do, Ho, Wo, ep)
                worst = max(worst, np.abs(backend.host(y_raw, v.shape) - ref_raw).max() / scale,
                            np.abs(backend.host(y_act, v.shape) - np.maximum(ref_raw * sc.reshape(shp) + sh.reshape(shp), 0)).max() / scale)
            else:                      # views: no ReLU, the activated output into channel slices of two wider tensors, no raw output
                o0, wide = 3, cout + 5
                big = backend.dev(np.full((n, wide, Do, Ho, Wo), 7.0, np.float32))
                big2 = backend.dev(np.full((n, wide, Do, Ho, Wo), -3.0, np.float32))
                ep.act = hip.View(backend.ptr(big, o0 * S), wide * S, 0, S, 1)
                ep.act2 = hip.View(backend.ptr(big2, 1 * S), wide * S, 0, S, 1)
                lib.wino_s2_output_forward(plan, backend.ptr(mbuf), c0, cout, Do, Ho, Wo, ep)
                ref_act = ref_raw * sc.reshape(shp) + sh.reshape(shp)
                assert (ref_act < 0).any()
                got, got2 = backend.host(big, (n, wide, Do, Ho, Wo)), backend.host(big2, (n, wide, Do, Ho, Wo))
                worst = max(worst, np.abs(got[:, o0:o0 + cout] - ref_act).max() / scale,
                            np.abs(got2[:, 1:1 + cout] - ref_act).max() / scale)
                assert (got[:, :o0] == 7.0).all() and (got[:, o0 + cout:] == 7.0).all()
                assert (got2[:, :1] == -3.0).all() and (got2[:, 1 + cout:] == -3.0).all()
        c0 += cout
    assert worst <= tol, worst
    return worst


@pytest.mark.parametrize("n,cin,couts,odims,num_cu", S2_CASES, ids=[f"s2_{i}" for i in range(len(S2_CASES))])
@pytest.mark.parametrize("mode", ["plain", "fused", "views"])
def test_wino_s2_route_matches_direct_conv(backend, n, cin, couts, odims, num_cu, mode):
    """Input transform -> 320 GEMMs (K = 8 cin) -> one output transform per member against the oracle's direct stride-2
    convolution.  Tolerance 5e-4 of the largest output (measured ~1e-4 at K = 1024: two nested eight-point transforms)."""
    run_wino_s2(backend, n, cin, couts, odims, num_cu, mode)


def test_wino_s2_weight_transform_is_the_polyphase_kronecker_product(backend):
    """u[(az, ay, ax)][co][((ci*2 + fz)*2 + fy)*2 + fx] = sum G5[az][tz] G8[ay][ty] G8[ax][tx] g_f[tz][ty][tx] with, per axis,
    g_1 = (w0, w2) (odd phase) and g_0 = (0, w1) (even phase); G = Cook-Toom evaluation rows over the row scales of B^T."""
    G5 = np.array([[1, 0], [1 / 2, 1 / 2], [1 / 6, -1 / 6], [8 / 3, 4 / 3], [0, 1 / 2]])
    G8 = np.array([[1, 0], [-2 / 9, -2 / 9], [-2 / 9, 2 / 9], [1 / 90, 2 / 90], [1 / 90, -2 / 90], [32 / 45, 16 / 45],
                   [32 / 45, -16 / 45], [0, 1]])
    rng = np.random.default_rng(7)
    w = rng.normal(size=(3, 2, 3, 3, 3)).astype(np.float32)
    u = np.empty((320, 3, 16), np.float32)
    backend.lib.wino_s2_weight_transform(w.ctypes.data, 3, 2, u.ctypes.data)
    sel = {1: [0, 2], 0: [1, 1]}
    ref = np.empty((5, 8, 8, 3, 2, 2, 2, 2))
    for fz in (0, 1):
        for fy in (0, 1):
            for fx in (0, 1):
                g = w.astype(np.float64)[:, :, sel[fz]][:, :, :, sel[fy]][:, :, :, :, sel[fx]].copy()
                if not fz:
                    g[:, :, 0] = 0
                if not fy:
                    g[:, :, :, 0] = 0
                if not fx:
                    g[:, :, :, :, 0] = 0
                ref[:, :, :, :, :, fz, fy, fx] = np.einsum("az,by,cx,oizyx->abcoi", G5, G8, G8, g)
    assert np.abs(u - ref.reshape(320, 3, 16)).max() < 1e-6


def test_wino_s2_rejects_wrong_plans_and_volumes(backend):
    lib = backend.lib
    p216 = lib.wgemm_plan(1, 32, 32, 1, 2, 2, 1, None, points=216)
    with pytest.raises(hip.EcoError, match="320"):
        lib.wino_s2_input_forward(p216, 0, 0, 8, 14, 14)
    p = lib.wgemm_plan(1, 32, 32, 1, 1, 1, 1, None, points=320)
    with pytest.raises(hip.EcoError, match="even"):
        lib.wino_s2_input_forward(p, 0, 0, 7, 14, 14)
    with pytest.raises(hip.EcoError, match="4x7x7"):
        lib.wino_s2_input_forward(p, 0, 0, 8, 16, 14)
    with pytest.raises(hip.EcoError, match="tiles"):
        lib.wino_s2_input_forward(p, 0, 0, 16, 14, 14)    # two depth tiles, plan has one
    ep = hip.ConvEpilogue()
    ep.residual, ep.raw, ep.act, ep.act2 = hip.null_view(), hip.plain_view(64, 8, 196), hip.null_view(), hip.null_view()
    with pytest.raises(hip.EcoError, match="channels"):
        lib.wino_s2_output_forward(p, 64, 28, 8, 4, 7, 7, ep)
    assert lib.wino_s2_lds_bytes(32, 2, 2, 2) == 2 * 10 * 30 * 32 * 4 and lib.wino_s2_lds_bytes(1, 1, 1, 1) == 10 * 16 * 20 * 4


S2_ECO = [  # id, n, cin, couts, OUTPUT volume: models_ECO_Lite/kinetics/deploy.prototxt:1262-1330 at num_segments 16 / 32
    ("res4a", 2, 128, (256, 256), (8, 14, 14)),
    ("res4a_n32", 1, 128, (256,), (16, 14, 14)),
]


@pytest.mark.gpu
@pytest.mark.parametrize("name,n,cin,couts,odims", S2_ECO, ids=[c[0] for c in S2_ECO])
def test_wino_s2_eco_trunk_layers(hip_backend, name, n, cin, couts, odims):
    """res4a_1 | res4a_down at their real size (K = 1024, 512 GEMM rows) on a ReLU'd input, the first member with the
    Eltwise residual + raw + BN + ReLU epilogue, against the oracle's direct convolution."""
    worst = run_wino_s2(hip_backend, n, cin, couts, odims, None, "fused", relu_x=True)
    print(f"{name}: {worst:.2e} of the largest output")


# ---- the 2-D form: F(7,2) x F(7,2) on 64 points, depth taps direct (kz = 3) or none (kz = 1) ---------------------------------
S2D_CASES = [  # n, cin, couts, kz, OUTPUT volume (Do, Ho, Wo), num_cu
    (2, 4, (32, 32), 3, (2, 7, 7), None),      # res5a-like: one tile per plane, depth taps incl. the plane above the volume
    (3, 8, (64,), 1, (1, 14, 7), None),        # a strided 2-D 3x3 conv (D = 1): K = 4 cin
    (5, 4, (96,), 3, (3, 7, 14), 1),           # odd output depth, ragged image groups, bm = 96, tiny "device"
    (2, 4, (32,), 1, (4, 7, 7), None),         # a (1,3,3) kernel with stride (1,2,2) on a 3-D blob: planes are positions
]


def run_wino_s2d(backend, n, cin, couts, kz, odims, num_cu, mode, relu_x=False, tol=5e-4):
    Do, Ho, Wo = odims
    D, H, W = (2 * Do if kz == 3 else Do), 2 * Ho, 2 * Wo
    rng = np.random.default_rng(cin + sum(couts) + Ho + Do + kz)
    x = rng.normal(size=(n, cin, D, H, W)).astype(np.float32)
    if relu_x:
        x = np.maximum(x, 0)
    ctot = sum(couts)
    w = (rng.normal(size=(ctot, cin, kz, 3, 3)) / np.sqrt(cin * 9 * kz)).astype(np.float32)
    b = rng.normal(size=ctot).astype(np.float32)
    ref = orc.convolution(x, w, b, (kz, 3, 3), (2 if kz == 3 else 1, 2, 2), (1 if kz == 3 else 0, 1, 1))
    assert ref.shape == (n, ctot, Do, Ho, Wo)
    lib = backend.lib
    TH, TW = Ho // 7, Wo // 7
    K = 4 * kz * cin
    plan = lib.wgemm_plan(n, K, ctot, Do, TH, TW, 1, num_cu, points=64)
    assert plan.points == 64 and plan.q == n * Do * TH * TW and plan.nstages == K // 16
    u = np.empty((64, ctot, K, 1), np.float32)
    lib.wino_s2d_weight_transform(w.ctypes.data, ctot, cin, kz, u.ctypes.data)
    up = np.empty(plan.u_elems, np.float32)
    lib.wgemm_pack_weights(plan, u.ctypes.data, up.ctypes.data)
    vbuf = backend.dev(np.full(plan.v_elems, np.nan, np.float32))
    mbuf = backend.empty((plan.m_elems,))
    lib.wino_s2d_input_forward(plan, backend.ptr(backend.dev(x)), backend.ptr(vbuf), kz, D, H, W)
    lib.wgemm_forward(plan, backend.ptr(vbuf), backend.ptr(backend.dev(up)), backend.ptr(mbuf))
    S = Do * Ho * Wo
    shp = (1, -1, 1, 1, 1)
    worst, c0 = 0.0, 0
    for k, cout in enumerate(couts):
        v = ref[:, c0:c0 + cout]
        scale = np.abs(v).max()
        ep = hip.ConvEpilogue()
        ep.bias = backend.ptr(backend.dev(b[c0:c0 + cout]))
        ep.residual, ep.raw, ep.act, ep.act2 = hip.null_view(), hip.null_view(), hip.null_view(), hip.null_view()
        ep.bn_scale = ep.bn_shift = None
        ep.relu = 0
        if mode == "plain" or k % 2 == 1:
            y_raw = backend.empty(v.shape)
            ep.raw = hip.plain_view(backend.ptr(y_raw), cout, S)
            lib.wino_s2d_output_forward(plan, backend.ptr(mbuf), c0, cout, Do, Ho, Wo, ep)
            worst = max(worst, np.abs(backend.host(y_raw, v.shape) - v).max() / scale)
        else:   # bias + residual + BN + ReLU into a channel slice of a wider (Concat) tensor, raw beside it
            res = rng.normal(size=v.shape).astype(np.float32)
            sc, sh = rng.uniform(0.5, 1.5, cout).astype(np.float32), rng.normal(size=cout).astype(np.float32)
            o0, wide = 8, cout + 24
            big = backend.dev(np.full((n, wide, Do, Ho, Wo), 7.0, np.float32))
            y_raw = backend.empty(v.shape)
            ep.residual = hip.plain_view(backend.ptr(backend.dev(res)), cout, S)
            ep.raw = hip.plain_view(backend.ptr(y_raw), cout, S)
            ep.bn_scale, ep.bn_shift, ep.relu = backend.ptr(backend.dev(sc)), backend.ptr(backend.dev(sh)), 1
            ep.act = hip.View(backend.ptr(big, o0 * S), wide * S, 0, S, 1)
            lib.wino_s2d_output_forward(plan, backend.ptr(mbuf), c0, cout, Do, Ho, Wo, ep)
            ref_raw = v + res
            got = backend.host(big, (n, wide, Do, Ho, Wo))
            worst = max(worst, np.abs(backend.host(y_raw, v.shape) - ref_raw).max() / scale,
                        np.abs(got[:, o0:o0 + cout] - np.maximum(ref_raw * sc.reshape(shp) + sh.reshape(shp), 0)).max() / scale)
            assert (got[:, :o0] == 7.0).all() and (got[:, o0 + cout:] == 7.0).all()
        c0 += cout
    assert worst <= tol, worst
    return worst


@pytest.mark.parametrize("n,cin,couts,kz,odims,num_cu", S2D_CASES, ids=[f"s2d_{i}" for i in range(len(S2D_CASES))])
@pytest.mark.parametrize("mode", ["plain", "fused"])
def test_wino_s2d_route_matches_direct_conv(backend, n, cin, couts, kz, odims, num_cu, mode):
    run_wino_s2d(backend, n, cin, couts, kz, odims, num_cu, mode)


def test_wino_s2d_rejects_wrong_plans(backend):
    lib = backend.lib
    p320 = lib.wgemm_plan(1, 32, 32, 1, 1, 1, 1, None, points=320)
    with pytest.raises(hip.EcoError, match="64"):
        lib.wino_s2d_input_forward(p320, 0, 0, 3, 8, 14, 14)
    p = lib.wgemm_plan(1, 48, 32, 4, 1, 1, 1, None, points=64)
    with pytest.raises(hip.EcoError, match="kz"):
        lib.wino_s2d_input_forward(p, 0, 0, 2, 8, 14, 14)
    with pytest.raises(hip.EcoError, match="planes"):
        lib.wino_s2d_input_forward(p, 0, 0, 3, 4, 14, 14)        # two output planes, the plan has four
    assert lib.wino_s2d_lds_bytes(32, 3, 4, 1, 1) == 8 * 8 * 16 * 20 * 4    # eight images' eight planes


S2D_ECO = [  # id, n, cin, couts, kz, OUTPUT volume
    ("res5a", 8, 256, (512, 512), 3, (4, 7, 7)),                 # models_ECO_Lite/kinetics/deploy.prototxt:1462-1530
    ("inception_3c_3x3", 32, 128, (160,), 1, (1, 14, 14)),       # models_ECO_Full/kinetics/deploy.prototxt:1854-1990
    ("inception_4e_double_3x3_2", 64, 256, (256,), 1, (1, 7, 7)),
]


@pytest.mark.gpu
@pytest.mark.parametrize("name,n,cin,couts,kz,odims", S2D_ECO, ids=[c[0] for c in S2D_ECO])
def test_wino_s2d_eco_layers(hip_backend, name, n, cin, couts, kz, odims):
    worst = run_wino_s2d(hip_backend, n, cin, couts, kz, odims, None, "fused", relu_x=True)
    print(f"{name}: {worst:.2e} of the largest output")
