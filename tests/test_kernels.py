"""Parity of every C-ABI operator against the CPU oracle, on both backends:
`emu` (CPU fiber emulator, small shapes, runs in the CPU suite) and `hip` (real MI355X,
`-m gpu`, larger shapes incl. the real ECO layer geometries).

Tolerance: fp32 results, relative to the tensor's max magnitude, 1e-5 (summation order is
the only difference between the MFMA fmaf chain and the oracle's sgemm)."""
import numpy as np
import pytest

import eco_oracle as orc
from eco_amd import hip

TOL = 1e-5


def relerr(got, ref):
    return float(np.abs(got - ref).max() / (np.abs(ref).max() + 1e-30))


def run_conv(be, n, cin, cout, insp, k, s, p, mode="plain", seed=0, num_cu=None, tweak=None):
    rng = np.random.default_rng(seed)
    nd = len(insp)
    x = rng.standard_normal((n, cin) + tuple(insp)).astype(np.float32)
    w = (rng.standard_normal((cout, cin) + tuple(k)) / np.sqrt(cin * np.prod(k))).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    ref = orc.convolution(x, w, b, k, s, p)
    outsp = ref.shape[2:]
    S = int(np.prod(outsp))
    lib = be.lib
    g = hip.conv_geom(n, cin, cout, insp, k, s, p, outsp)
    plan = lib.conv_plan(g, num_cu)
    if tweak:
        tweak(plan)
    wp = np.zeros(plan.wp_elems, np.float32)
    kt = np.zeros(plan.ktab_elems, np.int32)
    lib.conv_pack_weights(g, plan, w.ctypes.data, wp.ctypes.data, kt.ctypes.data)
    dx, dwp, dkt, db = be.dev(x), be.dev(wp), be.dev(kt), be.dev(b)
    ws = be.ptr(be.empty((plan.ws_bytes // 4,))) if plan.ws_bytes else None
    ep = hip.ConvEpilogue()
    ep.bias = be.ptr(db)
    ep.residual, ep.raw, ep.act = hip.null_view(), hip.null_view(), hip.null_view()
    ep.bn_scale = ep.bn_shift = None
    ep.relu = 0
    bshape = (1, cout) + (1,) * nd
    if mode == "plain":
        raw = be.empty(ref.shape)
        ep.raw = hip.plain_view(be.ptr(raw), cout, S)
        lib.conv_forward(g, plan, be.ptr(dx), be.ptr(dwp), be.ptr(dkt), ep, ws)
        assert relerr(be.host(raw, ref.shape), ref) < TOL
    elif mode == "fused":  # bias + residual + raw + BN + ReLU -> act
        res = rng.standard_normal(ref.shape).astype(np.float32)
        sc = rng.uniform(0.5, 1.5, cout).astype(np.float32)
        sh = rng.standard_normal(cout).astype(np.float32)
        dres, dsc, dsh = be.dev(res), be.dev(sc), be.dev(sh)
        raw, act = be.empty(ref.shape), be.empty(ref.shape)
        ep.residual = hip.plain_view(be.ptr(dres), cout, S)
        ep.raw = hip.plain_view(be.ptr(raw), cout, S)
        ep.act = hip.plain_view(be.ptr(act), cout, S)
        ep.bn_scale, ep.bn_shift, ep.relu = be.ptr(dsc), be.ptr(dsh), 1
        lib.conv_forward(g, plan, be.ptr(dx), be.ptr(dwp), be.ptr(dkt), ep, ws)
        exp_raw = ref + res
        exp_act = np.maximum(exp_raw * sc.reshape(bshape) + sh.reshape(bshape), 0)
        assert relerr(be.host(raw, ref.shape), exp_raw) < TOL
        assert relerr(be.host(act, ref.shape), exp_act) < TOL
    elif mode == "concat":  # act-only, written into channels [c0, c0+cout) of a wider tensor
        c0, ctot = 3, cout + 7
        big = be.dev(np.full((n, ctot) + tuple(outsp), 7.0, np.float32))
        ep.act = hip.View(be.ptr(big, c0 * S), ctot * S, 0, S, 1)
        ep.relu = 1
        lib.conv_forward(g, plan, be.ptr(dx), be.ptr(dwp), be.ptr(dkt), ep, ws)
        got = be.host(big, (n, ctot) + tuple(outsp))
        assert relerr(got[:, c0:c0 + cout], np.maximum(ref, 0)) < TOL
        assert (got[:, :c0] == 7.0).all() and (got[:, c0 + cout:] == 7.0).all()
    elif mode == "permute":  # [B*T, C, H, W] conv output stored as [B, C, T, H, W]
        T = 2
        assert n % T == 0 and nd == 2
        out = be.empty(ref.shape)
        ep.act = hip.View(be.ptr(out), cout * T * S, S, T * S, T)
        lib.conv_forward(g, plan, be.ptr(dx), be.ptr(dwp), be.ptr(dkt), ep, ws)
        exp = ref.reshape((n // T, T, cout) + tuple(outsp)).transpose(0, 2, 1, 3, 4)
        assert relerr(be.host(out, exp.shape), exp) < TOL
    return plan


# (n, cin, cout, in spatial, kernel, stride, pad) -- small enough for the emulator.
SMALL_CONVS = [
    (2, 3, 4, (6, 4), (3, 3), (2, 2), (0, 0)),             # reference TestSimpleConvolution shape
    (2, 3, 4, (5, 6, 4), (3, 3, 3), (2, 2, 2), (0, 0, 0)),  # reference TestSimple3DConvolution shape
    (2, 3, 4, (6, 4), (1, 1), (1, 1), (0, 0)),             # reference Test1x1Convolution
    (1, 8, 128, (4, 7, 7), (3, 3, 3), (1, 1, 1), (1, 1, 1)),  # bm=128 variant, res-block geometry
    (1, 6, 130, (3, 5, 5), (3, 3, 3), (2, 2, 2), (1, 1, 1)),  # 2 M-blocks, ragged cout, stride 2
    (3, 5, 96, (9, 9), (3, 3), (1, 1), (1, 1)),            # bm=96
    (2, 3, 64, (20, 20), (7, 7), (2, 2), (3, 3)),          # conv1 geometry (7x7 s2 p3), K=147 -> padded to 160
    (2, 16, 32, (7, 7), (1, 1), (1, 1), (0, 0)),           # 1x1, bm=32
    (1, 8, 160, (5, 5), (3, 3), (2, 2), (1, 1)),           # 2D stride 2 (ECO-Full 3c/4e), bm=96 x2
    (5, 4, 20, (3, 3), (3, 3), (1, 1), (1, 1)),            # tile spans several images (S_out=9)
    # cin % 16 == 0 -> the constant-tap kernel family (ECO_CONV_MODE_CTAP)
    (1, 16, 128, (4, 7, 7), (3, 3, 3), (1, 1, 1), (1, 1, 1)),  # res-block geometry, bm=128
    (1, 32, 130, (3, 5, 5), (3, 3, 3), (2, 2, 2), (1, 1, 1)),  # stride 2, ragged cout, 2 channel tiles
    (2, 16, 96, (9, 9), (3, 3), (1, 1), (1, 1)),               # bm=96
    (2, 32, 64, (10, 10), (3, 3), (2, 2), (1, 1)),             # bm=64, 2-D stride 2
    (3, 16, 20, (3, 3), (3, 3), (1, 1), (1, 1)),               # bm=32, tile spans images
    (1, 48, 40, (6, 6), (1, 1), (1, 1), (0, 0)),               # 1x1 with 3 channel tiles
    (1, 64, 128, (2, 6, 6), (3, 3, 3), (1, 1, 1), (1, 1, 1)),  # 1 tile, 108 stages -> split-K (plan.ksplit > 1)
    # span kernel (ECO_CONV_MODE_SPAN) corner cases
    (1, 16, 32, (2, 127), (3, 3), (1, 1), (1, 1)),             # widest row the span buffer takes (span_len = 512)
    (2, 16, 32, (3, 5, 5), (1, 3, 3), (1, 1, 1), (0, 1, 1)),   # 3-D input, kd = 1
    # depth-major position order: tiles inside the first / last depth plane skip the padded depth taps
    (3, 16, 32, (3, 10, 10), (3, 3, 3), (1, 1, 1), (1, 1, 1)),  # 4 tiles: d=0 | d=0..1 | d=1..2 | d=2
    (5, 16, 40, (1, 8, 8), (3, 3, 3), (1, 1, 1), (1, 1, 1)),    # one plane: only the centre depth tap is live
    (3, 64, 128, (2, 7, 7), (3, 3, 3), (1, 1, 1), (1, 1, 1)),   # split-K slices, plane size 49 (scalar reduce)
    # the same for the gather kernel: strided 3x3x3 and the (3,1,1) convs of the Winograd route
    (6, 16, 128, (6, 9, 9), (3, 3, 3), (2, 2, 2), (1, 1, 1)),   # out 3x5x5: tile 0 = plane 0 only -> depth tap 0 dead
    (8, 16, 128, (4, 4, 4), (3, 1, 1), (1, 1, 1), (1, 0, 0)),   # tiles = depth planes: first skips z=0, last skips z=2
    (8, 32, 128, (2, 4, 4), (3, 1, 1), (1, 1, 1), (1, 0, 0)),   # two planes, two live taps each
]


@pytest.mark.parametrize("cfg", SMALL_CONVS)
def test_conv_plain(backend, cfg):
    run_conv(backend, *cfg, mode="plain")


@pytest.mark.parametrize("cfg", [SMALL_CONVS[3], SMALL_CONVS[5], SMALL_CONVS[6], SMALL_CONVS[9], SMALL_CONVS[10],
                                 SMALL_CONVS[11], SMALL_CONVS[12], SMALL_CONVS[14], SMALL_CONVS[16], SMALL_CONVS[19],
                                 SMALL_CONVS[20], SMALL_CONVS[21], SMALL_CONVS[22], SMALL_CONVS[23], SMALL_CONVS[24]])
def test_conv_fused_epilogue(backend, cfg):
    run_conv(backend, *cfg, mode="fused", seed=1)


@pytest.mark.parametrize("cin", [6, 16])
def test_conv_concat_slice_store(backend, cin):
    run_conv(backend, 2, cin, 33, (6, 6), (3, 3), (1, 1), (1, 1), mode="concat", seed=2)


@pytest.mark.parametrize("cin", [6, 16])
def test_conv_permuted_store(backend, cin):
    run_conv(backend, 4, cin, 12, (5, 5), (3, 3), (1, 1), (1, 1), mode="permute", seed=3)


def test_conv_plan_choice(backend):
    lib = backend.lib
    for cout, bm in [(32, 32), (64, 64), (96, 96), (128, 128), (160, 96), (192, 96), (224, 128), (256, 128),
                     (320, 64), (352, 128), (512, 128)]:
        p = lib.conv_plan(hip.conv_geom(1, 8, cout, (8, 8), (3, 3), (1, 1), (1, 1), (8, 8)))
        assert p.bm == bm and p.bn in (128, 256) and p.kpad % p.kc == 0 and p.mpad >= cout and p.mpad % 4 == 0
        assert p.mode == 0  # cin=8: table mode
    assert lib.conv_plan(hip.conv_geom(1, 64, 64, (8, 8), (3, 3), (1, 1), (1, 1), (8, 8))).mode == 2   # span: 3x3 s1 same
    assert lib.conv_plan(hip.conv_geom(1, 64, 64, (8, 8), (3, 3), (2, 2), (1, 1), (4, 4))).mode == 1   # strided: ctap
    assert lib.conv_plan(hip.conv_geom(1, 64, 64, (8, 8), (1, 1), (1, 1), (0, 0), (8, 8))).mode == 1   # 1x1: ctap
    assert lib.conv_plan(hip.conv_geom(1, 16, 64, (4, 300), (3, 3), (1, 1), (1, 1), (4, 300))).mode == 1  # rows too wide
    # split-K: chosen when the tile count quantises badly over 256 CUs and the reduction is long
    p5 = lib.conv_plan(hip.conv_geom(32, 512, 512, (4, 7, 7), (3, 3, 3), (1, 1, 1), (1, 1, 1), (4, 7, 7)))   # res5b: 196 tiles
    p4 = lib.conv_plan(hip.conv_geom(32, 256, 256, (8, 14, 14), (3, 3, 3), (1, 1, 1), (1, 1, 1), (8, 14, 14)))  # res4b: 784 tiles
    p3 = lib.conv_plan(hip.conv_geom(32, 128, 128, (16, 28, 28), (3, 3, 3), (1, 1, 1), (1, 1, 1), (16, 28, 28)))  # res3b: 3136 tiles
    assert p5.ksplit > 1 and p5.ws_bytes == p5.ksplit * 512 * 32 * 196 * 4
    assert p4.ksplit > 1 and p4.split_tiles == 2 * 196 and p5.split_tiles == 4 * 25   # full split: every tile
    assert (p3.bm, p3.bn) == (128, 128) and (p4.bm, p4.bn) == (128, 256) and (p5.bm, p5.bn) == (128, 256)
    # res3b: 3136 tiles over 4 x 256 slots -> only the 64 stragglers of the 4th round are split, 16 ways
    assert (p3.split_tiles, p3.ksplit) == (64, 16) and p3.ws_bytes == 16 * 128 * 64 * 128 * 4
    with pytest.raises(hip.EcoError, match="workspace"):
        lib.conv_forward(hip.conv_geom(32, 512, 512, (4, 7, 7), (3, 3, 3), (1, 1, 1), (1, 1, 1), (4, 7, 7)), p5, 8, 8, 8,
                         _dummy_epilogue(), None)


# Tail split-K: with a tiny "device" (num_cu) the many-tile path is reachable at emulator sizes.
# (num_cu, conv cfg, expected split_tiles, expected ksplit)
TAIL_SPLIT = [
    (1, (1, 64, 64, (34, 34), (3, 3), (1, 1), (1, 1)), 1, 4),               # 5 tiles of 64x256 on 4 slots
    (1, (1, 16, 128, (2, 16, 18), (3, 3, 3), (1, 1, 1), (1, 1, 1)), 1, 3),  # 5 tiles of 128x128, last one ragged
    (2, (1, 16, 130, (2, 20, 20), (3, 3, 3), (1, 1, 1), (1, 1, 1)), 2, 3),  # 2 M-blocks x 4 N-blocks on 6 slots
]


@pytest.mark.parametrize("num_cu,cfg,split_tiles,ksplit", TAIL_SPLIT)
@pytest.mark.parametrize("mode", ["plain", "fused"])
def test_conv_tail_split(backend, num_cu, cfg, split_tiles, ksplit, mode):
    plan = run_conv(backend, *cfg, mode=mode, seed=5, num_cu=num_cu)
    assert (plan.split_tiles, plan.ksplit) == (split_tiles, ksplit) and plan.ws_bytes > 0


@pytest.mark.parametrize("mode", ["plain", "fused"])
def test_conv_gather_split_with_dead_depth_taps(backend, mode):
    """Gather kernel, depth-major order, every tile split: first / last plane tiles have 2 live depth taps of 3 and
    get ceil(ksplit*2/3) slices, numbered densely; the reduce kernel sums each position's own slice count."""
    cfg = (8, 128, 128, (4, 4, 4), (3, 1, 1), (1, 1, 1), (1, 0, 0))
    ntot = 8 * 64

    def force_split(plan):   # 128-wide tiles = one depth plane each, 3-way split of every tile
        plan.bn, plan.ksplit = 128, 3
        plan.split_tiles = -(-128 // plan.bm) * -(-ntot // plan.bn)
        plan.ws_bytes = 3 * 128 * ntot * 4
    plan = run_conv(backend, *cfg, mode=mode, seed=9, tweak=force_split)
    assert plan.mode == 1 and (plan.bm, plan.bn, plan.split_tiles) == (128, 128, 4)


def _dummy_epilogue():
    ep = hip.ConvEpilogue()
    ep.residual, ep.act = hip.null_view(), hip.null_view()
    ep.raw = hip.plain_view(8, 512, 196)
    return ep


def test_conv_rejects_bad_geometry(backend):
    lib = backend.lib
    with pytest.raises(hip.EcoError):  # wrong output size
        lib.conv_plan(hip.conv_geom(1, 3, 4, (6, 4), (3, 3), (2, 2), (0, 0), (3, 2)))
    with pytest.raises(hip.EcoError):  # 8x8 = 64 taps > 62
        lib.conv_plan(hip.conv_geom(1, 3, 4, (16, 16), (8, 8), (1, 1), (0, 0), (9, 9)))
    with pytest.raises(hip.EcoError):  # zero stride
        lib.conv_plan(hip.conv_geom(1, 3, 4, (6, 4), (3, 3), (0, 1), (0, 0), (2, 1)))
    assert "conv" in lib.last_error()


# ---- the real ECO layer geometries (GPU only: too slow for the emulator) --------------------
ECO_CONVS = [
    (8, 3, 64, (224, 224), (7, 7), (2, 2), (3, 3)),           # conv1_7x7_s2
    (8, 64, 192, (56, 56), (3, 3), (1, 1), (1, 1)),           # conv2_3x3
    (8, 192, 64, (28, 28), (1, 1), (1, 1), (0, 0)),           # inception_3a_1x1
    (8, 64, 96, (28, 28), (3, 3), (1, 1), (1, 1)),            # inception_3x_double_3x3_1
    (8, 192, 32, (28, 28), (1, 1), (1, 1), (0, 0)),           # inception_3a_pool_proj
    (1, 96, 128, (16, 28, 28), (3, 3, 3), (1, 1, 1), (1, 1, 1)),   # res3a_2n (one clip)
    (2, 128, 256, (16, 28, 28), (3, 3, 3), (2, 2, 2), (1, 1, 1)),  # res4a_1 / res4a_down
    (2, 256, 512, (8, 14, 14), (3, 3, 3), (2, 2, 2), (1, 1, 1)),   # res5a_1 / res5a_down
    (2, 512, 512, (4, 7, 7), (3, 3, 3), (1, 1, 1), (1, 1, 1)),     # res5b_*
]


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", ECO_CONVS)
def test_conv_eco_geometries(hip_backend, cfg):
    run_conv(hip_backend, *cfg, mode="fused", seed=4)


# ---- pooling ------------------------------------------------------------------------------
POOLS = [
    ("MAX", (2, 3), (7, 9), (3, 3), (2, 2), (0, 0)),            # pool1/pool2 style, ceil mode
    ("MAX", (1, 2), (3, 5), (2, 2), (1, 1), (0, 0)),            # reference TestForwardMax
    ("MAX", (2, 2), (6, 6), (3, 3), (2, 2), (2, 2)),            # padded max (reference :475-518 geometry)
    ("MAX", (1, 3), (7, 7), (3, 3), (1, 1), (1, 1)),            # inception_5b_pool (ECO-Full)
    ("AVE", (2, 3), (6, 6), (3, 3), (1, 1), (1, 1)),            # inception_3a_pool
    ("AVE", (1, 3), (7, 7), (3, 3), (1, 1), (1, 1)),            # inception_5a_pool (ECO-Full): unrolled 3x3 kernel
    ("MAX", (2, 2), (14, 14), (3, 3), (2, 2), (0, 0)),          # inception_4e_pool: 14 -> 7, overhanging last window
    ("AVE", (1, 2), (9, 7), (3, 3), (2, 2), (1, 1)),            # divisor counts covered padding, clipped at H + pad
    ("AVE", (1, 1), (3, 3, 3), (3, 3, 3), (1, 1, 1), (1, 1, 1)),  # reference 3-D AVE golden geometry
    ("AVE", (2, 5), (4, 7, 7), (4, 7, 7), (1, 1, 1), (0, 0, 0)),  # global_pool (wave-reduction kernel)
    ("AVE", (2, 1), (4, 10), (4, 1), (1, 1), (0, 0)),           # segment_consensus_st2 (kernel_h=N, kernel_w=1)
    ("AVE", (2, 6), (2, 2), (2, 2), (1, 1), (0, 0)),            # tiny global (generic kernel path)
    ("MAX", (1, 2), (5, 6, 7), (3, 3, 3), (2, 2, 2), (1, 1, 1)),  # 3-D max
    # shapes that take the float4 fast paths (W % 4 == 0)
    ("MAX", (2, 3), (10, 16), (3, 3), (2, 2), (0, 0)),          # even H: last window has only 2 rows
    ("MAX", (1, 2), (9, 8), (3, 3), (2, 2), (0, 0)),            # odd H: full last window; 1 quad per row
    ("MAX", (2, 2), (28, 28), (3, 3), (2, 2), (0, 0)),          # pool2-like 28 -> 14 (Wo=14: generic path)
    ("MAX", (1, 3), (56, 56), (3, 3), (2, 2), (0, 0)),          # pool2: 56 -> 28
    ("AVE", (2, 3), (6, 8), (3, 3), (1, 1), (1, 1)),
    ("AVE", (1, 2), (28, 28), (3, 3), (1, 1), (1, 1)),          # inception_3a_pool geometry
    ("AVE", (1, 2), (5, 12), (3, 3), (1, 1), (1, 1)),
    ("AVE", (2, 3), (14, 14), (3, 3), (1, 1), (1, 1)),          # ECO-Full inception_4x_pool: float2 path
    ("AVE", (1, 2), (5, 6), (3, 3), (1, 1), (1, 1)),            # W % 4 == 2
    ("MAX", (2, 2), (7, 12), (3, 3), (2, 2), (0, 0)),           # Wi % 4 == 0, Wo = 6: 2 outputs per thread
]


@pytest.mark.parametrize("method,nc,insp,k,s,p", POOLS)
def test_pool(backend, method, nc, insp, k, s, p):
    rng = np.random.default_rng(5)
    x = rng.standard_normal(nc + tuple(insp)).astype(np.float32)
    ref = orc.pooling(x, method, k, s, p)
    assert np.allclose(ref, orc.pooling_fast(x, method, k, s, p), rtol=1e-6, atol=1e-6)
    g = hip.pool_geom(nc[0], nc[1], insp, k, s, p, ref.shape[2:], method)
    y = backend.empty(ref.shape)
    backend.lib.pool_forward(g, backend.ptr(backend.dev(x)), backend.ptr(y))
    assert relerr(backend.host(y, ref.shape), ref) < TOL


@pytest.mark.parametrize("method,nc,insp,k,s,p", [
    ("MAX", (3, 8), (28, 28), (3, 3), (2, 2), (0, 0)),      # inception_3c_pool: the 8-byte vector kernel
    ("MAX", (2, 5), (14, 14), (3, 3), (2, 2), (0, 0)),      # inception_4e_pool: 14 -> 7, the 3x3 kernel
    ("MAX", (2, 4), (16, 16), (3, 3), (2, 2), (0, 0)),      # 16-byte vector kernel (8 outputs per row)
    ("AVE", (2, 3), (6, 8), (3, 3), (1, 1), (1, 1)),        # the AVE fast path is dense-only: falls to the 3x3 kernel
    ("AVE", (2, 2), (4, 5, 5), (2, 3, 3), (2, 2, 2), (0, 1, 1)),   # generic N-D kernel
])
def test_pool_into_a_concat_slice(backend, method, nc, insp, k, s, p):
    """eco_pool_forward_strided: the pooled blob lands in channels [c0, c0 + c) of a wider tensor, the rest untouched."""
    rng = np.random.default_rng(7)
    x = rng.standard_normal(nc + tuple(insp)).astype(np.float32)
    ref = orc.pooling(x, method, k, s, p)
    n, c = nc
    S = int(np.prod(ref.shape[2:]))
    c0, ctot = 4, c + 8
    g = hip.pool_geom(n, c, insp, k, s, p, ref.shape[2:], method)
    big = backend.dev(np.full((n, ctot) + ref.shape[2:], 7.0, np.float32))
    backend.lib.pool_forward_strided(g, backend.ptr(backend.dev(x)), backend.ptr(big, c0 * S), ctot * S)
    got = backend.host(big, (n, ctot) + ref.shape[2:])
    assert relerr(got[:, c0:c0 + c], ref) < TOL
    assert (got[:, :c0] == 7.0).all() and (got[:, c0 + c:] == 7.0).all()
    with pytest.raises(hip.EcoError, match="smaller than an image"):
        backend.lib.pool_forward_strided(g, 16, 16, c * S - 1)


def test_pool_rejects_wrong_output_shape(backend):
    g = hip.pool_geom(1, 1, (7, 7), (3, 3), (2, 2), (0, 0), (3, 3), "MAX")  # caffe ceil rule gives 4x4
    with pytest.raises(hip.EcoError):
        backend.lib.pool_forward(g, 0, 0)


# ---- elementwise / glue ----------------------------------------------------------------------
@pytest.mark.parametrize("shape", [(2, 5, 3, 4), (2, 3, 2, 3, 4)])
def test_bn_relu(backend, shape):
    rng = np.random.default_rng(6)
    x = rng.standard_normal(shape).astype(np.float32)
    C = shape[1]
    gamma, beta = rng.uniform(0.5, 1.5, C).astype(np.float32), rng.standard_normal(C).astype(np.float32)
    mean, var = rng.standard_normal(C).astype(np.float32), rng.uniform(0.5, 1.5, C).astype(np.float32)
    from eco_amd.engine import fold_bn
    a, b = fold_bn([gamma, beta, mean, var], 1e-5)
    ref = orc.bn_inference(x, gamma, beta, mean, var, 1e-5)
    inner = int(np.prod(shape[2:]))
    for relu in (0, 1):
        y = backend.empty(shape)
        backend.lib.bn_forward(backend.ptr(backend.dev(x)), backend.ptr(y), backend.ptr(backend.dev(a)),
                               backend.ptr(backend.dev(b)), shape[0], C, inner, relu)
        exp = orc.relu(ref) if relu else ref
        assert relerr(backend.host(y, shape), exp) < TOL


def test_relu_eltwise_concat_permute(backend):
    rng = np.random.default_rng(7)
    lib = backend.lib
    x = rng.standard_normal((3, 4, 5)).astype(np.float32)
    y = backend.empty(x.shape)
    lib.relu_forward(backend.ptr(backend.dev(x)), backend.ptr(y), x.size, 0.0)
    assert np.array_equal(backend.host(y, x.shape), orc.relu(x))
    lib.relu_forward(backend.ptr(backend.dev(x)), backend.ptr(y), x.size, 0.1)
    assert relerr(backend.host(y, x.shape), orc.relu(x, 0.1)) < TOL
    b = rng.standard_normal(x.shape).astype(np.float32)
    lib.eltwise_sum_forward(backend.ptr(backend.dev(x)), backend.ptr(backend.dev(b)), backend.ptr(y), x.size, 1.0, 1.0)
    assert np.array_equal(backend.host(y, x.shape), orc.eltwise_sum([x, b]))
    # concat along channels of two [2, c, 3, 2] tensors
    a1, a2 = rng.standard_normal((2, 3, 3, 2)).astype(np.float32), rng.standard_normal((2, 5, 3, 2)).astype(np.float32)
    out = backend.empty((2, 8, 3, 2))
    lib.concat_copy(backend.ptr(backend.dev(a1)), backend.ptr(out), 2, 3, 8, 0, 6)
    lib.concat_copy(backend.ptr(backend.dev(a2)), backend.ptr(out), 2, 5, 8, 3, 6)
    assert np.array_equal(backend.host(out, (2, 8, 3, 2)), orc.concat([a1, a2], 1))
    with pytest.raises(hip.EcoError):
        lib.concat_copy(backend.ptr(backend.dev(a2)), backend.ptr(out), 2, 5, 8, 4, 6)
    # the ECO Transpose1 permutation
    z = rng.standard_normal((2, 4, 3, 2, 5)).astype(np.float32)
    zp = backend.empty((2, 3, 4, 2, 5))
    lib.permute_forward(backend.ptr(backend.dev(z)), backend.ptr(zp), z.shape, [0, 2, 1, 3, 4])
    assert np.array_equal(backend.host(zp, (2, 3, 4, 2, 5)), orc.permute(z, [0, 2, 1, 3, 4]))
    with pytest.raises(hip.EcoError):
        lib.permute_forward(backend.ptr(backend.dev(z)), backend.ptr(zp), z.shape, [0, 2, 2, 3, 4])


def test_inner_product_tail_softmax(backend):
    rng = np.random.default_rng(8)
    lib = backend.lib
    M, N, K = 3, 10, 70
    x = rng.standard_normal((M, K)).astype(np.float32)
    w = rng.standard_normal((N, K)).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
    ref = orc.inner_product(x, w, b)
    y = backend.empty((M, N))
    lib.inner_product_forward(backend.ptr(backend.dev(x)), backend.ptr(backend.dev(w)), backend.ptr(backend.dev(b)),
                              backend.ptr(y), M, N, K)
    assert relerr(backend.host(y, (M, N)), ref) < TOL
    # fused global_pool + fc, including the column-offset/accumulate form used for concat(a, b) -> fc
    B, C, S, NO = 2, 12, 2 * 3 * 3, 70
    f = rng.standard_normal((B, C, 2, 3, 3)).astype(np.float32)
    w2 = rng.standard_normal((NO, C + 5)).astype(np.float32)
    b2 = rng.standard_normal(NO).astype(np.float32)
    pooled = orc.pooling(f, "AVE", (2, 3, 3), (1, 1, 1), (0, 0, 0)).reshape(B, C)
    ref2 = orc.inner_product(pooled, w2[:, 5:], b2)
    y2 = backend.empty((B, NO))
    lib.global_avgpool_fc_forward(backend.ptr(backend.dev(f)), backend.ptr(backend.dev(w2)),
                                  backend.ptr(backend.dev(b2)), backend.ptr(y2), B, C, S, NO, C + 5, 5, False)
    assert relerr(backend.host(y2, (B, NO)), ref2) < TOL
    # the ECO tail's own shape class: one frame of 65..256 positions per channel (res5b at num_segments 16 / 32: 4x7x7,
    # 8x7x7), enough channels for the four-channels-at-a-time pooling loop, a ragged channel count and >= 512 weights per
    # logit for the eight-loads-in-flight fc loop
    for C3, dhw in ((152, (4, 7, 7)), (520, (3, 5, 5)), (100, (8, 7, 7))):
        S3 = int(np.prod(dhw))
        f3 = rng.standard_normal((B, C3) + dhw).astype(np.float32)
        w3 = rng.standard_normal((NO, C3)).astype(np.float32)
        ref3 = orc.inner_product(f3.reshape(B, C3, S3).mean(2, dtype=np.float64).astype(np.float32), w3, b2)
        y3 = backend.empty((B, NO))
        lib.global_avgpool_fc_forward(backend.ptr(backend.dev(f3)), backend.ptr(backend.dev(w3)), backend.ptr(backend.dev(b2)),
                                      backend.ptr(y3), B, C3, S3, NO, C3, 0, False)
        assert relerr(backend.host(y3, (B, NO)), ref3) < TOL, (C3, dhw)
    sm = rng.standard_normal((2, 7, 3)).astype(np.float32)
    ys = backend.empty(sm.shape)
    lib.softmax_forward(backend.ptr(backend.dev(sm)), backend.ptr(ys), 2, 7, 3)
    assert relerr(backend.host(ys, sm.shape), orc.softmax(sm, 1)) < TOL


@pytest.mark.parametrize("shape,top_k,ignore", [((7, 13), 1, None), ((7, 13), 5, None), ((5, 400), 5, None),
                                                ((4, 9, 3), 2, None), ((9, 6), 1, 2), ((3, 70, 2, 2), 3, 5)])
def test_accuracy_and_softmax_loss(backend, shape, top_k, ignore):
    """AccuracyLayer / SoftmaxWithLossLayer against the oracle; scores are quantised so that ties occur and
    the std::greater<pair<score, class>> tie rule (larger class index first) decides some samples."""
    rng = np.random.default_rng(17)
    x = (rng.integers(-6, 7, shape) * 0.5).astype(np.float32)
    c = shape[1]
    outer, inner = shape[0], int(np.prod(shape[2:]))
    label = rng.integers(0, c, (outer, inner)).astype(np.float32)
    if ignore is not None:
        label[0, 0] = ignore
    dx, dl = backend.dev(x), backend.dev(label)
    out = backend.empty((2,))
    backend.lib.accuracy_forward(backend.ptr(dx), backend.ptr(dl), backend.ptr(out), outer, c, inner, top_k, ignore)
    ref = orc.accuracy(x, label, top_k, 1, ignore)
    assert backend.host(out, (2,))[0] == np.float32(ref)
    for normalize in (True, False):
        backend.lib.softmax_loss_forward(backend.ptr(dx), backend.ptr(dl), backend.ptr(out), outer, c, inner,
                                         normalize, ignore)
        refl = orc.softmax_loss(x, label, 1, normalize, ignore)
        assert abs(backend.host(out, (2,))[0] - refl) < 1e-5 * abs(refl)
    with pytest.raises(hip.EcoError, match="top_k"):
        backend.lib.accuracy_forward(backend.ptr(dx), backend.ptr(dl), backend.ptr(out), outer, c, inner, c + 1, None)


# ---- Winograd F(2x2,3x3) path: transforms + 16 batched (kd,1,1) convolutions == the direct convolution ----
def wino_conv(be, x, w, b, M, mode="plain", seed=0, num_cu=None):
    """Run the three-launch Winograd evaluation through the C ABI; returns (raw, act-or-None, expected act fn)."""
    lib = be.lib
    n, cin = x.shape[:2]
    cout, kd = w.shape[0], w.shape[2]
    D, H, W = x.shape[2:]
    P = (M + 2) ** 2
    TH, TW = -(-H // M), -(-W // M)
    u = np.zeros((P, cout, cin, kd), np.float32)
    lib.wino_weight_transform(w.ctypes.data, cout, cin, kd, M, u.ctypes.data)
    g = hip.conv_geom(n, cin, cout, (D, TH, TW), (kd, 1, 1), (1, 1, 1), (kd // 2, 0, 0), (D, TH, TW))
    plan = lib.conv_plan(g, num_cu)
    assert plan.mode in (0, 1)
    wps = np.zeros((P, plan.wp_elems), np.float32)
    kt = np.zeros(plan.ktab_elems, np.int32)
    for p in range(P):
        up = np.ascontiguousarray(u[p].reshape(cout, cin, kd, 1, 1))
        lib.conv_pack_weights(g, plan, up.ctypes.data, wps[p].ctypes.data, kt.ctypes.data)
    tiles_in, tiles_out = n * cin * D * TH * TW, n * cout * D * TH * TW
    dx, dwp, dkt = be.dev(x), be.dev(wps), be.dev(kt)
    v, m = be.empty((P * tiles_in,)), be.empty((P * tiles_out,))
    ws = be.ptr(be.empty((P * plan.ws_bytes // 4,))) if plan.ws_bytes else None
    lib.wino_input_forward(be.ptr(dx), be.ptr(v), n * cin * D, H, W, M)
    epg = hip.ConvEpilogue()
    epg.bias = None
    epg.residual, epg.act = hip.null_view(), hip.null_view()
    epg.bn_scale = epg.bn_shift = None
    epg.relu = 0
    epg.raw = hip.plain_view(be.ptr(m), cout, D * TH * TW)
    lib.conv_forward_batched(g, plan, be.ptr(v), be.ptr(dwp), be.ptr(dkt), epg, ws, P, tiles_in, plan.wp_elems,
                             tiles_out)
    S = D * H * W
    ep = hip.ConvEpilogue()
    db = be.dev(b)
    ep.bias = be.ptr(db)
    ep.residual, ep.raw, ep.act = hip.null_view(), hip.null_view(), hip.null_view()
    ep.bn_scale = ep.bn_shift = None
    ep.relu = 0
    shape = (n, cout, D, H, W)
    raw = be.empty(shape)
    ep.raw = hip.plain_view(be.ptr(raw), cout, S)
    extra = None
    if mode == "fused":
        rng = np.random.default_rng(seed)
        res = rng.standard_normal(shape).astype(np.float32)
        sc = rng.uniform(0.5, 1.5, cout).astype(np.float32)
        sh = rng.standard_normal(cout).astype(np.float32)
        dres, dsc, dsh, act = be.dev(res), be.dev(sc), be.dev(sh), be.empty(shape)
        ep.residual = hip.plain_view(be.ptr(dres), cout, S)
        ep.act = hip.plain_view(be.ptr(act), cout, S)
        ep.bn_scale, ep.bn_shift, ep.relu = be.ptr(dsc), be.ptr(dsh), 1
        extra = (res, sc, sh, act)
    lib.wino_output_forward(be.ptr(m), n, cout, D, H, W, M, ep)
    return be.host(raw, shape), extra, (v, u, plan)


@pytest.mark.parametrize("cfg", [(2, 16, 32, (3, 8, 8), 3), (1, 16, 130, (2, 7, 7), 3), (3, 32, 20, (1, 6, 10), 1),
                                 (1, 6, 8, (2, 5, 4), 3)])
@pytest.mark.parametrize("mode", ["plain", "fused"])
@pytest.mark.parametrize("M", [2, 4])
def test_winograd_path_matches_direct_conv(backend, cfg, mode, M):
    n, cin, cout, insp, kd = cfg
    rng = np.random.default_rng(23)
    x = rng.standard_normal((n, cin) + insp).astype(np.float32)
    w = (rng.standard_normal((cout, cin, kd, 3, 3)) / np.sqrt(cin * kd * 9)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    ref = orc.convolution(x, w, b, (kd, 3, 3), (1, 1, 1), (kd // 2, 1, 1))
    raw, extra, (v, u, plan) = wino_conv(backend, x, w, b, M, mode, seed=4)
    # transforms against their definitions (Lavin & Gray 2015, F(2x2,3x3) and F(4x4,3x3))
    if M == 2:
        G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], np.float64)
        BT = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], np.float64)
    else:
        G = np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6],
                      [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], np.float64)
        BT = np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0],
                       [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], np.float64)
    T = M + 2
    uref = np.einsum("ia,kczab,jb->ijkcz", G, w.astype(np.float64), G).reshape(T * T, cout, cin, kd)
    assert relerr(u, uref) < 2e-6
    H, W = insp[1:]
    TH, TW = -(-H // M), -(-W // M)
    xp = np.zeros((n, cin, insp[0], M * TH + 2, M * TW + 2), np.float64)
    xp[..., 1:H + 1, 1:W + 1] = x
    tiles = np.stack([np.stack([xp[..., i:i + M * TH:M, j:j + M * TW:M] for j in range(T)], 0) for i in range(T)], 0)
    vref = np.einsum("ia,ab...,jb->ij...", BT, tiles, BT).reshape(T * T, -1)
    assert relerr(backend.host(v, vref.shape), vref) < 2e-6
    tol = 2e-5 if M == 2 else 2e-4   # F(4x4): transform constants up to 8 amplify the fp32 rounding
    if mode == "plain":
        assert relerr(raw, ref) < tol
    else:
        res, sc, sh, act = extra
        exp_raw = ref + res
        bshape = (1, cout, 1, 1, 1)
        assert relerr(raw, exp_raw) < tol
        assert relerr(backend.host(act, ref.shape), np.maximum(exp_raw * sc.reshape(bshape) + sh.reshape(bshape), 0)) < tol


def test_batched_conv_equals_separate_launches(backend):
    """eco_conv_forward_batched: entry b uses x + b*stride_x, wp + b*stride_wp and the views moved by b*stride_out."""
    rng = np.random.default_rng(3)
    nb, n, cin, cout, insp = 3, 2, 16, 40, (4, 5, 5)
    g = hip.conv_geom(n, cin, cout, insp, (3, 1, 1), (1, 1, 1), (1, 0, 0), insp)
    lib = backend.lib
    plan = lib.conv_plan(g, 1)   # tiny device: 1 tile, long reduction -> split-K with a per-entry workspace
    xs = rng.standard_normal((nb, n, cin) + insp).astype(np.float32)
    wsn = (rng.standard_normal((nb, cout, cin, 3, 1, 1)) * 0.2).astype(np.float32)
    wps = np.zeros((nb, plan.wp_elems), np.float32)
    kt = np.zeros(plan.ktab_elems, np.int32)
    for b in range(nb):
        lib.conv_pack_weights(g, plan, wsn[b].ctypes.data, wps[b].ctypes.data, kt.ctypes.data)
    dx, dwp, dkt = backend.dev(xs), backend.dev(wps), backend.dev(kt)
    S = int(np.prod(insp))
    y = backend.empty((nb, n, cout) + insp)
    ep = hip.ConvEpilogue()
    ep.bias = None
    ep.residual, ep.act = hip.null_view(), hip.null_view()
    ep.bn_scale = ep.bn_shift = None
    ep.relu = 0
    ep.raw = hip.plain_view(backend.ptr(y), cout, S)
    ws = backend.ptr(backend.empty((nb * plan.ws_bytes // 4,))) if plan.ws_bytes else None
    lib.conv_forward_batched(g, plan, backend.ptr(dx), backend.ptr(dwp), backend.ptr(dkt), ep, ws, nb,
                             n * cin * S, plan.wp_elems, n * cout * S)
    got = backend.host(y, (nb, n, cout) + insp)
    for b in range(nb):
        ref = orc.convolution(xs[b], wsn[b], None, (3, 1, 1), (1, 1, 1), (1, 0, 0))
        assert relerr(got[b], ref) < TOL


# 1x1 stride-1 convolutions on the LDS-DMA GEMM kernel (ECO_CONV_MODE_POINT): reached at emulator sizes by planning for
# a one-CU device (the rule wants >= 4 * num_cu tiles of 256 positions)
POINT = [  # n, cin, cout, in_sp
    (3, 32, 64, (20, 20)),       # bm 64; 1200 positions: last tile ragged, tiles straddle images
    (2, 48, 96, (24, 24)),       # bm 96 (weight rows padded to 128 in LDS), three stages
    (5, 16, 160, (16, 16)),      # two M-blocks of 96, the second ragged; one stage
    (2, 64, 32, (3, 16, 16)),    # 3-D blob, bm 32
    (9, 32, 128, (12, 12)),      # bm 128 (4x2 wave tiles)
]


@pytest.mark.parametrize("n,cin,cout,insp", POINT, ids=[f"p{i}" for i in range(len(POINT))])
@pytest.mark.parametrize("mode", ["plain", "fused", "concat"])
def test_conv_point_kernel(backend, n, cin, cout, insp, mode):
    one = (1,) * len(insp)
    plan = run_conv(backend, n, cin, cout, insp, one, one, (0,) * len(insp), mode=mode, seed=9, num_cu=1)
    assert plan.mode == 3 and plan.bn == 256 and plan.ksplit == 1


def test_conv_point_kernel_eligibility(backend):
    lib = backend.lib
    # default device: the inception 1x1 layers of the benchmark qualify, a single clip's 7x7 layer does not
    assert lib.conv_plan(hip.conv_geom(512, 192, 64, (28, 28), (1, 1), (1, 1), (0, 0), (28, 28))).mode == 3
    assert lib.conv_plan(hip.conv_geom(16, 1024, 352, (7, 7), (1, 1), (1, 1), (0, 0), (7, 7))).mode == 1
    # plane size not a multiple of 4 (a lane's four positions would straddle images), strided 1x1: gather kernel
    assert lib.conv_plan(hip.conv_geom(64, 32, 64, (7, 7), (1, 1), (1, 1), (0, 0), (7, 7)), 1).mode == 1
    assert lib.conv_plan(hip.conv_geom(64, 32, 64, (8, 8), (1, 1), (2, 2), (0, 0), (4, 4)), 1).mode == 1


# ---- stream-K form of the gather kernel (csrc/eco_conv.hip, conv_streamk_kernel) --------------------------------------
def _force_streamk(wgs, bn=None):
    def tweak(plan):
        if bn:
            plan.bn = bn
        plan.ksplit, plan.split_tiles, plan.streamk_wgs = 1, 0, wgs
        plan.ws_bytes = (2 * wgs * plan.bm * plan.bn * 4 + 255) // 256 * 256 + 4 * wgs      # two partial blocks + a flag per workgroup
        plan.ktab_elems = plan.kpad + 4096
    return tweak


@pytest.mark.parametrize("mode", ["plain", "fused"])
@pytest.mark.parametrize("wgs", [1, 2, 3, 5, 7, 16])
def test_conv_streamk_shares_stages_across_tile_boundaries(backend, mode, wgs):
    """A strided 3x3x3 conv (res4a_1's shape, scaled down: 2 M-blocks x 4 columns of 128 positions, depth-major with dead
    depth taps in the first plane's columns) with its stages dealt to 1 .. 16 persistent workgroups: whole tiles, tiles
    shared by two workgroups, a tile spread over three and more (16 workgroups on 8 tiles), a workgroup whose whole range
    lies inside one tile -- every split the hand-off protocol knows."""
    cfg = (2, 32, 256, (8, 8, 8), (3, 3, 3), (2, 2, 2), (1, 1, 1))        # out 4x4x4: ntot = 2*64 = 128 -> bn 128: 1 column
    plan = run_conv(backend, *cfg, mode=mode, seed=wgs, tweak=_force_streamk(wgs, bn=128))
    assert plan.mode == 1 and plan.streamk_wgs == wgs
    cfg = (8, 32, 256, (8, 8, 8), (3, 3, 3), (2, 2, 2), (1, 1, 1))        # ntot = 512: four columns, one per depth plane
    plan = run_conv(backend, *cfg, mode=mode, seed=10 + wgs, tweak=_force_streamk(wgs, bn=128))
    assert (plan.bm, plan.bn) == (128, 128)


def test_conv_streamk_plan_is_opt_in(backend, monkeypatch):
    """The planner keeps the two-launch split for few-tile launches (stream-K measured no faster: the even shares lose
    the split order's L2 locality); a stream-K plan made by hand is what `eco_conv_forward` runs, with the workspace
    and table sizes the header states."""
    lib = backend.lib
    g = hip.conv_geom(32, 128, 256, (16, 28, 28), (3, 3, 3), (2, 2, 2), (1, 1, 1), (8, 14, 14))      # res4a_1: 392 tiles
    p = lib.conv_plan(g, 256)
    assert p.mode == 1 and (p.bm, p.bn) == (128, 256) and p.streamk_wgs == 0 and p.ksplit > 1 and p.split_tiles == 392
    _force_streamk(512)(p)
    assert hip.conv_kernel_name(p) == "eco::conv_streamk_kernel<2, 4, 2, 2, 16>"
    assert p.ws_bytes == 2 * 512 * 128 * 256 * 4 + 4 * 512
