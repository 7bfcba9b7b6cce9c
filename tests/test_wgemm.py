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
