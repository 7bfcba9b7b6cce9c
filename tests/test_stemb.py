"""The fused stem of the blocked bf16 path (csrc/eco_stemb.hip): conv 7x7/2 pad 3 (3 -> 32|64) + bias + folded BN + ReLU +
MAX pool 3x3/2 (ceil rule), fp32 frames in, pooled activations out in the blocked bf16 layout -- against the oracle's
layer sequence fed the same bf16-rounded frames and weights (so the products agree exactly; what remains is the fp32
accumulation order and the one rounding at the store: |err| <= 2^-8 |y| + noise), through the C ABI; and the engine
taking it for the ECO graphs at dtype="bf16"."""
import numpy as np
import pytest

import eco_oracle as orc
from eco_amd import blocked, fillers, hip, models
from eco_amd.netspec import NetSpec

BF16 = hip.DT_BF16


def _reference(x, w, b, sc, sh, bn, relu):
    v = orc.convolution(blocked.bf16_round(x), blocked.bf16_round(w), b, (7, 7), (2, 2), (3, 3))
    if bn:
        v = v * sc[None, :, None, None] + sh[None, :, None, None]
    if relu:
        v = np.maximum(v, 0)
    return orc.pooling(v, "MAX", (3, 3), (2, 2), (0, 0))


# max_wg: cap on the persistent workgroups (0 = two per CU); small caps make one workgroup walk several patches
@pytest.mark.parametrize("n,H,W,cout,bn,relu,max_wg", [(2, 64, 64, 64, True, 1, 0), (2, 64, 64, 64, True, 1, 3),
                                                       (1, 75, 52, 32, True, 0, 1), (3, 40, 36, 64, False, 1, 2), (2, 41, 50, 32, True, 1, 2), (1, 37, 43, 64, True, 0, 0),
                                                       (1, 224, 224, 64, True, 1, 0), (2, 224, 224, 64, True, 1, 5)])
def test_stemb_matches_layer_sequence(backend, n, H, W, cout, bn, relu, max_wg):
    if H == 224 and backend.kind == "emu":
        pytest.skip("full-size frame: GPU only")
    rng = np.random.default_rng(H + cout)
    x = rng.uniform(-120, 130, size=(n, 3, H, W)).astype(np.float32)
    w = (rng.normal(size=(cout, 3, 7, 7)) / 12).astype(np.float32)
    b = rng.normal(size=cout).astype(np.float32)
    sc, sh = rng.uniform(0.5, 1.5, cout).astype(np.float32), rng.normal(size=cout).astype(np.float32)
    ref = _reference(x, w, b, sc, sh, bn, relu)
    lib = backend.lib
    assert lib.stemb_weight_elems(cout) == 11 * 2 * cout * 8
    wp = np.empty(lib.stemb_weight_elems(cout), np.uint16)
    lib.stemb_pack_weights(w.ctypes.data, cout, wp.ctypes.data)
    y = backend.empty(ref.shape, np.uint16)
    lib.stemb_forward(backend.ptr(backend.dev(x)), backend.ptr(backend.dev(wp)), backend.ptr(backend.dev(b)),
                      backend.ptr(backend.dev(sc)) if bn else None, backend.ptr(backend.dev(sh)) if bn else None, relu,
                      backend.ptr(y), n, H, W, cout, max_workgroups=max_wg)
    got = blocked.from_blocked(backend.host(y, ref.shape), ref.shape, BF16)
    err = np.abs(got - ref)
    assert (err <= 2.0 ** -8 * np.abs(ref) + 2e-5 * np.abs(ref).max()).all(), float(err.max())


def test_stemb_packed_weights_layout(backend):
    """wp[s][g][m][e] = bf16(w[m][rho = 2s + g][kx = e]); tap 7 of every row and the 22nd row are zero."""
    cout = 32
    w = np.arange(cout * 147, dtype=np.float32).reshape(cout, 21, 7) / 64
    wp = np.full(backend.lib.stemb_weight_elems(cout), 0xFFFF, np.uint16)
    backend.lib.stemb_pack_weights(w.ctypes.data, cout, wp.ctypes.data)
    wp = wp.reshape(11, 2, cout, 8)
    assert not wp[..., 7].any() and not wp[10, 1].any()
    rows = wp.reshape(22, cout, 8)[:21, :, :7].transpose(1, 0, 2)
    assert np.array_equal(rows, blocked.bf16_bits(w).reshape(cout, 21, 7))


def test_stemb_rejects_other_widths(backend):
    with pytest.raises(hip.EcoError, match="32 or 64"):
        backend.lib.stemb_pack_weights(0, 48, 0)


def test_engine_fuses_the_blocked_stem(backend):
    """ECO-Lite at full width (64-channel conv1) on small frames, dtype="bf16": conv1+BN+ReLU+pool1 is one launch (no
    stem pack, no separate pool), pool1 matches the oracle with the blocked path's storage rounding, and stem=False
    restores the three-launch form with the same result to bf16 rounding."""
    from eco_amd.net import Net
    proto = models.eco_lite_deploy(num_segments=4, num_clips=1, num_classes=10, input_size=32, width_div=1)
    spec = NetSpec.from_prototxt(proto)
    params = fillers.synthetic_params(spec, seed=5)
    x = fillers.synthetic_frames(4, 32, 32, seed=2)
    kw = {"_backend": (backend.lib, backend.alloc)} if backend.kind == "emu" else {}
    net = Net(proto, params=params, dtype="bf16", **kw)
    labels = net.op_labels()
    assert any(l.startswith("conv1_7x7_s2+") and l.endswith("+pool1_3x3_s2") for l in labels)
    assert "pool1_3x3_s2" not in labels and not any("stem pack" in l for l in labels)
    net.blobs["data"].data[...] = x
    net.forward(end="pool1_3x3_s2")
    got = net.blobs["pool1_3x3_s2"].data.copy()
    p = params["conv1_7x7_s2"]
    bnp = params["conv1_7x7_s2_bn"]
    qparams = dict(params)
    qparams["conv1_7x7_s2"] = [blocked.bf16_round(p[0])] + list(p[1:])
    ref = orc.forward(spec, qparams, {"data": blocked.bf16_round(x)}, keep=["pool1_3x3_s2"])["pool1_3x3_s2"]
    assert bnp is not None
    err = np.abs(got - ref)
    assert (err <= 2.0 ** -8 * np.abs(ref) + 2e-5 * np.abs(ref).max()).all(), float(err.max())
    with pytest.raises(KeyError, match="stem launch"):
        net.blobs["conv1_7x7_s2_bn"].data
    net3 = Net(proto, params=params, dtype="bf16", stem=False, **kw)
    assert any("stem pack" in l for l in net3.op_labels())
    net3.blobs["data"].data[...] = x
    net3.forward(end="pool1_3x3_s2")
    three = net3.blobs["pool1_3x3_s2"].data
    assert (np.abs(three - got) <= 2.0 ** -7 * np.abs(got) + 2e-5 * np.abs(got).max()).all()
