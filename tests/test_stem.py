"""The fused stem kernel (csrc/eco_stem.hip): conv 7x7/2 pad 3 (3 -> 32|64) + bias + folded BN + ReLU + MAX pool 3x3/2
(ceil rule) against the oracle's layer sequence, through the C ABI; and the engine taking it for the ECO graphs."""
import numpy as np
import pytest

import eco_oracle as orc
from eco_amd import fillers, hip, models
from eco_amd.netspec import NetSpec
from tests.test_net import make_net, relerr


# max_wg: cap on the persistent workgroups (0 = two per CU); small caps make one workgroup walk several patches
@pytest.mark.parametrize("n,H,W,cout,bn,max_wg", [(2, 64, 64, 64, True, 0), (2, 64, 64, 64, True, 3),
                                                  (1, 75, 52, 32, True, 1), (3, 40, 36, 64, False, 2),
                                                  (1, 224, 224, 64, True, 0), (2, 224, 224, 64, True, 5)])
def test_stem_matches_layer_sequence(backend, n, H, W, cout, bn, max_wg):
    if H == 224 and backend.kind == "emu":
        pytest.skip("full-size frame: GPU only")
    rng = np.random.default_rng(H + cout)
    x = rng.uniform(-120, 130, size=(n, 3, H, W)).astype(np.float32)
    w = (rng.normal(size=(cout, 3, 7, 7)) / 12).astype(np.float32)
    b = rng.normal(size=cout).astype(np.float32)
    sc, sh = rng.uniform(0.5, 1.5, cout).astype(np.float32), rng.normal(size=cout).astype(np.float32)
    v = orc.convolution(x, w, b, (7, 7), (2, 2), (3, 3))
    if bn:
        v = v * sc[None, :, None, None] + sh[None, :, None, None]
    v = np.maximum(v, 0)
    ref = orc.pooling(v, "MAX", (3, 3), (2, 2), (0, 0))
    lib = backend.lib
    wp = np.empty(74 * cout * 2, np.float32)
    lib.stem_pack_weights(w.ctypes.data, cout, wp.ctypes.data)
    y = backend.empty(ref.shape)
    lib.stem_forward(backend.ptr(backend.dev(x)), backend.ptr(backend.dev(wp)), backend.ptr(backend.dev(b)), backend.ptr(backend.dev(sc)) if bn else None,
                     backend.ptr(backend.dev(sh)) if bn else None, 1, backend.ptr(y), n, H, W, cout, max_workgroups=max_wg)
    assert relerr(backend.host(y, ref.shape), ref) < 2e-5


def test_stem_rejects_other_widths(backend):
    with pytest.raises(hip.EcoError, match="32 or 64"):
        backend.lib.stem_pack_weights(0, 48, 0)


def test_engine_fuses_the_stem(backend):
    """ECO-Lite at full width (64-channel conv1) on small frames: conv1+BN+ReLU+pool1 is one launch, its blobs are not
    materialised, the net still matches the oracle, and stem=False restores the two-launch form."""
    proto = models.eco_lite_deploy(num_segments=4, num_clips=1, num_classes=10, input_size=32, width_div=1)
    spec = NetSpec.from_prototxt(proto)
    assert spec.layer("conv1_7x7_s2").geom["cout"] == 64
    # only the stem is compared here (the rest of a full-width net is too slow for the emulator): run to pool1
    params = fillers.synthetic_params(spec, seed=5)
    x = fillers.synthetic_frames(4, 32, 32, seed=2)
    net = make_net(backend, proto, params, True)
    assert any(l.startswith("conv1_7x7_s2+") and l.endswith("+pool1_3x3_s2") for l in net.op_labels())
    assert "pool1_3x3_s2" not in net.op_labels()
    net.blobs["data"].data[...] = x
    net.forward(end="pool1_3x3_s2")
    ref = orc.forward(spec, params, {"data": x}, keep=["pool1_3x3_s2"])["pool1_3x3_s2"]
    assert relerr(net.blobs["pool1_3x3_s2"].data, ref) < 2e-5
    with pytest.raises(KeyError, match="stem launch"):
        net.blobs["conv1_7x7_s2_bn"].data
