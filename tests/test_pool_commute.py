"""AVE pool 3x3/1/1 -> 1x1 conv -> BN -> ReLU with the two linear maps exchanged (inception_3a / 3b pool + pool_proj,
models_ECO_Lite/kinetics/deploy.prototxt:330-400): the conv runs on the block's input as one more member of the
block's sibling launch, `eco_avgpool_affine_forward` finishes on the conv's channels.  Kernel against the oracle's
layer sequence; engine against the whole-net oracle, with and without the rewrite."""
import numpy as np
import pytest

import eco_oracle as orc
from eco_amd import fillers, hip, models
from eco_amd.netspec import NetSpec
from tests.test_net import make_net, relerr


@pytest.mark.parametrize("n,c,H,W,relu,bn,slice_", [(2, 8, 28, 28, 1, True, False), (3, 5, 14, 14, 1, True, True),
                                                     (1, 4, 7, 7, 0, False, False), (2, 3, 9, 12, 1, True, True),
                                                     (1, 2, 1, 4, 1, True, False)])
def test_avgpool_affine_matches_layer_sequence(backend, n, c, H, W, relu, bn, slice_):
    rng = np.random.default_rng(n * 100 + c * 10 + W)
    x = rng.standard_normal((n, c, H, W)).astype(np.float32)
    b = rng.standard_normal(c).astype(np.float32)
    sc, sh = rng.uniform(0.5, 1.5, c).astype(np.float32), rng.standard_normal(c).astype(np.float32)
    ref = orc.pooling(x, "AVE", (3, 3), (1, 1), (1, 1)) + b[None, :, None, None]
    if bn:
        ref = ref * sc[None, :, None, None] + sh[None, :, None, None]
    if relu:
        ref = np.maximum(ref, 0)
    S = H * W
    c0, wide = (3, c + 5) if slice_ else (0, c)          # channels [c0, c0 + c) of a wider (Concat) tensor
    big = backend.dev(np.full((n, wide, H, W), 7.0, np.float32))
    dst = hip.View(backend.ptr(big, c0 * S), wide * S, 0, S, 1)
    backend.lib.avgpool_affine_forward(backend.ptr(backend.dev(x)), backend.ptr(backend.dev(b)),
                                       backend.ptr(backend.dev(sc)) if bn else None, backend.ptr(backend.dev(sh)) if bn else None,
                                       relu, dst, n, c, H, W)
    got = backend.host(big, (n, wide, H, W))
    assert relerr(got[:, c0:c0 + c], ref) < 2e-6
    assert (got[:, :c0] == 7.0).all() and (got[:, c0 + c:] == 7.0).all()


def test_avgpool_affine_rejects_bad_arguments(backend):
    with pytest.raises(hip.EcoError, match="avgpool_affine"):
        backend.lib.avgpool_affine_forward(0, None, None, None, 1, hip.null_view(), 1, 1, 4, 4)


@pytest.mark.parametrize("variant", ["lite", "full"])
def test_engine_runs_pool_proj_ahead_of_its_pool(backend, variant):
    """Channels / 2 keeps inception_3a / 3b's pool_proj at 16 / 32 output channels: 3b's (32) joins the block's three
    sibling 1x1 convs as a fourth member, 3a's (16: not a multiple of 32) keeps the reference order.  Logits and the
    block outputs still match the oracle; pool_commute=False restores pool -> conv."""
    gen = models.eco_lite_deploy if variant == "lite" else models.eco_full_deploy
    proto = gen(num_segments=4, num_clips=1, num_classes=10, input_size=32, width_div=2)
    spec = NetSpec.from_prototxt(proto)
    assert spec.layer("inception_3b_pool_proj").geom["cout"] == 32
    params = fillers.synthetic_params(spec, seed=11)
    x = fillers.synthetic_frames(4, 32, 32, seed=4)
    end = "inception_3b_output"                # the 2-D head up to the second block (the rest is slow on the emulator)
    ref = orc.forward(spec, params, {"data": x}, keep=["inception_3b_output", "inception_3a_output"])
    net = make_net(backend, proto, params, True)
    labels = net.op_labels()
    ahead = [l for l in labels if "inception_3b_pool_proj [ahead of inception_3b_pool]" in l]
    assert len(ahead) == 1 and ahead[0].count(" | ") == 3, labels           # one launch, four members
    assert any(l.startswith("inception_3b_pool+inception_3b_pool_proj") and "average" in l for l in labels)
    assert "inception_3b_pool" not in labels and "inception_3a_pool" in labels
    net.blobs["data"].data[...] = x
    net.forward(end=end)
    for name in ("inception_3a_output", "inception_3b_output"):
        assert relerr(net.blobs[name].data, ref[name]) < 2e-5, name
    with pytest.raises(KeyError, match="linear maps exchanged"):
        net.blobs["inception_3b_pool"].data
    # parameters changed after the build reach the concatenated group (the member's bias stays out of the conv)
    net.params["inception_3b_pool_proj"][1].data[...] += 0.5
    params2 = dict(params)                      # (the engine may share the caller's arrays: read the new values back)
    params2["inception_3b_pool_proj"] = [np.array(b.data) for b in net.params["inception_3b_pool_proj"]]
    net.forward(end=end)
    ref2 = orc.forward(spec, params2, {"data": x}, keep=["inception_3b_output"])
    assert relerr(net.blobs["inception_3b_output"].data, ref2["inception_3b_output"]) < 2e-5
    # the reference order, on request
    plain = make_net(backend, proto, params2, True, pool_commute=False)
    assert "inception_3b_pool" in plain.op_labels()
    plain.blobs["data"].data[...] = x
    plain.forward(end=end)
    assert relerr(plain.blobs["inception_3b_output"].data, ref2["inception_3b_output"]) < 2e-5
