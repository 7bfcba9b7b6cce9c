"""Regression tests for the round-2 advisor findings that can be exercised on one device."""
import numpy as np
import pytest

import eco_oracle as orc
from eco_amd import fillers, models
from eco_amd.netspec import NetSpec
from tests.test_net import make_net, relerr


def test_partial_forward_from_an_absorbed_sibling_reruns_its_group(backend):
    """engine.py `_try_fuse_siblings` emits every member at the first member's position.  A forward that STARTS at a
    later member (or at a layer fused into any earlier group) used to skip that layer's launch silently; now every
    launch knows the layers it stands for and runs when any of them lies in [start, end]."""
    proto = models.eco_lite_deploy(num_segments=4, num_clips=1, num_classes=10, input_size=32, width_div=2)
    spec = NetSpec.from_prototxt(proto)
    params = fillers.synthetic_params(spec, seed=3)
    x = fillers.synthetic_frames(4, 32, 32, seed=9)
    net = make_net(backend, proto, params, True)
    group = [l for l in net.op_labels() if l.startswith("inception_3a_1x1") and " | " in l]
    assert len(group) == 1 and "inception_3a_double_3x3_reduce" in group[0]
    names = [L.name for L in spec.layers]
    ops = net._engine.ops
    cover = next(m["layers"] for _, lab, _, m in ops if lab == group[0])
    assert names.index("inception_3a_double_3x3_reduce") in cover and names.index("inception_3a_3x3_reduce_bn") in cover
    net.blobs["data"].data[...] = x
    net.forward(end="inception_3a_output")
    ref = orc.forward(spec, params, {"data": x}, keep=["pool2_3x3_s2", "inception_3a_output"])
    assert relerr(net.blobs["inception_3a_output"].data, ref["inception_3a_output"]) < 2e-5
    # change the block's input on the host, then run from a MEMBER of the sibling group to the block's end
    x2 = (ref["pool2_3x3_s2"] * 0.5 + 1.0).astype(np.float32)
    net.blobs["pool2_3x3_s2"].data[...] = x2
    net.forward(start="inception_3a_double_3x3_reduce", end="inception_3a_output")
    # reference: the same block on the modified input
    i0, i1 = names.index("pool2_3x3_s2") + 1, names.index("inception_3a_output")   # from the Split behind pool2
    blobs = {"pool2_3x3_s2": x2}
    sub = orc.forward(_SubSpec(spec, i0, i1, "pool2_3x3_s2"), params, blobs, keep=["inception_3a_output"])
    # the branches at or behind the start layer (double_3x3: channels 64-111, pool_proj: 112-127) see the new input;
    # the reference would leave the 1x1 / 3x3 branches (layers ahead of `start`) as they were, the fused plan re-runs
    # the part of them that shares the member's launch -- either way they are not what this test is about
    c0 = spec.layer("inception_3a_1x1").geom["cout"] + spec.layer("inception_3a_3x3").geom["cout"]
    got = net.blobs["inception_3a_output"].data
    assert relerr(got[:, c0:], sub["inception_3a_output"][:, c0:]) < 2e-5
    assert relerr(got[:, c0:], ref["inception_3a_output"][:, c0:]) > 1e-2      # (and it did change)


class _SubSpec:
    """Layers [i0, i1] of a NetSpec as a net of their own (oracle-side helper)."""

    def __init__(self, spec, i0, i1, inp):
        self.layers = spec.layers[i0:i1 + 1]
        self.outputs = [self.layers[-1].tops[0]]
        self.inputs = [inp]


def test_plans_for_the_current_device_agree(backend):
    """eco_conv / eco_wgemm / eco_convb plans built with num_cu = 0 all ask the current device (256 CUs on MI355X
    and in the emulator build); round 2 hard-coded 256 in two of the three planners."""
    from eco_amd import hip
    lib = backend.lib
    assert lib.wgemm_plan(2, 64, 64, 1, 7, 7, 1, None).bn == lib.wgemm_plan(2, 64, 64, 1, 7, 7, 1, 256).bn
    g = hip.conv_geom(2, 64, 64, (14, 14), (3, 3), (1, 1), (1, 1), (14, 14))
    a, b = lib.convb_plan(g, hip.DT_BF16, None), lib.convb_plan(g, hip.DT_BF16, 256)
    assert (a.bm, a.bn, a.ksplit) == (b.bm, b.bn, b.ksplit)
