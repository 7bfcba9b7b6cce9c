"""Regression tests for the round-1 advisor findings: blob contents across a redundant
``net.reshape()``, fusion of a BN that is not the first consumer of a conv top, the hipGraph cache key,
alias blobs sharing one host mirror, and parameter-shape validation on rebuild.  Reference behaviours:
Net::Reshape keeps blob contents (net.cpp:843-849), layers run in file order (net.cpp:566-583), Reshape /
Split tops share their bottom's SyncedMemory (reshape_layer.cpp:88), InnerProduct CHECKs its weight shape."""
import numpy as np
import pytest

import eco_oracle as orc
from eco_amd import fillers, models
from eco_amd.net import Net
from eco_amd.netspec import NetSpec, NetSpecError
from tests.test_net import make_net, mini, relerr

TOL = 2e-5


def test_redundant_reshape_keeps_inputs(backend):
    proto = mini("lite")
    spec = NetSpec.from_prototxt(proto)
    params = fillers.synthetic_params(spec, seed=2)
    x = fillers.synthetic_frames(8, 32, 32, seed=4)
    ref = orc.forward(spec, params, {"data": x})["fc8"]
    net = make_net(backend, proto, params, True)
    net.blobs["data"].data[...] = x
    a = net.forward()["fc8"].copy()
    net.reshape()                       # pycaffe scripts call this redundantly
    b = net.forward()["fc8"].copy()     # the input's head is SYNCED: nothing is re-uploaded
    assert relerr(a, ref) < TOL and np.array_equal(a, b)
    # same with the frames resident on the device only (head = DEVICE)
    net2 = make_net(backend, proto, params, True)
    net2.blobs["data"].data[...] = x
    net2.forward()
    net2.blobs["data"]._host = None
    net2.blobs["data"]._head = 1        # _HEAD_DEVICE
    net2.reshape()
    assert np.array_equal(net2.forward()["fc8"], a)


PROTO_RELU_BEFORE_BN = """
name: "relu_before_bn"
input: "data" input_dim: 2 input_dim: 16 input_dim: 8 input_dim: 8
layer { name: "c1" type: "Convolution" bottom: "data" top: "c1"
        convolution_param { num_output: 16 kernel_size: 3 pad: 1 } }
layer { name: "r0" type: "ReLU" bottom: "c1" top: "c1" }
layer { name: "bn1" type: "BN" bottom: "c1" top: "bn1" bn_param { frozen: true } }
layer { name: "r1" type: "ReLU" bottom: "bn1" top: "bn1" }
"""


def test_bn_behind_inplace_layer_is_not_fused(backend):
    spec = NetSpec.from_prototxt(PROTO_RELU_BEFORE_BN)
    params = fillers.synthetic_params(spec, seed=5)
    x = np.random.default_rng(0).normal(size=(2, 16, 8, 8)).astype(np.float32)
    ref = orc.forward(spec, params, {"data": x})["bn1"]
    outs = {}
    for fuse in (False, True):
        net = make_net(backend, PROTO_RELU_BEFORE_BN, params, fuse)
        net.blobs["data"].data[...] = x
        outs[fuse] = net.forward()["bn1"].copy()
        assert relerr(outs[fuse], ref) < TOL, fuse
        if fuse:  # the in-place ReLU runs between the conv and the BN: the BN stays its own launch
            assert not any("c1+bn1" in l for l in net.op_labels())
    assert relerr(outs[True], outs[False]) < 1e-6


def test_alias_blob_shares_mirror_with_input(backend):
    """Reshape tops of an input alias its storage: writing the input after having read the alias must not be
    clobbered by a stale alias copy, and the alias sees what the input holds."""
    proto = """
    name: "alias"
    input: "data" input_dim: 4 input_dim: 16 input_dim: 4 input_dim: 4
    layer { name: "rs" type: "Reshape" bottom: "data" top: "data_reshape"
            reshape_param { shape { dim: 2 dim: 32 dim: 4 dim: 4 } } }
    layer { name: "c1" type: "Convolution" bottom: "data_reshape" top: "c1"
            convolution_param { num_output: 16 kernel_size: 1 } }
    """
    spec = NetSpec.from_prototxt(proto)
    params = fillers.synthetic_params(spec, seed=1)
    rng = np.random.default_rng(3)
    x1 = rng.normal(size=(4, 16, 4, 4)).astype(np.float32)
    x2 = rng.normal(size=(4, 16, 4, 4)).astype(np.float32)
    net = make_net(backend, proto, params, True)
    net.blobs["data"].data[...] = x1
    net.forward()
    assert np.array_equal(net.blobs["data_reshape"].data, x1.reshape(2, 32, 4, 4))   # alias read
    net.blobs["data"].data[...] = x2
    got = net.forward()["c1"].copy()
    ref = orc.forward(spec, params, {"data": x2})["c1"]
    assert relerr(got, ref) < TOL
    assert np.shares_memory(net.blobs["data_reshape"].data, net.blobs["data"].data)


def test_graph_cache_key_is_a_generation_counter(backend):
    proto = mini("lite")
    spec = NetSpec.from_prototxt(proto)
    net = make_net(backend, proto, fillers.synthetic_params(spec, seed=2), True)
    g0 = net._engine.generation
    net.reshape()
    g1 = net._engine.generation
    net.reshape()
    assert g0 < g1 < net._engine.generation
    net.params["fc8"][0].data[...] *= 2.0       # parameter edit -> re-upload -> new generation
    net._engine._sync_params()
    assert net._engine.generation > g1 + 1


def test_param_shapes_checked_on_reshape(backend):
    """A spatial reshape that changes an fc layer's K must raise (the reference CHECK-fails) instead of reading
    num_output*K_new floats from a buffer sized for K_old."""
    proto = """
    name: "fc_k"
    input: "data" input_dim: 2 input_dim: 4 input_dim: 6 input_dim: 6
    layer { name: "pool" type: "Pooling" bottom: "data" top: "pool"
            pooling_param { pool: AVE kernel_size: 3 stride: 3 } }
    layer { name: "fc" type: "InnerProduct" bottom: "pool" top: "fc" inner_product_param { num_output: 5 } }
    """
    spec = NetSpec.from_prototxt(proto)
    net = make_net(backend, proto, fillers.synthetic_params(spec, seed=1), True)
    net.blobs["data"].data[...] = 1.0
    net.forward()
    net.blobs["data"].reshape(2, 4, 9, 9)      # pool 2x2 -> 3x3: K 16 -> 36
    with pytest.raises(NetSpecError, match="parameter shapes"):
        net.reshape()
