"""Regression tests for the round-3 advisor findings.

1. (medium) pool_commute on the blocked (bf16) path: a commuted 1x1 conv that is the ONLY member of its launch, or that
   LEADS its sibling group (the pool branch listed before the other 1x1 convs), used to look its input up under the pool's
   top -- a blob that is never materialised once the pool launch is skipped -- and failed at build time (KeyError).
2. (low) a commuted conv whose epilogue needs a second / raw / residual destination must not silently lose it.
"""
import numpy as np
import pytest

import eco_oracle as orc
from eco_amd import fillers
from eco_amd.net import Net
from eco_amd.netspec import NetSpec

HEAD = """
input: "data" input_dim: 2 input_dim: 3 input_dim: 32 input_dim: 32
layer { name: "conv1" type: "Convolution" bottom: "data" top: "conv1" convolution_param { num_output: 64 kernel_size: 7 stride: 2 pad: 3 } }
layer { name: "conv1_bn" type: "BN" bottom: "conv1" top: "conv1_bn" }
layer { name: "conv1_relu" type: "ReLU" bottom: "conv1_bn" top: "conv1_bn" }
layer { name: "pool1" type: "Pooling" bottom: "conv1_bn" top: "pool1" pooling_param { pool: MAX kernel_size: 3 stride: 2 } }
"""
POOL_BRANCH = """
layer { name: "pool" type: "Pooling" bottom: "pool1" top: "pool" pooling_param { pool: AVE kernel_size: 3 stride: 1 pad: 1 } }
layer { name: "pool_proj" type: "Convolution" bottom: "pool" top: "pool_proj" convolution_param { num_output: 32 kernel_size: 1 } }
layer { name: "pool_proj_bn" type: "BN" bottom: "pool_proj" top: "pool_proj_bn" }
layer { name: "pool_proj_relu" type: "ReLU" bottom: "pool_proj_bn" top: "pool_proj_bn" }
"""
ONE_BY_ONE = """
layer { name: "b1" type: "Convolution" bottom: "pool1" top: "b1" convolution_param { num_output: 32 kernel_size: 1 } }
layer { name: "b1_bn" type: "BN" bottom: "b1" top: "b1_bn" }
layer { name: "b1_relu" type: "ReLU" bottom: "b1_bn" top: "b1_bn" }
"""
TAIL_SOLO = """
layer { name: "gp" type: "Pooling" bottom: "pool_proj_bn" top: "gp" pooling_param { pool: AVE kernel_size: 8 stride: 1 } }
layer { name: "gp_reshape" type: "Reshape" bottom: "gp" top: "gp_reshape" reshape_param { shape { dim: -1 dim: 32 } } }
layer { name: "fc8" type: "InnerProduct" bottom: "gp_reshape" top: "fc8" inner_product_param { num_output: 10 } }
"""
TAIL_GROUP = """
layer { name: "cat" type: "Concat" bottom: "pool_proj_bn" bottom: "b1_bn" top: "cat" }
layer { name: "gp" type: "Pooling" bottom: "cat" top: "gp" pooling_param { pool: AVE kernel_size: 8 stride: 1 } }
layer { name: "gp_reshape" type: "Reshape" bottom: "gp" top: "gp_reshape" reshape_param { shape { dim: -1 dim: 64 } } }
layer { name: "fc8" type: "InnerProduct" bottom: "gp_reshape" top: "fc8" inner_product_param { num_output: 10 } }
"""


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("case", ["solo", "leads_group"])
def test_commuted_conv_alone_or_leading_its_group(backend, dtype, case):
    proto = HEAD + POOL_BRANCH + (TAIL_SOLO if case == "solo" else ONE_BY_ONE + TAIL_GROUP)
    spec = NetSpec.from_prototxt(proto)
    params = fillers.synthetic_params(spec, seed=3)
    x = fillers.synthetic_frames(2, 32, 32, seed=9)
    kw = {"_backend": (backend.lib, backend.alloc)} if backend.kind == "emu" else {}
    if dtype != "f32":
        kw["dtype"] = dtype
    net = Net(proto, params=params, **kw)                       # used to raise KeyError('pool') at dtype="bf16"
    labels = net.op_labels()
    assert any("pool_proj [ahead of pool]" in l for l in labels), labels
    assert any(l.startswith("pool+") and "average" in l for l in labels), labels
    if case == "leads_group":
        lead = [l for l in labels if "pool_proj [ahead of pool]" in l][0]
        assert lead.startswith("pool_proj") and " | " in lead, lead      # the commuted conv is the group's first member
    out = net.forward(data=x)["fc8"].copy()
    ref = orc.forward(spec, params, {"data": x})["fc8"]
    tol = 1e-2 if dtype == "bf16" else 1e-4
    assert np.abs(out - ref).max() <= tol * np.abs(ref).max()
    # the reference order gives the same logits
    net2 = Net(proto, params=params, pool_commute=False, **kw)
    assert not any("ahead of" in l for l in net2.op_labels())
    out2 = net2.forward(data=x)["fc8"]
    assert np.abs(out2 - ref).max() <= tol * np.abs(ref).max()
