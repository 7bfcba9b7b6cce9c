"""Independent cross-check of the NumPy oracle against torch-CPU functional ops
(SURVEY.md section 7 'Cross-check mapping'): Caffe conv == F.conv2d/conv3d (cross-correlation, same
weight layout); MAX pool == F.max_pool2d(ceil_mode=True); AVE 3x3 s1 p1 == F.avg_pool2d(
count_include_pad=True); global_pool == mean over (D,H,W); BN(TEST) == F.batch_norm(eval);
Permute == Tensor.permute.  Also the whole reduced ECO-Lite net."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import eco_oracle as orc
from eco_amd import fillers, models
from eco_amd.netspec import NetSpec, pooled_dim


def t(x):
    return torch.from_numpy(np.ascontiguousarray(x))


@pytest.mark.parametrize("nd,k,s,p", [(2, 7, 2, 3), (2, 3, 1, 1), (2, 1, 1, 0), (2, 3, 2, 1), (3, 3, 1, 1), (3, 3, 2, 1)])
def test_conv(nd, k, s, p):
    rng = np.random.default_rng(0)
    sp = (11, 10) if nd == 2 else (5, 7, 6)
    x = rng.standard_normal((2, 5) + sp).astype(np.float32)
    w = rng.standard_normal((6, 5) + (k,) * nd).astype(np.float32)
    b = rng.standard_normal(6).astype(np.float32)
    y = orc.convolution(x, w, b, (k,) * nd, (s,) * nd, (p,) * nd)
    f = F.conv2d if nd == 2 else F.conv3d
    ref = f(t(x), t(w), t(b), stride=s, padding=p).numpy()
    assert y.shape == ref.shape and np.abs(y - ref).max() < 1e-4 * np.abs(ref).max()


@pytest.mark.parametrize("h", [112, 56, 28, 14, 7, 13])
def test_maxpool_ceil_mode(h):
    rng = np.random.default_rng(1)
    x = rng.standard_normal((2, 3, h, h)).astype(np.float32)
    y = orc.pooling_fast(x, "MAX", (3, 3), (2, 2), (0, 0))
    ref = F.max_pool2d(t(x), 3, 2, ceil_mode=True).numpy()
    assert y.shape == ref.shape == (2, 3, pooled_dim(h, 3, 2, 0), pooled_dim(h, 3, 2, 0))  # shapes asserted, not assumed
    assert np.array_equal(y, ref)
    y = orc.pooling_fast(x, "MAX", (3, 3), (1, 1), (1, 1))
    assert np.array_equal(y, F.max_pool2d(t(x), 3, 1, 1).numpy())


def test_avgpool_and_global():
    rng = np.random.default_rng(2)
    x = rng.standard_normal((2, 4, 28, 28)).astype(np.float32)
    y = orc.pooling_fast(x, "AVE", (3, 3), (1, 1), (1, 1))
    ref = F.avg_pool2d(t(x), 3, 1, 1, count_include_pad=True).numpy()
    assert np.abs(y - ref).max() < 1e-5
    v = rng.standard_normal((2, 6, 4, 7, 7)).astype(np.float32)
    g = orc.pooling_fast(v, "AVE", (4, 7, 7), (1, 1, 1), (0, 0, 0))
    assert g.shape == (2, 6, 1, 1, 1) and np.abs(g.reshape(2, 6) - v.mean((2, 3, 4))).max() < 1e-5
    y3 = orc.pooling(v, "AVE", (3, 3, 3), (1, 1, 1), (1, 1, 1))
    assert np.abs(y3 - F.avg_pool3d(t(v), 3, 1, 1, count_include_pad=True).numpy()).max() < 1e-5


@pytest.mark.parametrize("shape", [(3, 5, 6, 4), (2, 5, 3, 6, 4)])
def test_bn(shape):
    rng = np.random.default_rng(3)
    x = rng.standard_normal(shape).astype(np.float32)
    C = shape[1]
    g, b = rng.uniform(0.5, 1.5, C).astype(np.float32), rng.standard_normal(C).astype(np.float32)
    m, v = rng.standard_normal(C).astype(np.float32), rng.uniform(0.5, 1.5, C).astype(np.float32)
    y = orc.bn_inference(x, g, b, m, v, 1e-5)
    ref = F.batch_norm(t(x), t(m), t(v), t(g), t(b), training=False, eps=1e-5).numpy()
    assert np.abs(y - ref).max() < 1e-5
    from eco_amd.engine import fold_bn
    a, c = fold_bn([g, b, m, v], 1e-5)  # the folded form the HIP path uses
    bs = (1, C) + (1,) * (len(shape) - 2)
    assert np.abs(x * a.reshape(bs) + c.reshape(bs) - ref).max() < 1e-5


def torch_forward(spec, params, x):
    blobs = {"data": t(x)}
    for L in spec.layers:
        b = [blobs[n] for n in L.bottoms]
        g = L.geom
        if L.type == "Convolution":
            f = F.conv2d if g["nsp"] == 2 else F.conv3d
            y = f(b[0], t(params[L.name][0]), t(params[L.name][1]), stride=g["stride"], padding=g["pad"])
        elif L.type == "BN":
            p = params[L.name]
            y = F.batch_norm(b[0], t(p[2].reshape(-1)), t(p[3].reshape(-1)), t(p[0].reshape(-1)), t(p[1].reshape(-1)),
                             training=False, eps=1e-5)
        elif L.type == "ReLU":
            y = F.relu(b[0])
        elif L.type == "Pooling":
            if g["method"] == "MAX":
                y = F.max_pool2d(b[0], g["kernel"], g["stride"], g["pad"], ceil_mode=True)
            elif g["nsp"] == 2:
                y = F.avg_pool2d(b[0], g["kernel"], g["stride"], g["pad"], ceil_mode=True, count_include_pad=True)
            else:
                y = F.avg_pool3d(b[0], g["kernel"], g["stride"], g["pad"], count_include_pad=True)
        elif L.type == "Concat":
            y = torch.cat(b, g["axis"])
        elif L.type == "Eltwise":
            y = b[0] + b[1]
        elif L.type == "Reshape":
            y = b[0].reshape(L.top_shapes[0])
        elif L.type == "Permute":
            y = b[0].permute(*g["order"]).contiguous()
        elif L.type in ("Dropout",):
            y = b[0]
        elif L.type == "Split":
            for n in L.tops:
                blobs[n] = b[0]
            continue
        elif L.type == "InnerProduct":
            y = F.linear(b[0].reshape(g["M"], g["K"]), t(params[L.name][0]), t(params[L.name][1]))
        else:
            raise NotImplementedError(L.type)
        assert tuple(y.shape) == tuple(L.top_shapes[0]), L.name
        blobs[L.tops[0]] = y
    return blobs


@pytest.mark.parametrize("variant", ["lite", "full"])
def test_whole_net_against_torch(variant):
    gen = models.eco_lite_deploy if variant == "lite" else models.eco_full_deploy
    spec = NetSpec.from_prototxt(gen(num_segments=4, num_clips=2, num_classes=10, input_size=32, width_div=8))
    params = fillers.synthetic_params(spec, seed=5)
    x = fillers.synthetic_frames(8, 32, 32, seed=6)
    ref = torch_forward(spec, params, x)
    got = orc.forward(spec, params, {"data": x}, keep="all", fast_pool=False)
    for name in ("pool2_3x3_s2", "inception_3a_output", "res2b_bn", "res3b", "res5b_bn", "fc8"):
        r = ref[name].numpy()
        assert np.abs(got[name] - r).max() < 2e-5 * np.abs(r).max(), name
