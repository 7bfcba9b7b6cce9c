"""`.caffemodel` I/O (Net.copy_from / Net.save; net.cpp:852-883): pure-Python protobuf wire codec,
round trip, interoperability with the reference's own schema (container only), load-by-layer-name."""
import os
import sys

import numpy as np
import pytest

from eco_amd import caffemodel, fillers, models
from eco_amd.net import Net
from eco_amd.netspec import NetSpec
from tests.conftest import HAVE_REFERENCE, REFERENCE


def mini():
    return models.eco_lite_deploy(num_segments=4, num_clips=1, num_classes=10, input_size=32, width_div=8)


def test_roundtrip(tmp_path):
    spec = NetSpec.from_prototxt(mini())
    p = fillers.synthetic_params(spec, seed=1)
    f = str(tmp_path / "w.caffemodel")
    caffemodel.write_caffemodel(f, spec, p)
    q = caffemodel.read_caffemodel(f)
    assert set(q) == set(p)
    for k in p:
        assert len(p[k]) == len(q[k]) and all(a.shape == b.shape and np.array_equal(a, b) for a, b in zip(p[k], q[k]))
    with open(f, "rb") as fh:
        raw = fh.read()
    with open(f, "wb") as fh:
        fh.write(raw[:len(raw) // 2])
    with pytest.raises(caffemodel.CaffemodelError):
        caffemodel.read_caffemodel(f)


def test_net_save_and_copy_from(backend, tmp_path):
    proto = mini()
    spec = NetSpec.from_prototxt(proto)
    p = fillers.synthetic_params(spec, seed=2)
    kw = {"_backend": (backend.lib, backend.alloc)} if backend.kind == "emu" else {}
    a = Net(proto, params=p, **kw)
    x = fillers.synthetic_frames(4, 32, 32, seed=9)
    ya = a.forward(data=x)["fc8"].copy()
    f = str(tmp_path / "net.caffemodel")
    a.save(f)
    b = Net(proto, f, 1, **kw)                       # caffe.Net(prototxt, weights, caffe.TEST)
    yb = b.forward(data=x)["fc8"]
    assert np.array_equal(ya, yb)
    c = Net(proto, 1, **kw)                          # filler init (BN var 0 -> huge/inf logits), then load
    c.copy_from(f)
    assert np.array_equal(c.forward(data=x)["fc8"], ya)
    # a layer absent from the net is ignored; a blob-count mismatch is an error (net.cpp:861-870)
    extra = dict(p)
    extra["not_in_net"] = [np.zeros(3, np.float32)]
    caffemodel.write_caffemodel(f, type("S", (), {"name": "x", "layers": list(spec.layers) + [type("L", (), {"name": "not_in_net", "type": "X"})()]})(), extra)
    c.copy_from(f)
    bad = dict(p)
    bad["fc8"] = [p["fc8"][0]]
    caffemodel.write_caffemodel(f, spec, bad)
    with pytest.raises(ValueError, match="Incompatible number of blobs"):
        c.copy_from(f)


@pytest.mark.skipif(not HAVE_REFERENCE, reason="needs /root/reference")
def test_interop_with_reference_schema(tmp_path):
    os.environ.setdefault("PROTOCOL_BUFFERS_PYTHON_IMPLEMENTATION", "python")
    sys.path.insert(0, os.path.join(REFERENCE, "caffe_3d/python/caffe/proto"))
    try:
        import caffe_pb2
    except Exception as e:  # pragma: no cover
        pytest.skip(f"reference caffe_pb2 not importable here: {e}")
    spec = NetSpec.from_prototxt(mini())
    p = fillers.synthetic_params(spec, seed=3)
    f = str(tmp_path / "w.caffemodel")
    caffemodel.write_caffemodel(f, spec, p)
    net = caffe_pb2.NetParameter()
    net.ParseFromString(open(f, "rb").read())      # our file parses with the reference schema
    assert net.name == spec.name and [l.name for l in net.layer] == [L.name for L in spec.layers if L.name in p]
    for l in net.layer:
        for b, ref in zip(l.blobs, p[l.name]):
            assert list(b.shape.dim) == list(ref.shape)
            assert np.array_equal(np.array(b.data, np.float32).reshape(ref.shape), ref)
    # a file written by the reference schema (new-style shape, legacy 4-D dims, V1 `layers`) reads back
    n2 = caffe_pb2.NetParameter()
    l = n2.layer.add(); l.name = "fc"; l.type = "InnerProduct"
    b = l.blobs.add(); b.num, b.channels, b.height, b.width = 1, 1, 2, 3; b.data.extend([1, 2, 3, 4, 5, 6])
    l1 = n2.layers.add(); l1.name = "old"
    b = l1.blobs.add(); b.shape.dim.extend([2, 2]); b.data.extend([9, 8, 7, 6])
    open(f, "wb").write(n2.SerializeToString())
    r = caffemodel.read_caffemodel(f)
    assert r["fc"][0].shape == (1, 1, 2, 3) and r["old"][0].tolist() == [[9, 8], [7, 6]]


def test_bn_style_conversion_matches_reference_formulas():
    """python/bn_convert_style.py:17-24: var -> (var+eps)^-0.5 and back inv_std^-2 - eps, 4th blob of BN layers only."""
    spec = NetSpec.from_prototxt(mini())
    p = fillers.synthetic_params(spec, seed=3)
    bn = caffemodel.bn_layer_names(spec)
    assert bn and all(n.endswith("_bn") or n.endswith("bn") for n in bn)
    inv = caffemodel.convert_bn_style(p, bn, "var_to_inv_std", eps=1e-5)
    for n in bn:
        assert np.allclose(inv[n][3], 1.0 / np.sqrt(p[n][3].astype(np.float64) + 1e-5), rtol=1e-6)
        assert all(a is b for a, b in zip(inv[n][:3], p[n][:3]))
    back = caffemodel.convert_bn_style(inv, bn, "inv_std_to_var", eps=1e-5)
    for n in bn:
        assert np.allclose(back[n][3], p[n][3], rtol=2e-6, atol=1e-7)
    conv = [L.name for L in spec.layers if L.type == "Convolution"][0]
    assert inv[conv][0] is p[conv][0]
    with pytest.raises(ValueError, match="Unknown conversion"):
        caffemodel.convert_bn_style(p, bn, "nope")


def test_inv_std_style_file_loads_to_the_same_logits(backend, tmp_path):
    """A legacy inv-std-style .caffemodel read with bn_style="inv_std" gives the logits of the variance-style
    file; read as variances (the silent failure VERDICT r1 names) it does not."""
    proto = mini()
    spec = NetSpec.from_prototxt(proto)
    p = fillers.synthetic_params(spec, seed=4)
    kw = {"_backend": (backend.lib, backend.alloc)} if backend.kind == "emu" else {}
    x = fillers.synthetic_frames(4, 32, 32, seed=6)
    a = Net(proto, params=p, **kw)
    ya = a.forward(data=x)["fc8"].copy()
    f = str(tmp_path / "legacy.caffemodel")
    a.save(f, bn_style="inv_std")
    raw = caffemodel.read_caffemodel(f)
    some_bn = caffemodel.bn_layer_names(spec)[0]
    assert np.allclose(raw[some_bn][3], 1.0 / np.sqrt(p[some_bn][3] + 1e-5), rtol=1e-5)
    b = Net(proto, 1, **kw)
    b.copy_from(f, bn_style="inv_std")
    yb = b.forward(data=x)["fc8"].copy()
    assert np.abs(yb - ya).max() / np.abs(ya).max() < 1e-4
    b.copy_from(f)  # default style: the 4th blobs are taken for variances
    assert np.abs(b.forward(data=x)["fc8"] - ya).max() / np.abs(ya).max() > 1e-3
    with pytest.raises(ValueError, match="bn_style"):
        b.copy_from(f, bn_style="std")
