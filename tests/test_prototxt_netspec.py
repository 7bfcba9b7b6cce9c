"""Host logic: prototxt text-format parser, phase filtering, split insertion, shape rules, the
generated ECO graphs.  Where /root/reference exists (authoring container), the parser is
cross-checked against the reference's own generated schema (python/caffe/proto/caffe_pb2.py) and
the generated graphs against the reference prototxt files."""
import os
import sys

import numpy as np
import pytest

from eco_amd import models, prototxt
from eco_amd.netspec import TEST, TRAIN, NetSpec, NetSpecError, param_shapes, split_blob_name

from tests.conftest import HAVE_REFERENCE, REFERENCE

needs_ref = pytest.mark.skipif(not HAVE_REFERENCE, reason="needs /root/reference (authoring container)")


def test_parser_basics():
    m = prototxt.parse('''
      name: "n"  # comment
      input: "data" input_dim: 1 input_dim: 3 input_dim: 8 input_dim: 8
      layer { name: "c" type: "Convolution" bottom: "data" top: "c"
              convolution_param { num_output: 4 pad: [1, 1] kernel_size: [3,3] stride: 2
                                  weight_filler { type: "gaussian" std: 1e-2 } bias_term: false } }
      layer { name: 'p' type: "Pooling" bottom: "c" top: "p" pooling_param: { pool: AVE kernel_size: 2 } }
    ''')
    assert m.get("name") == "n" and m.getall("input_dim") == [1, 3, 8, 8]
    conv = m.getall("layer")[0].msg("convolution_param")
    assert conv.getall("pad") == [1, 1] and conv.getall("kernel_size") == [3, 3] and conv.get("stride") == 2
    assert conv.msg("weight_filler").get("std") == 0.01 and conv.get("bias_term") is False
    assert m.getall("layer")[1].msg("pooling_param").get("pool") == "AVE"
    assert isinstance(m.getall("layer")[1].msg("pooling_param").get("pool"), prototxt.Enum)
    again = prototxt.parse(m.to_text())  # round trip
    assert again.to_text() == m.to_text()
    for bad in ['layer { name: "x"', 'a: : 3', 'a { b: } }', '5: 3', 'a: "unterminated']:
        with pytest.raises(prototxt.ParseError):
            prototxt.parse(bad)


def test_phase_filter_and_splits():
    txt = '''
      input: "data" input_shape { dim: 2 dim: 4 dim: 6 dim: 6 }
      layer { name: "tr" type: "ReLU" bottom: "data" top: "a" include { phase: TRAIN } }
      layer { name: "te" type: "ReLU" bottom: "data" top: "a" include { phase: TEST } }
      layer { name: "x1" type: "ReLU" bottom: "a" top: "b" }
      layer { name: "x2" type: "ReLU" bottom: "a" top: "c" exclude { phase: TRAIN } }
      layer { name: "s" type: "Eltwise" bottom: "b" bottom: "c" top: "s" }
    '''
    spec = NetSpec.from_prototxt(txt, phase=TEST)
    names = [L.name for L in spec.layers]
    assert names == ["te", "a_te_0_split", "x1", "x2", "s"]          # insert_splits.cpp naming
    sp = spec.layer("a_te_0_split")
    assert sp.type == "Split" and sp.tops == [split_blob_name("te", "a", 0, 0), split_blob_name("te", "a", 0, 1)]
    assert spec.layer("x1").bottoms == ["a_te_0_split_0"] and spec.layer("x2").bottoms == ["a_te_0_split_1"]
    assert spec.outputs == ["s"] and spec.blob_shapes["s"] == (2, 4, 6, 6)
    with pytest.raises(NetSpecError):  # x2 excluded in TRAIN -> "c" unknown
        NetSpec.from_prototxt(txt, phase=TRAIN)
    with pytest.raises(NetSpecError, match="Unknown layer type"):
        NetSpec.from_prototxt('input: "d" input_dim: 1 input_dim: 1 input_dim: 1 input_dim: 1 '
                              'layer { name: "l" type: "LRN" bottom: "d" top: "l" }')
    with pytest.raises(NetSpecError, match="multiple sources"):
        NetSpec.from_prototxt('input: "d" input_dim: 1 input_dim: 1 input_dim: 1 input_dim: 1 '
                              'layer { name: "a" type: "ReLU" bottom: "d" top: "x" } '
                              'layer { name: "b" type: "ReLU" bottom: "d" top: "x" }')


def test_shape_rules():
    hdr = 'input: "d" input_shape { dim: 2 dim: 6 dim: 4 dim: 9 dim: 9 } '
    s = NetSpec.from_prototxt(hdr + 'layer { name: "c" type: "Convolution" bottom: "d" top: "c" convolution_param '
                                    '{ num_output: 5 kernel_size: [3,3,3] pad: [1,1,1] stride: [2,2,2] } }')
    assert s.blob_shapes["c"] == (2, 5, 2, 5, 5) and param_shapes(s.layer("c")) == [(5, 6, 3, 3, 3), (5,)]
    s = NetSpec.from_prototxt(hdr + 'layer { name: "r" type: "Reshape" bottom: "d" top: "r" reshape_param '
                                    '{ shape { dim: 0 dim: -1 dim: 9 } } }')
    assert s.blob_shapes["r"] == (2, 216, 9)                         # 0 copies, -1 inferred (reshape_layer.cpp)
    s = NetSpec.from_prototxt(hdr + 'layer { name: "p" type: "Permute" bottom: "d" top: "p" permute_param '
                                    '{ order: [0,2,1] } }')
    assert s.blob_shapes["p"] == (2, 4, 6, 9, 9) and s.layer("p").geom["order"] == [0, 2, 1, 3, 4]
    s = NetSpec.from_prototxt(hdr + 'layer { name: "g" type: "Pooling" bottom: "d" top: "g" pooling_param '
                                    '{ pool: AVE global_pooling: true } }')
    assert s.blob_shapes["g"] == (2, 6, 1, 1, 1)
    for bad, msg in [
        ('layer { name: "c" type: "Convolution" bottom: "d" top: "c" convolution_param { num_output: 5 kernel_size: [3,3] } }',
         "once per spatial"),
        ('layer { name: "c" type: "Convolution" bottom: "d" top: "c" convolution_param { num_output: 5 kernel_size: 3 group: 2 } }',
         "group"),
        ('layer { name: "r" type: "Reshape" bottom: "d" top: "r" reshape_param { shape { dim: -1 dim: 7 } } }', "divisible"),
        ('layer { name: "p" type: "Pooling" bottom: "d" top: "p" pooling_param { kernel_size: 2 pad: 2 } }', "pad"),
        ('layer { name: "p" type: "Permute" bottom: "d" top: "p" permute_param { order: [0,1,1] } }', "duplicate"),
    ]:
        with pytest.raises(NetSpecError, match=msg):
            NetSpec.from_prototxt(hdr + bad)


def test_generated_eco_graphs():
    lite = NetSpec.from_prototxt(models.eco_lite_deploy())
    assert len(lite.layers) == 116 and lite.inputs == ["data"] and lite.outputs == ["fc8"]
    from collections import Counter
    hist = Counter(L.type for L in lite.layers)
    assert (hist["Convolution"], hist["BN"], hist["ReLU"], hist["Pooling"], hist["Eltwise"], hist["Concat"],
            hist["Reshape"], hist["Permute"], hist["Dropout"], hist["InnerProduct"], hist["Split"]) == \
        (32, 30, 30, 5, 5, 2, 2, 1, 1, 1, 7)
    lite.reshape({"data": (512, 3, 224, 224)})
    assert lite.blob_shapes["res2b_bn"] == (32, 96, 16, 28, 28) and lite.blob_shapes["fc8"] == (32, 400)
    assert abs(lite.conv_fc_flops() / 1e9 - 2975.13) < 0.01
    full = NetSpec.from_prototxt(models.eco_full_deploy(num_clips=32))
    assert sum(L.type != "Split" for L in full.layers) == 281
    assert abs(full.conv_fc_flops() / 1e9 - 4122.43) < 0.01
    n32 = NetSpec.from_prototxt(models.eco_lite_deploy(num_segments=32, num_clips=32))
    assert n32.blob_shapes["global_pool"] == (32, 512, 1, 1, 1) and abs(n32.conv_fc_flops() / 1e9 - 5950.25) < 0.01
    n4 = NetSpec.from_prototxt(models.eco_lite_deploy(num_segments=4, num_clips=1))
    assert abs(n4.conv_fc_flops() / 1e9 - 23.24) < 0.01
    with pytest.raises(ValueError):
        models.eco_lite_deploy(num_segments=6)


@needs_ref
@pytest.mark.parametrize("gen,ref", [(models.eco_lite_deploy, "models_ECO_Lite/kinetics/deploy.prototxt"),
                                     (models.eco_full_deploy, "models_ECO_Full/kinetics/deploy.prototxt")])
def test_generated_graph_equals_reference_file(gen, ref):
    a = NetSpec.from_prototxt(gen())
    b = NetSpec.from_prototxt(os.path.join(REFERENCE, ref))
    assert a.name == b.name and a.inputs == b.inputs and a.input_shapes == b.input_shapes
    assert len(a.layers) == len(b.layers)
    for x, y in zip(a.layers, b.layers):
        assert (x.name, x.type, x.bottoms, x.tops, x.geom, x.top_shapes) == (y.name, y.type, y.bottoms, y.tops, y.geom, y.top_shapes)
        assert param_shapes(x) == param_shapes(y)


@needs_ref
def test_test_phase_of_trainval_prototxt_equals_generated_evaluator():
    """ECO_Lite.prototxt filtered to phase TEST (VideoData source, reshape_data, body, loss, top1, top5) is the
    graph models.test_phase_net builds around the deploy body."""
    b = NetSpec.from_prototxt(os.path.join(REFERENCE, "models_ECO_Lite/kinetics/ECO_Lite.prototxt"))  # TEST default
    a = NetSpec.from_prototxt(models.test_phase_net(models.eco_lite_deploy(num_clips=1), 16, batch_size=1))
    assert b.inputs == ["data", "label"] and b.input_shapes == {"data": (1, 48, 224, 224), "label": (1, 1, 1, 1)}
    assert a.inputs == b.inputs and a.input_shapes == b.input_shapes and sorted(b.outputs) == ["loss", "top1", "top5"]
    assert len(a.layers) == len(b.layers)
    for x, y in zip(a.layers, b.layers):
        assert (x.name, x.type, x.bottoms, x.tops, x.geom, x.top_shapes) == (y.name, y.type, y.bottoms, y.tops, y.geom, y.top_shapes)


@needs_ref
@pytest.mark.parametrize("ds,ncls", [("ucf101", 101), ("hmdb51", 51), ("something_something", 174)])
def test_other_datasets_differ_only_in_head(ds, ncls):
    b = NetSpec.from_prototxt(os.path.join(REFERENCE, "models_ECO_Lite", ds, "deploy.prototxt"))
    fc = [L for L in b.layers if L.type == "InnerProduct"][0]
    a = NetSpec.from_prototxt(models.eco_lite_deploy(num_classes=ncls, fc_name=fc.name))
    assert [(x.name, x.type, x.top_shapes) for x in a.layers] == [(y.name, y.type, y.top_shapes) for y in b.layers]


@needs_ref
@pytest.mark.parametrize("ref", ["models_ECO_Lite/kinetics/deploy.prototxt", "models_ECO_Full/kinetics/deploy.prototxt",
                                 "models_ECO_Lite/kinetics/ECO_Lite.prototxt"])
def test_parser_against_reference_schema(ref):
    """Parse the same file with the reference's own generated descriptor (caffe_pb2.py) and compare
    every field the hot path reads."""
    os.environ.setdefault("PROTOCOL_BUFFERS_PYTHON_IMPLEMENTATION", "python")
    sys.path.insert(0, os.path.join(REFERENCE, "caffe_3d/python/caffe/proto"))
    try:
        import caffe_pb2
        from google.protobuf import text_format
    except Exception as e:  # pragma: no cover
        pytest.skip(f"reference caffe_pb2 not importable here: {e}")
    path = os.path.join(REFERENCE, ref)
    pb = caffe_pb2.NetParameter()
    text_format.Merge(open(path).read(), pb)
    mine = prototxt.parse_file(path)
    layers = mine.getall("layer")
    assert len(layers) == len(pb.layer) and mine.get("name") == pb.name
    for a, b in zip(layers, pb.layer):
        assert (a.get("name"), a.get("type"), a.getall("bottom"), a.getall("top")) == (b.name, b.type, list(b.bottom), list(b.top))
        if b.type == "Convolution":
            p, q = a.msg("convolution_param"), b.convolution_param
            assert p.get("num_output") == q.num_output and p.getall("kernel_size") == list(q.kernel_size)
            assert p.getall("pad") == list(q.pad) and p.getall("stride") == list(q.stride)
        elif b.type == "Pooling":
            p, q = a.msg("pooling_param"), b.pooling_param
            assert str(p.get("pool", "MAX")) == caffe_pb2.PoolingParameter.PoolMethod.Name(q.pool)
            assert p.getall("kernel_size") == list(q.kernel_size) and p.getall("stride") == list(q.stride)
            assert p.getall("pad") == list(q.pad)
            assert (p.get("kernel_h", 0), p.get("kernel_w", 0)) == (q.kernel_h, q.kernel_w)
        elif b.type == "Reshape":
            assert a.msg("reshape_param").msg("shape").getall("dim") == list(b.reshape_param.shape.dim)
        elif b.type == "Permute":
            assert a.msg("permute_param").getall("order") == list(b.permute_param.order)
        elif b.type == "InnerProduct":
            assert a.msg("inner_product_param").get("num_output") == b.inner_product_param.num_output
        elif b.type == "Dropout":
            assert abs(a.msg("dropout_param").get("dropout_ratio") - b.dropout_param.dropout_ratio) < 1e-7
        assert len(a.getall("include")) == len(b.include)


def test_roofline_tool_rederives_survey_totals():
    """tools/roofline.py: algorithmic FLOPs and fused-model bytes of SURVEY.md section 8(d), from the shapes."""
    import importlib.util
    spec_ = importlib.util.spec_from_file_location("roofline", os.path.join(os.path.dirname(os.path.dirname(
        os.path.abspath(__file__))), "tools", "roofline.py"))
    rl = importlib.util.module_from_spec(spec_)
    spec_.loader.exec_module(rl)
    for gen, kw, gflop, fused_gb in [(models.eco_lite_deploy, dict(num_segments=16, num_clips=32), 2975.13, 18.54),
                                     (models.eco_full_deploy, dict(num_segments=16, num_clips=32), 4122.43, 31.30),
                                     (models.eco_lite_deploy, dict(num_segments=32, num_clips=32), 5950.25, 36.94),
                                     (models.eco_lite_deploy, dict(num_segments=4, num_clips=1), 23.24, 0.294)]:
        fl, fb, lb = rl.totals(NetSpec.from_prototxt(gen(**kw)))
        assert abs(fl / 1e9 - gflop) < 0.01 and abs(fb / 1e9 - fused_gb) < 0.006 and lb > fb


# --- graph parity against the committed extract of the reference's own prototxts ------------------------------
# tests/golden/reference_graphs.json is produced by tests/golden/make_reference_graphs.py from the reference
# files with the reference's generated schema and an independent restatement of FilterNet / InsertSplits / the
# per-layer Reshape rules (nothing of this package is imported there).  Unlike the needs_ref tests above it
# travels to the GPU box, so graph and shape inference is pinned there too.
def _reference_graphs():
    import json
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_graphs.json")) as f:
        return {g["file"]: g for g in json.load(f)["graphs"]}


def _geom_subset(L):
    """The fields of LayerSpec.geom that the extract records, in its spelling."""
    g, t = L.geom, L.type
    if t == "Convolution":
        return dict(kernel=list(g["kernel"]), stride=list(g["stride"]), pad=list(g["pad"]), cin=g["cin"], cout=g["cout"],
                    bias_term=g["bias_term"])
    if t == "Pooling":
        return dict(method=g["method"], kernel=list(g["kernel"]), stride=list(g["stride"]), pad=list(g["pad"]))
    if t == "BN":
        return dict(eps=g["eps"], frozen=g["frozen"], channels=g["channels"])
    if t == "ReLU":
        return dict(negative_slope=g["negative_slope"])
    if t == "Dropout":
        return dict(ratio=g["ratio"])
    if t == "Reshape":
        return dict(dims=[int(d) for d in L.param.msg("reshape_param").msg("shape").getall("dim")])
    if t == "Permute":
        return dict(order=list(g["order"]))
    if t == "Concat":
        return dict(axis=g["axis"])
    if t == "Eltwise":
        return dict(op=g["op"], coeff=list(g["coeff"]))
    if t == "InnerProduct":
        return dict(num_output=g["num_output"], K=g["K"], bias_term=g["bias_term"])
    if t == "Accuracy":
        return dict(top_k=g["top_k"])
    return {}


def _assert_spec_equals_extract(spec, ref):
    assert spec.name == ref["name"] and spec.inputs == ref["inputs"] and sorted(spec.outputs) == ref["outputs"]
    assert {k: list(v) for k, v in spec.input_shapes.items()} == ref["input_shapes"]
    assert len(spec.layers) == len(ref["layers"])
    for L, R in zip(spec.layers, ref["layers"]):
        assert (L.name, L.type, L.bottoms, L.tops) == (R["name"], R["type"], R["bottom"], R["top"])
        assert [list(s) for s in L.top_shapes] == R["top_shapes"], L.name
        mine, theirs = _geom_subset(L), R["geom"]
        assert set(mine) == set(theirs), L.name
        for k in mine:
            if isinstance(mine[k], float):
                assert abs(mine[k] - theirs[k]) <= 1e-6 * max(1.0, abs(theirs[k])), (L.name, k)
            else:
                assert mine[k] == theirs[k], (L.name, k, mine[k], theirs[k])


@pytest.mark.parametrize("backend_kind", ["cpu", pytest.param("gpu", marks=pytest.mark.gpu)])
@pytest.mark.parametrize("which", ["lite", "full", "ucf101", "hmdb51", "something_something"])
def test_generated_graph_equals_reference_extract(which, backend_kind):
    refs = _reference_graphs()
    if which == "lite":
        ref, spec = refs["models_ECO_Lite/kinetics/deploy.prototxt"], NetSpec.from_prototxt(models.eco_lite_deploy())
        assert len(ref["layers"]) == 116 and sum(r["type"] != "Split" for r in ref["layers"]) == 109
    elif which == "full":
        ref, spec = refs["models_ECO_Full/kinetics/deploy.prototxt"], NetSpec.from_prototxt(models.eco_full_deploy())
        assert sum(r["type"] != "Split" for r in ref["layers"]) == 281
    else:
        ref = refs[f"models_ECO_Lite/{which}/deploy.prototxt"]
        fc = [r for r in ref["layers"] if r["type"] == "InnerProduct"][0]
        drop = [r for r in ref["layers"] if r["type"] == "Dropout"][0]
        spec = NetSpec.from_prototxt(models.eco_lite_deploy(num_classes=fc["geom"]["num_output"], fc_name=fc["name"],
                                                            dropout_ratio=drop["geom"]["ratio"]))
        ref = dict(ref, name=spec.name)  # the dataset prototxts differ in net name, dropout ratio and fc only
    _assert_spec_equals_extract(spec, ref)


@pytest.mark.parametrize("backend_kind", ["cpu", pytest.param("gpu", marks=pytest.mark.gpu)])
def test_test_phase_evaluator_equals_reference_extract(backend_kind):
    """ECO_Lite.prototxt filtered to TEST (VideoData source -> inputs data/label, body, loss, top1, top5)."""
    ref = _reference_graphs()["models_ECO_Lite/kinetics/ECO_Lite.prototxt"]
    src = ref["source"]
    assert src["type"] == "VideoData" and src["top"] == ["data", "label"] and src["mean_value"] == [104.0, 117.0, 123.0]
    spec = NetSpec.from_prototxt(models.test_phase_net(models.eco_lite_deploy(num_clips=src["batch_size"]),
                                                       src["num_segments"], batch_size=src["batch_size"]))
    _assert_spec_equals_extract(spec, dict(ref, name=spec.name))
