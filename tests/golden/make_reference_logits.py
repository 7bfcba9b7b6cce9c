#!/usr/bin/env python
"""Full-width ECO logits from EXECUTED REFERENCE CODE -> tests/golden/reference_logits.json.

BASELINE.json configs[0] is "ECO-Lite num_segments=4, batch=1, random-init weights, synthetic 224x224 frames on caffe_3d
CPU forward".  The reference cannot be built whole in this image, but every layer type of its deploy
graphs now exists as the reference's own object code in oracle/_ref/libeco_ref.so (oracle/Makefile: im2col.cpp,
base_conv_layer.cpp, conv_layer.cpp, bn_layer.cpp, relu_layer.cpp, pooling_layer.cpp, concat_layer.cpp,
eltwise_layer.cpp, reshape_layer.cpp, permute_layer.cpp, inner_product_layer.cpp, compiled unmodified).  This script is
the missing Net::ForwardFromTo (net.cpp:566-583): it parses the reference's OWN deploy prototxt with the reference's OWN
generated schema (caffe_pb2.py), sets the clip geometry to num_segments=4 / one clip exactly where the authors' files carry
their num_segments=16 / 5-clip values (input_dim, the r2Dto3D reshape, the global_pool kernel; ECO-Full also
reshape_fc_st2 and segment_consensus_st2), and runs the layers in file order, each through the compiled class's
LayerSetUp / Reshape / Forward_cpu, on seeded weights and frames (tests/golden/ref_params.py).

Nothing of the product package (eco-efficient-video-understanding_amd/) and nothing of the NumPy oracle
(oracle/eco_oracle.py) is imported: the fixture is produced by reference object code + this 100-line walk only.

What the reference's CPU code cannot execute, and how it is evaluated here (both asserted, both recorded in the fixture):
  * BN on 5-D blobs (bn_layer.cpp:70-73 CHECK-fails through Blob::LegacyShape; the reference runs them on its cuDNN path,
    cudnn_bn_layer.cu) -> the compiled 4-D BNLayer on the blob folded to [n, c, d*h, w] (the arithmetic is per channel),
    eps = max(eps, 1e-5) as cudnn_bn_layer.cu:24 does;
  * 3-D pooling (pooling_layer.cpp:177-201 NOT_IMPLEMENTED for num_spatial_axes != 2; cuDNN Nd on the GPU): only
    `global_pool`, an AVE window that covers the whole (d, h, w) extent with no padding -> the compiled 2-D PoolingLayer
    on the blob folded to [n, c, d*h, w] with kernel [kd*kh, kw]: same elements, same summation order, same divisor.
  * Dropout in TEST phase is a copy (dropout_layer.cpp:37-50); Split never appears (multi-consumer blobs are read-only
    between their producer and consumers in these graphs, so sharing the array is what InsertSplits + ShareData does).

Run in the authoring container (needs /root/reference and oracle/_ref); the JSON is committed and is what the -m gpu test
tests/test_reference_logits.py compares the HIP logits with on the GPU box.
"""
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import make_reference_graphs as G      # noqa: E402  reference caffe_pb2 + text_format + filter_net + spatial()
import eco_ref                         # noqa: E402  ctypes binding of the compiled reference layers
import ref_params                      # noqa: E402

OUT = os.path.join(HERE, "reference_logits.json")
caffe_pb2, text_format = G.caffe_pb2, G.text_format
BN_MIN_EPS = 1e-5                      # CUDNN_BN_MIN_EPSILON, cudnn_bn_layer.cu:24


def load_net(rel_path, num_segments, num_clips):
    """The reference deploy file with the clip geometry set to (num_segments, num_clips).  The authors' files carry
    num_segments=16 in exactly these fields (models_ECO_Lite/kinetics/deploy.prototxt:4,1145,1669;
    models_ECO_Full/kinetics/deploy.prototxt:4,1254,4633,4634,4650) and edit them by hand for other lengths."""
    net = caffe_pb2.NetParameter()
    with open(os.path.join(G.REF, rel_path)) as f:
        text_format.Merge(f.read(), net)
    assert len(net.input) == 1 and len(net.input_dim) == 4
    edits = []
    net.input_dim[0] = num_segments * num_clips
    edits.append("input_dim[0]=%d" % net.input_dim[0])
    for lp in net.layer:
        if lp.name == "r2Dto3D":
            assert list(lp.reshape_param.shape.dim) == [-1, 16, 96, 28, 28]
            lp.reshape_param.shape.dim[1] = num_segments
            edits.append("r2Dto3D.dim[1]=%d" % num_segments)
        elif lp.name == "global_pool":
            assert list(lp.pooling_param.kernel_size) == [4, 7, 7] and num_segments % 4 == 0
            lp.pooling_param.kernel_size[0] = num_segments // 4
            edits.append("global_pool.kernel_size[0]=%d" % (num_segments // 4))
        elif lp.name == "reshape_fc_st2":
            assert list(lp.reshape_param.shape.dim) == [-1, 1, 16, 1024]
            lp.reshape_param.shape.dim[2] = num_segments
            edits.append("reshape_fc_st2.dim[2]=%d" % num_segments)
        elif lp.name == "segment_consensus_st2":
            assert lp.pooling_param.kernel_h == 16
            lp.pooling_param.kernel_h = num_segments
            edits.append("segment_consensus_st2.kernel_h=%d" % num_segments)
    return net, edits


def param_shapes(lp, bottom_shape):
    """Blob shapes as LayerSetUp creates them (base_conv_layer.cpp:138-150, inner_product_layer.cpp:26-43,
    bn_layer.cpp:24-41)."""
    if lp.type == "Convolution":
        p = lp.convolution_param
        nsp = len(bottom_shape) - 2
        k = G.spatial(p, "kernel_size", nsp, None)
        return [[int(p.num_output), bottom_shape[1]] + k] + ([[int(p.num_output)]] if p.bias_term else [])
    if lp.type == "InnerProduct":
        p = lp.inner_product_param
        K = int(np.prod(bottom_shape[p.axis:]))
        return [[int(p.num_output), K]] + ([[int(p.num_output)]] if p.bias_term else [])
    if lp.type == "BN":
        return [[1, bottom_shape[1], 1, 1]] * 4
    return []


def run_layer(lp, bt, blobs_of_layer, notes):
    """One layer through its compiled reference class (oracle/_ref)."""
    t = lp.type
    x = bt[0]
    if t == "Convolution":
        p = lp.convolution_param
        assert not (p.HasField("kernel_h") or p.HasField("stride_h") or p.HasField("pad_h")) and p.group == 1
        return eco_ref.convolution_layer(x, blobs_of_layer[0], blobs_of_layer[1] if p.bias_term else None,
                                         list(p.kernel_size), list(p.stride), list(p.pad), bool(p.force_nd_im2col))
    if t == "BN":
        p = lp.bn_param
        eps = float(p.eps)
        if x.ndim > 4:
            eps = max(eps, BN_MIN_EPS)
            notes["bn5d"] = notes.get("bn5d", 0) + 1
        return eco_ref.bn_inference(x, *blobs_of_layer, eps, bool(p.frozen))       # folds 5-D to [n,c,d*h,w]
    if t == "ReLU":
        return eco_ref.relu(x, float(lp.relu_param.negative_slope))
    if t == "Pooling":
        p = lp.pooling_param
        nsp = x.ndim - 2
        method = caffe_pb2.PoolingParameter.PoolMethod.Name(p.pool)
        k = list(x.shape[2:]) if p.global_pooling else G.spatial(p, "kernel_size", nsp, None)
        s, pd = G.spatial(p, "stride", nsp, 1), G.spatial(p, "pad", nsp, 0)
        if nsp == 2:
            return eco_ref.pooling(x, method, k, s, pd)
        assert nsp == 3 and method == "AVE" and k == list(x.shape[2:]) and pd == [0, 0, 0], (lp.name, k, x.shape)
        notes["pool3d_global"] = notes.get("pool3d_global", 0) + 1
        n, c, d, h, w = x.shape
        y = eco_ref.pooling(x.reshape(n, c, d * h, w), "AVE", [d * h, w], [1, 1], [0, 0])
        return y.reshape(n, c, 1, 1, 1)
    if t == "Concat":
        p = lp.concat_param
        return eco_ref.concat(bt, int(p.concat_dim) if p.HasField("concat_dim") else int(p.axis))
    if t == "Eltwise":
        p = lp.eltwise_param
        return eco_ref.eltwise(bt, caffe_pb2.EltwiseParameter.EltwiseOp.Name(p.operation),
                               [float(c) for c in p.coeff] or None)
    if t == "Reshape":
        p = lp.reshape_param
        return x.reshape(eco_ref.reshape_shape(x.shape, [int(d) for d in p.shape.dim], int(p.axis), int(p.num_axes)))
    if t == "Permute":
        return eco_ref.permute(x, [int(o) for o in lp.permute_param.order])
    if t == "Dropout":
        return x.copy()
    if t == "InnerProduct":
        p = lp.inner_product_param
        return eco_ref.inner_product(x, blobs_of_layer[0], blobs_of_layer[1] if p.bias_term else None, int(p.axis))
    raise SystemExit("layer type %s not on the ECO path" % t)


def forward(net, seed_params, seed_frames, keep_stats=True):
    """Net::ForwardFromTo (net.cpp:566-583) over the TEST-filtered layer list, each layer by compiled reference code."""
    shape = [int(d) for d in net.input_dim]
    blobs = {net.input[0]: ref_params.frames(shape[0], shape[2], shape[3], seed=seed_frames)}
    stats, plist, notes = {}, [], {}
    order = []
    t0 = time.time()
    for lp in G.filter_net(net, caffe_pb2.TEST):
        bt = [blobs[b] for b in lp.bottom]
        shapes = param_shapes(lp, list(bt[0].shape))
        pb = ref_params.layer_blobs(lp.name, lp.type, shapes, seed=seed_params) if shapes else []
        if shapes:
            plist.append(dict(name=lp.name, type=lp.type, shapes=shapes))
        assert len(lp.top) == 1
        y = run_layer(lp, bt, pb, notes)
        assert np.isfinite(y).all(), lp.name
        blobs[lp.top[0]] = y
        order.append(lp.top[0])
    if keep_stats:
        # fingerprint of every blob in its FINAL state (in-place ReLU / Dropout tops overwrite their bottoms)
        for name in dict.fromkeys(order):
            stats[name] = ref_params.blob_stats(blobs[name])
    return blobs, stats, plist, notes, time.time() - t0


CASES = [
    dict(key="eco_lite_n4_b1", file="models_ECO_Lite/kinetics/deploy.prototxt", num_segments=4, num_clips=1,
         baseline="BASELINE.json configs[0]"),
    dict(key="eco_full_n4_b1", file="models_ECO_Full/kinetics/deploy.prototxt", num_segments=4, num_clips=1,
         baseline="configs[3] graph at the configs[0] clip geometry"),
    dict(key="eco_lite_n8_b2", file="models_ECO_Lite/kinetics/deploy.prototxt", num_segments=8, num_clips=2,
         baseline="two clips, depth-2 global_pool window (clip independence through reference code)"),
    # the clip geometries of the headline configurations, one clip each (round 5): the 3-D trunk at 16 / 8 / 4 planes
    # (32 / 16 / 8), i.e. the depths at which the HIP path nests the minimal-filtering algorithm over depth as well
    dict(key="eco_lite_n16_b1", file="models_ECO_Lite/kinetics/deploy.prototxt", num_segments=16, num_clips=1,
         baseline="BASELINE.json configs[1] / configs[2] clip geometry (the authors' own num_segments), one clip"),
    dict(key="eco_lite_n32_b1", file="models_ECO_Lite/kinetics/deploy.prototxt", num_segments=32, num_clips=1,
         baseline="BASELINE.json configs[4] clip geometry (r2Dto3D 32x96x28x28, global_pool 8x7x7), one clip"),
]


def main():
    """No arguments: regenerate every net.  `--add`: keep the nets the committed JSON already holds (bit for bit) and
    compute only the missing ones."""
    assert eco_ref.has_conv_layer(), "oracle/_ref lacks the compiled ConvolutionLayer: make -C oracle"
    seed_params, seed_frames = 2024, 77
    have = {}
    if "--add" in sys.argv[1:] and os.path.exists(OUT):
        with open(OUT) as f:
            old = json.load(f)
        assert (old["seed_params"], old["seed_frames"], old["recipe"]) == (seed_params, seed_frames, ref_params.RECIPE)
        have = old["nets"]
    out = dict(generator="tests/golden/make_reference_logits.py", recipe=ref_params.RECIPE, seed_params=seed_params,
               seed_frames=seed_frames, blas="SciPy bundled OpenBLAS (cblas_sgemm), threads = library default",
               compiled=["util/im2col.cpp", "layers/base_conv_layer.cpp", "layers/conv_layer.cpp", "layers/bn_layer.cpp",
                         "layers/relu_layer.cpp", "layers/pooling_layer.cpp", "layers/concat_layer.cpp",
                         "layers/eltwise_layer.cpp", "layers/reshape_layer.cpp", "layers/permute_layer.cpp",
                         "layers/inner_product_layer.cpp"], nets={})
    for c in CASES:
        if c["key"] in have:
            out["nets"][c["key"]] = have[c["key"]]
            print("%-16s kept from the committed fixture" % c["key"])
            continue
        net, edits = load_net(c["file"], c["num_segments"], c["num_clips"])
        blobs, stats, plist, notes, dt = forward(net, seed_params, seed_frames)
        fc8 = blobs["fc8"]
        out["nets"][c["key"]] = dict(file=c["file"], baseline=c["baseline"], num_segments=c["num_segments"],
                                     num_clips=c["num_clips"], edits=edits, notes=notes,
                                     input_shape=[int(d) for d in net.input_dim], params=plist,
                                     fc8=[[float(v) for v in row] for row in fc8], blobs=stats)
        print("%-16s %3d blobs  fc8 %s  max|logit| %.4f  top-1 %s  %.1f s  notes %s"
              % (c["key"], len(stats), list(fc8.shape), float(np.abs(fc8).max()), fc8.argmax(axis=1).tolist(), dt, notes))
    with open(OUT, "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
