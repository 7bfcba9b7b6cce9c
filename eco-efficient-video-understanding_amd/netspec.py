"""Network description: prototxt -> filtered, split-inserted layer list + shapes.

Host-side restatement of the graph-building half of the reference runtime
(no arithmetic here):

* ``NetSpec.from_prototxt``  ~ ``Net<Dtype>::Init``            caffe_3d/src/caffe/net.cpp:39-316
* ``_filter_net``            ~ ``Net::FilterNet/StateMeetsRule`` net.cpp:319-400
* ``_insert_splits``         ~ ``InsertSplits``                 util/insert_splits.cpp:12-143
* per-layer ``Resolved*``    ~ each layer's ``LayerSetUp``/``Reshape``:
    Convolution  layers/base_conv_layer.cpp:13-261, conv_layer.cpp:12-25
    Pooling      layers/pooling_layer.cpp:17-163
    BN           layers/bn_layer.cpp:11-90
    Reshape      layers/reshape_layer.cpp:10-90
    Permute      layers/permute_layer.cpp:29-95
    Concat       layers/concat_layer.cpp:17-52
    Eltwise      layers/eltwise_layer.cpp:12-45
    InnerProduct layers/inner_product_layer.cpp:12-78
    Dropout/ReLU/Split/Softmax: shape-preserving.

Only the layer types that occur in the ECO prototxts are accepted; any other
``type`` raises ``NetSpecError`` (the reference LOG(FATAL)s on unknown types,
layer_factory.hpp:77-78).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

from . import prototxt
from .prototxt import Message

TRAIN = 0
TEST = 1
_PHASE_NAMES = {"TRAIN": TRAIN, "TEST": TEST}


class NetSpecError(ValueError):
    """Graph/shape error (the reference CHECK-fails / LOG(FATAL)s here)."""


def _prod(xs: Sequence[int]) -> int:
    p = 1
    for x in xs:
        p *= int(x)
    return p


@dataclass
class LayerSpec:
    name: str
    type: str
    bottoms: List[str]
    tops: List[str]
    param: Message
    # filled by NetSpec.reshape():
    bottom_shapes: List[Tuple[int, ...]] = field(default_factory=list)
    top_shapes: List[Tuple[int, ...]] = field(default_factory=list)
    geom: dict = field(default_factory=dict)  # resolved per-type geometry

    @property
    def inplace(self) -> bool:
        return bool(self.tops) and bool(self.bottoms) and self.tops[0] == self.bottoms[0]


# --------------------------------------------------------------------------
# phase filtering (net.cpp:319-400)
# --------------------------------------------------------------------------
def _phase_of(v) -> int:
    if isinstance(v, int):
        return v
    return _PHASE_NAMES[str(v)]


def _state_meets_rule(phase: int, level: int, stages: Sequence[str], rule: Message) -> bool:
    if rule.has("phase") and _phase_of(rule.get("phase")) != phase:
        return False
    if rule.has("min_level") and level < int(rule.get("min_level")):
        return False
    if rule.has("max_level") and level > int(rule.get("max_level")):
        return False
    for s in rule.getall("stage"):
        if s not in stages:
            return False
    for s in rule.getall("not_stage"):
        if s in stages:
            return False
    return True


def _filter_net(layers: List[Message], phase: int, level: int, stages: Sequence[str]) -> List[Message]:
    out = []
    for lp in layers:
        inc, exc = lp.getall("include"), lp.getall("exclude")
        if inc and exc:
            raise NetSpecError(f"layer {lp.get('name')!r}: specify either include rules or exclude rules; not both")
        included = not inc
        if included:
            for r in exc:
                if _state_meets_rule(phase, level, stages, r):
                    included = False
                    break
        else:
            for r in inc:
                if _state_meets_rule(phase, level, stages, r):
                    included = True
                    break
        if included:
            out.append(lp)
    return out


# --------------------------------------------------------------------------
# split insertion (util/insert_splits.cpp)
# --------------------------------------------------------------------------
def split_layer_name(layer_name: str, blob_name: str, blob_idx: int) -> str:
    return f"{blob_name}_{layer_name}_{blob_idx}_split"


def split_blob_name(layer_name: str, blob_name: str, blob_idx: int, split_idx: int) -> str:
    return f"{blob_name}_{layer_name}_{blob_idx}_split_{split_idx}"


def _insert_splits(inputs: List[str], layers: List[LayerSpec]) -> List[LayerSpec]:
    last_top: Dict[str, Tuple[int, int]] = {}
    src_of: Dict[Tuple[int, int], Tuple[int, int]] = {}
    count: Dict[Tuple[int, int], int] = {}
    lname = {-1: "input"}
    for i, b in enumerate(inputs):
        last_top[b] = (-1, i)
    for i, L in enumerate(layers):
        lname[i] = L.name
        for j, b in enumerate(L.bottoms):
            if b not in last_top:
                raise NetSpecError(f"Unknown blob input {b} to layer {L.name}")
            src_of[(i, j)] = last_top[b]
            count[last_top[b]] = count.get(last_top[b], 0) + 1
        for j, b in enumerate(L.tops):
            last_top[b] = (i, j)
        # a top used as a loss counts as one more consumer (insert_splits.cpp:47-57)
        lw = L.param.getall("loss_weight")
        for j in range(min(len(lw), len(L.tops))):
            if lw[j]:
                count[(i, j)] = count.get((i, j), 0) + 1

    def mk_split(layer_name: str, blob: str, idx: int, n: int) -> LayerSpec:
        p = Message()
        p.add("name", split_layer_name(layer_name, blob, idx))
        p.add("type", "Split")
        return LayerSpec(
            name=split_layer_name(layer_name, blob, idx),
            type="Split",
            bottoms=[blob],
            tops=[split_blob_name(layer_name, blob, idx, k) for k in range(n)],
            param=p,
        )

    out: List[LayerSpec] = []
    next_split: Dict[Tuple[int, int], int] = {}
    for i, b in enumerate(inputs):
        if count.get((-1, i), 0) > 1:
            out.append(mk_split("input", b, i, count[(-1, i)]))
    for i, L in enumerate(layers):
        bottoms = list(L.bottoms)
        for j, b in enumerate(bottoms):
            src = src_of[(i, j)]
            if count.get(src, 0) > 1:
                k = next_split.get(src, 0)
                next_split[src] = k + 1
                bottoms[j] = split_blob_name(lname[src[0]], b, src[1], k)
        out.append(LayerSpec(L.name, L.type, bottoms, list(L.tops), L.param))
        for j, b in enumerate(L.tops):
            if count.get((i, j), 0) > 1:
                out.append(mk_split(L.name, b, j, count[(i, j)]))
    return out


# --------------------------------------------------------------------------
# per-type geometry / shape rules
# --------------------------------------------------------------------------
def _spatial_param(p: Message, base: str, nsp: int, default: Optional[int], what: str,
                   hw_names: Tuple[str, str]) -> List[int]:
    """Resolve repeated ``kernel_size``/``stride``/``pad`` plus the 2-D ``*_h/_w`` forms
    (base_conv_layer.cpp:27-105, pooling_layer.cpp:37-100)."""
    h_name, w_name = hw_names
    if p.has(h_name) or p.has(w_name):
        if nsp != 2:
            raise NetSpecError(f"{what}: {h_name} & {w_name} can only be used for 2D")
        if p.getall(base):
            raise NetSpecError(f"{what}: either {base} or {h_name}/{w_name} should be specified; not both")
        return [int(p.get(h_name, 0 if default is None else default)),
                int(p.get(w_name, 0 if default is None else default))]
    vals = [int(v) for v in p.getall(base)]
    if not vals:
        if default is None:
            raise NetSpecError(f"{what}: {base} must be specified")
        return [default] * nsp
    if len(vals) == 1:
        return vals * nsp
    if len(vals) != nsp:
        raise NetSpecError(
            f"{what}: {base} must be specified once, or once per spatial dimension "
            f"({base} specified {len(vals)} times; {nsp} spatial dims)")
    return vals


def _canon_axis(axis: int, naxes: int, what: str) -> int:
    if axis < -naxes or axis >= naxes:
        raise NetSpecError(f"{what}: axis {axis} out of range for {naxes}-D blob")
    return axis + naxes if axis < 0 else axis


def _setup_convolution(L: LayerSpec) -> None:
    p = L.param.msg("convolution_param")
    bshape = L.bottom_shapes[0]
    axis = _canon_axis(int(p.get("axis", 1)), len(bshape), L.name)
    nsp = len(bshape) - axis - 1
    if nsp < 1 or nsp > 3:
        raise NetSpecError(f"{L.name}: {nsp} spatial axes unsupported (1..3)")
    if axis != 1:
        raise NetSpecError(f"{L.name}: only channel axis 1 is supported")
    kernel = _spatial_param(p, "kernel_size", nsp, None, L.name, ("kernel_h", "kernel_w"))
    stride = _spatial_param(p, "stride", nsp, 1, L.name, ("stride_h", "stride_w"))
    pad = _spatial_param(p, "pad", nsp, 0, L.name, ("pad_h", "pad_w"))
    if any(k <= 0 for k in kernel):
        raise NetSpecError(f"{L.name}: filter dimensions must be nonzero")
    if any(s <= 0 for s in stride):
        raise NetSpecError(f"{L.name}: stride dimensions must be nonzero")
    if int(p.get("group", 1)) != 1:
        raise NetSpecError(f"{L.name}: group != 1 not on the ECO path")
    if int(p.get("dilation", 1)) != 1:
        raise NetSpecError(f"{L.name}: dilation != 1 not on the ECO path")
    if not p.has("num_output") or int(p.get("num_output")) <= 0:
        raise NetSpecError(f"{L.name}: num_output must be > 0")
    cout = int(p.get("num_output"))
    cin = bshape[axis]
    out_sp = []
    for i in range(nsp):
        o = (bshape[axis + 1 + i] + 2 * pad[i] - kernel[i]) // stride[i] + 1  # conv_layer.cpp:19-22
        if o <= 0:
            raise NetSpecError(f"{L.name}: non-positive output dim")
        out_sp.append(o)
    L.geom = dict(kernel=kernel, stride=stride, pad=pad, cin=cin, cout=cout,
                  bias_term=bool(p.get("bias_term", True)), nsp=nsp)
    L.top_shapes = [tuple(bshape[:axis]) + (cout,) + tuple(out_sp)]


def conv_param_shapes(L: LayerSpec) -> List[Tuple[int, ...]]:
    g = L.geom
    shapes = [(g["cout"], g["cin"]) + tuple(g["kernel"])]  # base_conv_layer.cpp:138-141
    if g["bias_term"]:
        shapes.append((g["cout"],))
    return shapes


def pooled_dim(in_dim: int, k: int, s: int, p: int) -> int:
    """pooling_layer.cpp:131-147 (ceil rule + last-window clip when padded)."""
    o = int(math.ceil(float(in_dim + 2 * p - k) / s)) + 1
    if p:
        if (o - 1) * s >= in_dim + p:
            o -= 1
        if not ((o - 1) * s < in_dim + p):
            raise NetSpecError("pooling: last window starts outside the padded input")
    return o


def _setup_pooling(L: LayerSpec) -> None:
    p = L.param.msg("pooling_param")
    bshape = L.bottom_shapes[0]
    nsp = len(bshape) - 2
    if nsp < 1 or nsp > 3:
        raise NetSpecError(f"{L.name}: {nsp} spatial axes unsupported (1..3)")
    method = str(p.get("pool", "MAX"))
    if method not in ("MAX", "AVE"):
        raise NetSpecError(f"{L.name}: pooling method {method} not on the ECO path")
    global_pooling = bool(p.get("global_pooling", False))
    if global_pooling:
        if p.getall("kernel_size") or p.has("kernel_h") or p.has("kernel_w"):
            raise NetSpecError(f"{L.name}: with global_pooling, filter size cannot be specified")
        kernel = list(bshape[2:])
    else:
        kernel = _spatial_param(p, "kernel_size", nsp, None, L.name, ("kernel_h", "kernel_w"))
    stride = _spatial_param(p, "stride", nsp, 1, L.name, ("stride_h", "stride_w"))
    pad = _spatial_param(p, "pad", nsp, 0, L.name, ("pad_h", "pad_w"))
    for i in range(nsp):
        if global_pooling and not (pad[i] == 0 and stride[i] == 1):
            raise NetSpecError(f"{L.name}: with global_pooling only pad = 0 and stride = 1")
        if not pad[i] < kernel[i]:
            raise NetSpecError(f"{L.name}: pad must be smaller than kernel")
    out_sp = [pooled_dim(bshape[2 + i], kernel[i], stride[i], pad[i]) for i in range(nsp)]
    L.geom = dict(method=method, kernel=kernel, stride=stride, pad=pad, nsp=nsp)
    L.top_shapes = [tuple(bshape[:2]) + tuple(out_sp)]


def _setup_bn(L: LayerSpec) -> None:
    p = L.param.msg("bn_param")
    bshape = L.bottom_shapes[0]
    if len(bshape) < 2:
        raise NetSpecError(f"{L.name}: BN needs a channel axis")
    L.geom = dict(eps=float(p.get("eps", 1e-5)), momentum=float(p.get("momentum", 0.9)),
                  frozen=bool(p.get("frozen", False)), channels=bshape[1])
    L.top_shapes = [tuple(bshape)]


def _setup_reshape(L: LayerSpec) -> None:
    p = L.param.msg("reshape_param")
    bshape = list(L.bottom_shapes[0])
    dims = [int(d) for d in p.msg("shape").getall("dim")]
    in_axis = int(p.get("axis", 0))
    start = in_axis if in_axis >= 0 else len(bshape) + in_axis + 1
    if start < 0 or start > len(bshape):
        raise NetSpecError(f"{L.name}: axis {in_axis} out of range")
    num_axes = int(p.get("num_axes", -1))
    if num_axes < -1:
        raise NetSpecError(f"{L.name}: num_axes must be >= 0, or -1 for all")
    end = len(bshape) if num_axes == -1 else start + num_axes
    if end > len(bshape):
        raise NetSpecError(f"{L.name}: end_axis = axis + num_axes is out of range")
    top = bshape[:start] + dims + bshape[end:]
    inferred = -1
    const_count = 1
    copy_axes = []
    for i, d in enumerate(dims):
        if d == 0:
            copy_axes.append(i)
        elif d == -1:
            if inferred != -1:
                raise NetSpecError(f"{L.name}: new shape contains multiple -1 dims")
            inferred = i
        else:
            const_count *= d
    for i in copy_axes:
        if not len(bshape) > start + i:
            raise NetSpecError(f"{L.name}: new shape contains a 0, but there was no corresponding bottom axis to copy")
        top[start + i] = bshape[start + i]
    total = _prod(bshape)
    if inferred >= 0:
        explicit = const_count * _prod(bshape[:start]) * _prod(bshape[end:])
        for i in copy_axes:
            explicit *= top[start + i]
        if explicit == 0 or total % explicit != 0:
            raise NetSpecError(
                f"{L.name}: bottom count ({total}) must be divisible by the product of the specified dimensions ({explicit})")
        top[start + inferred] = total // explicit
    if _prod(top) != total:
        raise NetSpecError(f"{L.name}: output count must match input count")
    L.geom = {}
    L.top_shapes = [tuple(top)]


def _setup_permute(L: LayerSpec) -> None:
    p = L.param.msg("permute_param")
    bshape = L.bottom_shapes[0]
    order = [int(o) for o in p.getall("order")]
    for o in order:
        if o >= len(bshape):
            raise NetSpecError(f"{L.name}: order should be less than the input dimension")
    if len(set(order)) != len(order):
        raise NetSpecError(f"{L.name}: there are duplicate orders")
    for i in range(len(bshape)):  # permute_layer.cpp:47-52: unspecified axes keep their order
        if i not in order:
            order.append(i)
    L.geom = dict(order=order, need_permute=any(o != i for i, o in enumerate(order)))
    L.top_shapes = [tuple(bshape[o] for o in order)]


def _setup_concat(L: LayerSpec) -> None:
    p = L.param.msg("concat_param")
    b0 = L.bottom_shapes[0]
    if p.has("concat_dim"):
        axis = int(p.get("concat_dim"))
        if axis < 0:
            raise NetSpecError(f"{L.name}: concat_dim must be >= 0")
    else:
        axis = _canon_axis(int(p.get("axis", 1)), len(b0), L.name)
    tot = 0
    for s in L.bottom_shapes:
        if len(s) != len(b0) or any(s[i] != b0[i] for i in range(len(b0)) if i != axis):
            raise NetSpecError(f"{L.name}: all inputs must have the same shape, except at concat_axis")
        tot += s[axis]
    L.geom = dict(axis=axis)
    L.top_shapes = [tuple(b0[:axis]) + (tot,) + tuple(b0[axis + 1:])]


def _setup_eltwise(L: LayerSpec) -> None:
    p = L.param.msg("eltwise_param")
    op = str(p.get("operation", "SUM"))
    coeff = [float(c) for c in p.getall("coeff")]
    if coeff and len(coeff) != len(L.bottoms):
        raise NetSpecError(f"{L.name}: Eltwise takes one coefficient per bottom blob")
    if op != "SUM":
        raise NetSpecError(f"{L.name}: Eltwise {op} not on the ECO path (SUM only)")
    for s in L.bottom_shapes[1:]:
        if tuple(s) != tuple(L.bottom_shapes[0]):
            raise NetSpecError(f"{L.name}: Eltwise bottoms must have the same shape")
    L.geom = dict(op=op, coeff=coeff or [1.0] * len(L.bottoms))
    L.top_shapes = [tuple(L.bottom_shapes[0])]


def _setup_inner_product(L: LayerSpec) -> None:
    p = L.param.msg("inner_product_param")
    bshape = L.bottom_shapes[0]
    axis = _canon_axis(int(p.get("axis", 1)), len(bshape), L.name)
    n_out = int(p.get("num_output"))
    K = _prod(bshape[axis:])
    L.geom = dict(axis=axis, num_output=n_out, K=K, M=_prod(bshape[:axis]),
                  bias_term=bool(p.get("bias_term", True)))
    L.top_shapes = [tuple(bshape[:axis]) + (n_out,)]


def _setup_same(L: LayerSpec) -> None:
    L.geom = {}
    L.top_shapes = [tuple(L.bottom_shapes[0]) for _ in L.tops]


def _setup_dropout(L: LayerSpec) -> None:
    p = L.param.msg("dropout_param")
    L.geom = dict(ratio=float(p.get("dropout_ratio", 0.5)))
    L.top_shapes = [tuple(L.bottom_shapes[0])]


def _setup_relu(L: LayerSpec) -> None:
    p = L.param.msg("relu_param")
    L.geom = dict(negative_slope=float(p.get("negative_slope", 0.0)))
    L.top_shapes = [tuple(L.bottom_shapes[0])]


def _setup_softmax(L: LayerSpec) -> None:
    p = L.param.msg("softmax_param")
    L.geom = dict(axis=_canon_axis(int(p.get("axis", 1)), len(L.bottom_shapes[0]), L.name))
    L.top_shapes = [tuple(L.bottom_shapes[0])]


def _label_axis_geom(L: LayerSpec, axis: int) -> None:
    bs, ls = L.bottom_shapes[0], L.bottom_shapes[1]
    ax = _canon_axis(axis, len(bs), L.name)
    outer, inner = _prod(bs[:ax]), _prod(bs[ax + 1:])
    if outer * inner != _prod(ls):  # accuracy_layer.cpp:34-38, softmax_loss_layer.cpp:40-44
        raise NetSpecError(f"{L.name}: Number of labels must match number of predictions "
                           f"({outer * inner} predictions of shape {tuple(bs)}, {_prod(ls)} labels)")
    L.geom.update(axis=ax, outer=outer, classes=int(bs[ax]), inner=inner)


def _setup_accuracy(L: LayerSpec) -> None:
    """AccuracyLayer (layers/accuracy_layer.cpp:14-44): bottoms scores + labels, scalar (0-axis) top."""
    if len(L.bottoms) != 2 or len(L.tops) != 1:
        raise NetSpecError(f"{L.name}: Accuracy takes 2 bottoms and 1 top")
    p = L.param.msg("accuracy_param")
    ign = p.get("ignore_label", None)
    L.geom = dict(top_k=int(p.get("top_k", 1)), ignore_label=None if ign is None else int(ign))
    _label_axis_geom(L, int(p.get("axis", 1)))
    if L.geom["top_k"] > L.geom["classes"]:
        raise NetSpecError(f"{L.name}: top_k must be less than or equal to the number of classes.")
    L.top_shapes = [()]


def _setup_softmax_loss(L: LayerSpec) -> None:
    """SoftmaxWithLossLayer (layers/softmax_loss_layer.cpp:12-50, loss_layer.cpp): scalar loss top and,
    optionally, the softmax as a second top.  (Forward only: the loss weight matters to training alone.)"""
    if len(L.bottoms) != 2 or len(L.tops) not in (1, 2):
        raise NetSpecError(f"{L.name}: SoftmaxWithLoss takes 2 bottoms and 1 or 2 tops")
    lp = L.param.msg("loss_param")
    ign = lp.get("ignore_label", None)
    norm = lp.get("normalize", True)
    L.geom = dict(normalize=bool(norm) if not isinstance(norm, str) else norm.lower() == "true",
                  ignore_label=None if ign is None else int(ign))
    _label_axis_geom(L, int(L.param.msg("softmax_param").get("axis", 1)))
    L.top_shapes = [()] + ([tuple(L.bottom_shapes[0])] if len(L.tops) == 2 else [])


_SETUP = {
    "Convolution": _setup_convolution,
    "Pooling": _setup_pooling,
    "BN": _setup_bn,
    "ReLU": _setup_relu,
    "Reshape": _setup_reshape,
    "Permute": _setup_permute,
    "Concat": _setup_concat,
    "Eltwise": _setup_eltwise,
    "InnerProduct": _setup_inner_product,
    "Dropout": _setup_dropout,
    "Split": _setup_same,
    "Softmax": _setup_softmax,
    "Accuracy": _setup_accuracy,
    "SoftmaxWithLoss": _setup_softmax_loss,
}

SUPPORTED_LAYER_TYPES = tuple(_SETUP)


def param_shapes(L: LayerSpec) -> List[Tuple[int, ...]]:
    """Shapes of the layer's learnable blobs, in the reference's blob order."""
    if L.type == "Convolution":
        return conv_param_shapes(L)
    if L.type == "BN":  # bn_layer.cpp:20-41: scale, shift, running mean, running variance
        c = L.geom["channels"]
        return [(1, c)] * 4
    if L.type == "InnerProduct":  # inner_product_layer.cpp:27-46
        g = L.geom
        return [(g["num_output"], g["K"])] + ([(g["num_output"],)] if g["bias_term"] else [])
    return []


# --------------------------------------------------------------------------
class NetSpec:
    """Phase-filtered, split-inserted layer list with resolved shapes."""

    def __init__(self, name: str, inputs: List[str], input_shapes: Dict[str, Tuple[int, ...]],
                 layers: List[LayerSpec], phase: int) -> None:
        self.name = name
        self.inputs = inputs
        self.input_shapes = dict(input_shapes)
        self.layers = layers
        self.phase = phase
        self.blob_shapes: Dict[str, Tuple[int, ...]] = {}
        self.blob_names: List[str] = []
        self.reshape()

    # -- construction -------------------------------------------------------
    @classmethod
    def from_prototxt(cls, path_or_text: str, phase: int = TEST, level: int = 0,
                      stages: Sequence[str] = ()) -> "NetSpec":
        if "\n" in path_or_text or "{" in path_or_text:
            root = prototxt.parse(path_or_text)
        else:
            root = prototxt.parse_file(path_or_text)
        return cls.from_message(root, phase, level, stages)

    @classmethod
    def from_message(cls, root: Message, phase: int = TEST, level: int = 0,
                     stages: Sequence[str] = ()) -> "NetSpec":
        if root.getall("layers"):
            raise NetSpecError("V1 'layers' prototxt format is not supported (all ECO files are V2 'layer')")
        inputs = [str(s) for s in root.getall("input")]
        in_shapes: Dict[str, Tuple[int, ...]] = {}
        if root.getall("input_shape"):  # caffe.proto:70-71
            shp = root.getall("input_shape")
            if len(shp) != len(inputs):
                raise NetSpecError("exactly one input_shape must be specified per input")
            for n, s in zip(inputs, shp):
                in_shapes[n] = tuple(int(d) for d in s.getall("dim"))
        else:  # deprecated 4-D input_dim (caffe.proto:73, net.cpp:57-75)
            dims = [int(d) for d in root.getall("input_dim")]
            if len(dims) != 4 * len(inputs):
                raise NetSpecError("Must specify either input_shape OR 4 input_dims per input")
            for i, n in enumerate(inputs):
                in_shapes[n] = tuple(dims[4 * i:4 * i + 4])
        raw = _filter_net(root.getall("layer"), phase, level, tuple(stages))
        layers = []
        for lp in raw:
            t = str(lp.get("type"))
            if t == "VideoData":
                # The data source of the train/val prototxts.  The engine does not read frame folders: the
                # layer's tops become net inputs with the shapes VideoDataLayer::DataLayerSetUp gives them
                # (video_data_layer.cpp:107-119: data [batch, 3*new_length*num_segments, crop, crop] for RGB,
                # label [batch,1,1,1]); eco_amd.video.VideoInput fills `data` on the GPU.
                tops = [str(b) for b in lp.getall("top")]
                vp, tp = lp.msg("video_data_param"), lp.msg("transform_param")
                batch, segs, nl = int(vp.get("batch_size", 1)), int(vp.get("num_segments", 1)), int(vp.get("new_length", 1))
                crop = int(tp.get("crop_size", 0))
                if crop <= 0:
                    crop_h, crop_w = int(vp.get("new_height", 0)), int(vp.get("new_width", 0))
                    if crop_h <= 0 or crop_w <= 0:
                        raise NetSpecError(f"{lp.get('name')}: VideoData needs transform_param.crop_size or "
                                           "video_data_param.new_height/new_width to fix the input shape")
                else:
                    crop_h = crop_w = crop
                ch = (3 if str(vp.get("modality", "RGB")) == "RGB" else 2) * nl * segs
                for k, tname in enumerate(tops[:2]):
                    inputs.append(tname)
                    in_shapes[tname] = (batch, ch, crop_h, crop_w) if k == 0 else (batch, 1, 1, 1)
                continue
            layers.append(LayerSpec(name=str(lp.get("name")), type=t,
                                    bottoms=[str(b) for b in lp.getall("bottom")],
                                    tops=[str(b) for b in lp.getall("top")], param=lp))
        for L in layers:
            if L.type not in _SETUP:
                raise NetSpecError(f"Unknown layer type: {L.type} (layer {L.name!r}); "
                                   f"supported: {', '.join(SUPPORTED_LAYER_TYPES)}")
        layers = _insert_splits(inputs, layers)
        return cls(str(root.get("name", "")), inputs, in_shapes, layers, phase)

    # -- shapes -------------------------------------------------------------
    def reshape(self, input_shapes: Optional[Dict[str, Sequence[int]]] = None) -> None:
        """(Re)propagate shapes from the inputs (``Net::Reshape``, net.cpp:843-849)."""
        if input_shapes:
            for k, v in input_shapes.items():
                if k not in self.input_shapes:
                    raise NetSpecError(f"{k!r} is not a net input")
                self.input_shapes[k] = tuple(int(d) for d in v)
        shapes: Dict[str, Tuple[int, ...]] = {}
        names: List[str] = []
        for n in self.inputs:
            shapes[n] = tuple(self.input_shapes[n])
            names.append(n)
        for L in self.layers:
            for b in L.bottoms:
                if b not in shapes:
                    raise NetSpecError(f"Unknown blob input {b} (at layer {L.name})")
            L.bottom_shapes = [shapes[b] for b in L.bottoms]
            _SETUP[L.type](L)
            if len(L.top_shapes) != len(L.tops):
                raise NetSpecError(f"{L.name}: expected {len(L.top_shapes)} top(s), got {len(L.tops)}")
            for t, s in zip(L.tops, L.top_shapes):
                if t in L.bottoms:  # in-place (net.cpp:420-426)
                    if tuple(shapes[t]) != tuple(s):
                        raise NetSpecError(f"{L.name}: in-place top {t} changes shape")
                elif t in shapes:
                    raise NetSpecError(f"Top blob '{t}' produced by multiple sources.")
                else:
                    names.append(t)
                shapes[t] = tuple(s)
        self.blob_shapes = shapes
        self.blob_names = names

    # -- queries ------------------------------------------------------------
    @property
    def layer_names(self) -> List[str]:
        return [L.name for L in self.layers]

    @property
    def outputs(self) -> List[str]:
        """Blobs never consumed as a bottom, in ``std::set`` (sorted) order (net.cpp:262-269)."""
        avail = list(self.inputs)
        for L in self.layers:
            for b in L.bottoms:
                if b in avail:
                    avail.remove(b)
            for t in L.tops:
                if t not in avail:
                    avail.append(t)
        return sorted(avail)

    def layer(self, name: str) -> LayerSpec:
        for L in self.layers:
            if L.name == name:
                return L
        raise KeyError(name)

    def consumers(self, blob: str, after: int = -1) -> List[int]:
        return [i for i, L in enumerate(self.layers) if i > after and blob in L.bottoms]

    def fused_model_bytes(self, elem: int = 4) -> int:
        """Algorithmic bytes of one forward in the FUSED model of SURVEY.md section 8(d): each Convolution / InnerProduct
        reads its input and weights and writes its output once (bias / BN / ReLU / Dropout folded; Concat / Reshape / Split /
        Permute free), each Pooling reads its bottom and writes its top, each Eltwise costs one extra operand read, and a
        conv / Eltwise value that is needed raw AND through its BN + ReLU is written twice (res3a / res4a / res5a)."""
        consumers: Dict[str, List[LayerSpec]] = {}
        alias: Dict[str, str] = {}
        for L in self.layers:
            if L.type == "Split":
                for t in L.tops:
                    alias[t] = alias.get(L.bottoms[0], L.bottoms[0])
        for L in self.layers:
            if L.type != "Split":
                for b in L.bottoms:
                    consumers.setdefault(alias.get(b, b), []).append(L)
        tot = 0
        for L in self.layers:
            if L.type == "Convolution":
                g = L.geom
                tot += elem * (_prod(L.bottom_shapes[0]) + g["cout"] * g["cin"] * _prod(g["kernel"]) + _prod(L.top_shapes[0]))
            elif L.type == "InnerProduct":
                g = L.geom
                tot += elem * (g["M"] * g["K"] + g["num_output"] * g["K"] + g["M"] * g["num_output"])
            elif L.type == "Pooling":
                tot += elem * (_prod(L.bottom_shapes[0]) + _prod(L.top_shapes[0]))
            elif L.type == "Eltwise":
                tot += elem * _prod(L.top_shapes[0]) * (len(L.bottoms) - 1)
            if L.type in ("Convolution", "Eltwise"):
                cs = [c for c in consumers.get(alias.get(L.tops[0], L.tops[0]), []) if not (c.inplace and c.type == "Dropout")]
                if any(c.type == "BN" for c in cs) and len(cs) > 1:
                    tot += elem * _prod(L.top_shapes[0])
        return tot

    def conv_fc_flops(self) -> int:
        """2*MAC over Convolution + InnerProduct layers."""
        tot = 0
        for L in self.layers:
            if L.type == "Convolution":
                g = L.geom
                tot += 2 * _prod(L.top_shapes[0]) * g["cin"] * _prod(g["kernel"])
            elif L.type == "InnerProduct":
                g = L.geom
                tot += 2 * g["M"] * g["num_output"] * g["K"]
        return tot
