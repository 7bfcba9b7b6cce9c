"""Generators for the ECO deploy operator graphs (prototxt text).

The reference ships its nets as hand-written prototxt files
(models_ECO_Lite/*/deploy.prototxt, 109 layers; models_ECO_Full/*/deploy.prototxt,
281 layers) that differ per dataset only in net name, dropout ratio and fc
name/``num_output``, and per ``num_segments`` only in the ``r2Dto3D`` dim, the
``global_pool`` kernel depth (N/4) and, for ECO-Full, ``reshape_fc_st2`` /
``segment_consensus_st2`` (README.md:85-95).  Those files do not exist on the
GPU box, so the same graphs are *generated* here from the BN-Inception /
3D-ResNet-18 structure; layer and blob names are the reference's (weights are
matched by layer name, net.cpp:852-883).  tests/test_prototxt_netspec.py checks the
generated graph field-by-field against the reference files when
/root/reference is present, and against the committed extract
tests/golden/reference_graphs.json everywhere else.

Any user prototxt (including the reference's own files) can be passed to
``Net`` directly; these generators only remove the dependency on the files.
"""
from __future__ import annotations

from typing import List

_XAVIER = 'weight_filler { type: "xavier" } bias_filler { type: "constant" value: 0 }'
_BN_FILL = "slope_filler { type: \"constant\" value: 1 } bias_filler { type: \"constant\" value: 0 }"


class _Builder:
    def __init__(self, width_div: int = 1) -> None:
        self.out: List[str] = []
        self.width_div = int(width_div)

    def ch(self, c: int) -> int:
        """Channel count after the (test-only) width divisor."""
        return c if self.width_div == 1 else max(4, c // self.width_div)

    def emit(self, s: str) -> None:
        self.out.append(s)

    # -- primitive layers ----------------------------------------------------
    def conv(self, name: str, bottom: str, top: str, cout: int, k, s=1, p=0, nd: int = 2) -> None:
        def rep(field: str, v, default) -> str:
            if nd == 3:
                return f"{field}: [{v}, {v}, {v}] "
            return "" if v == default else f"{field}: {v} "
        self.emit(
            f'layer {{ name: "{name}" type: "Convolution" bottom: "{bottom}" top: "{top}" '
            f"param {{ lr_mult: 1 decay_mult: 1 }} param {{ lr_mult: 2 decay_mult: 0 }} "
            f"convolution_param {{ num_output: {self.ch(cout)} {rep('pad', p, 0)}{rep('kernel_size', k, None)}"
            f"{rep('stride', s, 1)}{_XAVIER} }} }}")

    def bn(self, name: str, bottom: str, top: str, frozen_field: bool = False) -> None:
        fr = "frozen: false " if frozen_field else ""
        self.emit(
            f'layer {{ name: "{name}" type: "BN" bottom: "{bottom}" top: "{top}" '
            f"param {{ lr_mult: 1 decay_mult: 0 }} param {{ lr_mult: 1 decay_mult: 0 }} "
            f"bn_param {{ {fr}{_BN_FILL} }} }}")

    def relu(self, name: str, blob: str) -> None:
        self.emit(f'layer {{ name: "{name}" type: "ReLU" bottom: "{blob}" top: "{blob}" }}')

    def pool(self, name: str, bottom: str, top: str, method: str, body: str) -> None:
        self.emit(f'layer {{ name: "{name}" type: "Pooling" bottom: "{bottom}" top: "{top}" '
                  f"pooling_param {{ pool: {method} {body} }} }}")

    def concat(self, name: str, bottoms: List[str], top: str, axis=None) -> None:
        b = " ".join(f'bottom: "{x}"' for x in bottoms)
        ax = f" concat_param {{ axis: {axis} }}" if axis is not None else ""
        self.emit(f'layer {{ name: "{name}" type: "Concat" {b} top: "{top}"{ax} }}')

    def eltwise(self, name: str, a: str, b: str, top: str) -> None:
        self.emit(f'layer {{ name: "{name}" type: "Eltwise" bottom: "{a}" bottom: "{b}" top: "{top}" }}')

    def reshape(self, name: str, bottom: str, top: str, dims) -> None:
        d = " ".join(f"dim: {x}" for x in dims)
        self.emit(f'layer {{ name: "{name}" type: "Reshape" bottom: "{bottom}" top: "{top}" '
                  f"reshape_param {{ shape {{ {d} }} }} }}")

    def dropout(self, name: str, blob: str, ratio: float) -> None:
        self.emit(f'layer {{ name: "{name}" type: "Dropout" bottom: "{blob}" top: "{blob}" '
                  f"dropout_param {{ dropout_ratio: {ratio} }} }}")

    # -- composites ----------------------------------------------------------
    def conv_bn_relu_2d(self, prefix: str, stem: str, bottom: str, cout: int, k, s=1, p=0,
                        relu_stem: str = "") -> str:
        """BN-Inception naming: conv ``{prefix}_{stem}``, BN ``..._bn``,
        ReLU ``{prefix}_relu_{stem}_inp`` (in place on the BN top)."""
        name = f"{prefix}_{stem}"
        self.conv(name, bottom, name, cout, k, s, p)
        self.bn(name + "_bn", name, name + "_bn")
        self.relu(f"{prefix}_relu_{relu_stem or stem}_inp", name + "_bn")
        return name + "_bn"

    def inception(self, prefix: str, bottom: str, c1, c3r, c3, cd3r, cd3, pool: str, cproj) -> str:
        """Regular (stride-1) inception block: 1x1 | 3x3 | double 3x3 | pool+proj."""
        tops = []
        tops.append(self.conv_bn_relu_2d(prefix, "1x1", bottom, c1, 1))
        r = self.conv_bn_relu_2d(prefix, "3x3_reduce", bottom, c3r, 1)
        tops.append(self.conv_bn_relu_2d(prefix, "3x3", r, c3, 3, 1, 1))
        r = self.conv_bn_relu_2d(prefix, "double_3x3_reduce", bottom, cd3r, 1)
        r = self.conv_bn_relu_2d(prefix, "double_3x3_1", r, cd3, 3, 1, 1)
        tops.append(self.conv_bn_relu_2d(prefix, "double_3x3_2", r, cd3, 3, 1, 1))
        self.pool(f"{prefix}_pool", bottom, f"{prefix}_pool", pool, "kernel_size: 3 stride: 1 pad: 1")
        tops.append(self.conv_bn_relu_2d(prefix, "pool_proj", f"{prefix}_pool", cproj, 1))
        self.concat(f"{prefix}_output", tops, f"{prefix}_output")
        return f"{prefix}_output"

    def res_conv(self, name: str, bottom: str, top: str, cout: int, stride: int) -> None:
        self.conv(name, bottom, top, cout, 3, stride, 1, nd=3)

    def bn_relu_3d(self, stem: str, bottom: str) -> str:
        self.bn(f"{stem}_bn", bottom, f"{stem}_bn", frozen_field=True)
        self.relu(f"{stem}_relu", f"{stem}_bn")
        return f"{stem}_bn"


def _head_2d(b: _Builder) -> str:
    """conv1 ... inception_3b_output (shared by Lite and Full)."""
    t = b.conv_bn_relu_2d("conv1", "7x7_s2", "data", 64, 7, 2, 3, relu_stem="7x7")
    b.pool("pool1_3x3_s2", t, "pool1_3x3_s2", "MAX", "kernel_size: 3 stride: 2")
    t = b.conv_bn_relu_2d("conv2", "3x3_reduce", "pool1_3x3_s2", 64, 1)
    t = b.conv_bn_relu_2d("conv2", "3x3", t, 192, 3, 1, 1)
    b.pool("pool2_3x3_s2", t, "pool2_3x3_s2", "MAX", "kernel_size: 3 stride: 2")
    t = b.inception("inception_3a", "pool2_3x3_s2", 64, 64, 64, 64, 96, "AVE", 32)
    t = b.inception("inception_3b", t, 64, 64, 96, 64, 96, "AVE", 64)
    return t


def _trunk_3d(b: _Builder, bottom_2d: str, num_segments: int, sp: int = 28) -> str:
    """r2Dto3D + Permute + 3D-ResNet-18 res3a..res5b (pre-activation residuals)."""
    b.reshape("r2Dto3D", bottom_2d, "res2b_bn_pre", [-1, num_segments, b.ch(96), sp, sp])
    b.emit('layer { name: "Transpose1" type: "Permute" bottom: "res2b_bn_pre" top: "res2b_bn" '
           "permute_param { order: [0,2,1,3,4] } }")
    b.res_conv("res3a_2n", "res2b_bn", "res3a", 128, 1)
    t = b.bn_relu_3d("res3a", "res3a")
    b.res_conv("res3b_1", t, "res3b_1", 128, 1)
    t = b.bn_relu_3d("res3b_1", "res3b_1")
    b.res_conv("res3b_2", t, "res3b_2", 128, 1)
    b.eltwise("res3b", "res3b_2", "res3a", "res3b")
    t = b.bn_relu_3d("res3b", "res3b")
    for stage, cout in (("res4", 256), ("res5", 512)):
        a, bb = stage + "a", stage + "b"
        b.res_conv(f"{a}_1", t, f"{a}_1", cout, 2)
        u = b.bn_relu_3d(f"{a}_1", f"{a}_1")
        b.res_conv(f"{a}_2", u, f"{a}_2", cout, 1)
        b.res_conv(f"{a}_down", t, f"{a}_down", cout, 2)
        b.eltwise(a, f"{a}_2", f"{a}_down", a)
        u = b.bn_relu_3d(a, a)
        b.res_conv(f"{bb}_1", u, f"{bb}_1", cout, 1)
        u = b.bn_relu_3d(f"{bb}_1", f"{bb}_1")
        b.res_conv(f"{bb}_2", u, f"{bb}_2", cout, 1)
        b.eltwise(bb, f"{bb}_2", a, bb)
        t = b.bn_relu_3d(bb, bb)
    return t


def _fc(b: _Builder, name: str, bottom: str, num_classes: int) -> None:
    b.emit(f'layer {{ name: "{name}" type: "InnerProduct" bottom: "{bottom}" top: "fc8" '
           f"param {{ lr_mult: 1 decay_mult: 1 }} param {{ lr_mult: 2 decay_mult: 0 }} "
           f"inner_product_param {{ num_output: {num_classes} {_XAVIER} }} }}")


def _header(name: str, frames: int, size: int) -> str:
    return (f'name: "{name}"\ninput: "data"\ninput_dim: {frames}\ninput_dim: 3\n'
            f"input_dim: {size}\ninput_dim: {size}\n")


def _check_segments(num_segments: int, input_size: int) -> None:
    if num_segments < 4 or num_segments % 4:
        raise ValueError("num_segments must be a positive multiple of 4 (global_pool depth = N/4)")
    if input_size < 32 or input_size % 32:
        raise ValueError("input_size must be a positive multiple of 32 (224 in every reference net)")


def eco_lite_deploy(num_segments: int = 16, num_clips: int = 5, num_classes: int = 400,
                    dropout_ratio: float = 0.3, fc_name: str = "fc8", net_name: str = "ECOLite",
                    input_size: int = 224, width_div: int = 1) -> str:
    """ECO-Lite deploy graph (models_ECO_Lite/kinetics/deploy.prototxt for the defaults).
    ``input_size`` != 224 / ``width_div`` != 1 give geometrically similar small nets for tests."""
    _check_segments(num_segments, input_size)
    b = _Builder(width_div)
    sp = input_size // 8
    t = _head_2d(b)
    t = b.conv_bn_relu_2d("inception_3c", "double_3x3_reduce", t, 64, 1)
    t = b.conv_bn_relu_2d("inception_3c", "double_3x3_1", t, 96, 3, 1, 1)
    t = _trunk_3d(b, t, num_segments, sp)
    b.pool("global_pool", t, "global_pool", "AVE",
           f"kernel_size: [{num_segments // 4}, {sp // 4}, {sp // 4}] stride: [1, 1, 1]")
    b.reshape("global_pool_reshape", "global_pool", "global_pool_reshape", [-1, b.ch(512)])
    b.dropout("dropout", "global_pool_reshape", dropout_ratio)
    _fc(b, fc_name, "global_pool_reshape", num_classes)
    return _header(net_name, num_clips * num_segments, input_size) + "\n".join(b.out) + "\n"


def eco_full_deploy(num_segments: int = 16, num_clips: int = 5, num_classes: int = 400,
                    dropout_ratio_3d: float = 0.5, dropout_ratio_2d: float = 0.6,
                    fc_name: str = "fc8N", net_name: str = "o3d", input_size: int = 224,
                    width_div: int = 1) -> str:
    """ECO-Full deploy graph (models_ECO_Full/kinetics/deploy.prototxt for the defaults):
    2D head -> {3D trunk, rest of BN-Inception 3c..5b as a per-frame 2D stream with
    segment consensus} -> concat(1024 + 512) -> fc."""
    _check_segments(num_segments, input_size)
    b = _Builder(width_div)
    sp = input_size // 8
    t3b = _head_2d(b)
    # inception_3c (stride-2 block); its double_3x3_1 output also feeds the 3D trunk
    r = b.conv_bn_relu_2d("inception_3c", "3x3_reduce", t3b, 128, 1)
    c3 = b.conv_bn_relu_2d("inception_3c", "3x3", r, 160, 3, 2, 1)
    r = b.conv_bn_relu_2d("inception_3c", "double_3x3_reduce", t3b, 64, 1)
    d1 = b.conv_bn_relu_2d("inception_3c", "double_3x3_1", r, 96, 3, 1, 1)
    t3d = _trunk_3d(b, d1, num_segments, sp)
    d2 = b.conv_bn_relu_2d("inception_3c", "double_3x3_2", d1, 96, 3, 2, 1)
    b.pool("inception_3c_pool", t3b, "inception_3c_pool", "MAX", "kernel_size: 3 stride: 2")
    b.concat("inception_3c_output", [c3, d2, "inception_3c_pool"], "inception_3c_output")
    t = "inception_3c_output"
    t = b.inception("inception_4a", t, 224, 64, 96, 96, 128, "AVE", 128)
    t = b.inception("inception_4b", t, 192, 96, 128, 96, 128, "AVE", 128)
    t = b.inception("inception_4c", t, 160, 128, 160, 128, 160, "AVE", 128)
    t = b.inception("inception_4d", t, 96, 128, 192, 160, 192, "AVE", 128)
    # inception_4e (stride-2 block)
    r = b.conv_bn_relu_2d("inception_4e", "3x3_reduce", t, 128, 1)
    c3 = b.conv_bn_relu_2d("inception_4e", "3x3", r, 192, 3, 2, 1)
    r = b.conv_bn_relu_2d("inception_4e", "double_3x3_reduce", t, 192, 1)
    r = b.conv_bn_relu_2d("inception_4e", "double_3x3_1", r, 256, 3, 1, 1)
    d2 = b.conv_bn_relu_2d("inception_4e", "double_3x3_2", r, 256, 3, 2, 1)
    b.pool("inception_4e_pool", t, "inception_4e_pool", "MAX", "kernel_size: 3 stride: 2")
    b.concat("inception_4e_output", [c3, d2, "inception_4e_pool"], "inception_4e_output")
    t = b.inception("inception_5a", "inception_4e_output", 352, 192, 320, 160, 224, "AVE", 128)
    t = b.inception("inception_5b", t, 352, 192, 320, 192, 224, "MAX", 128)
    b.pool("global_pool2D", t, "global_pool2D", "AVE", f"kernel_size: {sp // 4} stride: 1")
    b.dropout("dropout2D", "global_pool2D", dropout_ratio_2d)
    c2d = b.ch(352) + b.ch(320) + b.ch(224) + b.ch(128)  # inception_5b_output channels (1024)
    b.reshape("reshape_fc_st2", "global_pool2D", "reshape_fc_st2", [-1, 1, num_segments, c2d])
    b.pool("segment_consensus_st2", "reshape_fc_st2", "pool_fusion_st2", "AVE",
           f"kernel_h: {num_segments} kernel_w: 1")
    b.reshape("global_pool_reshape2D", "pool_fusion_st2", "pool_fusion_st2D", [-1, c2d])
    b.pool("global_pool", t3d, "global_pool", "AVE",
           f"kernel_size: [{num_segments // 4}, {sp // 4}, {sp // 4}] stride: [1, 1, 1]")
    b.reshape("global_pool_reshape", "global_pool", "global_pool_reshape", [-1, b.ch(512)])
    b.dropout("dropout", "global_pool_reshape", dropout_ratio_3d)
    b.concat("gn02_concat", ["pool_fusion_st2D", "global_pool_reshape"], "global_pool_gn02_reshape", axis=1)
    _fc(b, fc_name, "global_pool_gn02_reshape", num_classes)
    return _header(net_name, num_clips * num_segments, input_size) + "\n".join(b.out) + "\n"


def test_phase_net(deploy_text: str, num_segments: int, batch_size: int = 1, fc_name: str = "fc8",
                   input_size: int = 224, source: str = "val_frm.txt") -> str:
    """Wrap a deploy graph into the TEST-phase evaluator of the train/val prototxts
    (models_ECO_Lite/kinetics/ECO_Lite.prototxt:66-179,1883-1923): VideoData source (tops ``data`` +
    ``label``), ``reshape_data`` [B, 3N, H, W] -> [B*N, 3, H, W], the deploy body, then ``loss``
    (SoftmaxWithLoss), ``top1`` and ``top5`` (Accuracy), all ``include { phase: TEST }``."""
    body = deploy_text[deploy_text.index("layer {"):]
    if 'bottom: "data"' not in body:
        raise ValueError("deploy graph does not consume a blob named 'data'")
    body = body.replace('bottom: "data"', 'bottom: "reshape_data"')
    mean = "\n".join("    mean_value: [104]\n    mean_value: [117]\n    mean_value: [123]" for _ in range(num_segments))
    head = f"""name: "o3d"
layer {{
  name: "data"
  type: "VideoData"
  top: "data"
  top: "label"
  video_data_param {{
    source: "{source}"
    batch_size: {batch_size}
    new_length: 1
    num_segments: {num_segments}
    modality: RGB
    name_pattern: "img_%04d.jpg"
  }}
  transform_param {{
    crop_size: {input_size}
    mirror: false
{mean}
  }}
  include: {{ phase: TEST }}
}}
layer {{ name: "reshape_data" type: "Reshape" bottom: "data" top: "reshape_data" reshape_param {{ shape {{ dim: -1 dim: 3 dim: {input_size} dim: {input_size} }} }} }}
"""
    tail = f"""layer {{
  name: "loss"
  type: "SoftmaxWithLoss"
  bottom: "{fc_name}"
  bottom: "label"
  top: "loss"
  include {{
    phase: TEST
  }}
}}
layer {{
  name: "top1"
  type: "Accuracy"
  bottom: "{fc_name}"
  bottom: "label"
  top: "top1"
  accuracy_param {{ top_k : 1}}
  include {{
    phase: TEST
  }}
}}
layer {{
  name: "top5"
  type: "Accuracy"
  bottom: "{fc_name}"
  bottom: "label"
  top: "top5"
  accuracy_param {{ top_k: 5 }}
  include {{
    phase: TEST
  }}
}}
"""
    return head + body + tail
