"""ctypes binding of the C ABI declared in ``include/eco_hip.h``.

``load()`` opens the in-tree gfx950 build (``libeco_hip.so`` next to this
file) and *fails loudly* if it is missing or is not a device build -- there is
no CPU fallback anywhere in the product path.  ``EcoLib(path)`` can be pointed
at another build of the same ABI; the CPU test-suite uses that to bind
``tests/emu/libeco_emu.so`` (the fiber-emulated build of the same kernel
sources), never the product.

All pointer arguments are raw integer addresses (``tensor.data_ptr()`` /
``ndarray.ctypes.data``): the binding is agnostic of who owns the memory.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

ECO_OK = 0
ECO_ERR_INVALID = -1
ECO_ERR_RUNTIME = -2
POOL_MAX = 0
POOL_AVE = 1
ABI_VERSION = 19
MAX_SEG = 3   # ECO_MAX_SEG: extra output segments of a sibling launch
DT_BF16 = 1

_i32x3 = C.c_int32 * 3


class EcoError(RuntimeError):
    """A C-ABI call returned an error code (the reference would CHECK-fail)."""

    def __init__(self, code: int, msg: str) -> None:
        super().__init__(f"[eco_hip {code}] {msg}")
        self.code = code


class ConvGeom(C.Structure):
    _fields_ = [("n", C.c_int32), ("cin", C.c_int32), ("cout", C.c_int32),
                ("in_", _i32x3), ("kernel", _i32x3), ("stride", _i32x3),
                ("pad", _i32x3), ("out", _i32x3)]


class ConvPlan(C.Structure):
    _fields_ = [("bm", C.c_int32), ("bn", C.c_int32), ("kc", C.c_int32), ("k", C.c_int32),
                ("kpad", C.c_int32), ("mpad", C.c_int32),
                ("wp_elems", C.c_int64), ("ktab_elems", C.c_int64),
                ("mode", C.c_int32), ("ksplit", C.c_int32), ("ws_bytes", C.c_int64),
                ("split_tiles", C.c_int32), ("streamk_wgs", C.c_int32)]


class ConvBPlan(C.Structure):
    _fields_ = [("bm", C.c_int32), ("bn", C.c_int32), ("dt", C.c_int32), ("stem", C.c_int32),
                ("cblocks", C.c_int32), ("nstages", C.c_int32), ("mpad", C.c_int32), ("ksplit", C.c_int32),
                ("span_pieces", C.c_int32), ("pgrid", C.c_int32), ("wp_vecs", C.c_int64), ("ws_bytes", C.c_int64),
                ("tail_tiles", C.c_int32), ("tail_ksplit", C.c_int32)]


class WGemmPlan(C.Structure):
    _fields_ = [("n", C.c_int32), ("cin", C.c_int32), ("cout", C.c_int32), ("d", C.c_int32), ("th", C.c_int32),
                ("tw", C.c_int32), ("kd", C.c_int32), ("points", C.c_int32), ("bm", C.c_int32), ("bn", C.c_int32),
                ("nstages", C.c_int32), ("ksplit", C.c_int32), ("mblocks", C.c_int32), ("bmp", C.c_int32),
                ("q", C.c_int64), ("u_elems", C.c_int64), ("v_elems", C.c_int64), ("m_elems", C.c_int64)]


class View(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("stride_b", C.c_int64), ("stride_t", C.c_int64),
                ("stride_c", C.c_int64), ("t", C.c_int32)]


class ConvEpilogue(C.Structure):
    _fields_ = [("bias", C.c_void_p), ("residual", View), ("raw", View),
                ("bn_scale", C.c_void_p), ("bn_shift", C.c_void_p),
                ("relu", C.c_int32), ("act", View), ("act2", View),
                ("nseg", C.c_int32), ("seg_begin", C.c_int32 * MAX_SEG), ("seg_relu", C.c_int32 * MAX_SEG),
                ("seg_act", View * MAX_SEG)]


class PoolGeom(C.Structure):
    _fields_ = [("n", C.c_int32), ("c", C.c_int32), ("in_", _i32x3), ("kernel", _i32x3),
                ("stride", _i32x3), ("pad", _i32x3), ("out", _i32x3), ("method", C.c_int32)]


def _pad3(vals: Sequence[int], fill: int) -> "_i32x3":
    vals = list(vals)
    if not 1 <= len(vals) <= 3:
        raise ValueError("1..3 spatial dims supported")
    return _i32x3(*([fill] * (3 - len(vals)) + [int(v) for v in vals]))


def conv_geom(n: int, cin: int, cout: int, in_sp, kernel, stride, pad, out_sp) -> ConvGeom:
    return ConvGeom(int(n), int(cin), int(cout), _pad3(in_sp, 1), _pad3(kernel, 1),
                    _pad3(stride, 1), _pad3(pad, 0), _pad3(out_sp, 1))


def pool_geom(n: int, c: int, in_sp, kernel, stride, pad, out_sp, method: str) -> PoolGeom:
    return PoolGeom(int(n), int(c), _pad3(in_sp, 1), _pad3(kernel, 1), _pad3(stride, 1),
                    _pad3(pad, 0), _pad3(out_sp, 1), {"MAX": POOL_MAX, "AVE": POOL_AVE}[method])


_CONV_TILES = {(128, 128): (2, 2, 2, 2), (128, 256): (2, 4, 2, 2), (96, 256): (3, 2, 1, 4), (64, 256): (2, 2, 1, 4),
               (32, 256): (1, 2, 1, 4)}  # (bm, bn) -> TM,TN,WM,WN


def conv_kernel_name(plan: ConvPlan) -> str:
    """Name of the device kernel eco_conv_forward launches for this plan, as rocprofv3 prints it."""
    if plan.mode == 3:
        tm, tn, wm, wn = {128: (4, 2, 1, 4), 96: (3, 2, 1, 4), 64: (2, 2, 1, 4), 32: (1, 2, 1, 4)}[plan.bm]
        return f"eco::conv_point_kernel<{tm}, {tn}, {wm}, {wn}>"
    tm, tn, wm, wn = _CONV_TILES[(plan.bm, plan.bn)]
    if plan.mode == 2:
        return f"eco::conv_span_kernel<{tm}, {tn}, {wm}, {wn}>"
    if plan.streamk_wgs > 0:
        return f"eco::conv_streamk_kernel<{tm}, {tn}, {wm}, {wn}, {plan.kc}>"
    return f"eco::conv_mfma_kernel<{tm}, {tn}, {wm}, {wn}, {plan.kc}, {plan.mode}>"


def wgemm_kernel_name(plan: "WGemmPlan") -> str:
    tm, tn, wm, wn = {(128, 256): (4, 2, 1, 4), (96, 256): (3, 2, 1, 4), (64, 256): (2, 2, 1, 4), (32, 256): (1, 2, 1, 4),
                      (128, 128): (2, 2, 2, 2), (96, 128): (3, 1, 1, 4), (64, 128): (2, 1, 1, 4), (32, 128): (1, 1, 1, 4)}[(plan.bm, plan.bn)]
    return f"eco::wgemm_kernel<{tm}, {tn}, {wm}, {wn}>"


def convb_kernel_name(plan: "ConvBPlan") -> str:
    """Device kernel eco_convb_forward launches for this plan, as rocprofv3 prints it."""
    if plan.dt == DT_BF16 and plan.span_pieces:
        if plan.pgrid and plan.span_pieces <= 6 and os.environ.get("ECO_SPANP", "1") != "0":
            return f"eco::convb_spanp_kernel<{plan.bm // 32}>"      # persistent form (round 4)
        # (fallback of the persistent form since ABI v19: the per-tap LDS-DMA kernel on the span plan's 256-position tiles)
        tm, tn, wm, wn = {128: (4, 2, 1, 4), 96: (3, 2, 1, 4), 64: (2, 2, 1, 4), 32: (1, 2, 1, 4)}[plan.bm]
        return f"eco::convb_dma_kernel<{tm}, {tn}, {wm}, {wn}>"
    if plan.dt == DT_BF16:   # LDS-DMA kernel
        tm, tn, wm, wn = {(256, 128): (4, 2, 2, 2), (128, 256): (4, 2, 1, 4)}.get((plan.bm, plan.bn)) or _CONV_TILES[(plan.bm, plan.bn)]
        return f"eco::convb_dma_kernel<{tm}, {tn}, {wm}, {wn}>"
    tm, tn, wm, wn = _CONV_TILES[(plan.bm, plan.bn)]
    return f"eco::convb_kernel<{tm}, {tn}, {wm}, {wn}, 3>"


def pool_kernel_name(g: "PoolGeom", y_image_stride: int = 0, y_offset_elems: int = 0) -> str:
    """Device kernel eco_pool_forward / eco_pool_forward_strided picks for this geometry (mirrors the dispatch in
    csrc/eco_ops.hip; the float4 fast paths additionally need 16-byte aligned pointers, which torch allocations are).
    `y_image_stride` / `y_offset_elems`: the pooled blob written as a channel slice of a wider tensor (images that many
    floats apart, the slice starting that many floats into the allocation): the global-average and AVE 3x3/1/1 fast paths
    take dense outputs only, the MAX 3x3/2 one needs a slice whose gap and start are multiples of four floats."""
    two_d = g.in_[0] == 1 and g.kernel[0] == 1 and g.stride[0] == 1 and g.pad[0] == 0
    k, s, p = tuple(g.kernel[1:]), tuple(g.stride[1:]), tuple(g.pad[1:])
    s_out = g.out[0] * g.out[1] * g.out[2]
    y_extra = y_image_stride - g.c * s_out if y_image_stride else 0
    aligned = y_extra % 4 == 0 and y_offset_elems % 4 == 0
    if y_extra == 0 and all(g.kernel[i] == g.in_[i] and g.pad[i] == 0 and g.out[i] == 1 for i in range(3)) and \
            g.method == POOL_AVE and g.in_[0] * g.in_[1] * g.in_[2] >= 32:
        return "eco::global_avg_kernel"
    if two_d and aligned and g.method == POOL_MAX and k == (3, 3) and s == (2, 2) and p == (0, 0) and g.in_[2] % 4 == 0 \
            and g.out[2] % 2 == 0 and 2 * (g.out[2] - 1) + 2 <= g.in_[2]:
        return "eco::maxpool2d_k3s2_kernel<%d>" % (4 if g.out[2] % 4 == 0 else 2)
    if y_extra == 0 and two_d and aligned and g.method == POOL_AVE and k == (3, 3) and s == (1, 1) and p == (1, 1) and \
            g.in_[2] % 2 == 0 and g.in_[1] >= 2 and g.in_[2] >= 4:
        return "eco::avgpool2d_k3s1p1_kernel<%d>" % (4 if g.in_[2] % 4 == 0 else 2)
    if two_d and k == (3, 3):
        return "eco::pool2d_k3_kernel"
    return "eco::pool_kernel"


def plain_view(ptr: int, channels: int, spatial: int) -> View:
    """Dense [N, C, S] tensor."""
    return View(ptr, int(channels) * int(spatial), 0, int(spatial), 1)


def null_view() -> View:
    return View(None, 0, 0, 0, 1)


_SIGNATURES = {
    "eco_abi_version": (C.c_int, []),
    "eco_is_device_build": (C.c_int, []),
    "eco_last_error": (C.c_char_p, []),
    "eco_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "eco_set_device": (C.c_int, [C.c_int]),
    "eco_device_info": (C.c_int, [C.c_int, C.c_char_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_uint64)]),
    "eco_device_pci_bus_id": (C.c_int, [C.c_int, C.c_char_p, C.c_size_t]),
    "eco_source_digest": (C.c_char_p, []),
    "eco_conv_plan_create": (C.c_int, [C.POINTER(ConvGeom), C.POINTER(ConvPlan)]),
    "eco_conv_plan_create_ex": (C.c_int, [C.POINTER(ConvGeom), C.c_int32, C.POINTER(ConvPlan)]),
    "eco_conv_plan_create_batched": (C.c_int, [C.POINTER(ConvGeom), C.c_int32, C.c_int32, C.POINTER(ConvPlan)]),
    "eco_conv_pack_weights": (C.c_int, [C.POINTER(ConvGeom), C.POINTER(ConvPlan), C.c_void_p, C.c_void_p, C.c_void_p]),
    "eco_conv_forward": (C.c_int, [C.POINTER(ConvGeom), C.POINTER(ConvPlan), C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.POINTER(ConvEpilogue), C.c_void_p, C.c_void_p]),
    "eco_conv_forward_batched": (C.c_int, [C.POINTER(ConvGeom), C.POINTER(ConvPlan), C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.POINTER(ConvEpilogue), C.c_void_p, C.c_int32, C.c_int64, C.c_int64,
                                           C.c_int64, C.c_void_p]),
    "eco_wino_weight_transform": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "eco_wino_input_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "eco_wino_output_forward": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                          C.POINTER(ConvEpilogue), C.c_void_p]),
    "eco_pool_forward": (C.c_int, [C.POINTER(PoolGeom), C.c_void_p, C.c_void_p, C.c_void_p]),
    "eco_pool_forward_strided": (C.c_int, [C.POINTER(PoolGeom), C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "eco_bn_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64,
                                 C.c_int, C.c_void_p]),
    "eco_relu_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_void_p]),
    "eco_eltwise_sum_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_float,
                                          C.c_void_p]),
    "eco_concat_copy": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64,
                                  C.c_void_p]),
    "eco_permute_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                      C.c_void_p]),
    "eco_inner_product_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64,
                                            C.c_int64, C.c_void_p]),
    "eco_global_avgpool_fc_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64,
                                                C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_void_p]),
    "eco_global_avgpool_fc_seg_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64,
                                                    C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int,
                                                    C.c_void_p]),
    "eco_avgpool_affine_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(View),
                                             C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "eco_video_input_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                          C.c_int32, C.c_int32, C.POINTER(C.c_float), C.c_float, C.c_int32, C.c_void_p]),
    "eco_softmax_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p]),
    "eco_accuracy_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64,
                                       C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "eco_softmax_loss_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64,
                                           C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "eco_stem_pack_weights": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p]),
    "eco_stem_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                   C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "eco_stemb_weight_elems": (C.c_int64, [C.c_int32]),
    "eco_stemb_pack_weights": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p]),
    "eco_stemb_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                    C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "eco_wino_input_q4_forward": (C.c_int, [C.POINTER(WGemmPlan), C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "eco_wfused_weight_elems": (C.c_int64, [C.POINTER(WGemmPlan)]),
    "eco_wfused_pack_weights": (C.c_int, [C.POINTER(WGemmPlan), C.c_void_p, C.c_void_p]),
    "eco_wfused_forward": (C.c_int, [C.POINTER(WGemmPlan), C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                     C.POINTER(ConvEpilogue), C.c_void_p]),
    "eco_wgemm_plan_create": (C.c_int, [C.c_int32] * 9 + [C.POINTER(WGemmPlan)]),
    "eco_wgemm_pack_weights": (C.c_int, [C.POINTER(WGemmPlan), C.c_void_p, C.c_void_p]),
    "eco_wino_input_pk_forward": (C.c_int, [C.POINTER(WGemmPlan), C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "eco_wgemm_forward": (C.c_int, [C.POINTER(WGemmPlan), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "eco_wino_output_dm_forward": (C.c_int, [C.POINTER(WGemmPlan), C.c_void_p, C.c_int32, C.c_int32,
                                             C.POINTER(ConvEpilogue), C.c_void_p]),
    "eco_counters_reset": (C.c_int, [C.c_void_p]),
    "eco_counters_release_capture_slots": (C.c_int, []),
    "eco_counter_slot_probe": (C.c_int, [C.c_void_p]),
    "eco_wino3_weight_transform": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "eco_wino3_lds_bytes": (C.c_int64, [C.c_int32, C.c_int32, C.c_int32]),
    "eco_wino3_input_forward": (C.c_int, [C.POINTER(WGemmPlan), C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                          C.c_void_p]),
    "eco_wino3_output_forward": (C.c_int, [C.POINTER(WGemmPlan), C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                           C.POINTER(ConvEpilogue), C.c_void_p]),
    "eco_wino_s2_weight_transform": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "eco_wino_s2_lds_bytes": (C.c_int64, [C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "eco_wino_s2_input_forward": (C.c_int, [C.POINTER(WGemmPlan), C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                            C.c_void_p]),
    "eco_wino_s2_output_forward": (C.c_int, [C.POINTER(WGemmPlan), C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                             C.c_int32, C.POINTER(ConvEpilogue), C.c_void_p]),
    "eco_wino_s2d_weight_transform": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "eco_wino_s2d_lds_bytes": (C.c_int64, [C.c_int32] * 5),
    "eco_wino_s2d_input_forward": (C.c_int, [C.POINTER(WGemmPlan), C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                             C.c_int32, C.c_void_p]),
    "eco_wino_s2d_output_forward": (C.c_int, [C.POINTER(WGemmPlan), C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                              C.c_int32, C.POINTER(ConvEpilogue), C.c_void_p]),
    "eco_wfused_pool_scratch_elems": (C.c_int64, [C.POINTER(WGemmPlan)]),
    "eco_wfused_pool_forward": (C.c_int, [C.POINTER(WGemmPlan), C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                          C.POINTER(ConvEpilogue), C.c_void_p, C.c_void_p, C.c_void_p]),
    "eco_convb_plan_create": (C.c_int, [C.POINTER(ConvGeom), C.c_int32, C.c_int32, C.POINTER(ConvBPlan)]),
    "eco_convb_pack_weights": (C.c_int, [C.POINTER(ConvGeom), C.POINTER(ConvBPlan), C.c_void_p, C.c_void_p]),
    "eco_convb_forward": (C.c_int, [C.POINTER(ConvGeom), C.POINTER(ConvBPlan), C.c_void_p, C.c_void_p,
                                    C.POINTER(ConvEpilogue), C.c_void_p, C.c_void_p]),
    "eco_stem_pack_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "eco_poolb_forward": (C.c_int, [C.POINTER(PoolGeom), C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "eco_poolb_avg_affine_forward": (C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                               C.POINTER(View), C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "eco_global_avgpool_fc_b_forward": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                                  C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int,
                                                  C.c_void_p]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)


class EcoLib:
    """One loaded build of the ECO C ABI."""

    def __init__(self, path: str) -> None:
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        self.path = path
        self._dll = C.CDLL(path)
        for name, (res, args) in _SIGNATURES.items():
            try:
                fn = getattr(self._dll, name)
            except AttributeError as e:
                raise ImportError(f"{path} does not export {name}") from e
            fn.restype = res
            fn.argtypes = args
        v = self._dll.eco_abi_version()
        if v != ABI_VERSION:
            raise ImportError(f"{path}: ABI version {v}, binding expects {ABI_VERSION}")
        self.is_device_build = bool(self._dll.eco_is_device_build())

    # -- plumbing -------------------------------------------------------------
    def _check(self, rc: int) -> None:
        if rc != ECO_OK:
            raise EcoError(rc, (self._dll.eco_last_error() or b"").decode("utf-8", "replace"))

    def last_error(self) -> str:
        return (self._dll.eco_last_error() or b"").decode("utf-8", "replace")

    # -- device ---------------------------------------------------------------
    def device_count(self) -> int:
        n = C.c_int(0)
        self._check(self._dll.eco_device_count(C.byref(n)))
        return n.value

    def set_device(self, dev: int) -> None:
        self._check(self._dll.eco_set_device(int(dev)))

    def device_info(self, dev: int) -> dict:
        name = C.create_string_buffer(256)
        cu = C.c_int(0)
        mem = C.c_uint64(0)
        self._check(self._dll.eco_device_info(int(dev), name, 256, C.byref(cu), C.byref(mem)))
        return {"name": name.value.decode(), "num_cu": cu.value, "hbm_bytes": mem.value}

    def device_pci_bus_id(self, dev: int) -> str:
        """PCI address of the device as sysfs spells it ("0000:c1:00.0")."""
        buf = C.create_string_buffer(32)
        self._check(self._dll.eco_device_pci_bus_id(int(dev), buf, 32))
        return buf.value.decode()

    def source_digest(self) -> str:
        """SHA-256 of the sources this library was built from (csrc/Makefile), or "unknown"."""
        return (self._dll.eco_source_digest() or b"unknown").decode()

    # -- convolution ----------------------------------------------------------
    def conv_plan(self, g: ConvGeom, num_cu: Optional[int] = None, batch: int = 1) -> ConvPlan:
        p = ConvPlan()
        if num_cu is None and batch == 1:
            self._check(self._dll.eco_conv_plan_create(C.byref(g), C.byref(p)))
        elif batch == 1:
            self._check(self._dll.eco_conv_plan_create_ex(C.byref(g), int(num_cu), C.byref(p)))
        else:
            self._check(self._dll.eco_conv_plan_create_batched(C.byref(g), 0 if num_cu is None else int(num_cu),
                                                               int(batch), C.byref(p)))
        return p

    def conv_pack_weights(self, g: ConvGeom, p: ConvPlan, w_ptr: int, wp_ptr: int, ktab_ptr: int) -> None:
        self._check(self._dll.eco_conv_pack_weights(C.byref(g), C.byref(p), w_ptr, wp_ptr, ktab_ptr))

    def conv_forward(self, g: ConvGeom, p: ConvPlan, x: int, wp: int, ktab: int, ep: ConvEpilogue,
                     workspace: Optional[int] = None, stream: Optional[int] = None) -> None:
        self._check(self._dll.eco_conv_forward(C.byref(g), C.byref(p), x, wp, ktab, C.byref(ep), workspace, stream))

    def conv_forward_batched(self, g: ConvGeom, p: ConvPlan, x: int, wp: int, ktab: int, ep: ConvEpilogue,
                             workspace: Optional[int], batch: int, stride_x: int, stride_wp: int, stride_out: int,
                             stream: Optional[int] = None) -> None:
        self._check(self._dll.eco_conv_forward_batched(C.byref(g), C.byref(p), x, wp, ktab, C.byref(ep), workspace,
                                                       batch, stride_x, stride_wp, stride_out, stream))

    # -- Winograd F(MxM,3x3) front / back end (M = 2 or 4) ----------------------------------
    def wino_weight_transform(self, w_host: int, cout: int, cin: int, kd: int, tile_m: int, u_host: int) -> None:
        self._check(self._dll.eco_wino_weight_transform(w_host, cout, cin, kd, tile_m, u_host))

    def wino_input_forward(self, x: int, v: int, planes: int, h: int, w: int, tile_m: int, stream=None) -> None:
        self._check(self._dll.eco_wino_input_forward(x, v, planes, h, w, tile_m, stream))

    def wino_output_forward(self, m: int, n: int, cout: int, d: int, h: int, w: int, tile_m: int, ep: ConvEpilogue,
                            stream=None) -> None:
        self._check(self._dll.eco_wino_output_forward(m, n, cout, d, h, w, tile_m, C.byref(ep), stream))

    # -- the stem as one launch (csrc/eco_stem.hip) -----------------------------------
    def stem_pack_weights(self, w_host: int, cout: int, wp_host: int) -> None:
        self._check(self._dll.eco_stem_pack_weights(w_host, cout, wp_host))

    def stem_forward(self, x, wp, bias, bn_scale, bn_shift, relu, y, n, h, w, cout, stream=None,
                     max_workgroups: int = 0) -> None:
        self._check(self._dll.eco_stem_forward(x, wp, bias, bn_scale, bn_shift, int(relu), y, n, h, w, cout,
                                               max_workgroups, stream))

    # -- the stem of the blocked bf16 path as one launch (csrc/eco_stemb.hip) ----------
    def stemb_weight_elems(self, cout: int) -> int:
        return int(self._dll.eco_stemb_weight_elems(cout))

    def stemb_pack_weights(self, w_host: int, cout: int, wp_host: int) -> None:
        self._check(self._dll.eco_stemb_pack_weights(w_host, cout, wp_host))

    def stemb_forward(self, x, wp, bias, bn_scale, bn_shift, relu, y, n, h, w, cout, stream=None,
                      max_workgroups: int = 0) -> None:
        self._check(self._dll.eco_stemb_forward(x, wp, bias, bn_scale, bn_shift, int(relu), y, n, h, w, cout,
                                                max_workgroups, stream))

    # -- fused transformed-domain GEMM + output transform for the short-reduction 2-D layers --
    def wino_input_q4_forward(self, plan: "WGemmPlan", x: int, v: int, h: int, w: int, stream=None) -> None:
        self._check(self._dll.eco_wino_input_q4_forward(C.byref(plan), x, v, h, w, stream))

    def wfused_weight_elems(self, plan: "WGemmPlan") -> int:
        return int(self._dll.eco_wfused_weight_elems(C.byref(plan)))

    def wfused_pack_weights(self, plan: "WGemmPlan", u_host: int, up_host: int) -> None:
        self._check(self._dll.eco_wfused_pack_weights(C.byref(plan), u_host, up_host))

    def wfused_forward(self, plan: "WGemmPlan", v: int, up: int, h: int, w: int, ep: "ConvEpilogue", stream=None) -> None:
        self._check(self._dll.eco_wfused_forward(C.byref(plan), v, up, h, w, C.byref(ep), stream))

    def wfused_pool_scratch_elems(self, plan: "WGemmPlan") -> int:
        return int(self._dll.eco_wfused_pool_scratch_elems(C.byref(plan)))

    def wfused_pool_forward(self, plan: "WGemmPlan", v: int, up: int, h: int, w: int, ep: "ConvEpilogue", scratch: int, y: int,
                            stream=None) -> None:
        """eco_wfused_forward + the MAX 3x3 / 2 pooling that follows it (conv2_3x3 -> pool2): pooled blob only."""
        self._check(self._dll.eco_wfused_pool_forward(C.byref(plan), v, up, h, w, C.byref(ep), scratch, y, stream))

    # -- Winograd F(4x4,3x3) on the dedicated transformed-domain GEMM (csrc/eco_wgemm.hip) --
    def wgemm_plan(self, n, cin, cout, d, th, tw, kd, num_cu: Optional[int] = None, points: int = 36) -> "WGemmPlan":
        """points = 36: F(4x4,3x3), d = depth planes, kd depth taps direct; points = 216: F(4x4x4,3x3x3), d = depth TILES,
        kd = 1 (csrc/eco_wino3.hip)."""
        p = WGemmPlan()
        self._check(self._dll.eco_wgemm_plan_create(n, cin, cout, d, th, tw, kd, points, 0 if num_cu is None else int(num_cu),
                                                    C.byref(p)))
        return p

    def counters_reset(self, stream=None) -> None:
        """Zero the work counters of the dynamic-share launches (after a faulted kernel; include/eco_hip.h)."""
        self._check(self._dll.eco_counters_reset(stream))

    def counters_release_capture_slots(self) -> None:
        """Return the 192 capture slots to the pool: only when every graph captured so far has been destroyed."""
        self._check(self._dll.eco_counters_release_capture_slots())

    def counter_slot_probe(self, stream=None) -> int:
        """Slot (0..255) a dynamic-share launch on `stream` would take now, -1 = static shares (include/eco_hip.h)."""
        return int(self._dll.eco_counter_slot_probe(stream))

    # -- Winograd F(4x4x4,3x3x3): the 3-D trunk's transforms around the same GEMM (csrc/eco_wino3.hip) --
    def wino3_weight_transform(self, w_host: int, cout: int, cin: int, u_host: int) -> None:
        self._check(self._dll.eco_wino3_weight_transform(w_host, cout, cin, u_host))

    def wino3_lds_bytes(self, n: int, th: int, tw: int) -> int:
        return int(self._dll.eco_wino3_lds_bytes(n, th, tw))

    def wino3_input_forward(self, p: "WGemmPlan", x: int, v: int, d: int, h: int, w: int, stream=None) -> None:
        self._check(self._dll.eco_wino3_input_forward(C.byref(p), x, v, d, h, w, stream))

    def wino3_output_forward(self, p: "WGemmPlan", m: int, d: int, h: int, w: int, ep: ConvEpilogue, stream=None) -> None:
        self._check(self._dll.eco_wino3_output_forward(C.byref(p), m, d, h, w, C.byref(ep), stream))

    # -- stride-2 3x3x3 convolutions as polyphase F(4,2) x F(7,2) x F(7,2) problems on the same GEMM (csrc/eco_wino_s2.hip) --
    def wino_s2_weight_transform(self, w_host: int, cout: int, cin: int, u_host: int) -> None:
        """u[320][cout][8 cin] from w[cout][cin][3][3][3] (host)."""
        self._check(self._dll.eco_wino_s2_weight_transform(w_host, cout, cin, u_host))

    def wino_s2_lds_bytes(self, n: int, td: int, th: int, tw: int) -> int:
        return int(self._dll.eco_wino_s2_lds_bytes(n, td, th, tw))

    def wino_s2_input_forward(self, p: "WGemmPlan", x: int, v: int, d: int, h: int, w: int, stream=None) -> None:
        """(d, h, w): the INPUT volume; the plan is wgemm_plan(n, 8 * cin, ctot, d/8, h/14, w/14, 1, points=320)."""
        self._check(self._dll.eco_wino_s2_input_forward(C.byref(p), x, v, d, h, w, stream))

    def wino_s2_output_forward(self, p: "WGemmPlan", m: int, c0: int, cout: int, od: int, oh: int, ow: int, ep: ConvEpilogue,
                               stream=None) -> None:
        """Channels [c0, c0 + cout) of the GEMM's rows -> the (od, oh, ow) OUTPUT volume with the member's epilogue."""
        self._check(self._dll.eco_wino_s2_output_forward(C.byref(p), m, c0, cout, od, oh, ow, C.byref(ep), stream))

    # ... and the 2-D form: F(7,2) x F(7,2) on 64 points, depth taps (kz = 3) or none (kz = 1) in the reduction
    def wino_s2d_weight_transform(self, w_host: int, cout: int, cin: int, kz: int, u_host: int) -> None:
        """u[64][cout][4 kz cin] from w[cout][cin][kz][3][3] (host)."""
        self._check(self._dll.eco_wino_s2d_weight_transform(w_host, cout, cin, kz, u_host))

    def wino_s2d_lds_bytes(self, n: int, kz: int, od: int, th: int, tw: int) -> int:
        return int(self._dll.eco_wino_s2d_lds_bytes(n, kz, od, th, tw))

    def wino_s2d_input_forward(self, p: "WGemmPlan", x: int, v: int, kz: int, d: int, h: int, w: int, stream=None) -> None:
        """(d, h, w): the INPUT volume; the plan is wgemm_plan(n, 4 * kz * cin, ctot, od, h/14, w/14, 1, points=64)."""
        self._check(self._dll.eco_wino_s2d_input_forward(C.byref(p), x, v, kz, d, h, w, stream))

    def wino_s2d_output_forward(self, p: "WGemmPlan", m: int, c0: int, cout: int, od: int, oh: int, ow: int, ep: ConvEpilogue,
                                stream=None) -> None:
        self._check(self._dll.eco_wino_s2d_output_forward(C.byref(p), m, c0, cout, od, oh, ow, C.byref(ep), stream))

    def wgemm_pack_weights(self, p: "WGemmPlan", u_host: int, up_host: int) -> None:
        self._check(self._dll.eco_wgemm_pack_weights(C.byref(p), u_host, up_host))

    def wino_input_pk_forward(self, p: "WGemmPlan", x: int, v: int, h: int, w: int, stream=None) -> None:
        self._check(self._dll.eco_wino_input_pk_forward(C.byref(p), x, v, h, w, stream))

    def wgemm_forward(self, p: "WGemmPlan", v: int, up: int, m: int, stream=None) -> None:
        self._check(self._dll.eco_wgemm_forward(C.byref(p), v, up, m, stream))

    def wino_output_dm_forward(self, p: "WGemmPlan", m: int, h: int, w: int, ep: ConvEpilogue, stream=None) -> None:
        self._check(self._dll.eco_wino_output_dm_forward(C.byref(p), m, h, w, C.byref(ep), stream))

    # -- channel-blocked bf16-MFMA path (csrc/eco_blocked.hip) ------------------
    def convb_plan(self, g: ConvGeom, dt: int, num_cu: Optional[int] = None) -> "ConvBPlan":
        p = ConvBPlan()
        self._check(self._dll.eco_convb_plan_create(C.byref(g), int(dt), 0 if num_cu is None else int(num_cu), C.byref(p)))
        return p

    def convb_pack_weights(self, g: ConvGeom, p: "ConvBPlan", w_ptr: int, wp_ptr: int) -> None:
        self._check(self._dll.eco_convb_pack_weights(C.byref(g), C.byref(p), w_ptr, wp_ptr))

    def convb_forward(self, g: ConvGeom, p: "ConvBPlan", x: int, wp: int, ep: ConvEpilogue,
                      workspace: Optional[int] = None, stream: Optional[int] = None) -> None:
        self._check(self._dll.eco_convb_forward(C.byref(g), C.byref(p), x, wp, C.byref(ep), workspace, stream))

    def stem_pack_forward(self, x: int, y: int, frames: int, h: int, w: int, dt: int, stream=None) -> None:
        self._check(self._dll.eco_stem_pack_forward(x, y, frames, h, w, int(dt), stream))

    def poolb_forward(self, g: PoolGeom, dt: int, x: int, y: int, stream=None) -> None:
        self._check(self._dll.eco_poolb_forward(C.byref(g), int(dt), x, y, stream))

    def poolb_avg_affine_forward(self, dt, x, bias, bn_scale, bn_shift, relu, dst: "View", n, c, h, w, stream=None) -> None:
        """Blocked form of avgpool_affine_forward: x[n][c/8][h][w][8] -> 3x3/1/1 average, + bias, folded BN, ReLU -> view."""
        self._check(self._dll.eco_poolb_avg_affine_forward(int(dt), x, bias, bn_scale, bn_shift, int(relu), C.byref(dst),
                                                           n, c, h, w, stream))

    def global_avgpool_fc_b_forward(self, x, dt, w, bias, y, b, c, s, n_out, wk, c0=0, accumulate=False, stream=None) -> None:
        self._check(self._dll.eco_global_avgpool_fc_b_forward(x, int(dt), w, bias, y, b, c, s, n_out, wk, c0,
                                                              int(accumulate), stream))

    # -- stand-alone operators -----------------------------------------------
    def pool_forward(self, g: PoolGeom, x: int, y: int, stream=None) -> None:
        self._check(self._dll.eco_pool_forward(C.byref(g), x, y, stream))


    def pool_forward_strided(self, g: "PoolGeom", x: int, y: int, y_image_stride: int, stream=None) -> None:
        """eco_pool_forward into a channel slice of a wider tensor (a Concat top): images y_image_stride floats apart."""
        self._check(self._dll.eco_pool_forward_strided(C.byref(g), x, y, y_image_stride, stream))
    def bn_forward(self, x, y, scale, shift, n, c, inner, relu, stream=None) -> None:
        self._check(self._dll.eco_bn_forward(x, y, scale, shift, n, c, inner, int(relu), stream))

    def relu_forward(self, x, y, count, negative_slope=0.0, stream=None) -> None:
        self._check(self._dll.eco_relu_forward(x, y, count, float(negative_slope), stream))

    def eltwise_sum_forward(self, a, b, y, count, ca=1.0, cb=1.0, stream=None) -> None:
        self._check(self._dll.eco_eltwise_sum_forward(a, b, y, count, float(ca), float(cb), stream))

    def concat_copy(self, x, y, outer, cx, cy, c0, inner, stream=None) -> None:
        self._check(self._dll.eco_concat_copy(x, y, outer, cx, cy, c0, inner, stream))

    def permute_forward(self, x, y, in_shape: Sequence[int], order: Sequence[int], stream=None) -> None:
        n = len(in_shape)
        shp = (C.c_int32 * n)(*[int(s) for s in in_shape])
        ordr = (C.c_int32 * n)(*[int(o) for o in order])
        self._check(self._dll.eco_permute_forward(x, y, n, shp, ordr, stream))

    def inner_product_forward(self, x, w, bias, y, m, n, k, stream=None) -> None:
        self._check(self._dll.eco_inner_product_forward(x, w, bias, y, m, n, k, stream))

    def global_avgpool_fc_forward(self, x, w, bias, y, b, c, s, n_out, wk, c0=0, accumulate=False, stream=None) -> None:
        self._check(self._dll.eco_global_avgpool_fc_forward(x, w, bias, y, b, c, s, n_out, wk, c0,
                                                            int(accumulate), stream))

    def global_avgpool_fc_seg_forward(self, x, w, bias, y, b, t, c, s, n_out, wk, c0=0, accumulate=False, stream=None) -> None:
        self._check(self._dll.eco_global_avgpool_fc_seg_forward(x, w, bias, y, b, t, c, s, n_out, wk, c0,
                                                                int(accumulate), stream))

    def avgpool_affine_forward(self, x, bias, bn_scale, bn_shift, relu, dst: "View", n, c, h, w, stream=None) -> None:
        """AVE 3x3 stride 1 pad 1 of x[n,c,h,w], then (+ bias) * bn_scale + bn_shift, ReLU, into the strided view."""
        self._check(self._dll.eco_avgpool_affine_forward(x, bias, bn_scale, bn_shift, int(relu), C.byref(dst), n, c, h, w,
                                                         stream))

    def video_input_forward(self, frames, y, num_frames, height, width, crop_h, crop_w, h_off, w_off, mean, scale=1.0,
                            mirror=False, stream=None) -> None:
        m = (C.c_float * 3)(*[float(v) for v in mean])
        self._check(self._dll.eco_video_input_forward(frames, y, num_frames, height, width, crop_h, crop_w, h_off, w_off,
                                                      m, float(scale), int(bool(mirror)), stream))

    def softmax_forward(self, x, y, outer, c, inner, stream=None) -> None:
        self._check(self._dll.eco_softmax_forward(x, y, outer, c, inner, stream))

    def accuracy_forward(self, x, label, out, outer, c, inner, top_k, ignore_label=None, stream=None) -> None:
        self._check(self._dll.eco_accuracy_forward(x, label, out, outer, c, inner, top_k,
                                                   ignore_label is not None, ignore_label or 0, stream))

    def softmax_loss_forward(self, x, label, out, outer, c, inner, normalize=True, ignore_label=None,
                             stream=None) -> None:
        self._check(self._dll.eco_softmax_loss_forward(x, label, out, outer, c, inner, bool(normalize),
                                                       ignore_label is not None, ignore_label or 0, stream))


_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG_DIR, "libeco_hip.so")
_lib: Optional[EcoLib] = None


def load() -> EcoLib:
    """The product library.  No fallback: a missing or non-device build is an error."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                f"or `make -C {os.path.join(_PKG_DIR, 'csrc')}`. The ECO path has no CPU fallback.")
        # PyTorch-ROCm bundles its own libamdhip64.so.7 / libhsa-runtime64.so.1 (same sonames as
        # /opt/rocm/lib).  Whichever is dlopen'ed first serves the whole process, and torch only
        # works with its own copy -- so bring torch's runtime in before our library binds to it.
        import torch  # noqa: F401
        lib = EcoLib(LIB_PATH)
        if not lib.is_device_build:
            raise ImportError(f"{LIB_PATH} is not a gfx950 device build")
        _lib = lib
    return _lib
