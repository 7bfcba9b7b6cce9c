"""Execution plan for an ECO ``NetSpec`` on the HIP C ABI.

Replaces the reference's per-layer virtual dispatch (``Net::ForwardFromTo``,
caffe_3d/src/caffe/net.cpp:566-583 -> ``Layer::Forward``, layer.hpp:444-477)
with a static list of C-ABI launches built once per shape:

* ``fuse=False`` -- one launch per prototxt layer (Convolution, BN, ReLU, Pooling,
  Concat, Eltwise, Permute, InnerProduct, Softmax; Reshape/Split/Dropout(TEST) are
  aliases exactly as in the reference: reshape_layer.cpp:88, split_layer.cpp:26-32,
  dropout_layer.cpp:46-48).  Every blob of the prototxt is materialised and
  observable, like pycaffe.
* ``fuse=True`` (default) -- the MI355X plan: each Convolution absorbs the layers
  that follow it into its epilogue (bias, Eltwise-SUM residual, BN folded to a
  per-channel affine, ReLU), writes straight into its channel slice of a Concat
  top or through r2Dto3D+Permute with permuted strides, and the
  global_pool->reshape->dropout->fc tail is one launch.  Blobs that only exist
  inside a fused group are not materialised (asking for them raises).

Memory: every materialised blob gets its own HBM allocation (no in-place
sharing tricks like the reference's MemoryOptimize, net.cpp:1079+: 15 GB at
N=16 B=32 against 288 GB of HBM3E).  All launches go to one stream in layer
order; nothing synchronises.
"""
from __future__ import annotations

import ctypes as C
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import hip
from .netspec import LayerSpec, NetSpec, NetSpecError, param_shapes

_ALIAS_TYPES = ("Split", "Reshape", "Dropout")


def _prod(xs) -> int:
    p = 1
    for x in xs:
        p *= int(x)
    return p


class TorchAllocator:
    """Device memory + stream plumbing through PyTorch-ROCm (plumbing only)."""

    def __init__(self, device: Optional[int] = None) -> None:
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("no HIP device visible to PyTorch-ROCm; the ECO path has no CPU fallback")
        self.torch = torch
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else device)

    def empty(self, nelems: int, dtype=np.float32):
        # uint16 = raw bf16 bits (the blocked bf16 path); torch has no uint16 arithmetic, the buffer is int16
        tdt = {np.dtype(np.float32): self.torch.float32, np.dtype(np.int32): self.torch.int32,
               np.dtype(np.uint16): self.torch.int16}[np.dtype(dtype)]
        return self.torch.empty(max(int(nelems), 1), dtype=tdt, device=self.device)

    @staticmethod
    def ptr(h) -> int:
        return h.data_ptr()

    def upload(self, h, arr: np.ndarray) -> None:
        a = np.ascontiguousarray(arr).reshape(-1)
        if a.dtype == np.uint16:
            a = a.view(np.int16)
        h[: a.size].copy_(self.torch.from_numpy(a), non_blocking=False)

    def download(self, h, nelems: int) -> np.ndarray:
        a = h[:nelems].detach().cpu().numpy()
        return a.view(np.uint16) if a.dtype == np.int16 else a

    def stream(self) -> Optional[int]:
        return self.torch.cuda.current_stream(self.device).cuda_stream

    def synchronize(self) -> None:
        self.torch.cuda.synchronize(self.device)


class _Tensor:
    """A materialised blob: storage handle + logical shape (aliases share a handle)."""

    __slots__ = ("handle", "shape", "owner", "dt")

    def __init__(self, handle, shape, owner: str, dt: int = 0) -> None:
        self.handle = handle
        self.shape = tuple(int(s) for s in shape)
        self.owner = owner  # blob name that owns the storage
        self.dt = dt        # 0: fp32 in the reference's N,C,... layout; hip.DT_*: channel-blocked storage (blocked.py)

    @property
    def count(self) -> int:
        return _prod(self.shape)


def fold_bn(blobs: Sequence[np.ndarray], eps: float) -> Tuple[np.ndarray, np.ndarray]:
    """BN TEST branch as y = x*a + b: a = gamma/sqrt(var+eps), b = beta - mean*a
    (bn_layer.cpp:93-207; same algebra as python/gen_bn_inference.py:121-134).
    Folded in float64, stored fp32."""
    gamma, beta, mean, var = (np.asarray(b, np.float64).reshape(-1) for b in blobs[:4])
    a = gamma / np.sqrt(var + eps)
    b = beta - mean * a
    return a.astype(np.float32), b.astype(np.float32)


def bn_eps(L: LayerSpec) -> float:
    # 5-D blobs only run through cuDNN in the reference: eps = max(eps, CUDNN_BN_MIN_EPSILON)
    # (cudnn_bn_layer.cu:24); <=4-D uses bn_param.eps as is (bn_layer.cpp:159).
    e = L.geom["eps"]
    return max(e, 1e-5) if len(L.bottom_shapes[0]) > 4 else e


class Engine:
    def __init__(self, spec: NetSpec, lib: hip.EcoLib, alloc, fuse: bool = True, winograd=True,
                 num_cu: Optional[int] = None, dtype: str = "f32") -> None:
        self.spec = spec
        self.lib = lib
        self.alloc = alloc
        self.fuse = fuse
        # "f32": fp32 N,C,[D,]H,W blobs, fp32 MFMA kernels (eco_conv.hip).  "bf16": the channel-blocked
        # path on the bf16 matrix cores (eco_blocked.hip) with bf16 storage / fp32 storage and exactly split
        # operands; fused plan only, every convolution evaluated directly.
        if dtype not in ("f32", "bf16"):
            raise ValueError("dtype must be 'f32' or 'bf16'")
        self.dtype = dtype
        self.dt = {"f32": 0, "bf16": hip.DT_BF16}[dtype]
        if self.dt and not fuse:
            raise NetSpecError("the blocked bf16-MFMA path runs the fused plan only (fuse=True)")
        self.esize = 2 if self.dt == hip.DT_BF16 else 4        # bytes per stored activation element
        self.cblk = 8 if self.dt else 1                        # channels per view stride unit
        self.store_np = np.uint16 if self.dt == hip.DT_BF16 else np.float32
        # Winograd F(MxM,3x3) for the stride-1 3x3x3 convs of the 3-D trunk: False = direct evaluation, 2 / 4 =
        # output tile M, True = pick M per layer (4 where the planes are large enough to tile by 4)
        if winograd not in (False, True, 2, 4):
            raise ValueError("winograd must be False, True, 2 or 4")
        self.winograd = winograd
        self.wino_min_tiles = 64   # size rule of winograd=True: tile positions per transform point
        self.wfused = True         # ... and, for the short-reduction 2-D layers, GEMM + output transform in one kernel
        # (the kernel takes up to 224 input channels; measured on ECO-Full at 32 clips: cut-off 96 / 128 / 160 / 224
        # -> 26.57 / 26.19 / 26.26 / 26.75 ms per step -- above 128 the unfused GEMM's larger tiles win)
        self.wfused_max_cin = 128
        self.wgemm = True          # F(4x4,3x3) GEMMs on the dedicated dense kernel (False: round-1 gather-kernel route)
        # 3x3x3 layers: nest the minimal-filtering algorithm over depth as well, F(4x4x4,3x3x3) (csrc/eco_wino3.hip): half
        # the transformed-domain multiplies of F(4x4,3x3) + direct depth taps, V / M 1.33x / 1.5x larger.  Taken where the
        # depth tiles (4 planes) waste little (8*ceil(D/4) <= 3*D: D = 3, 4, 6, 7, 8, ...) and a transform point still has
        # wino3_min_positions tile positions (at one clip res4 / res5 have 32 / 4: they stay on the 2-D route)
        self.wino3 = True
        self.wino3_min_positions = 128
        # STRIDE-2 3x3x3 layers (res4a_1 / res4a_down) as eight polyphase F(4,2) x F(7,2) x F(7,2) problems on the same GEMM
        # (csrc/eco_wino_s2.hip): 13.1 multiplies per output and input channel instead of 27, where the output volume tiles
        # by 4 x 7 x 7, the input is exactly twice as large and a transform point still has wino_s2_min_positions tile
        # positions; else (res5a at 32 clips has 32 and would stream 1.3 GB of transformed weights) the 2-D form, F(7,2) x F(7,2) with
        # every output plane a position and the depth taps in the reduction -- also for strided 2-D 3x3 convs where a cost estimate
        # says the transforms pay (_ws2_form).  Convs of one geometry on the same bottom share the input transform and the GEMM.
        self.wino_s2 = True
        self.wino_s2_min_positions = 128
        self._ws2_groups: Dict[str, dict] = {}
        # conv2_3x3 -> pool2: the fused 2-D Winograd kernel writes partial window maxima instead of the conv output
        # (csrc/eco_wgemm.hip, wino_pool9_store + pool9_finish_kernel)
        self.wpool = True
        self.pool_into_concat = True   # a pooling feeding only a channel Concat writes its slice itself (ECO-Full 3c / 4e)
        self.stem = True           # conv1 + BN + ReLU + pool1 as one launch (False: conv kernel + pooling kernel)
        self.siblings = True       # 1x1 convs reading the same bottom as one launch (fp32 path)
        # AVE pool 3x3/1/1 -> 1x1 conv (inception_3a/3b pool + pool_proj): both maps are linear, so the conv runs first
        # -- on the block's input, as one more member of the block's sibling launch -- and the window average (+ bias,
        # BN, ReLU) on its cout output channels instead of on the cin input channels (csrc/eco_ops.hip,
        # avgpool2d_k3s1p1_affine_kernel).  fp32 path; logits differ from the reference order by fp32 rounding.
        self.pool_commute = True
        # ... and a residual block's strided first conv with its projection shortcut (res4a_1 | res4a_down, res5a_1 | res5a_down) as
        # one direct launch; the shortcut then keeps its raw value and the Eltwise rides on the block's second conv.  Round 2 left
        # this off: at 32 clips the fp32 512-channel launch quantised worse over the CUs than two 256-channel ones (1.73-1.89 ms
        # against 2 x 0.80; res5a: 0.80 against 0.85, profiles/r02_notes.md).  Round 6: at 32 clips the fp32 pairs run as polyphase
        # groups (_ws2_form) and never get here; what does get here gains -- fp32 at small batches (one clip per step: 1.150 ->
        # 1.083 ms, two: 1.525 -> 1.483, four: 2.076 -> 2.054; tools/exp/b1_sibling_blocks.py) and the blocked bf16 path, whose
        # strided launches are bound by operand traffic through L2 (configs[4]: res4a 0.196 + 0.214 -> 0.335 ms, res5a 0.115 +
        # 0.125 -> 0.171, step 6.47 -> 6.35; tools/exp/bf16_sibling_blocks.py): the pair reads (gathers) its positions once.
        self.sibling_blocks = True
        self.num_cu = num_cu       # None = the device's 256 CUs (tests shrink it to reach split-K paths)
        self.params: Dict[str, List[np.ndarray]] = {}
        self._param_dev: Dict[str, dict] = {}    # layer name -> device-side state
        self._group_dev: Dict[str, dict] = {}    # sibling-conv group -> device-side state (concatenated members)
        self._groups: Dict[str, dict] = {}
        self._dirty_groups: set = set()
        self._dirty_params: set = set()
        self.tensors: Dict[str, _Tensor] = {}
        self.fused_away: Dict[str, str] = {}     # blob -> reason
        # (layer idx, label, fn(stream), meta {"kernel": str, "flops": int, "bytes": int})
        self.ops: List[Tuple[int, str, Callable[[Optional[int]], None], dict]] = []
        self._keep = []                          # ctypes structs referenced by closures
        self._built = False
        self.generation = 0                      # bumped by every build() / parameter upload (hipGraph cache key)

    # ------------------------------------------------------------------ params
    def set_params(self, params: Dict[str, List[np.ndarray]]) -> None:
        for L in self.spec.layers:
            shapes = param_shapes(L)
            if not shapes:
                continue
            if L.name not in params:
                raise KeyError(f"no parameters for layer {L.name!r}")
            blobs = params[L.name]
            if len(blobs) != len(shapes):  # net.cpp:869-870
                raise ValueError(f"Incompatible number of blobs for layer {L.name}: {len(blobs)} vs {len(shapes)}")
            out = []
            for b, s in zip(blobs, shapes):
                b = np.asarray(b, dtype=np.float32)
                if b.size != _prod(s):
                    raise ValueError(f"layer {L.name}: parameter shape {b.shape} does not match {s}")
                out.append(np.ascontiguousarray(b.reshape(s)))
            self.params[L.name] = out
            self._dirty_params.add(L.name)

    def mark_param_dirty(self, layer_name: str) -> None:
        self._dirty_params.add(layer_name)

    def _upload_f32(self, arr: np.ndarray):
        h = self.alloc.empty(arr.size, np.float32)
        self.alloc.upload(h, arr.astype(np.float32, copy=False))
        return h

    def _sync_params(self) -> None:
        """(Re)upload parameters whose host copy changed; repack conv weights."""
        if not self._dirty_params and not self._dirty_groups:
            return
        self.generation += 1
        self._sync_groups(set(self._dirty_params))
        self._sync_ws2(set(self._dirty_params))
        for name in list(self._dirty_params):
            L = self.spec.layer(name)
            st = self._param_dev.setdefault(name, {})
            blobs = self.params[name]
            if L.type == "Convolution" and self.dt:
                bp = st["bplan"]
                wpb = np.empty(bp.wp_vecs * 8, np.uint16)
                w = np.ascontiguousarray(blobs[0], np.float32)
                self.lib.convb_pack_weights(st["geom"], bp, w.ctypes.data, wpb.ctypes.data)
                self.alloc.upload(st["wp"], wpb)
                if L.geom["bias_term"]:
                    self.alloc.upload(st["bias"], blobs[1])
                if "stemb_wp" in st:   # conv1 + pool1 as one launch (csrc/eco_stemb.hip)
                    swp = np.empty(self.lib.stemb_weight_elems(L.geom["cout"]), np.uint16)
                    self.lib.stemb_pack_weights(w.ctypes.data, L.geom["cout"], swp.ctypes.data)
                    self.alloc.upload(st["stemb_wp"], swp)
            elif L.type == "Convolution" and "ws2" in st:
                # (weights live in the group's transformed-domain image, _sync_ws2; the direct kernel's are not needed)
                if L.geom["bias_term"]:
                    self.alloc.upload(st["bias"], blobs[1])
            elif L.type == "Convolution":
                g: hip.ConvGeom = st["geom"]
                plan: hip.ConvPlan = st["plan"]
                wp = np.empty(plan.wp_elems, np.float32)
                kt = np.empty(plan.ktab_elems, np.int32)
                w = np.ascontiguousarray(blobs[0], np.float32)
                self.lib.conv_pack_weights(g, plan, w.ctypes.data, wp.ctypes.data, kt.ctypes.data)
                self.alloc.upload(st["wp"], wp)
                self.alloc.upload(st["ktab"], kt)
                if L.geom["bias_term"]:
                    self.alloc.upload(st["bias"], blobs[1])
                if "stem_wp" in st:   # conv1 + pool1 as one launch (csrc/eco_stem.hip)
                    swp = np.empty(74 * L.geom["cout"] * 2, np.float32)
                    self.lib.stem_pack_weights(w.ctypes.data, L.geom["cout"], swp.ctypes.data)
                    self.alloc.upload(st["stem_wp"], swp)
                wn = st.get("wino")
                if wn is not None and wn.get("kind") == "wgemm3":  # u[p] = ((G x G x G) g)[p], 216 points, K = cin
                    cout, cin = L.geom["cout"], L.geom["cin"]
                    u = np.empty((216, cout, cin, 1), np.float32)
                    self.lib.wino3_weight_transform(w.ctypes.data, cout, cin, u.ctypes.data)
                    up = np.zeros(wn["up_elems"], np.float32)
                    self.lib.wgemm_pack_weights(wn["plan"], u.ctypes.data, up.ctypes.data)
                    self.alloc.upload(wn["up"], up)
                elif wn is not None and wn.get("kind") == "wgemm":   # u[p] = (G g G^T)[p] packed for the dense GEMM kernel
                    cout, cin, kd = L.geom["cout"], L.geom["cin"], wn["kd"]
                    u = np.empty((36, cout, cin, kd), np.float32)
                    self.lib.wino_weight_transform(w.ctypes.data, cout, cin, kd, 4, u.ctypes.data)
                    up = np.zeros(wn["up_elems"], np.float32)
                    if wn["fused"]:
                        self.lib.wfused_pack_weights(wn["plan"], u.ctypes.data, up.ctypes.data)
                    else:
                        self.lib.wgemm_pack_weights(wn["plan"], u.ctypes.data, up.ctypes.data)
                    self.alloc.upload(wn["up"], up)
                elif wn is not None:  # u[p] = (G g G^T)[p], each point packed for the (kd,1,1) gather kernel
                    cout, cin, kd = L.geom["cout"], L.geom["cin"], wn["kd"]
                    P = wn["points"]
                    u = np.empty((P, cout, cin, kd), np.float32)
                    self.lib.wino_weight_transform(w.ctypes.data, cout, cin, kd, wn["M"], u.ctypes.data)
                    wps = np.empty((P, wn["plan"].wp_elems), np.float32)
                    ktw = np.empty(wn["plan"].ktab_elems, np.int32)
                    for pt in range(P):
                        self.lib.conv_pack_weights(wn["geom"], wn["plan"], u[pt].ctypes.data, wps[pt].ctypes.data,
                                                   ktw.ctypes.data)
                    self.alloc.upload(wn["wp"], wps)
                    self.alloc.upload(wn["ktab"], ktw)
            elif L.type == "BN":
                a, b = fold_bn(blobs, bn_eps(L))
                self.alloc.upload(st["scale"], a)
                self.alloc.upload(st["shift"], b)
            elif L.type == "InnerProduct":
                self.alloc.upload(st["w"], blobs[0])
                if L.geom["bias_term"]:
                    self.alloc.upload(st["bias"], blobs[1])
            self._dirty_params.discard(name)

    # ------------------------------------------------------------------ build
    def build(self) -> None:
        """Allocate blobs and record the launch list for the spec's current shapes."""
        spec = self.spec
        # Net::Reshape keeps blob contents (net.cpp:843-849, Blob::Reshape only reallocates when the count
        # grows): storage whose element count is unchanged is carried over to the new plan, so a redundant
        # net.reshape() neither loses resident inputs nor hands kernels uninitialised buffers.
        self._old_tensors = self.tensors
        self.tensors = {}
        self.fused_away = {}
        self.ops = []
        self._keep = []
        self._groups = {}
        self._dirty_groups = set()
        self.generation += 1
        # Split tops are names for their bottom (the fused plan's dataflow and the stride-2 Winograd groups work on real blobs)
        self._alias_src: Dict[str, str] = {}
        for L in spec.layers:
            if L.type == "Split":
                for t in L.tops:
                    self._alias_src[t] = self._resolve(L.bottoms[0])
        # device-side parameter storage (sizes depend on geometry)
        for L in spec.layers:
            st = self._param_dev.setdefault(L.name, {})
            shapes = param_shapes(L)
            if shapes and L.name in self.params:
                # a reshape may change a parameter's shape (e.g. fc K after a spatial reshape with a fixed
                # global_pool kernel); the reference CHECK-fails in LayerSetUp / Reshape, never reads past the blob
                have = [tuple(b.shape) for b in self.params[L.name]]
                want = [tuple(int(d) for d in sh) for sh in shapes]
                if [_prod(h) for h in have] != [_prod(w) for w in want]:
                    raise NetSpecError(f"layer {L.name}: parameter shapes {have} do not match the reshaped "
                                       f"geometry {want}")
            if L.type == "Convolution":
                g = L.geom
                geom = hip.conv_geom(L.bottom_shapes[0][0], g["cin"], g["cout"], L.bottom_shapes[0][2:], g["kernel"],
                                     g["stride"], g["pad"], L.top_shapes[0][2:])
                if self.dt:
                    self._plan_blocked_conv(L, st, geom)
                    continue
                plan = self.lib.conv_plan(geom, self.num_cu)
                old = st.get("plan")
                if old is None or (old.wp_elems, old.ktab_elems) != (plan.wp_elems, plan.ktab_elems):
                    st["wp"] = self.alloc.empty(plan.wp_elems, np.float32)
                    st["ktab"] = self.alloc.empty(plan.ktab_elems, np.int32)
                    if g["bias_term"]:
                        st["bias"] = self.alloc.empty(g["cout"], np.float32)
                st["geom"], st["plan"] = geom, plan
                st.pop("wino", None) if not self._wino_eligible(L) else self._plan_wino(L, st)
                if self._stem_geometry(L) and "stem_wp" not in st:
                    st["stem_wp"] = self.alloc.empty(74 * g["cout"] * 2, np.float32)
                self._dirty_params.add(L.name)  # the gather table depends on the input dims
            elif L.type == "BN" and st.get("size") != L.geom["channels"]:
                st["scale"] = self.alloc.empty(L.geom["channels"], np.float32)
                st["shift"] = self.alloc.empty(L.geom["channels"], np.float32)
                st["size"] = L.geom["channels"]
                self._dirty_params.add(L.name)
            elif L.type == "InnerProduct" and st.get("size") != (L.geom["num_output"], L.geom["K"]):
                st["w"] = self.alloc.empty(L.geom["num_output"] * L.geom["K"], np.float32)
                if L.geom["bias_term"]:
                    st["bias"] = self.alloc.empty(L.geom["num_output"], np.float32)
                st["size"] = (L.geom["num_output"], L.geom["K"])
                self._dirty_params.add(L.name)
        self._plan_ws2_groups()
        # one scratch buffer serves every split-K convolution (launches are serial on one stream); the same
        # goes for the Winograd path's transformed input / output volumes
        ws_bytes = max([st["plan"].ws_bytes for st in self._param_dev.values() if "plan" in st] +
                       [st["bplan"].ws_bytes for st in self._param_dev.values() if "bplan" in st] +
                       [st["wino"]["points"] * st["wino"]["plan"].ws_bytes for st in self._param_dev.values()
                        if "wino" in st and st["wino"]["kind"] == "gather"] + [0])
        for key in ("v_elems", "m_elems"):
            need = max([st["wino"][key] for st in self._param_dev.values() if "wino" in st] +
                       [getattr(grp["plan"], key) for grp in self._ws2_groups.values()] + [0])
            if need > getattr(self, "_wino_" + key, 0):
                setattr(self, "_wino_buf_" + key, self.alloc.empty(need, np.float32))
                setattr(self, "_wino_" + key, need)
        if ws_bytes > getattr(self, "_ws_bytes", 0):
            self._ws = self.alloc.empty((ws_bytes + 3) // 4, np.float32)
            self._ws_bytes = ws_bytes
        for n in spec.inputs:
            self._materialize(n, spec.blob_shapes[n], plain=True)
        if self.fuse:
            self._build_fused()
        else:
            for i, L in enumerate(spec.layers):
                self._emit_unfused(i, L)
        self._old_tensors = {}
        self._built = True

    # -- storage helpers -------------------------------------------------------
    def _materialize(self, name: str, shape, plain: bool = False) -> _Tensor:
        """Storage for blob `name`.  In the blocked modes every activation with a channel axis is channel-blocked
        in the path's storage type; `plain` keeps the reference's fp32 layout (net inputs, the logits)."""
        if name in self.tensors:
            t = self.tensors[name]
            if _prod(t.shape) != _prod(shape):
                raise NetSpecError(f"blob {name}: storage of {t.shape} reused for {shape}")
            return t
        dt = 0 if (plain or not self.dt) else self.dt
        if dt and (len(shape) < 2 or shape[1] % 8):
            raise NetSpecError(f"blob {name} {tuple(shape)}: the blocked path needs a channel count that is a multiple of 8")
        old = getattr(self, "_old_tensors", {}).get(name)
        if old is not None and old.owner == name and old.count == _prod(shape) and old.dt == dt:
            t = _Tensor(old.handle, shape, name, dt)   # same element count: contents survive the rebuild
        else:
            t = _Tensor(self.alloc.empty(_prod(shape), self.store_np if dt else np.float32), shape, name, dt)
        self.tensors[name] = t
        return t

    def _alias(self, name: str, src: str, shape) -> None:
        s = self.tensors[src]
        if s.dt and tuple(shape[:2]) != tuple(s.shape[:2]):
            raise NetSpecError(f"blob {name}: a Reshape that moves the channel axis of the channel-blocked blob {src} "
                               "is not available on the blocked path")
        self.tensors[name] = _Tensor(s.handle, shape, s.owner, s.dt)

    def _ptr(self, name: str, offset_elems: int = 0) -> int:
        if name not in self.tensors:
            why = self.fused_away.get(name, "not produced")
            raise KeyError(f"blob {name!r} is not materialised ({why}); build the net with fuse=False to observe it")
        t = self.tensors[name]
        return self.alloc.ptr(t.handle) + (self.esize if t.dt else 4) * int(offset_elems)

    def _view(self, name: str, channels: int, spatial: int, c0: int = 0, ctot: Optional[int] = None) -> "hip.View":
        """Dense [N, C, S] view of blob `name` (or of channels [c0, c0+channels) of its ctot channels) in the
        stride units of the active path: elements (fp32 path) or 8-channel blocks (blocked paths)."""
        ctot = channels if ctot is None else ctot
        if self.cblk > 1 and (c0 % self.cblk or ctot % self.cblk):
            raise NetSpecError(f"blob {name}: channel offset {c0} / count {ctot} is not a multiple of {self.cblk}")
        return hip.View(self._ptr(name, c0 * spatial), (ctot // self.cblk) * spatial, 0, spatial, 1)

    def _no_blocked(self, L: LayerSpec) -> None:
        raise NetSpecError(f"layer {L.name} ({L.type}) has no stand-alone kernel on the blocked {self.dtype} path: that "
                           "path runs fused ECO graphs (conv+BN+ReLU+Eltwise epilogues, Concat / Permute as views, "
                           "Pooling, the pool+fc tail); use dtype='f32' for this net")

    def _pdev(self, layer: str, key: str) -> int:
        return self.alloc.ptr(self._param_dev[layer][key])

    def _add(self, idx: int, label: str, fn: Callable[[Optional[int]], None], meta: Optional[dict] = None) -> None:
        self.ops.append((idx, label, fn, meta or {"kernel": label.split("[")[0], "flops": 0, "bytes": 0}))

    # ------------------------------------------------------------------ unfused
    def _emit_unfused(self, i: int, L: LayerSpec) -> None:
        lib = self.lib
        t = L.type
        bshape = L.bottom_shapes[0] if L.bottom_shapes else None
        if t in _ALIAS_TYPES:
            if L.bottoms[0] not in self.tensors:  # alias of a blob that lives only inside a fused group
                for top in L.tops:
                    self.fused_away[top] = self.fused_away.get(L.bottoms[0], "alias of a non-materialised blob")
                return
            for top, shp in zip(L.tops, L.top_shapes):
                if top != L.bottoms[0]:
                    self._alias(top, L.bottoms[0], shp)
                else:
                    self.tensors[top] = _Tensor(self.tensors[top].handle, shp, self.tensors[top].owner)
            return
        top = L.tops[0]
        inplace = top in L.bottoms
        if self.dt and t not in ("Convolution", "Pooling"):
            self._no_blocked(L)
        if not inplace:
            self._materialize(top, L.top_shapes[0])
        if t == "Convolution":
            st = self._param_dev[L.name]
            ep = hip.ConvEpilogue()
            ep.bias = self._pdev(L.name, "bias") if L.geom["bias_term"] else None
            ep.residual = hip.null_view()
            ep.raw = self._view(top, L.geom["cout"], _prod(L.top_shapes[0][2:]))
            ep.bn_scale = None
            ep.bn_shift = None
            ep.relu = 0
            ep.act = hip.null_view()
            self._emit_conv(i, L, ep, L.name)
        elif t == "BN":
            x, y = self._ptr(L.bottoms[0]), self._ptr(top)
            n, c, inner = bshape[0], bshape[1], _prod(bshape[2:])
            sc, sh = self._pdev(L.name, "scale"), self._pdev(L.name, "shift")
            self._add(i, L.name, lambda s, x=x, y=y, sc=sc, sh=sh, n=n, c=c, inner=inner:
                      lib.bn_forward(x, y, sc, sh, n, c, inner, 0, s))
        elif t == "ReLU":
            x, y, cnt, slope = self._ptr(L.bottoms[0]), self._ptr(top), _prod(bshape), L.geom["negative_slope"]
            self._add(i, L.name, lambda s, x=x, y=y, cnt=cnt, slope=slope: lib.relu_forward(x, y, cnt, slope, s))
        elif t == "Pooling":
            self._emit_pool(i, L, self._ptr(L.bottoms[0]), self._ptr(top))
        elif t == "Concat":
            self._emit_concat(i, L, skip=())
        elif t == "Eltwise":
            cf = L.geom["coeff"]
            y, cnt = self._ptr(top), _prod(bshape)
            a, b = self._ptr(L.bottoms[0]), self._ptr(L.bottoms[1])
            self._add(i, L.name, lambda s, a=a, b=b, y=y, cnt=cnt, ca=cf[0], cb=cf[1]:
                      lib.eltwise_sum_forward(a, b, y, cnt, ca, cb, s))
            for k in range(2, len(L.bottoms)):  # y += c_k * bottom_k
                bk = self._ptr(L.bottoms[k])
                self._add(i, L.name, lambda s, bk=bk, y=y, cnt=cnt, ck=cf[k]:
                          lib.eltwise_sum_forward(y, bk, y, cnt, 1.0, ck, s))
        elif t == "Permute":
            x, y = self._ptr(L.bottoms[0]), self._ptr(top)
            shp, order = list(bshape), list(L.geom["order"])
            if len(shp) > 6:
                raise NetSpecError(f"{L.name}: Permute of {len(shp)} axes unsupported")
            self._add(i, L.name, lambda s, x=x, y=y, shp=shp, order=order: lib.permute_forward(x, y, shp, order, s))
        elif t == "InnerProduct":
            g = L.geom
            x, y = self._ptr(L.bottoms[0]), self._ptr(top)
            w = self._pdev(L.name, "w")
            b = self._pdev(L.name, "bias") if g["bias_term"] else None
            self._add(i, L.name, lambda s, x=x, w=w, b=b, y=y, m=g["M"], n=g["num_output"], k=g["K"]:
                      lib.inner_product_forward(x, w, b, y, m, n, k, s))
        elif t == "Softmax":
            ax = L.geom["axis"]
            x, y = self._ptr(L.bottoms[0]), self._ptr(top)
            outer, c, inner = _prod(bshape[:ax]), bshape[ax], _prod(bshape[ax + 1:])
            self._add(i, L.name, lambda s, x=x, y=y, outer=outer, c=c, inner=inner:
                      lib.softmax_forward(x, y, outer, c, inner, s))
        elif t in ("Accuracy", "SoftmaxWithLoss"):
            g = L.geom
            x, lab, y = self._ptr(L.bottoms[0]), self._ptr(L.bottoms[1]), self._ptr(top)
            if t == "Accuracy":
                self._add(i, L.name, lambda s, x=x, lab=lab, y=y, g=g: lib.accuracy_forward(
                    x, lab, y, g["outer"], g["classes"], g["inner"], g["top_k"], g["ignore_label"], s))
            else:
                self._add(i, L.name, lambda s, x=x, lab=lab, y=y, g=g: lib.softmax_loss_forward(
                    x, lab, y, g["outer"], g["classes"], g["inner"], g["normalize"], g["ignore_label"], s))
                if len(L.tops) == 2:  # the softmax itself as second top
                    p = self._ptr(L.tops[1])
                    self._add(i, L.name, lambda s, x=x, p=p, g=g: lib.softmax_forward(
                        x, p, g["outer"], g["classes"], g["inner"], s))
        else:  # pragma: no cover
            raise NetSpecError(f"no HIP launcher for layer type {t}")

    # -- Winograd F(MxM,3x3) path, M = 4 by default (csrc/eco_wino.hip) ---------------------------------
    @staticmethod
    def _wino_dims(L: LayerSpec):
        """(n, D, H, W, kd) of a stride-1 pad-1 (3x)3x3 convolution's input, D = kd = 1 for 2-D blobs."""
        shp = L.bottom_shapes[0]
        return (shp[0], shp[2], shp[3], shp[4], 3) if len(shp) == 5 else (shp[0], 1, shp[2], shp[3], 1)

    def _wino_eligible(self, L: LayerSpec) -> bool:
        """Stride-1, pad-1 3x3x3 / 3x3 convolutions with at least 64 input channels (a multiple of 16) and
        cout <= 4*cin: every such conv of ECO-Lite / ECO-Full.  Measured against the direct span kernel (32
        clips): 3-D trunk 2.89 -> 1.18 ms (res3b), 2-D convs 64->64 0.32 -> 0.27, 96->96 0.58 -> 0.46, conv2_3x3
        (64->192, 56x56) 3.24 -> 2.33 ms -- even there, where the transformed output volume is 2.8 GB."""
        g = L.geom
        nd = len(L.bottom_shapes[0]) - 2
        if self.dt:
            return False
        if not (self.winograd and nd in (2, 3) and tuple(g["kernel"]) == (3,) * nd and
                tuple(g["stride"]) == (1,) * nd and tuple(g["pad"]) == (1,) * nd and g["cin"] % 16 == 0 and
                tuple(L.top_shapes[0][2:]) == tuple(L.bottom_shapes[0][2:])):
            return False
        if g["cin"] < 64 or g["cout"] > 4 * g["cin"]:
            return False
        # each transform point is a GEMM over n*D*ceil(H/M)*ceil(W/M) tile positions in 128/256-wide tiles: below
        # wino_min_tiles positions the layer is mostly tile padding and runs direct (split-K span kernel).  Measured
        # per step at 1 / 2 / 4 clips: threshold 256: 1.47 / 1.90 / 2.93 ms, 64: 1.43 / 1.90 / 2.86, 16: 1.57 / 2.00 /
        # 2.86; an explicit winograd=2/4 overrides the size rule
        if self.winograd is True:
            n, D, H, W, _ = self._wino_dims(L)
            return n * D * -(-H // 4) * -(-W // 4) >= self.wino_min_tiles
        return True

    def _wino3_pays(self, n: int, D: int, TH: int, TW: int) -> bool:
        """F(4x4x4,3x3x3) instead of F(4x4,3x3) + direct depth taps for a 3x3x3 layer on n x D x (4 TH) x (4 TW) volumes?"""
        TD = -(-D // 4)
        if not self.wino3 or 8 * TD > 3 * D:
            return False
        if self.winograd is True and n * TD * TH * TW < self.wino3_min_positions:   # an explicit winograd=4 overrides
            return False
        return self.lib.wino3_lds_bytes(n, TH, TW) <= 152 * 1024

    def _plan_wino(self, L: LayerSpec, st: dict) -> None:
        g = L.geom
        n, D, H, W, kd = self._wino_dims(L)
        # output tile: F(4x4,3x3) does 4x fewer multiplies than direct (F(2x2): 2.25x) and its transformed
        # volumes are 2.25x the activations (F(2x2): 4x); planes that do not tile by 4 pay ceil() padding
        M = self.winograd if self.winograd in (2, 4) else 4
        T = M + 2
        TH, TW = -(-H // M), -(-W // M)
        old = st.get("wino")
        if M == 4 and self.wgemm and kd == 3 and self._wino3_pays(n, D, TH, TW):
            # F(4x4x4,3x3x3): the same dense GEMM kernel on 216 points with K = cin; transforms in csrc/eco_wino3.hip
            TD = -(-D // 4)
            plan = self.lib.wgemm_plan(n, g["cin"], g["cout"], TD, TH, TW, 1, self.num_cu, points=216)
            wn = dict(kind="wgemm3", plan=plan, M=4, points=216, TD=TD, TH=TH, TW=TW, kd=kd, fused=False,
                      v_elems=plan.v_elems, m_elems=plan.m_elems, up_elems=plan.u_elems)
            if old is not None and old.get("kind") == "wgemm3" and old.get("up_elems") == plan.u_elems:
                wn["up"] = old["up"]
            else:
                wn["up"] = self.alloc.empty(plan.u_elems, np.float32)
            st["wino"] = wn
            return
        if M == 4 and self.wgemm:
            # F(4x4,3x3) on the dedicated dense GEMM (csrc/eco_wgemm.hip): pair-interleaved depth-major V, LDS-DMA staging
            plan = self.lib.wgemm_plan(n, g["cin"], g["cout"], D, TH, TW, kd, self.num_cu)
            wn = dict(kind="wgemm", plan=plan, M=4, points=36, TH=TH, TW=TW, kd=kd, v_elems=plan.v_elems, m_elems=plan.m_elems)
            # short-reduction 2-D layers (conv2_3x3, the inception 3x3 convs): GEMM and output transform in one
            # kernel, the transformed-domain products never leave LDS (csrc/eco_wgemm.hip, wfused_kernel)
            wn["fused"] = bool(self.wfused and kd == 1 and D == 1 and g["cin"] % 32 == 0 and
                               64 <= g["cin"] <= self.wfused_max_cin and g["cout"] % 32 == 0)
            elems = self.lib.wfused_weight_elems(plan) if wn["fused"] else plan.u_elems
            if wn["fused"]:
                # (no M in HBM; the pooling-fused form parks 9 floats per tile and channel in the same scratch buffer)
                wn["m_elems"] = self.lib.wfused_pool_scratch_elems(plan) if (self.wpool and H % 4 == 0 and W % 4 == 0) else 0
            if old is not None and old.get("kind") == "wgemm" and old.get("fused") == wn["fused"] and \
                    old.get("up_elems") == elems:
                wn["up"] = old["up"]
            else:
                wn["up"] = self.alloc.empty(elems, np.float32)
            wn["up_elems"] = elems
            st["wino"] = wn
            return
        gw = hip.conv_geom(n, g["cin"], g["cout"], (D, TH, TW), (kd, 1, 1), (1, 1, 1), (kd // 2, 0, 0), (D, TH, TW))
        plan = self.lib.conv_plan(gw, self.num_cu, batch=T * T)   # the T*T points share one launch
        wn = dict(kind="gather", geom=gw, plan=plan, M=M, points=T * T, TH=TH, TW=TW, kd=kd,
                  v_elems=T * T * n * g["cin"] * D * TH * TW, m_elems=T * T * n * g["cout"] * D * TH * TW)
        if old is not None and old.get("kind") == "gather" and (old["points"], old["plan"].wp_elems, old["plan"].ktab_elems) == \
                (T * T, plan.wp_elems, plan.ktab_elems):
            wn["wp"], wn["ktab"] = old["wp"], old["ktab"]
        else:
            wn["wp"] = self.alloc.empty(T * T * plan.wp_elems, np.float32)
            wn["ktab"] = self.alloc.empty(plan.ktab_elems, np.int32)
        st["wino"] = wn

    def _emit_wino_conv(self, i: int, L: LayerSpec, ep: "hip.ConvEpilogue", label: str, nbytes: int) -> None:
        st = self._param_dev[L.name]
        wn = st["wino"]
        lib = self.lib
        n, D, H, W, kd = self._wino_dims(L)
        cin, cout = L.geom["cin"], L.geom["cout"]
        x = self._ptr(L.bottoms[0])
        v = self.alloc.ptr(self._wino_buf_v_elems)
        m = self.alloc.ptr(self._wino_buf_m_elems) if getattr(self, "_wino_m_elems", 0) else None
        if wn["kind"] == "wgemm3":
            plan, up = wn["plan"], self.alloc.ptr(wn["up"])
            self._keep.append((plan, ep))
            pos = n * wn["TD"] * wn["TH"] * wn["TW"]                # positions per transform point
            v_bytes, m_bytes = 4 * 216 * cin * pos, 4 * 216 * plan.ksplit * cout * pos
            x_bytes, w_bytes = 4 * n * cin * D * H * W, 4 * 27 * cin * cout
            tag = "F(4x4x4,3x3x3)"
            self._add(i, f"{label} [winograd {tag} input transform]", lambda s, plan=plan, x=x, v=v, D=D, H=H, W=W:
                      lib.wino3_input_forward(plan, x, v, D, H, W, s),
                      {"kernel": "eco::wino3_input_kernel", "flops": 0, "bytes": x_bytes + v_bytes})
            self._add(i, f"{label} [216 transformed-domain GEMMs, K = {cin}]", lambda s, plan=plan, v=v, up=up, m=m:
                      lib.wgemm_forward(plan, v, up, m, s),
                      {"kernel": hip.wgemm_kernel_name(plan), "flops": 2 * 216 * pos * cout * cin,
                       "useful_flops": 2 * 216 * (n * D * H * W / 64.0) * cout * cin,
                       "bytes": v_bytes + 4 * 216 * cout * cin + m_bytes})
            self._add(i, f"{label} [winograd {tag} output transform]", lambda s, plan=plan, m=m, D=D, H=H, W=W, ep=ep:
                      lib.wino3_output_forward(plan, m, D, H, W, ep, s),
                      {"kernel": "eco::wino3_output_kernel", "flops": 0, "bytes": m_bytes + nbytes - x_bytes - w_bytes})
            return
        if wn["kind"] == "wgemm":
            plan, up = wn["plan"], self.alloc.ptr(wn["up"])
            self._keep.append((plan, ep))
            tiles = n * D * wn["TH"] * wn["TW"]                     # positions per transform point
            v_bytes = 4 * 36 * cin * (D + 2 * (kd // 2)) * n * wn["TH"] * wn["TW"]
            m_bytes = 4 * 36 * plan.ksplit * cout * tiles
            tag = "F(4x4,3x3)"
            if wn["fused"]:
                self._add(i, f"{label} [winograd {tag} input transform]", lambda s, plan=plan, x=x, v=v, H=H, W=W:
                          lib.wino_input_q4_forward(plan, x, v, H, W, s),
                          {"kernel": "eco::wino_input_q4_kernel", "flops": 0, "bytes": 4 * n * cin * D * H * W + v_bytes})
            else:
                self._add(i, f"{label} [winograd {tag} input transform]", lambda s, plan=plan, x=x, v=v, H=H, W=W:
                          lib.wino_input_pk_forward(plan, x, v, H, W, s),
                          {"kernel": "eco::wino_input_pk_kernel", "flops": 0, "bytes": 4 * n * cin * D * H * W + v_bytes})
            if wn["fused"]:
                self._add(i, f"{label} [36 transformed-domain GEMMs, K = {cin}, + winograd {tag} output transform]",
                          lambda s, plan=plan, v=v, up=up, H=H, W=W, ep=ep: lib.wfused_forward(plan, v, up, H, W, ep, s),
                          {"kernel": "eco::wfused_kernel", "flops": 2 * 36 * tiles * cout * cin,
                           "useful_flops": 2 * 36 * (n * D * H * W / 16.0) * cout * cin,   # without the tiles' overhang
                           "bytes": v_bytes + 4 * 36 * cout * cin + nbytes - 4 * (n * cin * D * H * W + 9 * cin * cout)})
                return
            self._add(i, f"{label} [36 transformed-domain GEMMs, K = {cin * kd}]", lambda s, plan=plan, v=v, up=up, m=m:
                      lib.wgemm_forward(plan, v, up, m, s),
                      {"kernel": hip.wgemm_kernel_name(plan), "flops": 2 * 36 * tiles * cout * cin * kd,
                       "useful_flops": 2 * 36 * (n * D * H * W / 16.0) * cout * cin * kd,
                       "bytes": v_bytes + 4 * 36 * cout * cin * kd + m_bytes})
            self._add(i, f"{label} [winograd {tag} output transform]", lambda s, plan=plan, m=m, H=H, W=W, ep=ep:
                      lib.wino_output_dm_forward(plan, m, H, W, ep, s),
                      {"kernel": "eco::wino_output_dm_kernel", "flops": 0,
                       "bytes": m_bytes + nbytes - 4 * (n * cin * D * H * W + 9 * kd * cin * cout)})
            return
        gw, plan, M, P = wn["geom"], wn["plan"], wn["M"], wn["points"]
        wp, kt = self.alloc.ptr(wn["wp"]), self.alloc.ptr(wn["ktab"])
        ws = self.alloc.ptr(self._ws) if plan.ws_bytes else None
        tin, tout = wn["v_elems"] // P, wn["m_elems"] // P
        epg = hip.ConvEpilogue()
        epg.bias = None
        epg.residual, epg.act = hip.null_view(), hip.null_view()
        epg.bn_scale = epg.bn_shift = None
        epg.relu = 0
        epg.raw = hip.plain_view(m, cout, D * wn["TH"] * wn["TW"])
        self._keep.append((gw, plan, ep, epg))
        tag = f"F({M}x{M},3x3)"
        self._add(i, f"{label} [winograd {tag} input transform]", lambda s, x=x, v=v, pl=n * cin * D, H=H, W=W, M=M:
                  lib.wino_input_forward(x, v, pl, H, W, M, s),
                  {"kernel": f"eco::wino_input_kernel<{M}>", "flops": 0, "bytes": 4 * (n * cin * D * H * W + P * tin)})
        self._add(i, f"{label} [{P} transformed ({kd},1,1) convs]",
                  lambda s, gw=gw, plan=plan, v=v, wp=wp, kt=kt, epg=epg, ws=ws, tin=tin, tout=tout, P=P:
                  lib.conv_forward_batched(gw, plan, v, wp, kt, epg, ws, P, tin, plan.wp_elems, tout, s),
                  {"kernel": hip.conv_kernel_name(plan), "flops": 2 * P * tout * cin * kd,
                   "bytes": 4 * (P * tin + P * cout * cin * kd + P * tout)})
        self._add(i, f"{label} [winograd {tag} output transform]",
                  lambda s, m=m, n=n, cout=cout, D=D, H=H, W=W, M=M, ep=ep: lib.wino_output_forward(m, n, cout, D, H, W, M, ep, s),
                  {"kernel": f"eco::wino_output_kernel<{M}>", "flops": 0,
                   "bytes": 4 * P * tout + nbytes - 4 * (n * cin * D * H * W + 9 * kd * cin * cout)})

    # -- stride-2 3x3x3 convolutions on the polyphase Winograd route (csrc/eco_wino_s2.hip) ----------------------------
    def _ws2_form(self, L: LayerSpec):
        """Which polyphase form runs a stride-2 3x3(x3) convolution, or None (the direct strided kernel):
          ("3d", TD, TH, TW)      3x3x3, stride 2, pad 1 on an input exactly twice the output volume, the output volume tiling by
                                  4 x 7 x 7 with enough tiles: F(4,2) x F(7,2) x F(7,2), 320 points, K = 8 cin (res4a_1 / res4a_down
                                  at num_segments 16 / 32: 8 x 14 x 14 outputs, models_ECO_Lite/kinetics/deploy.prototxt:1262-1330)
          ("2d", kz, Do, TH, TW)  output planes tiling by 7 x 7: F(7,2) x F(7,2), 64 points, every output plane a position, the depth
                                  taps in the reduction (K = 4 kz cin): res5a_1 / res5a_down (4 x 7 x 7 outputs: kz = 3) and the
                                  strided 2-D 3x3 convs of ECO-Full's inception_3c / 4e (kz = 1), where the estimate below says
                                  the transforms pay."""
        g = L.geom
        if self.dt or not self.winograd or not self.wino_s2 or not self.wgemm or L.type != "Convolution" or g.get("group", 1) != 1:
            return None
        nd = len(L.bottom_shapes[0]) - 2
        if nd not in (2, 3) or tuple(g["kernel"])[-2:] != (3, 3) or tuple(g["stride"])[-2:] != (2, 2) or tuple(g["pad"])[-2:] != (1, 1):
            return None
        n, cin = L.bottom_shapes[0][:2]
        H, W = L.bottom_shapes[0][-2:]
        Ho, Wo = L.top_shapes[0][-2:]
        if (H, W) != (2 * Ho, 2 * Wo) or Ho % 7 or Wo % 7 or cin % 4 or cin < 16:
            return None
        TH, TW = Ho // 7, Wo // 7
        forced = self.winograd is not True                      # an explicit winograd=4 overrides the size rules
        if nd == 3 and (g["kernel"][0], g["stride"][0], g["pad"][0]) == (3, 2, 1):
            D, Do = L.bottom_shapes[0][2], L.top_shapes[0][2]
            if D != 2 * Do:
                return None
            if Do % 4 == 0 and (forced or n * (Do // 4) * TH * TW >= self.wino_s2_min_positions) and \
                    self.lib.wino_s2_lds_bytes(n, Do // 4, TH, TW) <= 152 * 1024:
                return ("3d", Do // 4, TH, TW)
            kz = 3
        elif nd == 2 or (g["kernel"][0], g["stride"][0], g["pad"][0]) == (1, 1, 0):
            Do, kz = (1 if nd == 2 else L.top_shapes[0][2]), 1
        else:
            return None
        if not forced:
            if n * Do * TH * TW < self.wino_s2_min_positions:
                return None
            # do the transforms pay?  direct: 27 / 9 multiplies per output and input channel on the gather kernel (0.55 of the
            # MFMA peak measured on these layers); here 64 * 4 kz / 49 of them on the dense GEMM (0.65) + V, M and the blobs
            # through the transform kernels (4.5 TB/s measured).  Small output-channel counts lose: inception_3c_double_3x3_2
            cout = g["cout"]
            outs = n * Do * Ho * Wo
            t_direct = 2.0 * outs * cout * cin * 9 * kz / (0.55 * 157.3e12)
            pos = n * Do * TH * TW
            t_ws = 2.0 * 64 * pos * cout * 4 * kz * cin / (0.65 * 157.3e12) + \
                4.0 * (_prod(L.bottom_shapes[0]) + 64 * pos * (4 * kz * cin + cout) + outs * cout) / 4.5e12
            if t_direct < 1.15 * t_ws:
                return None
        if self.lib.wino_s2d_lds_bytes(n, kz, Do, TH, TW) > 152 * 1024:
            return None
        return ("2d", kz, Do, TH, TW)

    def _plan_ws2_groups(self) -> None:
        """Group the eligible convs by (bottom blob, geometry) -- in the fused plan a residual block's first conv and its
        projection shortcut become ONE transformed-domain problem of cout_1 + cout_2 rows -- and plan each group's GEMM."""
        layers = self.spec.layers
        old = self._ws2_groups
        self._ws2_groups = {}
        for L in layers:
            self._param_dev.get(L.name, {}).pop("ws2", None)
        found: Dict[tuple, List[int]] = {}
        for i, L in enumerate(layers):
            form = self._ws2_form(L) if L.type == "Convolution" and "wino" not in self._param_dev[L.name] else None
            if form is not None:
                k = (self._resolve(L.bottoms[0]), tuple(L.bottom_shapes[0]), tuple(L.top_shapes[0][2:]), form)
                k = k if self.fuse else k + (i,)
                # a layer that rewrites the shared bottom in place between two members splits the group
                if k in found and any(layers[j].inplace and self._resolve(layers[j].bottoms[0]) == k[0]
                                      for j in range(found[k][0], i)):
                    k = k + (i,)
                found.setdefault(k, []).append(i)
        for k, idxs in found.items():
            Ls = [layers[j] for j in idxs]
            key = "|".join(Lc.name for Lc in Ls)
            n, cin = Ls[0].bottom_shapes[0][:2]
            couts = [Lc.geom["cout"] for Lc in Ls]
            form = k[3]
            if form[0] == "3d":
                plan = self.lib.wgemm_plan(n, 8 * cin, sum(couts), form[1], form[2], form[3], 1, self.num_cu, points=320)
            else:
                plan = self.lib.wgemm_plan(n, 4 * form[1] * cin, sum(couts), form[2], form[3], form[4], 1, self.num_cu, points=64)
            grp = dict(plan=plan, convs=[Lc.name for Lc in Ls], idxs=idxs, couts=couts, cin=cin, form=form)
            o = old.get(key)
            grp["up"] = o["up"] if o is not None and o["plan"].u_elems == plan.u_elems else self.alloc.empty(plan.u_elems, np.float32)
            self._ws2_groups[key] = grp
            for Lc in Ls:
                self._param_dev[Lc.name]["ws2"] = key
                self._dirty_params.add(Lc.name)

    def _sync_ws2(self, dirty) -> None:
        """(Re)build the transformed-domain weight image of the groups with a changed member: the members' weights
        concatenated along cout -> u[320][ctot][8 cin] -> the GEMM kernel's packed layout."""
        for key, grp in self._ws2_groups.items():
            if not (dirty & set(grp["convs"])) or any(n not in self.params for n in grp["convs"]):
                continue
            plan, cin = grp["plan"], grp["cin"]
            ctot = sum(grp["couts"])
            w = np.ascontiguousarray(np.concatenate(
                [np.asarray(self.params[n][0], np.float32).reshape(c, -1) for n, c in zip(grp["convs"], grp["couts"])], 0))
            if grp["form"][0] == "3d":
                u = np.empty(320 * ctot * 8 * cin, np.float32)
                self.lib.wino_s2_weight_transform(w.ctypes.data, ctot, cin, u.ctypes.data)
            else:
                kz = grp["form"][1]
                u = np.empty(64 * ctot * 4 * kz * cin, np.float32)
                self.lib.wino_s2d_weight_transform(w.ctypes.data, ctot, cin, kz, u.ctypes.data)
            up = np.empty(plan.u_elems, np.float32)
            self.lib.wgemm_pack_weights(plan, u.ctypes.data, up.ctypes.data)
            del u
            self.alloc.upload(grp["up"], up)

    def _try_fuse_ws2(self, i, L, layers, consumers, outputs, sole_consumer, bn_relu_after, absorbed, concat_skip) -> bool:
        """The first member of a stride-2 Winograd group emits the whole group here (every member's only input is the shared
        bottom): one input transform, one GEMM, one output transform per member with the member's own epilogue.  A
        projection shortcut emitted ahead of the block's second conv keeps its raw value; its Eltwise then rides on that
        conv (the later producer), exactly as for the sibling launches of the direct kernels."""
        key = self._param_dev[L.name].get("ws2")
        if key is None:
            return False
        grp = self._ws2_groups[key]
        if grp["idxs"][0] != i:
            return False
        if any(j in absorbed for j in grp["idxs"]):   # a member already runs inside another group: everyone back to the direct kernel
            for name in grp["convs"]:
                self._param_dev[name].pop("ws2", None)
                self._dirty_params.add(name)
            del self._ws2_groups[key]
            return False
        members = []
        for j in grp["idxs"]:
            Lj = layers[j]
            self._emit_pos = i
            try:
                ep, label = self._conv_epilogue(j, Lj, layers, consumers, outputs, sole_consumer, bn_relu_after, absorbed,
                                                concat_skip)
            finally:
                self._emit_pos = None
            if j != i:
                absorbed[j] = L.name
            members.append((Lj, ep, label))
        self._emit_ws2(i, key, members)
        return True

    def _emit_ws2(self, i: int, key: str, members) -> None:
        grp = self._ws2_groups[key]
        plan, up = grp["plan"], self.alloc.ptr(grp["up"])
        lib = self.lib
        L0 = members[0][0]
        form = grp["form"]
        n, cin = L0.bottom_shapes[0][:2]
        D = L0.bottom_shapes[0][2] if len(L0.bottom_shapes[0]) == 5 else 1
        H, W = L0.bottom_shapes[0][-2:]
        Do = L0.top_shapes[0][2] if len(L0.top_shapes[0]) == 5 else 1
        Ho, Wo = L0.top_shapes[0][-2:]
        ctot = sum(grp["couts"])
        x = self._ptr(L0.bottoms[0])
        v = self.alloc.ptr(self._wino_buf_v_elems)
        m = self.alloc.ptr(self._wino_buf_m_elems)
        pos = n * plan.d * plan.th * plan.tw                       # positions per transform point
        S = Do * Ho * Wo
        P, K = plan.points, plan.cin
        v_bytes, m_bytes = 4 * P * K * pos, 4 * P * plan.ksplit * ctot * pos
        x_bytes = 4 * n * cin * D * H * W
        names = " | ".join(lb for _, _, lb in members)
        self._keep.append(plan)
        if form[0] == "3d":
            tag = "stride-2 winograd F(4,2)xF(7,2)xF(7,2)"
            k_in, k_out = "eco::wino_s2_input_kernel", "eco::wino_s2_output_kernel<true>"
            fin = lambda s: lib.wino_s2_input_forward(plan, x, v, D, H, W, s)
            fout = lib.wino_s2_output_forward
        else:
            kz = form[1]
            tag = "stride-2 winograd F(7,2)xF(7,2)" + (", depth taps direct" if kz == 3 else "")
            k_in, k_out = "eco::wino_s2d_input_kernel", "eco::wino_s2_output_kernel<false>"
            fin = lambda s: lib.wino_s2d_input_forward(plan, x, v, kz, D, H, W, s)
            fout = lib.wino_s2d_output_forward
        self._add(i, f"{names} [{tag} input transform]", fin, {"kernel": k_in, "flops": 0, "bytes": x_bytes + v_bytes})
        self._add(i, f"{names} [{P} transformed-domain GEMMs, K = {K}]", lambda s: lib.wgemm_forward(plan, v, up, m, s),
                  {"kernel": hip.wgemm_kernel_name(plan), "flops": 2 * P * pos * ctot * K,
                   "bytes": v_bytes + 4 * P * ctot * K + m_bytes, "siblings": len(members)})
        for (Lj, ep, label), c0 in zip(members, np.cumsum([0] + grp["couts"][:-1])):
            cout = Lj.geom["cout"]
            self._keep.append(ep)
            touched = bool(ep.raw.ptr) + bool(ep.act.ptr) + bool(ep.residual.ptr) + bool(ep.act2.ptr)
            self._add(i, f"{label} [{tag} output transform]",
                      lambda s, c0=int(c0), cout=cout, ep=ep: fout(plan, m, c0, cout, Do, Ho, Wo, ep, s),
                      {"kernel": k_out, "flops": 0, "bytes": m_bytes * cout // ctot + 4 * n * cout * S * touched})

    # -- the stem: conv1_7x7_s2 + BN + ReLU + pool1_3x3_s2 as one launch (csrc/eco_stem.hip) --
    def _stem_geometry(self, L: LayerSpec) -> bool:
        g = L.geom
        return (self.fuse and not self.dt and self.stem and len(L.bottom_shapes[0]) == 4 and g["cin"] == 3 and
                g["cout"] in (32, 64) and list(g["kernel"]) == [7, 7] and list(g["stride"]) == [2, 2] and
                list(g["pad"]) == [3, 3] and min(L.top_shapes[0][2:]) >= 3)

    def _stemb_geometry(self, L: LayerSpec) -> bool:
        """The same stem on the blocked bf16 path (csrc/eco_stemb.hip)."""
        g = L.geom
        return (self.fuse and self.dt == hip.DT_BF16 and self.stem and len(L.bottom_shapes[0]) == 4 and g["cin"] == 3 and
                g["cout"] in (32, 64) and list(g["kernel"]) == [7, 7] and list(g["stride"]) == [2, 2] and
                list(g["pad"]) == [3, 3] and min(L.top_shapes[0][2:]) >= 3)

    def _try_fuse_stem(self, i, L, ep, act_blob, label, layers, consumers, outputs, absorbed) -> bool:
        """conv (stem geometry) + BN + ReLU whose activated blob feeds only a MAX 3x3 stride-2 unpadded Pooling:
        one launch writes the pooled blob; the conv's own output is never stored."""
        blocked_stem = self._stemb_geometry(L)
        if act_blob is None or act_blob in outputs or ep.raw.ptr or ep.residual.ptr or \
                not (blocked_stem or self._stem_geometry(L)):
            return False
        cs = [c for c in consumers.get(act_blob, []) if absorbed.get(c) != L.name]
        if len(cs) != 1 or layers[cs[0]].type != "Pooling":
            return False
        Lp = layers[cs[0]]
        gp = Lp.geom
        if gp["method"] != "MAX" or list(gp["kernel"]) != [3, 3] or list(gp["stride"]) != [2, 2] or any(gp["pad"]):
            return False
        st = self._param_dev[L.name]
        n, _, H, W = L.bottom_shapes[0]
        cout = L.geom["cout"]
        self._materialize(Lp.tops[0], Lp.top_shapes[0])
        absorbed[cs[0]] = L.name
        self.fused_away[act_blob] = f"only exists inside the fused stem launch {L.name}+{Lp.name}"
        x, y = self._ptr(L.bottoms[0]), self._ptr(Lp.tops[0])
        bias, sc, sh, relu = ep.bias, ep.bn_scale, ep.bn_shift, ep.relu
        lib = self.lib
        n_conv = _prod(L.top_shapes[0])
        if blocked_stem:   # fp32 frames in, pooled blob out in the blocked bf16 layout
            if self.tensors[L.bottoms[0]].dt:
                raise NetSpecError(f"{L.name}: the 3-channel stem reads the fp32 frames, got a blocked blob")
            wp = self.alloc.ptr(st["stemb_wp"])
            self._add(i, f"{label}+{Lp.name}", lambda s: lib.stemb_forward(x, wp, bias, sc, sh, relu, y, n, H, W, cout, s),
                      {"kernel": "eco::stemb_kernel", "flops": 2 * n_conv * 147,
                       "bytes": 4 * _prod(L.bottom_shapes[0]) + 2 * (147 * cout + _prod(Lp.top_shapes[0]))})
            return True
        wp = self.alloc.ptr(st["stem_wp"])
        self._add(i, f"{label}+{Lp.name}", lambda s: lib.stem_forward(x, wp, bias, sc, sh, relu, y, n, H, W, cout, s),
                  {"kernel": "eco::stem_kernel", "flops": 2 * n_conv * 147,
                   "bytes": 4 * (_prod(L.bottom_shapes[0]) + 147 * cout + _prod(Lp.top_shapes[0]))})
        return True

    def _try_fuse_wpool(self, i, L, ep, act_blob, label, layers, consumers, outputs, absorbed) -> bool:
        """A stride-1 3x3 2-D conv on the fused Winograd kernel (+ BN + ReLU) whose activated blob feeds ONLY a MAX 3x3
        stride-2 unpadded Pooling, on planes that tile by 4 -- conv2_3x3 -> pool2 (deploy.prototxt:103-128): the kernel
        stores the nine partial window maxima of every 4x4 tile (9 floats instead of 16) and a small second launch folds
        neighbouring tiles into the pooled blob; the 1.2 GB conv output (32 clips) is never written or read back."""
        st = self._param_dev[L.name]
        wn = st.get("wino")
        if not self.wpool or act_blob is None or act_blob in outputs or ep.raw.ptr or ep.residual.ptr or self.dt or \
                wn is None or wn.get("kind") != "wgemm" or not wn.get("fused") or wn.get("m_elems", 0) <= 0:
            return False   # (m_elems: the partial-maxima scratch _plan_wino sized for this form; 0 = planned without it)
        cs = [c for c in consumers.get(act_blob, []) if absorbed.get(c) != L.name]
        if len(cs) != 1 or layers[cs[0]].type != "Pooling":
            return False
        Lp = layers[cs[0]]
        gp = Lp.geom
        n, cin, H, W = L.bottom_shapes[0]
        if gp["method"] != "MAX" or list(gp["kernel"]) != [3, 3] or list(gp["stride"]) != [2, 2] or any(gp["pad"]) or \
                H % 4 or W % 4 or list(Lp.top_shapes[0][2:]) != [H // 2, W // 2]:
            return False
        cout = L.geom["cout"]
        plan, up = wn["plan"], self.alloc.ptr(wn["up"])
        lib = self.lib
        self._materialize(Lp.tops[0], Lp.top_shapes[0])
        absorbed[cs[0]] = L.name
        self.fused_away[act_blob] = f"only exists inside the fused launch pair {L.name}+{Lp.name}"
        x, y = self._ptr(L.bottoms[0]), self._ptr(Lp.tops[0])
        v = self.alloc.ptr(self._wino_buf_v_elems)
        scratch = self.alloc.ptr(self._wino_buf_m_elems)
        self._keep.append((plan, ep))
        tiles = n * wn["TH"] * wn["TW"]
        v_bytes = 4 * 36 * cin * tiles
        p9_bytes = 4 * 9 * cout * tiles
        self._add(i, f"{label} [winograd F(4x4,3x3) input transform]", lambda s: lib.wino_input_q4_forward(plan, x, v, H, W, s),
                  {"kernel": "eco::wino_input_q4_kernel", "flops": 0, "bytes": 4 * n * cin * H * W + v_bytes})
        self._add(i, f"{label}+{Lp.name} [36 transformed-domain GEMMs, K = {cin}, + winograd F(4x4,3x3) output transform "
                     f"+ partial window maxima]",
                  lambda s: lib.wfused_pool_forward(plan, v, up, H, W, ep, scratch, y, s),
                  {"kernel": "eco::wfused_kernel", "flops": 2 * 36 * tiles * cout * cin,
                   "useful_flops": 2 * 36 * (n * H * W / 16.0) * cout * cin,
                   # (the launch pair: V and the weights in, the pooled blob out; the partial maxima's round trip is extra)
                   "bytes": v_bytes + 4 * 36 * cout * cin + 4 * _prod(Lp.top_shapes[0]), "scratch_bytes": 2 * p9_bytes})
        return True

    # -- blocked bf16-MFMA path (csrc/eco_blocked.hip) --------------------------------
    def _plan_blocked_conv(self, L: LayerSpec, st: dict, geom) -> None:
        g = L.geom
        bp = self.lib.convb_plan(geom, self.dt, self.num_cu)
        old = st.get("bplan")
        if old is None or old.wp_vecs != bp.wp_vecs:
            st["wp"] = self.alloc.empty(bp.wp_vecs * 8, np.uint16)
            if g["bias_term"]:
                st["bias"] = self.alloc.empty(g["cout"], np.float32)
        if bp.stem and self._stemb_geometry(L):   # conv1 + pool1 as one launch (csrc/eco_stemb.hip)
            if "stemb_wp" not in st:
                st["stemb_wp"] = self.alloc.empty(self.lib.stemb_weight_elems(g["cout"]), np.uint16)
        elif bp.stem:
            self._stem_pack_buffer(L, st)
        st["geom"], st["bplan"] = geom, bp
        st.pop("plan", None)
        st.pop("wino", None)
        self._dirty_params.add(L.name)

    def _stem_pack_buffer(self, L: LayerSpec, st: dict) -> None:
        """Zero-padded pixel-interleaved copy of the fp32 frames (eco_stem_pack_forward) for the unfused blocked stem."""
        n, _, H, W = L.bottom_shapes[0]
        need = n * (H + 6) * (W + 8) * 4
        if st.get("stem_elems") != need:
            st["stem_buf"] = self.alloc.empty(need, self.store_np)
            st["stem_elems"] = need

    def _emit_blocked_conv(self, i: int, L: LayerSpec, ep: "hip.ConvEpilogue", label: str,
                           src_name: Optional[str] = None) -> None:
        st = self._param_dev[L.name]
        g, bp = st["geom"], st["bplan"]
        lib, dt = self.lib, self.dt
        in_name = src_name if src_name is not None else L.bottoms[0]   # (src_name: a conv that runs ahead of its AVE pool)
        src = self.tensors[in_name]
        wp = self.alloc.ptr(st["wp"])
        ws = self.alloc.ptr(self._ws) if bp.ws_bytes else None
        self._keep.append((g, bp, ep))
        n_out = _prod(L.top_shapes[0])
        k = L.geom["cin"] * _prod(L.geom["kernel"])
        es = self.esize
        outs = bool(ep.raw.ptr) + bool(ep.act.ptr) + bool(ep.residual.ptr)
        if bp.stem:
            if src.dt:
                raise NetSpecError(f"{L.name}: the 3-channel stem reads the fp32 frames, got a blocked blob")
            n, _, H, W = L.bottom_shapes[0]
            self._stem_pack_buffer(L, st)   # (a stem whose pool could not be fused gets its buffer here)
            x, buf = self._ptr(L.bottoms[0]), self.alloc.ptr(st["stem_buf"])
            self._add(i, f"{label} [stem pack]", lambda s, x=x, buf=buf, n=n, H=H, W=W: lib.stem_pack_forward(x, buf, n, H, W, dt, s),
                      {"kernel": "eco::stem_pack_kernel", "flops": 0, "bytes": 4 * n * 3 * H * W + es * st["stem_elems"]})
            self._add(i, label, lambda s, g=g, bp=bp, buf=buf, wp=wp, ep=ep, ws=ws: lib.convb_forward(g, bp, buf, wp, ep, ws, s),
                      {"kernel": hip.convb_kernel_name(bp), "flops": 2 * n_out * k,
                       "bytes": es * st["stem_elems"] + 2 * k * L.geom["cout"] + es * n_out * outs})
            return
        if not src.dt:
            raise NetSpecError(f"{L.name}: input blob {in_name} is not channel-blocked")
        x = self._ptr(in_name)
        self._add(i, label, lambda s, g=g, bp=bp, x=x, wp=wp, ep=ep, ws=ws: lib.convb_forward(g, bp, x, wp, ep, ws, s),
                  {"kernel": hip.convb_kernel_name(bp), "flops": 2 * n_out * k,
                   "bytes": es * _prod(L.bottom_shapes[0]) + 2 * k * L.geom["cout"] + es * n_out * outs})

    def _emit_conv(self, i: int, L: LayerSpec, ep: "hip.ConvEpilogue", label: str, src: Optional[str] = None) -> None:
        if self.dt:
            self._emit_blocked_conv(i, L, ep, label, src_name=src)
            return
        st = self._param_dev[L.name]
        g, plan = st["geom"], st["plan"]
        x = self._ptr(src if src is not None else L.bottoms[0])   # (src: a conv that runs ahead of its AVE pool)
        wp, kt = self.alloc.ptr(st["wp"]), self.alloc.ptr(st["ktab"])
        self._keep.append((g, plan, ep))
        lib = self.lib
        ws = self.alloc.ptr(self._ws) if plan.ws_bytes else None
        n_out = _prod(L.top_shapes[0])
        k = L.geom["cin"] * _prod(L.geom["kernel"])
        # algorithmic bytes (fused model): input + weights + each tensor the epilogue touches, once
        nbytes = 4 * (_prod(L.bottom_shapes[0]) + k * L.geom["cout"]
                      + n_out * (bool(ep.raw.ptr) + bool(ep.act.ptr) + bool(ep.residual.ptr)))
        if "wino" in st:
            self._emit_wino_conv(i, L, ep, label, nbytes)
            return
        if "ws2" in st and src is None and len(self._ws2_groups[st["ws2"]]["convs"]) == 1:
            self._emit_ws2(i, st["ws2"], [(L, ep, label)])
            return
        meta = {"kernel": hip.conv_kernel_name(plan), "flops": 2 * n_out * k, "bytes": nbytes}
        self._add(i, label, lambda s, g=g, plan=plan, x=x, wp=wp, kt=kt, ep=ep, ws=ws:
                  lib.conv_forward(g, plan, x, wp, kt, ep, ws, s), meta)

    def _emit_pool(self, i: int, L: LayerSpec, x: int, y: int) -> None:
        g = L.geom
        b = L.bottom_shapes[0]
        pg = hip.pool_geom(b[0], b[1], b[2:], g["kernel"], g["stride"], g["pad"], L.top_shapes[0][2:], g["method"])
        self._keep.append(pg)
        lib = self.lib
        if self.dt:
            dt = self.dt
            if not self.tensors[L.bottoms[0]].dt:
                raise NetSpecError(f"{L.name}: input blob {L.bottoms[0]} is not channel-blocked")
            self._add(i, L.name, lambda s, pg=pg, x=x, y=y: lib.poolb_forward(pg, dt, x, y, s),
                      {"kernel": "eco::poolb_k3_kernel" if len(b) == 4 and list(g["kernel"]) == [3, 3] else "eco::poolb_kernel",
                       "flops": 0, "bytes": self.esize * (_prod(b) + _prod(L.top_shapes[0]))})
            return
        self._add(i, L.name, lambda s, pg=pg, x=x, y=y: lib.pool_forward(pg, x, y, s),
                  {"kernel": hip.pool_kernel_name(pg), "flops": 0,
                   "bytes": 4 * (_prod(b) + _prod(L.top_shapes[0]))})

    def _try_pool_into_concat(self, i, L, layers, consumers, outputs, absorbed, concat_skip) -> bool:
        """A pooling whose top feeds only a channel Concat (inception_3c_pool / inception_4e_pool of ECO-Full: the MAX pool
        branch of a stride-2 block, deploy.prototxt:1960-1990) writes its channels of the Concat top itself
        (eco_pool_forward_strided): no pooled blob, no concat_copy launch."""
        top = L.tops[0]
        if self.dt or not self.pool_into_concat or top in outputs or len(L.bottom_shapes[0]) < 4:
            return False
        cs = [c for c in consumers.get(top, []) if c not in absorbed]
        if len(cs) != 1 or layers[cs[0]].type != "Concat" or layers[cs[0]].geom["axis"] != 1:
            return False
        Lc = layers[cs[0]]
        names = [self._resolve(b) for b in Lc.bottoms]
        if names.count(top) != 1 or self._resolve(L.bottoms[0]) not in self.tensors:
            return False
        k = names.index(top)
        ctot = Lc.top_shapes[0][1]
        c0 = sum(bs[1] for bs in Lc.bottom_shapes[:k])
        S = _prod(L.top_shapes[0][2:])
        if Lc.tops[0] not in self.tensors:
            self._materialize(Lc.tops[0], Lc.top_shapes[0])
        concat_skip.setdefault(cs[0], []).append(k)
        self.fused_away[top] = f"written directly into channels [{c0},{c0 + L.top_shapes[0][1]}) of {Lc.tops[0]}"
        g, b = L.geom, L.bottom_shapes[0]
        pg = hip.pool_geom(b[0], b[1], b[2:], g["kernel"], g["stride"], g["pad"], L.top_shapes[0][2:], g["method"])
        self._keep.append(pg)
        x, y = self._ptr(L.bottoms[0]), self._ptr(Lc.tops[0], c0 * S)
        lib = self.lib
        self._add(i, f"{L.name} [into {Lc.tops[0]}]", lambda s: lib.pool_forward_strided(pg, x, y, ctot * S, s),
                  {"kernel": hip.pool_kernel_name(pg, ctot * S, c0 * S), "flops": 0,
                   "bytes": 4 * (_prod(b) + _prod(L.top_shapes[0]))})
        return True

    def _emit_concat(self, i: int, L: LayerSpec, skip: Sequence[int]) -> None:
        ax = L.geom["axis"]
        tshape = L.top_shapes[0]
        outer, cy, inner = _prod(tshape[:ax]), tshape[ax], _prod(tshape[ax + 1:])
        y = self._ptr(L.tops[0])
        c0 = 0
        lib = self.lib
        for k, (b, bs) in enumerate(zip(L.bottoms, L.bottom_shapes)):
            cx = bs[ax]
            if k not in skip:
                if self.dt:
                    self._no_blocked(L)
                x = self._ptr(b)
                self._add(i, f"{L.name}[{k}]", lambda s, x=x, y=y, outer=outer, cx=cx, cy=cy, c0=c0, inner=inner:
                          lib.concat_copy(x, y, outer, cx, cy, c0, inner, s))
            c0 += cx

    # ------------------------------------------------------------------ fused plan
    def _resolve(self, blob: str) -> str:
        """Follow Split aliases back to the blob that a real layer produced."""
        return self._alias_src.get(blob, blob)

    def _build_fused(self) -> None:
        spec = self.spec
        layers = spec.layers
        # --- dataflow over "real" blobs: Split tops are names for their bottom (self._alias_src, build()) -------------
        consumers: Dict[str, List[int]] = {}
        for i, L in enumerate(layers):
            if L.type == "Split":
                continue
            for b in L.bottoms:
                consumers.setdefault(self._resolve(b), []).append(i)
        outputs = set(spec.outputs)
        # last layer that writes each blob (in-place layers included).  A fused group is emitted at its first layer's
        # position, never later than any layer it absorbs, so "producer index < position" means "written by then".
        self._producer: Dict[str, int] = {}
        for i, L in enumerate(layers):
            for t in L.tops:
                self._producer[self._resolve(t)] = i
        self._emit_pos: Optional[int] = None
        self._index_of = {L.name: k for k, L in enumerate(layers)}

        def sole_consumer(blob: str, typ: str) -> Optional[int]:
            cs = consumers.get(blob, [])
            if len(cs) == 1 and layers[cs[0]].type == typ and blob not in outputs:
                return cs[0]
            return None

        def bn_relu_after(blob: str) -> Optional[Tuple[int, int]]:
            """(bn idx, relu idx) if `blob` feeds a non-in-place BN whose top is first hit by an in-place ReLU.
            The BN must see the value the epilogue holds: an in-place layer on `blob` ahead of the BN in layer
            order (e.g. conv -> in-place ReLU -> BN) changes it first, so such a BN is not fused."""
            for ci in consumers.get(blob, []):
                Lb = layers[ci]
                if Lb.inplace:
                    return None          # the blob is rewritten before any later consumer reads it
                if Lb.type != "BN":
                    continue
                tb = Lb.tops[0]
                cs = consumers.get(tb, [])
                if cs and layers[cs[0]].type == "ReLU" and layers[cs[0]].inplace and \
                        layers[cs[0]].geom["negative_slope"] == 0.0 and cs[0] > ci:
                    return ci, cs[0]
            return None

        absorbed: Dict[int, str] = {}       # layer idx -> label of the group that runs it
        concat_skip: Dict[int, List[int]] = {}
        # AVE pool 3x3/1/1 whose only consumer is a 1x1 conv seen only through BN + ReLU: conv idx -> (pool idx, source)
        self._commute: Dict[int, Tuple[int, str]] = {}
        self._commute_pools: Dict[int, int] = {}
        if self.pool_commute and self.siblings:   # (both layouts: eco_avgpool_affine_forward / eco_poolb_avg_affine_forward)
            for pi, P in enumerate(layers):
                if P.type != "Pooling" or len(P.bottom_shapes[0]) != 4 or P.geom["method"] != "AVE" or \
                        list(P.geom["kernel"]) != [3, 3] or list(P.geom["stride"]) != [1, 1] or list(P.geom["pad"]) != [1, 1] or \
                        tuple(P.top_shapes[0]) != tuple(P.bottom_shapes[0]) or P.bottoms[0] in P.tops:
                    continue
                ci = sole_consumer(P.tops[0], "Convolution")
                if ci is None:
                    continue
                Cv = layers[ci]
                g = Cv.geom
                if not (all(k == 1 for k in g["kernel"]) and all(k == 1 for k in g["stride"]) and not any(g["pad"]) and
                        g.get("group", 1) == 1 and g["cout"] % 32 == 0 and "wino" not in self._param_dev[Cv.name]):
                    continue
                # the conv's value must exist only inside its fused BN + ReLU (no Eltwise, no second reader, not an output)
                if sole_consumer(Cv.tops[0], "BN") is None or bn_relu_after(Cv.tops[0]) is None:
                    continue
                self._commute[ci] = (pi, self._resolve(P.bottoms[0]))
                self._commute_pools[pi] = ci
        for i, L in enumerate(layers):
            if i in absorbed:
                continue
            if L.type == "Pooling" and i in self._commute_pools:
                continue                     # runs behind its 1x1 conv (emitted with the conv, below)
            if L.type == "Convolution":
                if self._try_fuse_ws2(i, L, layers, consumers, outputs, sole_consumer, bn_relu_after, absorbed, concat_skip):
                    pass
                elif not self._try_fuse_siblings(i, L, layers, consumers, outputs, sole_consumer, bn_relu_after,
                                                 absorbed, concat_skip):
                    self._fuse_conv(i, L, layers, consumers, outputs, sole_consumer, bn_relu_after, absorbed,
                                    concat_skip)
            elif L.type == "Pooling" and (self._try_fuse_tail(i, L, layers, sole_consumer, absorbed) or
                                          self._try_fuse_two_stream_tail(i, L, layers, absorbed) or
                                          self._try_pool_into_concat(i, L, layers, consumers, outputs, absorbed, concat_skip)):
                pass
            elif L.type == "Concat":
                if L.tops[0] not in self.tensors:
                    self._materialize(L.tops[0], L.top_shapes[0])
                self._emit_concat(i, L, skip=concat_skip.get(i, ()))
            else:
                self._emit_unfused(i, L)
        # layers each launch stands for: its own and the ones its group absorbed (partial forwards, Engine.forward)
        index_of = self._index_of
        cover: Dict[int, set] = {}
        for k in absorbed:
            lead = k
            while lead in absorbed and index_of[absorbed[lead]] != lead:   # a member's BN -> the member -> the group's first conv
                lead = index_of[absorbed[lead]]
            cover.setdefault(lead, set()).add(k)
        for idx, _label, _fn, meta in self.ops:
            meta["layers"] = {idx} | cover.get(idx, set())

    def _fuse_conv(self, i, L, layers, consumers, outputs, sole_consumer, bn_relu_after, absorbed, concat_skip) -> None:
        ep, label = self._conv_epilogue(i, L, layers, consumers, outputs, sole_consumer, bn_relu_after, absorbed,
                                        concat_skip)
        if ep is not None:
            self._emit_conv(i, L, ep, label)

    def _conv_epilogue(self, i, L, layers, consumers, outputs, sole_consumer, bn_relu_after, absorbed, concat_skip):
        """Decide what the conv's epilogue absorbs and where its outputs go: (epilogue, label), or (None, label) when
        the whole group was emitted here already (the stem)."""
        cout = L.geom["cout"]
        S = _prod(L.top_shapes[0][2:])
        ep = hip.ConvEpilogue()
        ep.bias = self._pdev(L.name, "bias") if L.geom["bias_term"] else None
        ep.residual = hip.null_view()
        ep.raw = hip.null_view()
        ep.act = hip.null_view()
        ep.bn_scale = None
        ep.bn_shift = None
        ep.relu = 0
        label = L.name
        value = L.tops[0]  # blob holding "v" of the epilogue
        # 1. Eltwise SUM absorbed when this conv's top feeds nothing else and the other operand exists already
        ei = sole_consumer(value, "Eltwise")
        if ei is not None and ei not in absorbed:
            E = layers[ei]
            others = [b for b in E.bottoms if self._resolve(b) != value]
            pos = i if self._emit_pos is None else self._emit_pos   # a sibling member runs at its group's position
            # the other operand must have been written by then: storage alone is not enough (Concat / Permute
            # destinations are materialised before all their producers have run -- round-2 advisor finding)
            prod = self._producer.get(self._resolve(others[0]), -1) if len(others) == 1 else -1
            while prod in absorbed and self._index_of[absorbed[prod]] != prod:
                prod = self._index_of[absorbed[prod]]   # a layer absorbed into a group is written where the group runs
            if len(E.bottoms) == 2 and len(others) == 1 and all(c == 1.0 for c in E.geom["coeff"]) \
                    and self._resolve(others[0]) in self.tensors and prod < pos:
                r = self._resolve(others[0])
                ep.residual = self._view(r, cout, S)
                self.fused_away[value] = f"summed into {E.tops[0]} inside the epilogue of {L.name}"
                absorbed[ei] = L.name
                value = E.tops[0]
                label += "+" + E.name
        # 2. BN + in-place ReLU on the value
        br = bn_relu_after(value)
        act_blob = None
        if br is not None and br[0] not in absorbed:
            bi, ri = br
            Lb = layers[bi]
            ep.bn_scale = self._pdev(Lb.name, "scale")
            ep.bn_shift = self._pdev(Lb.name, "shift")
            ep.relu = 1
            absorbed[bi] = L.name
            absorbed[ri] = L.name
            act_blob = Lb.tops[0]
            label += "+" + Lb.name + "+" + layers[ri].name
        # 3. raw output: needed if anything else reads the value
        rest = [c for c in consumers.get(value, []) if absorbed.get(c) != L.name]
        if act_blob is None or rest or value in outputs:
            self._materialize(value, L.top_shapes[0])
            ep.raw = self._view(value, cout, S)
        else:
            self.fused_away[value] = f"only exists inside the fused epilogue of {L.name}"
        # 3b. the stem: conv1 + BN + ReLU + pool1 as one launch
        if self._try_fuse_stem(i, L, ep, act_blob, label, layers, consumers, outputs, absorbed):
            return None, label
        # 3c. conv2_3x3 + BN + ReLU + pool2: the fused 2-D Winograd kernel leaves partial window maxima instead of its output
        if self._try_fuse_wpool(i, L, ep, act_blob, label, layers, consumers, outputs, absorbed):
            return None, label
        # 4. activated output and its destination
        if act_blob is not None:
            dest = self._act_destination(act_blob, L, layers, consumers, outputs, absorbed, concat_skip)
            if dest is None:
                self._materialize(act_blob, L.top_shapes[0])
                ep.act = self._view(act_blob, cout, S)
                # a blob with several consumers, one of them r2Dto3D + Permute (ECO-Full feeds
                # inception_3c_double_3x3_1_bn to its 2-D stream AND to the 3-D trunk): second destination
                second = self._permuted_destination(act_blob, L, layers, consumers, outputs, absorbed, sole=False)
                if second is not None:
                    ep.act2 = second
                    label += "+" + "+".join(layers[c].name for c, v in absorbed.items()
                                            if v == L.name and layers[c].type in ("Reshape", "Permute"))
            else:
                ep.act = dest
        return ep, label

    # -- sibling convolutions: one launch for convs of one geometry that read the same bottom --------------------
    def _try_fuse_siblings(self, i, L, layers, consumers, outputs, sole_consumer, bn_relu_after, absorbed,
                           concat_skip) -> bool:
        """Convs of one geometry reading the same blob -- the 1x1 / 3x3_reduce / double_3x3_reduce convs of an
        Inception block (models_ECO_Lite/kinetics/deploy.prototxt:130-330), a residual block's first conv and its
        projection shortcut (res4a_1 / res4a_down, res5a_1 / res5a_down: deploy.prototxt:1090-1180) -- concatenated
        along the output channel are one GEMM that reads (gathers) the input once; each member's 32-row tiles write
        to its own destination (eco_conv_epilogue::nseg).  A member is either seen only through its fused BN + ReLU
        or keeps its raw value (a shortcut whose Eltwise then rides on the block's second conv)."""
        if not self.siblings:
            return False

        def eligible(Lc) -> bool:
            g = Lc.geom
            return g["cout"] % 32 == 0 and g.get("group", 1) == 1 and "wino" not in self._param_dev[Lc.name] and \
                not (self.stem and self._stem_geometry(Lc))

        def same(Lc) -> bool:
            return all(list(Lc.geom[k]) == list(L.geom[k]) for k in ("kernel", "stride", "pad")) and \
                Lc.geom["cin"] == L.geom["cin"] and Lc.bottom_shapes[0] == L.bottom_shapes[0]

        commuted = getattr(self, "_commute", {})
        # a conv that runs ahead of its AVE pool (pool_commute) reads the pool's input
        src = commuted[i][1] if i in commuted else self._resolve(L.bottoms[0])
        if not eligible(L):
            return False
        if not self.sibling_blocks and not all(k == 1 for k in L.geom["kernel"]):
            return False
        members = [i]
        for j in sorted(set(consumers.get(src, [])) | {c for c, (_, sb) in commuted.items() if sb == src}):
            Lj = layers[j]
            if j <= i or j in absorbed or Lj.type != "Convolution" or len(members) == hip.MAX_SEG + 1:
                continue
            if eligible(Lj) and same(Lj) and \
                    not any(layers[k].inplace and self._resolve(layers[k].bottoms[0]) == src for k in range(i, j)):
                members.append(j)
        if len(members) < 2 and i not in commuted:
            return False
        # every member is emitted at this position (its only input is the shared bottom): decide its epilogue now
        found = []
        after = []                                        # pool launches that follow the conv launch (pool_commute)
        for j in members:
            Lj = layers[j]
            br = bn_relu_after(Lj.tops[0])
            self._emit_pos = i
            try:
                ep, label = self._conv_epilogue(j, Lj, layers, consumers, outputs, sole_consumer, bn_relu_after,
                                                absorbed, concat_skip)
            finally:
                self._emit_pos = None
            if j != i:
                absorbed[j] = L.name
            if ep is None:
                continue                                  # (emitted by _conv_epilogue itself)
            plain = not ep.residual.ptr and not ep.act2.ptr
            if j in commuted and (ep.act2.ptr or ep.raw.ptr or ep.residual.ptr or not ep.act.ptr):
                # the pool-affine kernels write ONE activated destination: an epilogue with a second / raw / residual
                # destination cannot ride on them (the pre-pass only registers convs whose BN+ReLU blob has a single
                # consumer, so this is a planner invariant, not a user error)
                raise NetSpecError(f"{Lj.name}: pool_commute cannot express this epilogue (act2 / raw / residual set); "
                                   f"build the net with pool_commute=False")
            if j in commuted:
                # conv first, on the pool's input and without its bias, raw into a scratch tensor; the window average,
                # bias, BN and ReLU follow on the conv's channels and write the destination the epilogue chose
                pi = commuted[j][0]
                Pj = layers[pi]
                absorbed[pi] = L.name
                z = Lj.name + "/before_" + Pj.name
                self._materialize(z, Lj.top_shapes[0])
                self.fused_away[Pj.tops[0]] = f"{Pj.name} runs behind {Lj.name} (linear maps exchanged) on {z}"
                n_, c_, h_, w_ = Lj.top_shapes[0]
                zv = self._view(z, c_, h_ * w_)
                after.append((self._ptr(z), ep, f"{Pj.name}+{label} [average, bias, BN, ReLU]", (n_, c_, h_, w_)))
                ep2 = hip.ConvEpilogue()
                ep2.bias = None
                ep2.residual, ep2.act, ep2.act2 = hip.null_view(), hip.null_view(), hip.null_view()
                ep2.bn_scale = ep2.bn_shift = None
                ep2.relu = 0
                ep2.raw = zv
                found.append((Lj, ep2, f"{Lj.name} [ahead of {Pj.name}]", zv, 0, None, True))
            elif plain and ep.act.ptr and ep.act.t == 1 and not ep.raw.ptr and br is not None:
                found.append((Lj, ep, label, ep.act, int(ep.relu), layers[br[0]].name, False))
            elif plain and ep.raw.ptr and ep.raw.t == 1 and not ep.act.ptr:
                found.append((Lj, ep, label, ep.raw, 0, None, False))
            else:                                         # a destination the segmented epilogue cannot express
                self._emit_conv(i, Lj, ep, label)
        if len(found) == 1:
            self._emit_conv(i, found[0][0], found[0][1], found[0][2], src=src)
        elif found:
            self._emit_sibling_conv(i, found, src)
        lib = self.lib
        for zp, ep, label, (n_, c_, h_, w_) in after:
            self._keep.append(ep)
            if self.dt:
                dt = self.dt
                self._add(i, label, lambda s, zp=zp, ep=ep, n_=n_, c_=c_, h_=h_, w_=w_: lib.poolb_avg_affine_forward(
                    dt, zp, ep.bias, ep.bn_scale, ep.bn_shift, ep.relu, ep.act, n_, c_, h_, w_, s),
                    {"kernel": "eco::poolb_avg_affine_kernel", "flops": 0, "bytes": 2 * self.esize * n_ * c_ * h_ * w_})
                continue
            self._add(i, label, lambda s, zp=zp, ep=ep, n_=n_, c_=c_, h_=h_, w_=w_: lib.avgpool_affine_forward(
                zp, ep.bias, ep.bn_scale, ep.bn_shift, ep.relu, ep.act, n_, c_, h_, w_, s),
                {"kernel": "eco::avgpool2d_k3s1p1_affine_kernel", "flops": 0, "bytes": 8 * n_ * c_ * h_ * w_})
        return True

    def _emit_sibling_conv(self, i, found, src=None) -> None:
        Ls = [f[0] for f in found]
        L = Ls[0]
        key = "|".join(Lc.name for Lc in Ls)
        couts = [Lc.geom["cout"] for Lc in Ls]
        ctot = sum(couts)
        geom = hip.conv_geom(L.bottom_shapes[0][0], L.geom["cin"], ctot, L.bottom_shapes[0][2:], L.geom["kernel"],
                             L.geom["stride"], L.geom["pad"], L.top_shapes[0][2:])
        st = self._group_dev.setdefault(key, {})
        old = st.get("plan")
        if self.dt:
            plan = self.lib.convb_plan(geom, self.dt, self.num_cu)
            if old is None or (old.wp_vecs, st.get("ctot")) != (plan.wp_vecs, ctot):
                st["wp"] = self.alloc.empty(plan.wp_vecs * 8, np.uint16)
        else:
            plan = self.lib.conv_plan(geom, self.num_cu)
            if old is None or (old.wp_elems, old.ktab_elems, st.get("ctot")) != (plan.wp_elems, plan.ktab_elems, ctot):
                st["wp"] = self.alloc.empty(plan.wp_elems, np.float32)
                st["ktab"] = self.alloc.empty(plan.ktab_elems, np.int32)
        if st.get("ctot") != ctot:
            for k in ("bias", "scale", "shift"):
                st[k] = self.alloc.empty(ctot, np.float32)
            st["ctot"] = ctot
        st["geom"], st["plan"] = geom, plan
        if plan.ws_bytes > getattr(self, "_ws_bytes", 0):
            self._keep.append(getattr(self, "_ws", None))   # launches recorded so far hold the old buffer's address
            self._ws = self.alloc.empty((plan.ws_bytes + 3) // 4, np.float32)
            self._ws_bytes = plan.ws_bytes
        self._groups[key] = {"convs": [Lc.name for Lc in Ls], "bns": [f[5] for f in found], "st": st,
                             "nobias": [f[6] for f in found]}
        self._dirty_groups.add(key)
        ep = hip.ConvEpilogue()
        ep.bias = self.alloc.ptr(st["bias"])
        ep.bn_scale, ep.bn_shift = self.alloc.ptr(st["scale"]), self.alloc.ptr(st["shift"])
        ep.residual, ep.raw, ep.act2 = hip.null_view(), hip.null_view(), hip.null_view()
        ep.act, ep.relu = found[0][3], found[0][4]
        ep.nseg = len(Ls) - 1
        begin = 0
        for s in range(1, len(Ls)):
            begin += couts[s - 1]
            ep.seg_begin[s - 1] = begin
            ep.seg_relu[s - 1] = found[s][4]
            ep.seg_act[s - 1] = found[s][3]
        x = self._ptr(src if src is not None else L.bottoms[0])
        wp = self.alloc.ptr(st["wp"])
        self._keep.append((geom, plan, ep))
        lib = self.lib
        k = L.geom["cin"] * _prod(L.geom["kernel"])
        n_out = sum(_prod(Lc.top_shapes[0]) for Lc in Ls)
        es = self.esize
        meta = {"kernel": hip.convb_kernel_name(plan) if self.dt else hip.conv_kernel_name(plan), "flops": 2 * n_out * k,
                "bytes": es * (_prod(L.bottom_shapes[0]) + n_out) + (2 if self.dt else 4) * k * ctot, "siblings": len(Ls)}
        if self.dt:
            in_name = src if src is not None else L.bottoms[0]   # (a commuted conv leading its group: its bottom, the
            if not self.tensors[in_name].dt:                     #  pool's top, is never materialised)
                raise NetSpecError(f"{L.name}: input blob {in_name} is not channel-blocked")

            def run(s, g=geom, plan=plan, x=x, wp=wp, ep=ep):
                lib.convb_forward(g, plan, x, wp, ep, self.alloc.ptr(self._ws) if plan.ws_bytes else None, s)
        else:
            kt = self.alloc.ptr(st["ktab"])

            def run(s, g=geom, plan=plan, x=x, wp=wp, kt=kt, ep=ep):
                lib.conv_forward(g, plan, x, wp, kt, ep, self.alloc.ptr(self._ws) if plan.ws_bytes else None, s)
        self._add(i, " | ".join(f[2] for f in found), run, meta)

    def _sync_groups(self, dirty) -> None:
        """Repack the concatenated weights / bias / folded BN vectors of sibling groups with a changed member."""
        for key, grp in self._groups.items():
            if key not in self._dirty_groups and not (dirty & set(grp["convs"] + [b for b in grp["bns"] if b])):
                continue
            st = grp["st"]
            g, plan = st["geom"], st["plan"]
            w = np.ascontiguousarray(np.concatenate(
                [np.asarray(self.params[n][0], np.float32).reshape(self.spec.layer(n).geom["cout"], -1)
                 for n in grp["convs"]], 0))
            if self.dt:
                wpb = np.empty(plan.wp_vecs * 8, np.uint16)
                self.lib.convb_pack_weights(g, plan, w.ctypes.data, wpb.ctypes.data)
                self.alloc.upload(st["wp"], wpb)
            else:
                wp = np.empty(plan.wp_elems, np.float32)
                kt = np.empty(plan.ktab_elems, np.int32)
                self.lib.conv_pack_weights(g, plan, w.ctypes.data, wp.ctypes.data, kt.ctypes.data)
                self.alloc.upload(st["wp"], wp)
                self.alloc.upload(st["ktab"], kt)
            bias, scale, shift = [], [], []
            for n, bn, nobias in zip(grp["convs"], grp["bns"], grp["nobias"]):
                Lc = self.spec.layer(n)
                c = Lc.geom["cout"]
                # (a conv that runs ahead of its AVE pool adds its bias behind the pool)
                bias.append(np.asarray(self.params[n][1], np.float32).ravel() if Lc.geom["bias_term"] and not nobias
                            else np.zeros(c, np.float32))
                if bn is None:                        # the member keeps its raw value
                    scale.append(np.ones(c, np.float32))
                    shift.append(np.zeros(c, np.float32))
                else:
                    a, b = fold_bn(self.params[bn], bn_eps(self.spec.layer(bn)))
                    scale.append(np.asarray(a, np.float32))
                    shift.append(np.asarray(b, np.float32))
            self.alloc.upload(st["bias"], np.concatenate(bias))
            self.alloc.upload(st["scale"], np.concatenate(scale))
            self.alloc.upload(st["shift"], np.concatenate(shift))
            self._dirty_groups.discard(key)

    def _act_destination(self, act_blob, L, layers, consumers, outputs, absorbed, concat_skip):
        """Strided destination for a fused conv's activated output, or None for a dense tensor."""
        if act_blob in outputs:
            return None
        cs = [c for c in consumers.get(act_blob, []) if absorbed.get(c) != L.name]
        if len(cs) != 1:
            return None
        Lc = layers[cs[0]]
        cout = L.geom["cout"]
        tshape = L.top_shapes[0]
        S = _prod(tshape[2:])
        # (a) channel slice of a Concat top
        if Lc.type == "Concat" and Lc.geom["axis"] == 1:
            k = [self._resolve(b) for b in Lc.bottoms].index(act_blob)
            if [self._resolve(b) for b in Lc.bottoms].count(act_blob) != 1:
                return None
            ctot = Lc.top_shapes[0][1]
            c0 = sum(bs[1] for bs in Lc.bottom_shapes[:k])
            if Lc.tops[0] not in self.tensors:
                self._materialize(Lc.tops[0], Lc.top_shapes[0])
            concat_skip.setdefault(cs[0], []).append(k)
            self.fused_away[act_blob] = f"written directly into channels [{c0},{c0 + cout}) of {Lc.tops[0]}"
            return self._view(Lc.tops[0], cout, S, c0, ctot)
        # (b) r2Dto3D Reshape [-1,T,C,H,W] followed by Permute [0,2,1,3,4]
        return self._permuted_destination(act_blob, L, layers, consumers, outputs, absorbed, sole=True)

    def _permuted_destination(self, act_blob, L, layers, consumers, outputs, absorbed, sole: bool):
        """View that writes a conv's [B*T,C,H,W] output through r2Dto3D (Reshape [-1,T,C,H,W]) + Permute
        [0,2,1,3,4] straight into the [B,C,T,H,W] volume (reshape_layer.cpp:88, permute_layer.cpp:9-26), or None.
        sole=True: the Reshape is the blob's only consumer (the blob itself is then never stored); sole=False: it
        is one of several (the view becomes the epilogue's second destination)."""
        tshape = L.top_shapes[0]
        S = _prod(tshape[2:])
        cs = [c for c in consumers.get(act_blob, []) if absorbed.get(c) != L.name]
        if act_blob in outputs or len(tshape) != 4 or (sole and len(cs) != 1) or (not sole and len(cs) < 2):
            return None
        for ci in cs:
            Lc = layers[ci]
            if Lc.type != "Reshape" or len(Lc.top_shapes[0]) != 5:
                continue
            rs = Lc.top_shapes[0]
            if tuple(rs[2:]) != tuple(tshape[1:]):
                continue
            pcs = consumers.get(Lc.tops[0], [])
            if len(pcs) != 1 or Lc.tops[0] in outputs:
                continue
            Lp = layers[pcs[0]]
            if Lp.type != "Permute" or list(Lp.geom["order"]) != [0, 2, 1, 3, 4]:
                continue
            T, Cc = rs[1], rs[2]
            self._materialize(Lp.tops[0], Lp.top_shapes[0])
            absorbed[ci] = L.name
            absorbed[pcs[0]] = L.name
            why = f"written through {Lc.name}+{Lp.name} directly into {Lp.tops[0]}"
            if sole:
                self.fused_away[act_blob] = why
            self.fused_away[Lc.tops[0]] = why
            return hip.View(self._ptr(Lp.tops[0]), (Cc // self.cblk) * T * S, S, T * S, T)
        return None

    def _try_fuse_tail(self, i, L, layers, sole_consumer, absorbed) -> bool:
        """global AVE pool -> Reshape [-1,C] -> (Dropout) -> InnerProduct as one launch."""
        g = L.geom
        b = L.bottom_shapes[0]
        if g["method"] != "AVE" or list(g["kernel"]) != list(b[2:]) or any(p != 0 for p in g["pad"]) \
                or any(o != 1 for o in L.top_shapes[0][2:]):
            return False
        ri = sole_consumer(L.tops[0], "Reshape")
        if ri is None:
            return False
        Lr = layers[ri]
        if tuple(Lr.top_shapes[0]) != (b[0], b[1]):
            return False
        chain = [ri]
        blob = Lr.tops[0]
        # in-place Dropout (TEST = identity) may sit on the reshaped blob
        cs = [c for c in self._consumers_of(blob)]
        drop = [c for c in cs if layers[c].type == "Dropout" and layers[c].inplace]
        fcs = [c for c in cs if layers[c].type == "InnerProduct"]
        if len(fcs) != 1 or len(cs) != len(drop) + 1 or blob in self.spec.outputs:
            return False
        Lf = layers[fcs[0]]
        if Lf.geom["K"] != b[1] or Lf.geom["M"] != b[0]:
            return False
        chain += drop + fcs
        for c in chain:
            absorbed[c] = L.name
        for nm in (L.tops[0], blob):
            self.fused_away[nm] = f"only exists inside the fused {L.name}+{Lf.name} tail"
        self._materialize(Lf.tops[0], Lf.top_shapes[0], plain=True)
        x, y = self._ptr(L.bottoms[0]), self._ptr(Lf.tops[0])
        w = self._pdev(Lf.name, "w")
        bias = self._pdev(Lf.name, "bias") if Lf.geom["bias_term"] else None
        B, Cc, S, n_out = b[0], b[1], _prod(b[2:]), Lf.geom["num_output"]
        lib = self.lib
        if self.dt:
            dt = self.dt
            self._add(i, f"{L.name}+{Lf.name}", lambda s, x=x, w=w, bias=bias, y=y, B=B, Cc=Cc, S=S, n_out=n_out:
                      lib.global_avgpool_fc_b_forward(x, dt, w, bias, y, B, Cc, S, n_out, Cc, 0, False, s),
                      {"kernel": "eco::global_avgpool_fc_b_kernel", "flops": 2 * B * n_out * Cc,
                       "bytes": self.esize * B * Cc * S + 4 * (n_out * Cc + B * n_out)})
            return True
        self._add(i, f"{L.name}+{Lf.name}", lambda s, x=x, w=w, bias=bias, y=y, B=B, Cc=Cc, S=S, n_out=n_out:
                  lib.global_avgpool_fc_forward(x, w, bias, y, B, Cc, S, n_out, Cc, 0, False, s),
                  {"kernel": "eco::global_avgpool_fc_kernel", "flops": 2 * B * n_out * Cc,
                   "bytes": 4 * (B * Cc * S + n_out * Cc + B * n_out)})
        return True

    def _try_fuse_two_stream_tail(self, i, L, layers, absorbed) -> bool:
        """ECO-Full's tail (models_ECO_Full/kinetics/deploy.prototxt:4607-4690) as two launches:

            2-D stream: global_pool2D (AVE over the whole h x w plane of [B*T,C2,h,w]) -> dropout -> reshape
                        [-1,1,T,C2] -> segment_consensus (AVE, kernel (T,1)) -> reshape [-1,C2]        \
            3-D stream: global_pool (AVE over the whole d x h x w volume of [B,C3,...]) -> reshape -> dropout -> Concat -> fc

        = fc over [mean_{t,h,w} x2d | mean_{d,h,w} x3d]: one pool+fc launch per stream on its columns of the fc
        weights, the second accumulating into the logits.  Matched from the 2-D pool; every blob in between must
        have no other consumer."""
        spec = self.spec
        g, b = L.geom, L.bottom_shapes[0]
        if self.dt or len(b) != 4 or g["method"] != "AVE" or list(g["kernel"]) != list(b[2:]) or \
                any(p != 0 for p in g["pad"]) or any(o != 1 for o in L.top_shapes[0][2:]):
            return False
        chain: List[int] = []

        def only_next(blob: str, typ: str) -> Optional[int]:
            """The single non-Dropout consumer of `blob` if it has type `typ` (in-place Dropouts are absorbed)."""
            cs = self._consumers_of(blob)
            drops = [c for c in cs if layers[c].type == "Dropout" and layers[c].inplace]
            rest = [c for c in cs if c not in drops]
            if len(rest) != 1 or layers[rest[0]].type != typ or blob in spec.outputs:
                return None
            chain.extend(drops)
            return rest[0]

        r1 = only_next(L.tops[0], "Reshape")
        if r1 is None:
            return False
        shp = layers[r1].top_shapes[0]
        if len(shp) != 4 or shp[1] != 1 or shp[3] != b[1] or shp[0] * shp[2] != b[0]:
            return False
        B, T, C2 = shp[0], shp[2], b[1]
        p2 = only_next(layers[r1].tops[0], "Pooling")
        if p2 is None:
            return False
        Lp2 = layers[p2]
        if Lp2.geom["method"] != "AVE" or list(Lp2.geom["kernel"]) != [T, 1] or any(x != 0 for x in Lp2.geom["pad"]) or \
                tuple(Lp2.top_shapes[0]) != (B, 1, 1, C2):
            return False
        r2 = only_next(Lp2.tops[0], "Reshape")
        if r2 is None or tuple(layers[r2].top_shapes[0]) != (B, C2):
            return False
        cat = only_next(layers[r2].tops[0], "Concat")
        if cat is None:
            return False
        Lcat = layers[cat]
        srcs = [self._resolve(x) for x in Lcat.bottoms]
        if Lcat.geom["axis"] != 1 or len(srcs) != 2 or srcs[0] != layers[r2].tops[0]:
            return False
        # the other operand: Reshape [B,C3] of a whole-volume AVE pool
        prod = [k for k, Lk in enumerate(layers) if srcs[1] in Lk.tops and not Lk.inplace]
        if len(prod) != 1 or layers[prod[0]].type != "Reshape":
            return False
        r3 = prod[0]
        pp = [k for k, Lk in enumerate(layers) if self._resolve(layers[r3].bottoms[0]) in Lk.tops and not Lk.inplace]
        if len(pp) != 1 or layers[pp[0]].type != "Pooling":
            return False
        p3 = pp[0]
        Lp3 = layers[p3]
        b3 = Lp3.bottom_shapes[0]
        if Lp3.geom["method"] != "AVE" or list(Lp3.geom["kernel"]) != list(b3[2:]) or any(x != 0 for x in Lp3.geom["pad"]) or \
                b3[0] != B or tuple(layers[r3].top_shapes[0]) != (B, b3[1]) or self._resolve(Lp3.bottoms[0]) not in self.tensors:
            return False
        C3 = b3[1]
        if only_next(Lp3.tops[0], "Reshape") != r3 or only_next(layers[r3].tops[0], "Concat") != cat:
            return False
        fc = only_next(Lcat.tops[0], "InnerProduct")
        if fc is None:
            return False
        Lf = layers[fc]
        if Lf.geom["K"] != C2 + C3 or Lf.geom["M"] != B:
            return False
        if any(c <= i for c in (r1, p2, r2, cat, r3, p3, fc) if c != i and c < i):
            return False
        for c in chain + [r1, p2, r2, cat, r3, p3, fc]:
            absorbed[c] = L.name
        for c in (i, r1, p2, r2, cat, r3, p3):
            for nm in layers[c].tops:
                self.fused_away[nm] = f"only exists inside the fused two-stream tail {L.name}+{Lp3.name}+{Lf.name}"
        self._materialize(Lf.tops[0], Lf.top_shapes[0], plain=True)
        x2, x3, y = self._ptr(L.bottoms[0]), self._ptr(Lp3.bottoms[0]), self._ptr(Lf.tops[0])
        w = self._pdev(Lf.name, "w")
        bias = self._pdev(Lf.name, "bias") if Lf.geom["bias_term"] else None
        S2, S3, n_out, wk = _prod(b[2:]), _prod(b3[2:]), Lf.geom["num_output"], C2 + C3
        lib = self.lib
        self._add(i, f"{Lp3.name}+{Lf.name} [3-D stream]", lambda s: lib.global_avgpool_fc_seg_forward(
            x3, w, bias, y, B, 1, C3, S3, n_out, wk, C2, False, s),
            {"kernel": "eco::global_avgpool_fc_kernel", "flops": 2 * B * n_out * C3, "bytes": 4 * (B * C3 * S3 + n_out * C3 + B * n_out)})
        self._add(i, f"{L.name}+{Lp2.name}+{Lf.name} [2-D stream, accumulate]", lambda s: lib.global_avgpool_fc_seg_forward(
            x2, w, None, y, B, T, C2, S2, n_out, wk, 0, True, s),
            {"kernel": "eco::global_avgpool_fc_kernel", "flops": 2 * B * n_out * C2, "bytes": 4 * (B * T * C2 * S2 + n_out * C2 + 2 * B * n_out)})
        return True

    def _consumers_of(self, blob: str) -> List[int]:
        return [i for i, L in enumerate(self.spec.layers)
                if L.type != "Split" and any(self._resolve(b) == blob for b in L.bottoms)]

    # ------------------------------------------------------------------ run
    def forward(self, start: int = 0, end: Optional[int] = None, stream: Optional[int] = None) -> None:
        if not self._built:
            self.build()
        self._sync_params()
        end = len(self.spec.layers) - 1 if end is None else end
        if stream is None and hasattr(self.alloc, "stream"):
            stream = self.alloc.stream()
        # A launch runs when any layer it stands for lies in [start, end]: starting at a layer that was absorbed into an
        # earlier group (a sibling conv, a fused BN) re-runs that group instead of silently skipping the layer.
        for idx, _label, fn, meta in self.ops:
            if any(start <= k <= end for k in meta.get("layers", (idx,))):
                fn(stream)

    def op_labels(self) -> List[str]:
        return [op[1] for op in self.ops]

    def profile(self, iters: int = 3) -> List[dict]:
        """Per-launch device time (ms, mean over ``iters``) measured with HIP events on the
        launch stream (the tools/caffe.cpp:276-360 `caffe time` idea, per launch instead of
        per layer).  Needs the torch allocator (events are torch.cuda.Event on its stream)."""
        torch = self.alloc.torch
        self._sync_params()
        stream = self.alloc.stream()
        n = len(self.ops)
        tot = [0.0] * n
        for _ in range(iters):
            evs = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
            evs[0].record()
            for j, (_i, _l, fn, _m) in enumerate(self.ops):
                fn(stream)
                evs[j + 1].record()
            torch.cuda.synchronize()
            for j in range(n):
                tot[j] += evs[j].elapsed_time(evs[j + 1])
        return [dict(label=op[1], ms=tot[j] / iters, **op[3]) for j, op in enumerate(self.ops)]
