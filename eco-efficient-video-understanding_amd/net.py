"""pycaffe-compatible ``Net`` / ``Blob`` surface over the MI355X HIP engine.

Mirrors, for the inference path, what the reference exposes through
Boost.Python + ``pycaffe.py`` (caffe_3d/python/caffe/_caffe.cpp:99-108,171-260;
caffe_3d/python/caffe/pycaffe.py:21-98):

    net = Net(model_prototxt, [weights], TEST)
    net.blobs['data'].reshape(B*N, 3, 224, 224); net.reshape()
    net.blobs['data'].data[...] = frames
    out = net.forward()                 # {'fc8': ndarray}
    net.params['conv1_7x7_s2'][0].data  # weights, reference layout

``Blob.data`` follows ``SyncedMemory`` (caffe_3d/src/caffe/syncedmem.cpp:21-70):
a host ndarray mirrored lazily with the HBM copy; touching ``.data`` moves the
"head" to the host, the next forward uploads it.  ``Blob.tensor`` is the zero-copy
extension: a torch view of the HBM buffer (for callers that keep frames resident).

Errors: shape/graph errors raise ``NetSpecError`` and C-ABI failures raise
``hip.EcoError`` where the reference would CHECK-fail and abort.
There is no CPU execution path: without the HIP library / a GPU, constructing a
``Net`` raises.
"""
from __future__ import annotations

import os
from collections import OrderedDict
from typing import Dict, List, Optional, Sequence

import numpy as np

from . import fillers, hip
from .engine import Engine, TorchAllocator
from .netspec import TEST, TRAIN, NetSpec, NetSpecError, param_shapes

_HEAD_HOST, _HEAD_DEVICE, _HEAD_SYNCED, _HEAD_NONE = 0, 1, 2, 3

_device_id: Optional[int] = None


def set_mode_gpu() -> None:
    """caffe.set_mode_gpu (_caffe.cpp:213): the only mode of this path."""


def set_mode_cpu() -> None:
    """caffe.set_mode_cpu: not available -- this package is the GPU path only."""
    raise RuntimeError("eco_amd has no CPU execution mode (the CPU oracle lives in oracle/ for tests only)")


def set_device(device_id: int) -> None:
    """caffe.set_device (_caffe.cpp:216 -> Caffe::SetDevice, common.cpp:140-160)."""
    global _device_id
    hip.load().set_device(int(device_id))
    import torch
    torch.cuda.set_device(int(device_id))
    _device_id = int(device_id)


def get_device() -> Optional[int]:
    return _device_id


class Blob:
    """Activation or parameter blob (include/caffe/blob.hpp:24-282, _caffe.cpp:171-205)."""

    def __init__(self, net: "Net", name: str, shape: Sequence[int], param_of: Optional[str] = None) -> None:
        self._net = net
        self._name = name
        self._shape = tuple(int(s) for s in shape)
        self._param_of = param_of
        self._host: Optional[np.ndarray] = None
        self._head = _HEAD_NONE
        # Reshape / Split / Dropout(TEST) tops share their bottom's SyncedMemory in the reference
        # (Blob::ShareData, reshape_layer.cpp:88): such a blob keeps no mirror of its own, its data is a
        # reshaped view of the storage owner's host copy and its head is the owner's.
        self._root: Optional["Blob"] = None

    # -- shape accessors ----------------------------------------------------
    @property
    def shape(self):
        return tuple(self._shape)

    @property
    def count(self) -> int:
        return int(np.prod(self._shape)) if self._shape else 1

    def _legacy(self, i: int) -> int:  # blob.hpp:140-160 LegacyShape
        if len(self._shape) > 4:
            raise NetSpecError("Cannot use legacy accessors on Blobs with > 4 axes.")
        return self._shape[i] if i < len(self._shape) else 1

    num = property(lambda self: self._legacy(0))
    channels = property(lambda self: self._legacy(1))
    height = property(lambda self: self._legacy(2))
    width = property(lambda self: self._legacy(3))

    def reshape(self, *dims) -> None:
        """Blob.reshape (_caffe.cpp:193-205).  Only net inputs may be reshaped; call
        ``net.reshape()`` afterwards to propagate."""
        if self._param_of is not None:
            raise NetSpecError("parameter blobs cannot be reshaped")
        if self._name not in self._net.inputs:
            raise NetSpecError(f"only net input blobs can be reshaped by the user ({self._name!r} is produced by a layer)")
        self._shape = tuple(int(d) for d in dims)
        self._host = None
        self._head = _HEAD_NONE
        self._net._pending_input_shapes[self._name] = self._shape

    # -- data -----------------------------------------------------------------
    @property
    def data(self) -> np.ndarray:
        if self._param_of is not None:
            self._net._engine.mark_param_dirty(self._param_of)  # mutable_cpu_data semantics
            return self._host
        if self._name not in self._net._engine.tensors:
            self._net._engine._ptr(self._name)  # raises KeyError with the fused-away reason
        if self._root is not None:
            return self._root.data.reshape(self._shape)
        if self._host is None or self._host.shape != self._shape:
            self._host = np.zeros(self._shape, np.float32)
            if self._head == _HEAD_NONE:
                self._head = _HEAD_HOST
        if self._head == _HEAD_DEVICE:
            self._host[...] = self._net._download(self._name).reshape(self._shape)
        self._head = _HEAD_HOST
        return self._host

    @property
    def tensor(self):
        """Zero-copy torch view of the HBM buffer (extension; torch plumbing only)."""
        if self._param_of is not None:
            raise AttributeError("parameter blobs have no device view")
        t = self._net._engine.tensors.get(self._name)
        if t is None:
            self._net._engine._ptr(self._name)  # raises with the fused-away reason
        if t.dt:
            raise AttributeError(f"blob {self._name!r} is stored channel-blocked ({self._net._engine.dtype} path); "
                                 "only fp32 N,C,... blobs (net inputs, logits) have a device view -- use .data")
        owner = self._root if self._root is not None else self
        self._net._flush_host(owner._name)
        owner._head = _HEAD_DEVICE
        return t.handle[: self.count].view(self._shape)


class _LayerView:
    def __init__(self, spec_layer, blobs) -> None:
        self.type = spec_layer.type
        self.name = spec_layer.name
        self.blobs = blobs


class Net:
    def __init__(self, model, weights=None, phase=TEST, *, fuse: bool = True, winograd: bool = True,
                 dtype: str = "f32", device: Optional[int] = None,
                 params: Optional[Dict[str, List[np.ndarray]]] = None, seed: int = 0, pool_commute: bool = True,
                 stem: bool = True, _backend=None, _num_cu: Optional[int] = None) -> None:
        # pycaffe accepts Net(model, phase) and Net(model, weights, phase)
        if isinstance(weights, int) and not isinstance(weights, bool):
            weights, phase = None, weights
        if phase not in (TEST, TRAIN):
            raise ValueError("phase must be caffe.TRAIN or caffe.TEST")
        if phase != TEST:
            raise NetSpecError("only the TEST-phase forward path is implemented (training is out of scope)")
        self._spec = NetSpec.from_prototxt(model, phase=phase)
        if _backend is not None:  # test hook: (EcoLib, allocator) of the CPU emulator build
            lib, alloc = _backend
        else:
            lib = hip.load()  # raises if libeco_hip.so is missing: no fallback
            if device is None:
                device = _device_id
            alloc = TorchAllocator(device)
        self._lib = lib
        self._alloc = alloc
        # winograd=False evaluates every convolution directly (the reference's arithmetic order up to the
        # summation order); the default (True) routes every stride-1 3x3(x3) conv with cin >= 64 through
        # Winograd F(4x4,3x3) when the batch is large enough (engine._wino_eligible); 2 / 4 force the tile size
        # dtype: "f32" (default) = the reference's fp32 blobs on the fp32 MFMA kernels; "bf16" = bf16 storage and
        # bf16 MFMA with fp32 accumulation (BASELINE configs[4]), which keeps activations channel-blocked internally;
        # .data still hands out N,C,... fp32 arrays.
        # pool_commute=False keeps AVE pool -> 1x1 conv in the reference's order (default: conv first, on the block's
        # input inside the sibling launch; the average then runs on the conv's channels -- engine.pool_commute)
        self._engine = Engine(self._spec, lib, alloc, fuse=fuse, winograd=winograd, num_cu=_num_cu, dtype=dtype)
        self._engine.pool_commute = bool(pool_commute)
        self._engine.stem = bool(stem)   # False: conv1 and pool1 as separate launches (debug / A-B measurements)
        self._pending_input_shapes: Dict[str, tuple] = {}
        self._engine.set_params(params if params is not None else fillers.filler_params(self._spec, seed))
        self._engine.build()
        self._make_blobs()
        if weights is not None:
            self.copy_from(weights)

    # -- structure ------------------------------------------------------------
    def _make_blobs(self) -> None:
        old = getattr(self, "blobs", {})
        self.blobs: "OrderedDict[str, Blob]" = OrderedDict()
        for name in self._spec.blob_names:
            shape = self._spec.blob_shapes[name]
            b = old.get(name)
            if b is not None and b.shape == tuple(shape):
                self.blobs[name] = b
            else:
                self.blobs[name] = Blob(self, name, shape)
        for name, b in self.blobs.items():   # aliases share the mirror / head of the blob that owns the storage
            t = self._engine.tensors.get(name)
            b._root = self.blobs[t.owner] if t is not None and t.owner != name and t.owner in self.blobs else None
            if b._root is not None:
                b._host, b._head = None, _HEAD_NONE
        self.params: "OrderedDict[str, List[Blob]]" = OrderedDict()
        for L in self._spec.layers:
            if param_shapes(L):
                lst = []
                for i, arr in enumerate(self._engine.params[L.name]):
                    pb = Blob(self, f"{L.name}[{i}]", arr.shape, param_of=L.name)
                    pb._host = arr
                    lst.append(pb)
                self.params[L.name] = lst
        self.layers = [_LayerView(L, self.params.get(L.name, [])) for L in self._spec.layers]

    @property
    def inputs(self) -> List[str]:
        return list(self._spec.inputs)

    @property
    def outputs(self) -> List[str]:
        return list(self._spec.outputs)

    @property
    def _layer_names(self) -> List[str]:
        return self._spec.layer_names

    @property
    def _blob_names(self) -> List[str]:
        return list(self.blobs.keys())

    @property
    def name(self) -> str:
        return self._spec.name

    # -- memory sync ------------------------------------------------------------
    def _download(self, name: str) -> np.ndarray:
        eng = self._engine
        eng._ptr(name)  # raises if fused away
        if hasattr(self._alloc, "synchronize"):
            self._alloc.synchronize()
        t = eng.tensors[name]
        raw = self._alloc.download(t.handle, t.count)
        if t.dt:   # channel-blocked storage -> the reference's N,C,... fp32 layout
            from . import blocked
            return blocked.from_blocked(np.asarray(raw), t.shape, t.dt).reshape(-1)
        return raw

    def _flush_host(self, name: str) -> None:
        b = self.blobs[name]
        if b._root is not None:   # an alias: its storage owner carries the mirror
            return
        if b._head == _HEAD_HOST and b._host is not None and name in self._engine.tensors:
            t = self._engine.tensors[name]
            if t.dt:
                from . import blocked
                self._alloc.upload(t.handle, blocked.to_blocked(b._host.reshape(t.shape), t.dt))
            else:
                self._alloc.upload(t.handle, b._host)
            b._head = _HEAD_SYNCED

    # -- API ----------------------------------------------------------------
    def reshape(self) -> None:
        """Net::Reshape (net.cpp:843-849): propagate input-blob reshapes through the graph."""
        self._spec.reshape(self._pending_input_shapes)
        self._pending_input_shapes = {}
        self._engine.build()   # storage of unchanged element count (and its contents) is carried over
        self._graph = None     # a captured launch list refers to the old plan's buffers
        self._graph_key = None
        self._make_blobs()

    def set_params(self, params: Dict[str, List[np.ndarray]]) -> None:
        """Load a full parameter set {layer name: [blobs in reference order]} (extension)."""
        self._engine.set_params(params)
        self._make_blobs()

    def copy_from(self, path: str, bn_style: str = "variance", bn_eps: float = 1e-5) -> None:
        """Net::CopyTrainedLayersFrom (net.cpp:852-883): match by layer name.  ``bn_style`` names what the
        file's BN layers keep in their fourth blob: "variance" (this fork's layer, bn_layer.cpp:38-41) or
        "inv_std" (the legacy style; converted on load per python/bn_convert_style.py:21-24 with ``bn_eps``)."""
        from . import caffemodel
        if bn_style not in caffemodel.BN_STYLES:
            raise ValueError(f"bn_style must be one of {caffemodel.BN_STYLES}")
        loaded = caffemodel.read_caffemodel(path)
        if bn_style == "inv_std":
            loaded = caffemodel.convert_bn_style(loaded, caffemodel.bn_layer_names(self._spec), "inv_std_to_var", bn_eps)
        merged = {k: list(v) for k, v in self._engine.params.items()}
        for lname, blobs in loaded.items():
            if lname not in merged:
                continue  # "Ignoring source layer"
            if len(blobs) != len(merged[lname]):
                raise ValueError(f"Incompatible number of blobs for layer {lname}")
            merged[lname] = [np.asarray(b, np.float32).reshape(t.shape) for b, t in zip(blobs, merged[lname])]
        self.set_params(merged)

    def save(self, path: str, bn_style: str = "variance", bn_eps: float = 1e-5) -> None:
        """Net::ToProto + WriteProtoToBinaryFile; ``bn_style="inv_std"`` writes the legacy BN style
        (python/bn_convert_style.py:17-20)."""
        from . import caffemodel
        if bn_style not in caffemodel.BN_STYLES:
            raise ValueError(f"bn_style must be one of {caffemodel.BN_STYLES}")
        params = self._engine.params
        if bn_style == "inv_std":
            params = caffemodel.convert_bn_style(params, caffemodel.bn_layer_names(self._spec), "var_to_inv_std", bn_eps)
        caffemodel.write_caffemodel(path, self._spec, params)

    def _forward(self, start: int, end: int) -> None:
        if self._pending_input_shapes:
            self.reshape()
        # SyncedMemory::to_gpu for what the launches will read: the net inputs always; layer tops only when the
        # forward starts mid-net (a full forward overwrites every one of them, so a blob the caller merely
        # inspected through .data is not uploaded again)
        for name in self.blobs:
            if start > 0 or name in self._spec.inputs:
                self._flush_host(name)
        self._engine.forward(start, end)
        for name, b in self.blobs.items():
            if name in self._engine.tensors and name not in self._spec.inputs:
                b._head = _HEAD_DEVICE
        # aliases of inputs written in place are device-headed too
        for name in self._spec.inputs:
            if self.blobs[name]._head == _HEAD_NONE:
                self.blobs[name]._head = _HEAD_DEVICE

    def forward(self, blobs=None, start=None, end=None, **kwargs):
        """pycaffe ``_Net_forward`` (pycaffe.py:52-98)."""
        if blobs is None:
            blobs = []
        names = self._layer_names
        start_ind = names.index(start) if start is not None else 0
        if end is not None:
            end_ind = names.index(end)
            outputs = set([end] + list(blobs))
        else:
            end_ind = len(names) - 1
            outputs = set(self.outputs + list(blobs))
        if kwargs:
            if set(kwargs.keys()) != set(self.inputs):
                raise Exception("Input blob arguments do not match net inputs.")
            for in_, arr in kwargs.items():
                if arr.shape[0] != self.blobs[in_].shape[0]:
                    raise Exception("Input is not batch sized")
                self.blobs[in_].data[...] = arr
        self._forward(start_ind, end_ind)
        return {out: self.blobs[out].data for out in outputs}

    def set_input_device(self, name: str, tensor) -> None:
        """Fill net input ``name`` from a torch tensor already on the device (fp32, any shape with the blob's
        element count) without a host round trip; the blob's head moves to the device (extension used by
        bench.py and the multi-GPU driver)."""
        if name not in self._spec.inputs:
            raise KeyError(f"{name!r} is not a net input")
        if self._pending_input_shapes:
            self.reshape()
        dst = self.blobs[name].tensor          # marks the head DEVICE
        if tensor.numel() != dst.numel():
            raise ValueError(f"input {name}: {tensor.numel()} elements given, blob holds {dst.numel()}")
        dst.copy_(tensor.reshape(dst.shape))

    def forward_device(self, graph: bool = False):
        """Run the whole net on whatever is resident in HBM; returns nothing and does not
        synchronise (extension used by bench.py / the multi-GPU driver).  ``graph=True`` replays the
        launch list as one hipGraph (captured on first use; re-captured after reshape / parameter
        edits) -- worth it when the step is launch-bound, e.g. single-clip online recognition."""
        if graph:
            self._graph_replay()
        else:
            self._engine.forward()
        for name, b in self.blobs.items():
            if name in self._engine.tensors:
                b._head = _HEAD_DEVICE

    def _graph_replay(self) -> None:
        import torch
        eng = self._engine
        if eng._dirty_params or eng._dirty_groups:
            eng._sync_params()                       # uploads happen outside the capture (bumps the generation)
        key = eng.generation                         # monotonic: a stale capture can never match a new plan
        if getattr(self, "_graph_key", None) != key or getattr(self, "_graph", None) is None:
            eng.forward()                            # warm-up on the normal stream
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            # thread-local capture mode: only THIS thread's calls are checked against the capture -- a communicator's
            # watchdog thread (RCCL at N > 1) polling its events must not invalidate it
            with torch.cuda.graph(g, capture_error_mode="thread_local"):   # capture stream becomes torch's current stream
                eng.forward()
            self._graph, self._graph_key = g, eng.generation
        self._graph.replay()

    def op_labels(self) -> List[str]:
        return self._engine.op_labels()
