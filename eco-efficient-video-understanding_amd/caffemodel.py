"""Minimal reader/writer for binary ``.caffemodel`` files (protobuf wire format, no protoc).

Only what ``Net::CopyTrainedLayersFrom`` / ``Net::ToProto`` need for the ECO path
(caffe_3d/src/caffe/net.cpp:852-883,945-960; blob.cpp:472-505; schema caffe.proto:5-20,62-97,282-301):

    NetParameter { name = 1; repeated LayerParameter layer = 100; repeated V1LayerParameter layers = 2 }
    LayerParameter { name = 1; type = 2; repeated BlobProto blobs = 7 }
    V1LayerParameter { name = 4; repeated BlobProto blobs = 6 }
    BlobProto { shape = 7 { repeated int64 dim = 1 [packed] }; repeated float data = 5 [packed];
                legacy num/channels/height/width = 1..4 }

Weights are matched to the net by *layer name*, exactly like the reference.

BN blob styles.  This fork's BN layer keeps (scale, shift, running mean, running **variance**)
(bn_layer.cpp:29-41,138-164); older releases of the same layer kept the running **inverse std**
1/sqrt(var + eps) in the fourth blob, and the reference ships the converter between the two
(caffe_3d/python/bn_convert_style.py:13-30; gen_bn_inference.py:121-134 takes the style as a flag).
``convert_bn_style`` is that converter on a parameter dict; ``Net.copy_from(path, bn_style="inv_std")``
applies it while loading so that an inv-std-style file is not silently read as variances.
"""
from __future__ import annotations

import struct
from typing import Dict, Iterator, List, Tuple

import numpy as np


class CaffemodelError(ValueError):
    pass


BN_STYLES = ("variance", "inv_std")


def convert_bn_style(params: Dict[str, List[np.ndarray]], bn_layers, conversion: str,
                     eps: float = 1e-5) -> Dict[str, List[np.ndarray]]:
    """bn_convert_style.py:13-30 on a {layer: [blobs]} dict: for every layer in ``bn_layers`` (the reference
    takes every layer whose name ends in ``_bn``; pass those names, or the BN-typed layers of a NetSpec)
    rewrite the fourth blob, ``var_to_inv_std``: (var + eps)^-0.5, ``inv_std_to_var``: inv_std^-2 - eps.
    Returns a new dict; the other blobs are shared."""
    if conversion not in ("var_to_inv_std", "inv_std_to_var"):
        raise ValueError(f"Unknown conversion {conversion}")
    out = {k: list(v) for k, v in params.items()}
    for name in bn_layers:
        if name not in out:
            continue
        blobs = out[name]
        if len(blobs) != 4:
            raise CaffemodelError(f"BN layer {name}: expected 4 blobs (scale, shift, mean, var), got {len(blobs)}")
        b3 = np.asarray(blobs[3], np.float32)
        if conversion == "var_to_inv_std":
            new = np.power(b3 + np.float32(eps), np.float32(-0.5))
        else:
            new = np.power(b3, np.float32(-2)) - np.float32(eps)
        blobs[3] = new.astype(np.float32)
    return out


def bn_layer_names(spec) -> List[str]:
    """BN-typed layers of a NetSpec (a superset of the reference's ``name.endswith('_bn')`` rule for every
    ECO prototxt, where all BN layers are named ``*_bn``)."""
    return [L.name for L in spec.layers if L.type == "BN"]


def _varint(buf: bytes, pos: int) -> Tuple[int, int]:
    result = shift = 0
    while True:
        if pos >= len(buf):
            raise CaffemodelError("truncated varint")
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7
        if shift > 70:
            raise CaffemodelError("varint too long")


def _fields(buf: bytes) -> Iterator[Tuple[int, int, object]]:
    """Yield (field number, wire type, value); length-delimited values are memoryview slices."""
    mv = memoryview(buf)
    pos, n = 0, len(buf)
    while pos < n:
        key, pos = _varint(buf, pos)
        fno, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 1:
            v = bytes(mv[pos:pos + 8]); pos += 8
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            if pos + ln > n:
                raise CaffemodelError("truncated length-delimited field")
            v = mv[pos:pos + ln]; pos += ln
        elif wt == 5:
            v = bytes(mv[pos:pos + 4]); pos += 4
        else:
            raise CaffemodelError(f"unsupported wire type {wt}")
        yield fno, wt, v


def _parse_blob(buf) -> np.ndarray:
    buf = bytes(buf)
    shape: List[int] = []
    legacy = {}
    chunks: List[np.ndarray] = []
    for fno, wt, v in _fields(buf):
        if fno == 7 and wt == 2:                      # BlobShape
            for f2, w2, v2 in _fields(bytes(v)):
                if f2 == 1 and w2 == 2:               # packed int64 dims
                    b2, p = bytes(v2), 0
                    while p < len(b2):
                        d, p = _varint(b2, p)
                        shape.append(d)
                elif f2 == 1 and w2 == 0:
                    shape.append(v2)
        elif fno == 5 and wt == 2:                    # packed float data
            chunks.append(np.frombuffer(v, dtype="<f4"))
        elif fno == 5 and wt == 5:                    # unpacked float
            chunks.append(np.frombuffer(v, dtype="<f4"))
        elif fno in (1, 2, 3, 4) and wt == 0:
            legacy[fno] = v
    data = np.concatenate(chunks) if chunks else np.zeros(0, np.float32)
    if not shape and legacy:                          # blob.cpp:478-486: legacy 4-D
        shape = [legacy.get(i, 1) for i in (1, 2, 3, 4)]
    if not shape:
        shape = [data.size]
    if int(np.prod(shape)) != data.size:
        raise CaffemodelError(f"blob shape {shape} does not match {data.size} values")
    return data.astype(np.float32).reshape(shape)


def read_caffemodel(path: str) -> Dict[str, List[np.ndarray]]:
    """{layer name: [blob, ...]} for every layer that carries blobs."""
    with open(path, "rb") as f:
        buf = f.read()
    out: Dict[str, List[np.ndarray]] = {}
    for fno, wt, v in _fields(buf):
        if wt != 2 or fno not in (100, 2):
            continue
        name_field, blob_field = (1, 7) if fno == 100 else (4, 6)
        name, blobs = None, []
        for f2, w2, v2 in _fields(bytes(v)):
            if f2 == name_field and w2 == 2:
                name = bytes(v2).decode("utf-8")
            elif f2 == blob_field and w2 == 2:
                blobs.append(_parse_blob(v2))
        if name is not None and blobs:
            out[name] = blobs
    return out


def _enc_varint(v: int) -> bytes:
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _ld(fno: int, payload: bytes) -> bytes:
    return _enc_varint((fno << 3) | 2) + _enc_varint(len(payload)) + payload


def write_caffemodel(path: str, spec, params: Dict[str, List[np.ndarray]]) -> None:
    """Serialise name/type/blobs of every parameterised layer (``Net::ToProto`` subset)."""
    body = _ld(1, spec.name.encode("utf-8"))
    for L in spec.layers:
        if L.name not in params:
            continue
        lp = _ld(1, L.name.encode("utf-8")) + _ld(2, L.type.encode("utf-8"))
        for b in params[L.name]:
            b = np.ascontiguousarray(b, dtype="<f4")
            dims = b"".join(_enc_varint(int(d)) for d in b.shape)
            lp += _ld(7, _ld(7, _ld(1, dims)) + _ld(5, b.tobytes()))
        body += _ld(100, lp)
    with open(path, "wb") as f:
        f.write(body)
