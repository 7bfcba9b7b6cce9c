"""Host-side mirror of the channel-blocked ("NC8") storage of csrc/eco_blocked.hip.

The blocked bf16 path keeps activations as ``X[n][c/8][spatial...][c%8]`` (bf16 bits).  Callers never see that layout: inputs and logits are plain fp32, and when a caller
looks at an intermediate blob (``net.blobs[name].data``) the raw storage is converted back to the reference's
``N,C,[D,]H,W`` fp32 here.  NumPy only -- plumbing, no arithmetic beyond the bf16 <-> fp32 bit moves."""
from __future__ import annotations

from typing import Sequence

import numpy as np

DT_BF16 = 1
STORAGE = {DT_BF16: np.uint16}


def bf16_bits(x: np.ndarray) -> np.ndarray:
    """fp32 -> bf16 bits, round to nearest even (what v_cvt_pk_bf16_f32 does for finite values)."""
    u = np.ascontiguousarray(x, np.float32).view(np.uint32)
    return ((u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) >> np.uint32(16)).astype(np.uint16)


def bf16_to_f32(bits: np.ndarray) -> np.ndarray:
    return (np.ascontiguousarray(bits, np.uint16).astype(np.uint32) << np.uint32(16)).view(np.float32)


def bf16_round(x: np.ndarray) -> np.ndarray:
    """fp32 values rounded to the nearest bf16 (still fp32)."""
    return bf16_to_f32(bf16_bits(x)).reshape(np.shape(x))


def to_blocked(x: np.ndarray, dt: int) -> np.ndarray:
    """N,C,spatial... fp32 -> flat storage array of N,C/8,spatial...,8 in the path's storage type."""
    x = np.ascontiguousarray(x, np.float32)
    n, c = x.shape[:2]
    if c % 8:
        raise ValueError(f"{c} channels are not a multiple of the 8-channel block")
    sp = int(np.prod(x.shape[2:], dtype=np.int64))
    b = x.reshape(n, c // 8, 8, sp).transpose(0, 1, 3, 2)
    b = np.ascontiguousarray(b).reshape(-1)
    return bf16_bits(b) if dt == DT_BF16 else b


def from_blocked(raw: np.ndarray, shape: Sequence[int], dt: int) -> np.ndarray:
    """Inverse of to_blocked: storage array -> fp32 array of the logical N,C,spatial... shape."""
    shape = tuple(int(s) for s in shape)
    n, c = shape[:2]
    sp = int(np.prod(shape[2:], dtype=np.int64))
    f = bf16_to_f32(raw) if dt == DT_BF16 else np.asarray(raw, np.float32)
    f = f.reshape(-1)[: n * c * sp].reshape(n, c // 8, sp, 8).transpose(0, 1, 3, 2)
    return np.ascontiguousarray(f).reshape(shape)
