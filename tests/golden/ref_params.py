"""Seeded weights / frames shared by tests/golden/make_reference_logits.py (which pushes them through the COMPILED
reference layers) and the tests that push them through the HIP path and the NumPy oracle.  TEST INFRASTRUCTURE.

Deliberately independent of the product package: NumPy only, no eco_amd import, so that the fixture
tests/golden/reference_logits.json does not depend on the product's fillers or graph layer.  Every parameter blob is drawn
from its own generator seeded with crc32(layer name) ^ seed ^ blob index, i.e. the values depend on the layer's NAME and
blob SHAPE only -- not on layer order, not on which other layers exist.

Distributions (the "random-init weights" of BASELINE.json configs[0]; the reference's fillers are caffe_3d/include/caffe/
filler.hpp -- msra :187-210 for weights; its BN statistics start at mean 0 / variance 0, which makes every BN multiply by
1/sqrt(eps) and overflows 30 BN deep, so statistics are drawn non-degenerate here):
  Convolution / InnerProduct weight  N(0, sqrt(2 / fan_in)), fan_in = count / shape[0]
  Convolution / InnerProduct bias    U(-0.1, 0.1)
  BN slope U(0.5, 1.5), bias U(-0.1, 0.1), running mean U(-0.1, 0.1), running variance U(0.5, 1.5)
"""
import math
import zlib

import numpy as np

RECIPE = "ref_params/1"


def _rng(layer_name: str, blob_index: int, seed: int) -> np.random.Generator:
    return np.random.default_rng([zlib.crc32(layer_name.encode()) & 0xFFFFFFFF, int(seed) & 0xFFFFFFFF, int(blob_index)])


def layer_blobs(layer_name: str, layer_type: str, shapes, seed: int = 2024):
    """The parameter blobs of one layer: shapes = list of blob shapes as the reference's LayerSetUp creates them
    (Convolution [cout, cin, k...] (+ [cout]); InnerProduct [out, in] (+ [out]); BN four [1, C, 1, 1])."""
    out = []
    for i, shp in enumerate(shapes):
        shp = tuple(int(d) for d in shp)
        r = _rng(layer_name, i, seed)
        if layer_type in ("Convolution", "InnerProduct"):
            if i == 0:
                fan_in = int(np.prod(shp)) // shp[0]
                out.append(r.normal(0.0, math.sqrt(2.0 / fan_in), size=shp).astype(np.float32))
            else:
                out.append(r.uniform(-0.1, 0.1, size=shp).astype(np.float32))
        elif layer_type == "BN":
            lo, hi = [(0.5, 1.5), (-0.1, 0.1), (-0.1, 0.1), (0.5, 1.5)][i]
            out.append(r.uniform(lo, hi, size=shp).astype(np.float32))
        else:
            raise ValueError(layer_type)
    return out


def frames(num_frames: int, height: int = 224, width: int = 224, seed: int = 77) -> np.ndarray:
    """VideoData TEST-phase output contract (video_data_layer.cpp:107-119, data_transformer.cpp:179-199): [F, 3, H, W]
    fp32 BGR planes, pixel values in [0, 255) minus the channel means (104, 117, 123)."""
    r = np.random.default_rng([0xF7A3E5, int(seed), int(num_frames), int(height), int(width)])
    x = r.uniform(0.0, 255.0, size=(num_frames, 3, height, width)).astype(np.float32)
    x -= np.array([104.0, 117.0, 123.0], np.float32).reshape(1, 3, 1, 1)
    return x


def blob_stats(a: np.ndarray, nsample: int = 16) -> dict:
    """Order-sensitive fingerprint of a blob: shape, float64 sum / abs-sum / a position-weighted sum, max|x| and `nsample`
    values at evenly spread flat positions (so a transposed or shifted blob does not pass)."""
    a = np.ascontiguousarray(a, np.float32)
    f = a.reshape(-1).astype(np.float64)
    n = f.size
    w = (np.arange(n, dtype=np.float64) % 251.0 + 1.0) / 251.0
    idx = np.unique(np.linspace(0, n - 1, min(nsample, n)).astype(np.int64))
    return dict(shape=[int(d) for d in a.shape], sum=float(f.sum()), abs_sum=float(np.abs(f).sum()),
                wsum=float((f * w).sum()), max_abs=float(np.abs(f).max()) if n else 0.0,
                idx=[int(i) for i in idx], val=[float(a.reshape(-1)[i]) for i in idx])


def check_stats(a: np.ndarray, st: dict, rtol: float):
    """Compare a blob with its fixture fingerprint; returns (ok, message).  Sums are compared against rtol * abs_sum
    (rounding of a sum of n terms), samples against rtol * max_abs."""
    a = np.ascontiguousarray(a, np.float32)
    if [int(d) for d in a.shape] != st["shape"]:
        return False, f"shape {list(a.shape)} != {st['shape']}"
    got = blob_stats(a, nsample=len(st["idx"]))
    scale = max(st["abs_sum"], 1e-30)
    for k in ("sum", "abs_sum", "wsum"):
        if abs(got[k] - st[k]) > rtol * scale:
            return False, f"{k}: {got[k]!r} vs {st[k]!r} (abs_sum {st['abs_sum']!r})"
    flat = a.reshape(-1)
    tol = rtol * max(st["max_abs"], 1e-30)
    for i, v in zip(st["idx"], st["val"]):
        if abs(float(flat[i]) - v) > tol:
            return False, f"value at flat index {i}: {float(flat[i])!r} vs {v!r} (tol {tol:.3e})"
    return True, ""
