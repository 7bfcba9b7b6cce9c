"""Parameter initialisation ("random-init weights") for the ECO layers.

Distributions restate the reference fillers (caffe_3d/include/caffe/filler.hpp):
``constant`` (:30-48), ``uniform`` (:52-66), ``gaussian`` (:70-108, dense form),
``xavier`` (:145-168: U(-sqrt(3/n), +sqrt(3/n)), n = fan_in = count/shape[0] by
default, fan_out or their average per ``variance_norm``), ``msra`` (:187-210:
N(0, sqrt(2/n))).  The reference draws from boost::mt19937 seeded from the
process RNG (common.cpp), which is not reproducible here, so values come from a
seeded ``numpy.random.Generator`` -- same distributions, different stream.

BN (layers/bn_layer.cpp:24-41): scale/shift from ``slope_filler``/``bias_filler``,
running mean 0, running variance 0 (1 if frozen).  With variance 0 every BN
multiplies by 1/sqrt(eps) ~ 316 and a 30-BN-deep net overflows, so
``synthetic_params`` (the benchmark/parity weight set) overrides
BN statistics with non-degenerate values and gives biases a non-zero range so
that every term of the fused epilogue is exercised.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import numpy as np

from .netspec import NetSpec, param_shapes
from .prototxt import Message


def fill(shape, filler: Optional[Message], rng: np.random.Generator) -> np.ndarray:
    filler = filler if filler is not None else Message()
    kind = str(filler.get("type", "constant"))
    count = int(np.prod(shape))
    if kind == "constant":
        return np.full(shape, float(filler.get("value", 0.0)), dtype=np.float32)
    if kind == "uniform":
        lo, hi = float(filler.get("min", 0.0)), float(filler.get("max", 1.0))
        return rng.uniform(lo, hi, size=shape).astype(np.float32)
    if kind == "gaussian":
        if int(filler.get("sparse", -1)) >= 0:
            raise ValueError("gaussian filler: sparse >= 0 not supported")
        return rng.normal(float(filler.get("mean", 0.0)), float(filler.get("std", 1.0)),
                          size=shape).astype(np.float32)
    if kind in ("xavier", "msra"):
        fan_in = count // shape[0]
        fan_out = count // shape[1] if len(shape) > 1 else count
        norm = str(filler.get("variance_norm", "FAN_IN"))
        n = {"FAN_IN": fan_in, "FAN_OUT": fan_out, "AVERAGE": (fan_in + fan_out) / 2.0}[norm]
        if kind == "xavier":
            s = math.sqrt(3.0 / n)
            return rng.uniform(-s, s, size=shape).astype(np.float32)
        return rng.normal(0.0, math.sqrt(2.0 / n), size=shape).astype(np.float32)
    raise ValueError(f"Unknown filler name: {kind}")


def filler_params(spec: NetSpec, seed: int = 0) -> Dict[str, List[np.ndarray]]:
    """Initialise every layer exactly as its prototxt fillers say (``LayerSetUp``)."""
    rng = np.random.default_rng(seed)
    out: Dict[str, List[np.ndarray]] = {}
    for L in spec.layers:
        shapes = param_shapes(L)
        if not shapes:
            continue
        if L.type == "Convolution":
            p = L.param.msg("convolution_param")
            blobs = [fill(shapes[0], p.get("weight_filler"), rng)]
            if len(shapes) > 1:
                blobs.append(fill(shapes[1], p.get("bias_filler"), rng))
        elif L.type == "InnerProduct":
            p = L.param.msg("inner_product_param")
            blobs = [fill(shapes[0], p.get("weight_filler"), rng)]
            if len(shapes) > 1:
                blobs.append(fill(shapes[1], p.get("bias_filler"), rng))
        elif L.type == "BN":
            p = L.param.msg("bn_param")
            blobs = [fill(shapes[0], p.get("slope_filler"), rng),
                     fill(shapes[1], p.get("bias_filler"), rng),
                     np.zeros(shapes[2], np.float32),
                     np.full(shapes[3], 1.0 if L.geom["frozen"] else 0.0, np.float32)]
        else:  # pragma: no cover
            raise AssertionError(L.type)
        out[L.name] = blobs
    return out


def synthetic_params(spec: NetSpec, seed: int = 4321, weight_init: str = "msra") -> Dict[str, List[np.ndarray]]:
    """Seeded benchmark/parity weights.

    conv/fc weights: ``msra`` (default; keeps activations O(1) through 30 BN+ReLU
    stages so a *relative* logit tolerance is not vacuous) or ``xavier``;
    biases U(-0.1, 0.1); BN scale U(0.5, 1.5), shift U(-0.1, 0.1), running mean
    U(-0.1, 0.1), running variance U(0.5, 1.5)."""
    rng = np.random.default_rng(seed)
    wf = Message()
    wf.add("type", weight_init)
    out: Dict[str, List[np.ndarray]] = {}
    for L in spec.layers:
        shapes = param_shapes(L)
        if not shapes:
            continue
        if L.type in ("Convolution", "InnerProduct"):
            blobs = [fill(shapes[0], wf, rng)]
            if len(shapes) > 1:
                blobs.append(rng.uniform(-0.1, 0.1, size=shapes[1]).astype(np.float32))
        else:  # BN
            blobs = [rng.uniform(0.5, 1.5, size=shapes[0]).astype(np.float32),
                     rng.uniform(-0.1, 0.1, size=shapes[1]).astype(np.float32),
                     rng.uniform(-0.1, 0.1, size=shapes[2]).astype(np.float32),
                     rng.uniform(0.5, 1.5, size=shapes[3]).astype(np.float32)]
        out[L.name] = blobs
    return out


def synthetic_frames(num_frames: int, height: int = 224, width: int = 224, seed: int = 1234) -> np.ndarray:
    """VideoData output contract (layers/video_data_layer.cpp:107-119,
    data_transformer.cpp:179-199): ``[F, 3, H, W]`` fp32, BGR planes, uniform
    [0,255) pixel values with the channel means (104, 117, 123) subtracted."""
    rng = np.random.default_rng(seed)
    x = rng.uniform(0.0, 255.0, size=(num_frames, 3, height, width)).astype(np.float32)
    x -= np.array([104.0, 117.0, 123.0], np.float32).reshape(1, 3, 1, 1)
    return x
