#!/usr/bin/env python
"""Algorithmic work of the ECO forward path per layer, from the prototxt shapes alone (no GPU):
FLOPs = 2*MACs of Convolution + InnerProduct; bytes under the *fused* model (each conv/fc: input +
weights + output once; BN/ReLU/bias/dropout folded; concat/reshape/split/permute free; each pool: in + out;
each eltwise: one extra operand read; one extra write where a raw sum feeds both a BN and a later sum) and
under the *layer-by-layer* model (every layer reads its bottoms and writes its tops once).  These are the
figures SURVEY.md section 8(d) quotes and bench.py's roofline uses; `main` re-derives and asserts them.

    python tools/roofline.py [--variant lite|full] [--segments 16] [--clips 32] [--per-layer]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from eco_amd import models  # noqa: E402
from eco_amd.netspec import NetSpec, param_shapes  # noqa: E402

PEAK_FP32_MFMA = 157.3e12   # /opt/skills/guides/MI355X_MICROARCH.md
PEAK_HBM = 8.0e12


def _n(shape):
    p = 1
    for d in shape:
        p *= int(d)
    return p


def layer_table(spec: NetSpec):
    """[(name, type, flops, fused_bytes, layerwise_bytes)] in execution order."""
    consumers = {}
    for L in spec.layers:
        for b in L.bottoms:
            consumers.setdefault(b, []).append(L)
    rows = []
    for L in spec.layers:
        nin = sum(_n(s) for s in L.bottom_shapes)
        nout = sum(_n(s) for s in L.top_shapes)
        npar = sum(_n(s) for s in param_shapes(L))
        flops = fused = 0
        if L.type == "Convolution":
            g = L.geom
            flops = 2 * _n(L.top_shapes[0]) * g["cin"] * _n(g["kernel"])
            fused = 4 * (nin + npar + nout)
        elif L.type == "InnerProduct":
            g = L.geom
            flops = 2 * g["M"] * g["num_output"] * g["K"]
            fused = 4 * (nin + npar + nout)
        elif L.type == "Pooling":
            fused = 4 * (nin + nout)
        elif L.type == "Eltwise":
            fused = 4 * _n(L.top_shapes[0])      # the operand the producing conv's epilogue has to read
        elif L.type == "BN":
            fused = 4 * npar
        layerwise = 4 * (nin + nout + npar) if L.type not in ("Split", "Reshape", "Dropout") else 0
        rows.append([L.name, L.type, flops, fused, layerwise])
    # A sum that feeds both its BN and the next sum (or a conv output that feeds a BN and a sum) is written
    # twice by the fused epilogue, raw and activated: blobs consumed (through their Split) by both types
    for L in spec.layers:
        if L.type == "Split":
            kinds = set()
            for t in L.tops:
                kinds |= {c.type for c in consumers.get(t, [])}
            if "BN" in kinds and "Eltwise" in kinds:
                rows.append([L.bottoms[0] + " (raw + activated)", "dual-write", 0, 4 * _n(L.top_shapes[0]), 0])
    return rows


def totals(spec: NetSpec):
    rows = layer_table(spec)
    return (sum(r[2] for r in rows), sum(r[3] for r in rows), sum(r[4] for r in rows))


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--variant", choices=["lite", "full"], default="lite")
    ap.add_argument("--segments", type=int, default=16)
    ap.add_argument("--clips", type=int, default=32)
    ap.add_argument("--per-layer", action="store_true")
    args = ap.parse_args()
    gen = models.eco_lite_deploy if args.variant == "lite" else models.eco_full_deploy
    spec = NetSpec.from_prototxt(gen(num_segments=args.segments, num_clips=args.clips))
    rows = layer_table(spec)
    if args.per_layer:
        print(f"{'layer':36s} {'type':14s} {'GFLOP':>10s} {'fused MB':>10s} {'flop/B':>8s} {'bound':>5s} {'floor us':>9s}")
        for name, typ, fl, fb, _ in rows:
            if not fl and not fb:
                continue
            t_f, t_b = fl / PEAK_FP32_MFMA, fb / PEAK_HBM
            print(f"{name:36s} {typ:14s} {fl / 1e9:10.3f} {fb / 1e6:10.2f} {fl / fb if fb else 0:8.1f} "
                  f"{'mfma' if t_f >= t_b else 'hbm':>5s} {max(t_f, t_b) * 1e6:9.1f}")
    fl, fb, lb = sum(r[2] for r in rows), sum(r[3] for r in rows), sum(r[4] for r in rows)
    print(f"ECO-{args.variant} N={args.segments} B={args.clips}: {fl / 1e9:.2f} GFLOP ({fl / 1e9 / args.clips:.2f} per clip), "
          f"fused {fb / 1e9:.2f} GB, layer-by-layer {lb / 1e9:.2f} GB, {fl / fb:.0f} flop/B; "
          f"floors: {fl / PEAK_FP32_MFMA * 1e3:.2f} ms fp32 MFMA, {fb / PEAK_HBM * 1e3:.2f} ms HBM "
          f"-> ceiling {args.clips / max(fl / PEAK_FP32_MFMA, fb / PEAK_HBM):.0f} clips/s")


if __name__ == "__main__":
    main()
