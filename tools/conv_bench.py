#!/usr/bin/env python
"""Per-layer convolution micro-benchmark on the GPU: every distinct conv geometry of ECO-Lite at
the BASELINE configs[1] size (N=16, 32 clips), timed through the C ABI with HIP events.
Usage: python tools/conv_bench.py [--iters 5] [--filter res3]   (GPU box only)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch

from eco_amd import hip, models
from eco_amd.netspec import NetSpec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--filter", default="")
    ap.add_argument("--clips", type=int, default=32)
    ap.add_argument("--segments", type=int, default=16)
    ap.add_argument("--ksplit", type=int, default=0, help="override the split-K factor of fully split plans (experiments)")
    ap.add_argument("--no-span", action="store_true", help="run span-eligible plans with the CTAP gather kernel (A/B)")
    args = ap.parse_args()
    lib = hip.load()
    spec = NetSpec.from_prototxt(models.eco_lite_deploy(num_segments=args.segments, num_clips=args.clips))
    dev = torch.device("cuda", 0)
    seen = {}
    rows = []
    for L in spec.layers:
        if L.type != "Convolution" or args.filter not in L.name:
            continue
        g = L.geom
        key = (tuple(L.bottom_shapes[0]), g["cout"], tuple(g["kernel"]), tuple(g["stride"]), tuple(g["pad"]))
        if key in seen:
            seen[key].append(L.name)
            continue
        seen[key] = [L.name]
        bs, ts = L.bottom_shapes[0], L.top_shapes[0]
        geom = hip.conv_geom(bs[0], g["cin"], g["cout"], bs[2:], g["kernel"], g["stride"], g["pad"], ts[2:])
        plan = lib.conv_plan(geom)
        if args.no_span and plan.mode == 2:
            plan.mode = 1  # same packed-weight order
        if args.ksplit and plan.ksplit > 1 and plan.split_tiles == -(-g["cout"] // plan.bm) * -(-(int(np.prod(ts)) // g["cout"]) // plan.bn):
            plan.ksplit = args.ksplit
            plan.ws_bytes = args.ksplit * int(np.prod(ts)) * 4
        w = (np.random.default_rng(0).standard_normal((g["cout"], g["cin"]) + tuple(g["kernel"])) * 0.05).astype(np.float32)
        wp = np.zeros(plan.wp_elems, np.float32)
        kt = np.zeros(plan.ktab_elems, np.int32)
        lib.conv_pack_weights(geom, plan, w.ctypes.data, wp.ctypes.data, kt.ctypes.data)
        x = torch.randn(bs, device=dev)
        dwp, dkt = torch.from_numpy(wp).to(dev), torch.from_numpy(kt).to(dev)
        bias = torch.randn(g["cout"], device=dev)
        sc, sh = torch.rand(g["cout"], device=dev) + 0.5, torch.randn(g["cout"], device=dev)
        y = torch.empty(ts, device=dev)
        S = int(np.prod(ts[2:]))
        ep = hip.ConvEpilogue()
        ep.bias = bias.data_ptr()
        ep.residual, ep.raw = hip.null_view(), hip.null_view()
        ep.bn_scale, ep.bn_shift, ep.relu = sc.data_ptr(), sh.data_ptr(), 1
        ep.act = hip.plain_view(y.data_ptr(), g["cout"], S)
        stream = torch.cuda.current_stream().cuda_stream
        wsb = torch.empty(max(plan.ws_bytes // 4, 1), device=dev)
        run = lambda: lib.conv_forward(geom, plan, x.data_ptr(), dwp.data_ptr(), dkt.data_ptr(), ep,
                                       wsb.data_ptr() if plan.ws_bytes else None, stream)
        run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.iters
        flops = 2.0 * np.prod(ts) * g["cin"] * np.prod(g["kernel"])
        rows.append((L.name, plan.bm, plan.bn, flops / 1e9, ms, flops / ms / 1e9, key, plan.ksplit, plan.mode))
        del x, y
    tot_f = tot_t = 0.0
    print(f"{'layer':34s} bm  bn   GFLOP     ms    TFLOP/s  x  (frac of 157.3)")
    for name, bm, bn, gf, ms, tf, key, ks, mode in rows:
        mult = len(seen[key])
        tot_f += gf * mult
        tot_t += ms * mult
        print(f"{name:34s} {bm:3d} {bn:3d} {gf:8.1f} {ms:7.3f} {tf:8.1f}  x{mult}  {tf / 157.3:.3f}  ksplit={ks} mode={mode}")
    print(f"TOTAL conv: {tot_f:.1f} GFLOP in {tot_t:.2f} ms = {tot_f / tot_t:.1f} TFLOP/s ({tot_f / tot_t / 157.3:.3f} of peak)")


if __name__ == "__main__":
    main()
