#!/usr/bin/env python
"""`caffe time`-style report for the HIP path (tools/caffe.cpp:276-360 `time()`, forward only): average
forward time of every launch of the plan over --iterations, with the algorithmic GFLOP / MB of the launch
and where it sits against the MI355X roofline.  GPU box only.

    python tools/eco_time.py --variant lite --segments 16 --clips 32 --iterations 10 [--no-fuse]
    python tools/eco_time.py --model path/to/deploy.prototxt [--weights x.caffemodel]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

PEAK_MFMA = {"f32": 157.3e12, "bf16": 2500e12}
PEAK_HBM = 8.0e12


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", help="deploy prototxt (default: generated ECO graph)")
    ap.add_argument("--weights", help=".caffemodel (default: seeded synthetic parameters)")
    ap.add_argument("--variant", choices=["lite", "full"], default="lite")
    ap.add_argument("--segments", type=int, default=16)
    ap.add_argument("--clips", type=int, default=32)
    ap.add_argument("--iterations", type=int, default=10)
    ap.add_argument("--no-fuse", action="store_true", help="one launch per prototxt layer, like the reference executor")
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="f32")
    ap.add_argument("--no-winograd", action="store_true")
    args = ap.parse_args()

    import eco_amd as caffe
    from eco_amd import fillers, models
    from eco_amd.netspec import NetSpec

    caffe.set_device(0)
    if args.model:
        proto = args.model
    else:
        gen = models.eco_lite_deploy if args.variant == "lite" else models.eco_full_deploy
        proto = gen(num_segments=args.segments, num_clips=args.clips)
    spec = NetSpec.from_prototxt(proto)
    kw = dict(fuse=not args.no_fuse, dtype=args.dtype, winograd=not args.no_winograd)
    if args.weights:
        net = caffe.Net(proto, args.weights, caffe.TEST, **kw)
    else:
        net = caffe.Net(proto, caffe.TEST, params=fillers.synthetic_params(spec), **kw)
    import torch
    for name in spec.inputs:   # the VideoData contract's value range (mean-subtracted pixels)
        net.set_input_device(name, torch.from_numpy(fillers.synthetic_frames(spec.input_shapes[name][0])).cuda()
                             if len(spec.input_shapes[name]) == 4 and spec.input_shapes[name][1] == 3
                             else torch.zeros(spec.input_shapes[name]).cuda())
    net.forward_device()
    prof = net._engine.profile(args.iterations)
    print(f"*** Benchmark begins ***  Testing for {args.iterations} iterations.")
    tot = 0.0
    for p in prof:
        ms = p["ms"]
        tot += ms
        fl, by = p.get("flops", 0), p.get("bytes", 0)
        floor = max(fl / PEAK_MFMA[args.dtype], by / PEAK_HBM) * 1e3
        frac = f"{floor / ms:5.2f}" if ms > 0 and floor > 0 else "    -"
        print(f"{p['label']:>44s}\tforward: {ms:8.4f} ms.  {fl / 1e9:9.3f} GFLOP {by / 1e6:9.2f} MB  "
              f"roofline frac {frac}  [{p.get('kernel', '')}]")
    flops = spec.conv_fc_flops()
    print(f"Average Forward pass: {tot:.4f} ms.  ({flops / 1e9:.1f} GFLOP of direct convolutions -> "
          f"{flops / tot / 1e9:.1f} TFLOP/s equivalent; launches execute {sum(p.get('flops', 0) for p in prof) / 1e9:.1f} GFLOP)")
    print("*** Benchmark ends ***")


if __name__ == "__main__":
    main()
