#!/usr/bin/env python
"""ECO-Lite inference benchmark on MI355X: clips/sec (whole job), roofline of the dominant
kernel, and the CPU oracle timed beside it.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload = BASELINE.json configs[1]: ECO-Lite, num_segments=16, 32 clips per GPU
(512 frames of 3x224x224 fp32), random-init (seeded) weights, synthetic frames already
resident in HBM when the timed region starts.  A "step" is one forward pass of the whole
path over the per-GPU clip batch; with N>1 the clip batch is sharded across ranks (weak
scaling, 32 clips per GPU) and each step ends with the one collective of the path, an RCCL
all-gather of the [32,400] logits.  value = clips processed by all ranks / max-over-ranks time.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# MI355X peaks from /opt/skills/guides/MI355X_MICROARCH.md
PEAK_FP32_MFMA_TFLOPS = 157.3
PEAK_HBM_GBS = 8000.0


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--clips-per-gpu", type=int, default=32)
    ap.add_argument("--segments", type=int, default=16)
    ap.add_argument("--variant", choices=["lite", "full"], default="lite")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-winograd", action="store_true", help="evaluate every convolution directly (A/B runs)")
    ap.add_argument("--cpu-clips", type=int, default=0, help="clips in the CPU sample (0 = auto, 10-30 s)")
    ap.add_argument("--profile-iters", type=int, default=3)
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N>1 must be launched with torch.distributed.run --nproc-per-node N")
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE {world}")

    import eco_amd as caffe
    from eco_amd import models, fillers
    from eco_amd import dist as eco_dist
    from eco_amd.netspec import NetSpec

    # ECO_BENCH_DEVICE / ECO_BENCH_BACKEND exist only to exercise the N>1 code path on a 1-GPU box
    # (all ranks on one device, gloo instead of RCCL); the driver's multi-GPU runs use neither.
    dev_index = int(os.environ.get("ECO_BENCH_DEVICE", local_rank))
    backend = os.environ.get("ECO_BENCH_BACKEND", "nccl")
    caffe.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    eco_dist.init_process_group(backend, device=dev if backend == "nccl" else None)  # RCCL over xGMI; no-op at world 1

    B, N = args.clips_per_gpu, args.segments
    gen = models.eco_lite_deploy if args.variant == "lite" else models.eco_full_deploy
    proto = gen(num_segments=N, num_clips=B)
    spec = NetSpec.from_prototxt(proto)
    params = fillers.synthetic_params(spec)            # same weights on every rank (same seed)
    net = caffe.Net(proto, caffe.TEST, params=params, winograd=not args.no_winograd)
    # rank r owns clips [r*B, (r+1)*B) of the global batch: a different seed per rank
    frames = fillers.synthetic_frames(B * N, seed=1234 + rank)
    net.blobs["data"].tensor.copy_(torch.from_numpy(frames).to(dev))
    logits = net.blobs["fc8"].tensor
    n_cls = logits.shape[1]
    gathered = torch.empty(world * B, n_cls, device=dev) if world > 1 else None

    def step() -> None:
        net.forward_device()
        if world > 1:
            eco_dist.all_gather_logits(logits, out=gathered)

    def fence() -> None:
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = 1e3 * elapsed / args.steps
    clips_per_s = world * B * args.steps / elapsed

    if rank != 0:
        dist.barrier()  # keep the communicator alive until rank 0 has finished reporting
        dist.destroy_process_group()
        return

    # ---- roofline of the dominant kernel: per-launch HIP-event times on the launch stream ----
    prof = net._engine.profile(args.profile_iters)
    by_kernel = {}
    for p in prof:
        k = by_kernel.setdefault(p["kernel"], dict(ms=0.0, flops=0, bytes=0, launches=0))
        k["ms"] += p["ms"]; k["flops"] += p["flops"]; k["bytes"] += p["bytes"]; k["launches"] += 1
    total_ms = sum(k["ms"] for k in by_kernel.values())
    dom_name, dom = max(by_kernel.items(), key=lambda kv: kv[1]["ms"])
    total_flops = spec.conv_fc_flops()
    t_flops = dom["flops"] / (PEAK_FP32_MFMA_TFLOPS * 1e12)
    t_bytes = dom["bytes"] / (PEAK_HBM_GBS * 1e9)
    if t_flops >= t_bytes:
        ach = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
        roofline = {"bound": "mfma", "achieved": round(ach, 3), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(ach / PEAK_FP32_MFMA_TFLOPS, 4), "traffic": None}
    else:
        ach = dom["bytes"] / (dom["ms"] * 1e-3) / 1e9
        roofline = {"bound": "hbm", "achieved": round(ach, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                    "frac": round(ach / PEAK_HBM_GBS, 4), "traffic": None}
    # HBM traffic of that kernel from the committed PMC summary (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in
    # separate passes, gfx950 FETCH half-count corrected; tools/summarize_profiles.py) -- bench.py cannot run
    # the profiler on itself, so this is the figure of the last profiled build, or null.
    try:
        with open(os.path.join(ROOT, "profiles", "hbm_traffic_latest.json")) as f:
            tr = json.load(f)
        roofline["traffic"] = round(tr["kernels"][dom_name]["hbm_bytes_per_launch"] / 1e9, 4)
        roofline["traffic_unit"] = "GB per launch (PMC, " + tr["source"] + ")"
        roofline["algorithmic_gb_per_launch"] = round(dom["bytes"] / dom["launches"] / 1e9, 4)
    except Exception:
        roofline["traffic"] = None
    roofline.update({
        "kernel": dom_name, "launches_per_step": dom["launches"],
        "timing": "HIP events on the launch stream around each eco_conv_forward call; for split-K plans that "
                  "includes the conv_splitk_reduce_kernel launch that follows the main kernel",
        "avg_launch_ms": round(dom["ms"] / dom["launches"], 4),
        "algorithmic_gflop_per_launch": round(dom["flops"] / dom["launches"] / 1e9, 3),
        "kernel_share_of_step": round(dom["ms"] / total_ms, 4),
        # gflop = algorithmic flops of the direct convolutions (what the reference computes); executed_gflop =
        # MFMA flops the launches actually issue (Winograd F(4x4,3x3) on the 3-D trunk needs 4x fewer)
        "whole_step": {"gflop": round(total_flops / 1e9, 2),
                       "tflops": round(total_flops / (ms_per_step * 1e-3) / 1e12, 2),
                       "frac_of_fp32_mfma_peak": round(total_flops / (ms_per_step * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4),
                       "executed_gflop": round(sum(p["flops"] for p in prof) / 1e9, 2),
                       "executed_tflops": round(sum(p["flops"] for p in prof) / (ms_per_step * 1e-3) / 1e12, 2)},
        "per_kernel_ms": {k: round(v["ms"], 3) for k, v in sorted(by_kernel.items(), key=lambda kv: -kv[1]["ms"])},
    })

    # ---- CPU baseline: the NumPy oracle (caffe cost structure: per-image im2col + SGEMM) ----
    cpu = None
    parity = None
    if not args.no_cpu_baseline and world == 1:  # reported on rank 0 at N=1 only
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import eco_oracle  # checker/baseline only; never on the product path
        try:
            cores = len(os.sched_getaffinity(0))
        except AttributeError:
            cores = os.cpu_count()
        spec1 = NetSpec.from_prototxt(gen(num_segments=N, num_clips=1))
        done, t_cpu, refs = 0, 0.0, []
        max_clips = args.cpu_clips or 8
        while done < max_clips:
            x1 = frames[done * N:(done + 1) * N]
            t1 = time.perf_counter()
            refs.append(eco_oracle.forward(spec1, params, {"data": x1})["fc8"])
            t_cpu += time.perf_counter() - t1
            done += 1
            if not args.cpu_clips and t_cpu >= 10.0:
                break
        cpu = {"value": round(done / t_cpu, 4), "unit": "clips/sec", "cores": cores, "kind": "port",
               "sample": f"{done} clip(s) of the same workload (num_segments={N}, the first clips of rank 0's "
                         f"batch), NumPy oracle with OpenBLAS sgemm on all cores, {t_cpu:.1f} s"}
        ref = np.concatenate(refs, 0)
        got = logits[:done].detach().cpu().numpy()
        parity = {"clips_checked": done, "max_rel_err": float(np.abs(got - ref).max() / np.abs(ref).max()),
                  "max_abs_logit": float(np.abs(ref).max()), "top1_agree": bool((got.argmax(1) == ref.argmax(1)).all())}

    line = {
        "metric": "clips/sec (whole node), ECO-%s N=%d 224x224 bs%d; top-1 logits vs CPU ref" % (
            "Lite" if args.variant == "lite" else "Full", N, B),
        "value": round(clips_per_s, 2), "unit": "clips/sec", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "ECO-%s num_segments=%d batch=%d/GPU fp32 (BASELINE.json configs[1]%s), "
                               "random-init seeded weights, synthetic 224x224 frames resident in HBM" % (
                                   "Lite" if args.variant == "lite" else "Full", N, B,
                                   "" if world == 1 else f" x{world} GPUs = configs[2] sharding"),
                   "global_batch": world * B, "num_segments": N, "parallelism": f"clip-batch dp{world}",
                   "launches_per_step": len(prof), "collective": "none" if world == 1 else "RCCL all-gather of logits"},
        "roofline": roofline, "cpu_baseline": cpu, "parity": parity,
    }
    print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
