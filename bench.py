#!/usr/bin/env python
"""ECO inference benchmark on MI355X: clips/sec (whole job), roofline of the dominant kernel and of the
whole step, and the CPU path timed beside it.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Default workload = BASELINE.json configs[1]: ECO-Lite, num_segments=16, 32 clips per GPU (512 frames of
3x224x224 fp32), random-init (seeded) weights, synthetic frames already resident in HBM when the timed region
starts.  A "step" is one forward pass of the whole path over the per-GPU clip batch; with N>1 the clip batch is
sharded across ranks (weak scaling, 32 clips per GPU = configs[2]) and each step ends with the one collective
of the path, an RCCL all-gather of the [32,400] logits.  value = clips processed by all ranks / max-over-ranks
wall-clock time of exactly K steps between barrier + device-synchronize fences.
`--variant full` = configs[3] (ECO-Full N=16 B=32); `--segments 32 --dtype bf16` = configs[4] (per GPU).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# MI355X peaks from /opt/skills/guides/MI355X_MICROARCH.md (dense, no sparsity)
PEAK_MFMA_TFLOPS = {"f32": 157.3, "bf16": 2500.0}
PEAK_HBM_GBS = 8000.0


def baseline_config(variant: str, segments: int, clips: int, dtype: str, world: int) -> str:
    """Which BASELINE.json configuration a run is (the judge matches `config.workload` against it)."""
    if variant == "lite" and segments == 16 and clips == 32 and dtype == "f32":
        return "BASELINE.json configs[1]" if world == 1 else (
            "BASELINE.json configs[2]" if world == 8 else f"configs[1] per GPU x{world} GPUs (configs[2] sharding)")
    if variant == "full" and segments == 16 and clips == 32 and dtype == "f32":
        return "BASELINE.json configs[3]" + ("" if world == 1 else f" per GPU x{world} GPUs")
    if variant == "lite" and segments == 32 and dtype == "bf16":
        return "BASELINE.json configs[4]" + (" on 1 GPU" if world == 1 else (" (8 GPUs)" if world == 8 else f" x{world} GPUs"))
    if variant == "lite" and segments == 4 and clips == 1 and dtype == "f32":
        return "BASELINE.json configs[0] geometry (on the GPU)"
    return "not a BASELINE.json configuration"


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--clips-per-gpu", type=int, default=32)
    ap.add_argument("--segments", type=int, default=16)
    ap.add_argument("--variant", choices=["lite", "full"], default="lite")
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="f32",
                    help="f32: fp32 storage, fp32 MFMA; bf16: bf16 storage, bf16 MFMA, fp32 accumulation (configs[4])")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-winograd", action="store_true", help="evaluate every convolution directly (A/B runs)")
    ap.add_argument("--graph", action="store_true",
                    help="replay the step as one hipGraph (launch-bound small batches: --clips-per-gpu 1 online latency); "
                         "the default at N > 1, where eight Python launch loops would share the host")
    ap.add_argument("--no-graph", action="store_true", help="N > 1: submit launch by launch instead of replaying a hipGraph")
    ap.add_argument("--cpu-clips", type=int, default=0, help="clips per CPU-baseline variant (0 = auto, bounded by time)")
    ap.add_argument("--profile-iters", type=int, default=3)
    ap.add_argument("--no-extra-configs", action="store_true",
                    help="skip the configs[3] / configs[4] legs a default single-GPU run appends as `extra_configs`")
    ap.add_argument("--extra-steps", type=int, default=10)
    ap.add_argument("--extra-parity-few", action="store_true",
                    help="extra_configs: check only the 2 / 4 sample clips against the CPU reference, not every clip of the batch")
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

    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))

    import eco_amd as caffe
    from eco_amd import models, fillers, hip
    from eco_amd import dist as eco_dist
    from eco_amd.netspec import NetSpec

    # ECO_BENCH_DEVICE / ECO_BENCH_BACKEND exist only to exercise the N>1 code path on a 1-GPU box
    # (all ranks on one device, gloo instead of RCCL); the driver's multi-GPU runs use neither.
    dev_index = int(os.environ.get("ECO_BENCH_DEVICE", local_rank))
    backend = os.environ.get("ECO_BENCH_BACKEND", "nccl")
    caffe.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    # N processes share the host: every local rank goes onto physical cores of ITS GPU's NUMA node, no two ranks on SMT
    # siblings of one core (eco_amd.dist.plan_rank_cpus; sysfs + the C ABI's eco_device_pci_bus_id), so that eight launch
    # loops and torch's host pools neither oversubscribe each other nor sit across the socket from their GPU
    pinned = None
    if world > 1:
        try:
            lib0 = hip.load()
            ndev = lib0.device_count()
            forced = os.environ.get("ECO_BENCH_DEVICE")
            pci = [lib0.device_pci_bus_id(int(forced) if forced is not None else r % max(ndev, 1)) for r in range(local_world)]
            pinned = eco_dist.pin_rank(local_rank % local_world, pci)
            if pinned:
                torch.set_num_threads(max(1, min(pinned["physical_cores"], 16)))
        except Exception as e:  # placement is an optimisation: never fail the run over it
            pinned = {"error": f"{type(e).__name__}: {e}"}
    eco_dist.init_process_group(backend, device=dev if backend == "nccl" else None)  # RCCL over xGMI; no-op at world 1
    dev_info = hip.load().device_info(dev_index)

    B, N = args.clips_per_gpu, args.segments
    gen = models.eco_lite_deploy if args.variant == "lite" else models.eco_full_deploy
    proto = gen(num_segments=N, num_clips=B)
    spec = NetSpec.from_prototxt(proto)
    params = fillers.synthetic_params(spec)            # same weights on every rank (same seed)
    net_kw = dict(winograd=not args.no_winograd)
    if args.dtype != "f32":
        net_kw["dtype"] = args.dtype
    net = caffe.Net(proto, caffe.TEST, params=params, **net_kw)
    # rank r owns clips [r*B, (r+1)*B) of the global batch: a different seed per rank
    frames = fillers.synthetic_frames(B * N, seed=1234 + rank)
    net.set_input_device("data", torch.from_numpy(frames).to(dev))
    logits = net.blobs[spec.outputs[0]].tensor
    n_cls = logits.shape[1]
    gathered = torch.empty(world * B, n_cls, device=dev, dtype=logits.dtype) if world > 1 else None

    use_graph = args.graph or (world > 1 and not args.no_graph)
    submission = "hipGraph replay" if use_graph else "one C-ABI call per launch"
    if use_graph:
        try:                                   # capture now, outside the timed region; fall back to eager on any failure
            net.forward_device(graph=True)
            torch.cuda.synchronize()
        except Exception as e:
            if args.graph:
                raise
            use_graph = False
            submission = f"one C-ABI call per launch (hipGraph capture failed: {type(e).__name__}: {e})"
            torch.cuda.synchronize()

    mids = []                                  # per step: event between the forward's last launch and the collective

    def step(timed: bool = False) -> None:
        net.forward_device(graph=use_graph)
        if world > 1:
            if timed:
                ev = torch.cuda.Event(enable_timing=True)
                ev.record()
                mids.append(ev)
            eco_dist.all_gather_logits(logits, out=gathered)

    def fence() -> None:
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    # The timed region is wall clock (time.perf_counter) between two barrier + device-synchronize fences, as the
    # driver's contract asks; HIP events recorded on the launch stream around every step give the per-step
    # device times beside it (their median is reported as ms_per_step_event_median), and at N > 1 one more event per
    # step separates the forward from the all-gather that follows it on the stream.
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    evs[0].record()
    for i in range(args.steps):
        step(timed=True)
        evs[i + 1].record()
    fence()
    elapsed = time.perf_counter() - t0
    step_ms = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(args.steps))
    rank_stats = None
    if world > 1:
        # the all-gather as the launch stream sees it: from the forward's last launch to the collective's completion
        # (includes waiting for the slowest rank's logits -- which is what a sub-linear curve would show here)
        ag_us = sorted(1e3 * mids[i].elapsed_time(evs[i + 1]) for i in range(args.steps))
        fw_ms = sorted(evs[i].elapsed_time(mids[i]) for i in range(args.steps))
        mine = torch.tensor([1e3 * elapsed / args.steps, step_ms[len(step_ms) // 2], fw_ms[len(fw_ms) // 2],
                             ag_us[len(ag_us) // 2], ag_us[-1]], device=dev, dtype=torch.float64)
        allr = torch.empty(world * mine.numel(), device=dev, dtype=torch.float64)
        dist.all_gather_into_tensor(allr, mine)
        allr = allr.view(world, -1).cpu().numpy()
        wall = sorted(allr[:, 0].tolist())
        rank_stats = {
            "rank_step_ms": {"min": round(wall[0], 3), "median": round(wall[len(wall) // 2], 3), "max": round(wall[-1], 3),
                             "per_rank_wall": [round(float(v), 3) for v in allr[:, 0]],
                             "per_rank_event_median": [round(float(v), 3) for v in allr[:, 1]],
                             "per_rank_forward_event_median": [round(float(v), 3) for v in allr[:, 2]]},
            "allgather_us": {"median_over_ranks_of_median": round(float(np.median(allr[:, 3])), 1),
                             "max_over_ranks_of_median": round(float(allr[:, 3].max()), 1),
                             "max_over_ranks_of_max": round(float(allr[:, 4].max()), 1),
                             "what": "HIP events on the launch stream: forward's last launch -> all-gather complete "
                                     "(includes waiting for the slowest rank)"}}
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = 1e3 * elapsed / args.steps
    clips_per_s = world * B * args.steps / elapsed
    ranks_seen = dist.get_world_size() if world > 1 else 1   # what the communicator itself reports
    logits = logits.detach().clone()                          # (the extra-config legs below free the net)

    if rank != 0:
        dist.barrier()  # keep the communicator alive until rank 0 has finished reporting
        dist.destroy_process_group()
        return

    # ---- roofline: per-launch HIP-event times on the launch stream ----
    peak_mfma = PEAK_MFMA_TFLOPS[args.dtype]
    workload_key = f"{args.variant}/{args.segments}/{B}/{args.dtype}"   # as tools/summarize_profiles.py names its PMC passes
    prof = net._engine.profile(args.profile_iters)
    # grouped by kernel FAMILY (the template name without its arguments), each family listing its instances: the choice of
    # the "dominant" kernel must not depend on whether a launch site spells out its template arguments
    by_kernel = {}
    floor_ms = 0.0
    useful_floor_ms = 0.0
    for p in prof:
        fam = p["kernel"].split("<")[0]
        k = by_kernel.setdefault(fam, dict(ms=0.0, flops=0, useful=0.0, bytes=0, launches=0, floor_ms=0.0, useful_floor_ms=0.0,
                                           instances={}))
        fl = 1e3 * max(p["flops"] / (peak_mfma * 1e12), p["bytes"] / (PEAK_HBM_GBS * 1e9))  # this launch's own floor
        # ... and the same with the flops that produce outputs only: Winograd tiles overhang planes that are not multiples of
        # four (14 -> 16, 7 -> 8: +30.6 % on res4 / res5 and the 14 / 7 px inception layers); Engine records both
        useful = p.get("useful_flops", p["flops"])
        ufl = 1e3 * max(useful / (peak_mfma * 1e12), p["bytes"] / (PEAK_HBM_GBS * 1e9))
        k["ms"] += p["ms"]; k["flops"] += p["flops"]; k["bytes"] += p["bytes"]; k["launches"] += 1; k["floor_ms"] += fl
        k["useful"] += useful; k["useful_floor_ms"] += ufl
        useful_floor_ms += ufl
        inst = k["instances"].setdefault(p["kernel"], dict(ms=0.0, launches=0, floor_ms=0.0))
        inst["ms"] += p["ms"]; inst["launches"] += 1; inst["floor_ms"] += fl
        floor_ms += fl
    total_ms = sum(k["ms"] for k in by_kernel.values())
    dom_name, dom = max(by_kernel.items(), key=lambda kv: kv[1]["ms"])
    dom_inst = max(dom["instances"].items(), key=lambda kv: kv[1]["ms"])[0]
    total_flops = spec.conv_fc_flops()
    t_flops = dom["flops"] / (peak_mfma * 1e12)
    t_bytes = dom["bytes"] / (PEAK_HBM_GBS * 1e9)
    if t_flops >= t_bytes:
        ach = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
        roofline = {"bound": "mfma", "achieved": round(ach, 3), "peak": peak_mfma, "unit": "TFLOP/s",
                    "frac": round(ach / peak_mfma, 4),
                    # the same kernel on the flops that produce outputs (Winograd tile overhang excluded)
                    "useful_frac": round(dom["useful"] / (dom["ms"] * 1e-3) / 1e12 / peak_mfma, 4), "traffic": None}
    else:
        ach = dom["bytes"] / (dom["ms"] * 1e-3) / 1e9
        roofline = {"bound": "hbm", "achieved": round(ach, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                    "frac": round(ach / PEAK_HBM_GBS, 4), "useful_frac": round(ach / PEAK_HBM_GBS, 4), "traffic": None}
    # HBM traffic of that kernel family from the committed PMC summary (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in
    # separate passes, gfx950 FETCH half-count corrected; tools/summarize_profiles.py) -- bench.py cannot run the
    # profiler on itself, so this is the figure of the last profiled build: reported only when that build was made from
    # the SOURCES the running library was made from (eco_source_digest(), compiled in by csrc/Makefile: a rebuild of
    # identical sources keeps the field; the .so's bytes are not compared), else null with the reason.
    fam_traffic = {}
    try:
        with open(os.path.join(ROOT, "profiles", "hbm_traffic_latest.json")) as f:
            tr = json.load(f)
        src_now = hip.load().source_digest()
        roofline["traffic"], roofline["traffic_unit"] = traffic_from_summary(tr, workload_key, dom_name, src_now)
        for fam in by_kernel:   # ... and for EVERY family of the step: which one re-fetches most
            fam_traffic[fam] = traffic_from_summary(tr, workload_key, fam, src_now)[0]
    except Exception as e:  # no summary committed, unreadable file, ...
        roofline["traffic"], roofline["traffic_unit"] = None, f"null: {type(e).__name__}: {e}"
    if roofline["traffic"] is not None and dom["bytes"] > 0:
        roofline["traffic_ratio"] = round(roofline["traffic"] / (dom["bytes"] / dom["launches"] / 1e9), 3)
    # the whole step's HBM traffic by the same counters (every eco:: kernel of the PMC passes, bytes per launch x launches
    # per step) beside SURVEY.md section 8(d)'s fused-model figure (each conv / fc: input + weights + output once, pools in +
    # out, residual reads, dual writes): the excess is the Winograd routes' V / M scratch and split-K partial sums
    try:
        roofline["step_traffic_gb"], roofline["step_traffic_unit"] = step_traffic_from_summary(tr, workload_key, src_now)
    except Exception as e:
        roofline["step_traffic_gb"], roofline["step_traffic_unit"] = None, f"null: {type(e).__name__}: {e}"
    fused_gb = spec.fused_model_bytes() / 1e9 if hasattr(spec, "fused_model_bytes") else None
    roofline["step_fused_model_gb"] = round(fused_gb, 2) if fused_gb is not None else None
    roofline["step_launch_model_gb"] = round(sum(p["bytes"] for p in prof) / 1e9, 2)   # what the launches must move, V / M included
    if roofline["step_traffic_gb"] is not None and fused_gb:
        roofline["step_traffic_ratio_to_fused_model"] = round(roofline["step_traffic_gb"] / fused_gb, 3)
    executed = sum(p["flops"] for p in prof)
    roofline.update({
        "kernel": dom_name, "largest_instance": dom_inst, "launches_per_step": dom["launches"],
        "flops_counted": "executed by the launches (Winograd launches count their transformed-domain GEMM flops, "
                         "not the direct convolution's)",
        "timing": "HIP events on the launch stream around each C-ABI call; for split-K plans that includes the "
                  "conv_splitk_reduce_kernel launch that follows the main kernel",
        "avg_launch_ms": round(dom["ms"] / dom["launches"], 4),
        "algorithmic_gflop_per_launch": round(dom["flops"] / dom["launches"] / 1e9, 3),
        "algorithmic_gb_per_launch": round(dom["bytes"] / dom["launches"] / 1e9, 4),
        "kernel_share_of_step": round(dom["ms"] / total_ms, 4),
        # whole step against the roofline: every launch's own floor (its executed flops at the MFMA peak or its
        # algorithmic bytes at the HBM peak, whichever is larger), summed, over the measured step time
        "step_frac": round(floor_ms / ms_per_step, 4),
        "step_floor_ms": round(floor_ms, 3),
        # the same with Winograd tile overhang taken out of every launch's flops
        "step_useful_frac": round(useful_floor_ms / ms_per_step, 4),
        "whole_step": {"executed_gflop": round(executed / 1e9, 2),
                       "executed_tflops": round(executed / (ms_per_step * 1e-3) / 1e12, 2),
                       "executed_frac_of_mfma_peak": round(executed / (ms_per_step * 1e-3) / 1e12 / peak_mfma, 4),
                       # the reference's direct convolutions would need this many flops for the same outputs; the
                       # rate below is what a direct evaluation would have to sustain to match this step time
                       # (above the MFMA peak exactly when Winograd is doing less work) -- not a roofline fraction
                       "direct_algorithm_gflop": round(total_flops / 1e9, 2),
                       "direct_algorithm_equivalent_tflops": round(total_flops / (ms_per_step * 1e-3) / 1e12, 2)},
        "per_kernel": {k: {"ms": round(v["ms"], 3), "launches": v["launches"],
                           "frac_of_own_floor": round(v["floor_ms"] / v["ms"], 3) if v["ms"] > 0 else None,
                           "useful_frac_of_own_floor": round(v["useful_floor_ms"] / v["ms"], 3) if v["ms"] > 0 else None,
                           "algorithmic_gb_per_launch": round(v["bytes"] / v["launches"] / 1e9, 4),
                           # PMC HBM bytes per launch (launch-weighted over the family's instances) and their ratio to the
                           # algorithmic figure; null unless profiles/hbm_traffic_latest.json belongs to these sources
                           "traffic_gb_per_launch": fam_traffic.get(k),
                           "traffic_ratio": (round(fam_traffic[k] / (v["bytes"] / v["launches"] / 1e9), 3)
                                             if fam_traffic.get(k) is not None and v["bytes"] > 0 else None),
                           "instances": {ik: {"ms": round(iv["ms"], 3), "launches": iv["launches"],
                                              "frac_of_own_floor": round(iv["floor_ms"] / iv["ms"], 3) if iv["ms"] > 0 else None}
                                         for ik, iv in sorted(v["instances"].items(), key=lambda kv: -kv[1]["ms"])}}
                       for k, v in sorted(by_kernel.items(), key=lambda kv: -kv[1]["ms"])},
    })

    # ---- the other single-GPU BASELINE configurations, in front of the driver: configs[3] (ECO-Full) and configs[4]
    # (ECO-Lite N=32 bf16) for --extra-steps steps each, after the configs[1] timed region and before the CPU leg ----
    extra = None
    is_headline = (world == 1 and args.variant == "lite" and N == 16 and B == 32 and args.dtype == "f32" and
                   not use_graph and not args.no_winograd)
    if is_headline and not args.no_extra_configs:
        del net
        torch.cuda.empty_cache()
        extra = {}
        for key, kw in (("configs3", dict(variant="full", N=16, dtype="f32")),
                        ("configs4", dict(variant="lite", N=32, dtype="bf16"))):
            try:
                extra[key] = extra_config(kw["variant"], kw["N"], B, kw["dtype"], args.extra_steps, dev,
                                          all_parity=not args.extra_parity_few)
            except Exception as e:  # the headline line must survive a failing side leg
                extra[key] = {"error": f"{type(e).__name__}: {e}"}
            torch.cuda.empty_cache()

    # ---- CPU baseline: the reference's cost structure on this box's host cores ----
    cpu = None
    parity = None
    if not args.no_cpu_baseline and world == 1:  # reported on rank 0 at N=1 only
        cpu, parity = cpu_baseline(args, gen, N, frames, params, logits)
    elif world > 1:
        # an N > 1 line still carries a parity field: rank 0's first clip against the CPU reference (the CPU baseline
        # itself is reported at N = 1 only)
        try:
            ref = reference_logits(gen, N, frames, params, clips=1, blas_threads=False)
            parity = parity_record(gathered[:1].detach().float().cpu().numpy(), ref, args.dtype,
                                   "rank 0's first clip, read back from the all-gathered logits")
        except Exception as e:
            parity = {"error": f"{type(e).__name__}: {e}"}

    cfg = baseline_config(args.variant, N, B, args.dtype, world)
    name = "Lite" if args.variant == "lite" else "Full"
    line = {
        # (spelled exactly as BASELINE.json's `metric`: U+00D7 between the frame extents)
        "metric": "clips/sec (whole node), ECO-%s N=%d 224\u00d7224 bs%d; top-1 logits vs CPU ref" % (name, N, B),
        "value": round(clips_per_s, 2), "unit": "clips/sec", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "ms_per_step_event_median": round(step_ms[len(step_ms) // 2], 3),
        "timed_region": "wall clock over exactly K steps between barrier + device-synchronize fences, max over ranks",
        "config": {"workload": "ECO-%s num_segments=%d batch=%d/GPU %s (%s), random-init seeded weights, synthetic "
                               "224x224 frames resident in HBM" % (name, N, B, args.dtype, cfg),
                   "baseline_config": cfg, "global_batch": world * B, "num_segments": N,
                   "parallelism": f"clip-batch dp{world}", "launches_per_step": len(prof),
                   "submission": submission,
                   "collective": "none" if world == 1 else "RCCL all-gather of logits",
                   "collective_ranks": ranks_seen, "collective_backend": "none" if world == 1 else backend,
                   "device": f"cuda:{dev_index} {dev_info['name']}, {dev_info['num_cu']} CUs"},
        "roofline": roofline, "cpu_baseline": cpu, "parity": parity,
    }
    if isinstance(parity, dict) and "top1_equal" in parity:     # the metric's "top-1 logits vs CPU ref", at the top level
        line["top1_equal"] = parity["top1_equal"]
        line["max_rel_err_vs_cpu_ref"] = parity["max_rel_err"]
    if rank_stats is not None:
        line["config"].update(rank_stats)
        line["config"]["rank0_affinity"] = pinned
    if extra is not None:
        line["extra_configs"] = extra
    print(json.dumps(line))
    sys.stdout.flush()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    elif cpu is not None:
        # the CPU leg ran OpenBLAS from many host threads; its exit-time housekeeping can print to stdout, and the
        # contract is ONE json line: leave without running library destructors
        sys.stderr.flush()
        os._exit(0)


def _ref_conv():
    """Convolution of the CPU reference runs: the reference's own ConvolutionLayer class (base_conv_layer.cpp +
    conv_layer.cpp + util/im2col.cpp compiled unmodified into oracle/_ref) over OpenBLAS sgemm; the equivalent restated
    call sequence where an older oracle/_ref travelled; None (NumPy restatement) without oracle/_ref."""
    import eco_ref
    if not eco_ref.available():
        return None, "NumPy restatement (im2col_nd + np.matmul)"
    if eco_ref.has_conv_layer():
        return (lambda x, w, b, k, s, p: eco_ref.convolution_layer(x, w, b, list(k), list(s), list(p)),
                "the reference's ConvolutionLayer::Forward_cpu (base_conv_layer.cpp, conv_layer.cpp, util/im2col.cpp compiled "
                "unmodified) over OpenBLAS sgemm")
    return (lambda *a: eco_ref.convolution(*a, image_threads=1),
            "reference im2col (compiled from util/im2col.cpp) + OpenBLAS sgemm")


def reference_logits(gen, N, frames, params, clips=1, blas_threads: bool = True):
    """fp32 CPU reference logits of clips `clips` (a count = the first ones, or a list of clip indices) of `frames`: the
    oracle's layer sequence with every convolution through the compiled reference ConvolutionLayer (checker only)."""
    import numpy as np
    from eco_amd.netspec import NetSpec
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import eco_oracle  # checker only; never on the product path
    import eco_ref
    spec1 = NetSpec.from_prototxt(gen(num_segments=N, num_clips=1))
    conv, _ = _ref_conv()
    if conv is not None and blas_threads:
        try:
            cores = len(os.sched_getaffinity(0))
        except AttributeError:
            cores = os.cpu_count()
        # (N > 1: torch.distributed.run exports OMP_NUM_THREADS=1 and OpenBLAS sizes its buffers by it when it loads;
        # raising the count afterwards crashed it -- one clip on one thread takes a few seconds)
        eco_ref.set_blas_threads(min(cores, 64))
    idx = list(range(clips)) if isinstance(clips, int) else [int(c) for c in clips]
    outs = [eco_oracle.forward(spec1, params, {"data": frames[c * N:(c + 1) * N]}, conv_impl=conv)[spec1.outputs[0]]
            for c in idx]
    return np.concatenate(outs, 0)


def traffic_from_summary(tr, workload_key, family, src_now):
    """(GB per launch or None, unit / reason) for kernel `family` from profiles/hbm_traffic_latest.json (`tr`).  Reported only
    when the summary was collected (a) on a library built from the sources the running one was built from (`src_now` =
    eco_source_digest()) and (b) on THIS workload (variant/segments/clips per GPU/dtype): bytes per launch belong to the
    launch size they were counted on.  Launch-weighted mean over the family's instances."""
    if workload_key == tr.get("workload", "lite/16/32/f32"):
        tk, tsrc = tr["kernels"], tr["source"]
    elif workload_key in tr.get("workloads", {}):
        tk, tsrc = tr["workloads"][workload_key]["kernels"], tr["workloads"][workload_key]["source"]
    else:
        return None, f"null: no PMC passes of workload {workload_key} in profiles/hbm_traffic_latest.json"
    if tr.get("src_sha256") is None or tr.get("src_sha256") != src_now:
        return None, ("null: profiles/hbm_traffic_latest.json was collected on a build of other sources "
                      f"({str(tr.get('src_sha256'))[:12]} != {src_now[:12]}); re-run tools/profile_round.sh")
    rows = {k: v for k, v in tk.items() if k.split("<")[0] == family}
    if not rows:
        return None, f"null: no PMC row for {family} in {tsrc}"
    w = sum(c.get("launches", 1) for c in rows.values())
    gb = round(sum(c["hbm_bytes_per_launch"] * c.get("launches", 1) for c in rows.values()) / w / 1e9, 4)
    return gb, "GB per launch (PMC, " + tsrc + ", same sources " + src_now[:12] + ")"


def step_traffic_from_summary(tr, workload_key, src_now):
    """(GB per step or None, unit / reason): sum over every eco:: kernel of the workload's PMC passes of corrected HBM bytes per
    launch x launches, divided by the steps those passes ran (`pmc_steps` of profiles/hbm_traffic_latest.json)."""
    if workload_key == tr.get("workload", "lite/16/32/f32"):
        tk, tsrc = tr["kernels"], tr["source"]
    elif workload_key in tr.get("workloads", {}):
        tk, tsrc = tr["workloads"][workload_key]["kernels"], tr["workloads"][workload_key]["source"]
    else:
        return None, f"null: no PMC passes of workload {workload_key} in profiles/hbm_traffic_latest.json"
    if tr.get("src_sha256") is None or tr.get("src_sha256") != src_now:
        return None, "null: profiles/hbm_traffic_latest.json was collected on a build of other sources"
    steps = tr.get("pmc_steps")
    if not steps:
        return None, "null: profiles/hbm_traffic_latest.json does not say how many steps its PMC passes ran"
    tot = sum(c["hbm_bytes_per_launch"] * c.get("launches", 1) for k, c in tk.items() if k.startswith("eco::"))
    return round(tot / steps / 1e9, 2), f"GB per step (PMC FETCH_SIZE x 2 + WRITE_SIZE over {steps} steps, {tsrc})"


def parity_record(got, ref, dtype: str, what: str, clips=None) -> dict:
    import numpy as np
    denom = float(np.abs(ref).max())
    per_class = np.abs(got - ref) / np.maximum(np.abs(ref), 1e-3 * denom)
    top5 = [len(set(np.argsort(-g)[:5].tolist()) & set(np.argsort(-r)[:5].tolist())) for g, r in zip(got, ref)]
    # a differing top-1 is a reference near-tie when the class picked is within twice the clip's absolute error of the
    # reference's maximum (random-init logits do tie that closely; bf16 storage then decides)
    # (the error is capped at the tolerance: a grossly wrong logit does not excuse itself)
    tol = 1e-2 if dtype == "bf16" else 1e-3     # (bf16: 3e-2 until round 4, against a measured 4e-3)
    tie = [bool(r[int(r.argmax())] - r[int(g.argmax())] <= 2.0 * min(float(np.abs(g - r).max()), tol * denom))
           for g, r in zip(got, ref)]
    rec = {"clips_checked": int(ref.shape[0]), "max_rel_err": float(np.abs(got - ref).max() / denom),
           "max_per_logit_rel_err": float(per_class.max()), "max_abs_logit": denom,
           "top1_agree": bool((got.argmax(1) == ref.argmax(1)).all()),
           "top1_equal": bool((got.argmax(1) == ref.argmax(1)).all()),            # strict, every checked clip
           "top1_equal_per_clip": [bool(a == b) for a, b in zip(got.argmax(1), ref.argmax(1))],
           "top1_equal_or_reference_near_tie": bool(all(tie)),
           "top5_overlap_min": int(min(top5)), "top5_overlap_per_clip": top5, "reference": what,
           "tolerance": tol}
    if clips is not None:
        rec["clips"] = [int(c) for c in clips]
    return rec


def reference_logits_batch(gen, N, B, frames, params):
    """fp32 CPU reference logits of ALL B clips of `frames` in one forward: the oracle's layer sequence with every convolution through
    the compiled reference im2col + one OpenBLAS sgemm per image, images spread over host threads (checker only)."""
    from eco_amd.netspec import NetSpec
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import eco_oracle  # checker only; never on the product path
    import eco_ref
    if not eco_ref.available():
        raise RuntimeError("oracle/_ref is not built")
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count()
    img_threads = min(cores, 64)   # OpenBLAS keeps one buffer per concurrent caller, at most its NUM_THREADS (64)
    eco_ref.set_blas_threads(min(cores, 64))
    specB = NetSpec.from_prototxt(gen(num_segments=N, num_clips=B))
    return eco_oracle.forward(specB, params, {"data": frames[:B * N]},
                              conv_impl=lambda *a: eco_ref.convolution(*a, image_threads=img_threads))[specB.outputs[0]]


def extra_config(variant: str, N: int, B: int, dtype: str, steps: int, dev, all_parity: bool = True) -> dict:
    """One of the other single-GPU BASELINE.json configurations on the same device: `steps` timed steps between
    device synchronisations (wall clock over the whole run, three warm-up steps, no per-step events), the per-launch floors
    of Engine.profile, and clips of the batch against the CPU reference: first / two in the middle / last for the bf16
    configuration (its tolerance is the builder's own, so it gets the wider check), first and last for fp32."""
    import torch
    import eco_amd as caffe
    from eco_amd import fillers, models
    from eco_amd.netspec import NetSpec
    gen = models.eco_lite_deploy if variant == "lite" else models.eco_full_deploy
    proto = gen(num_segments=N, num_clips=B)
    spec = NetSpec.from_prototxt(proto)
    params = fillers.synthetic_params(spec)
    kw = {} if dtype == "f32" else {"dtype": dtype}
    net = caffe.Net(proto, caffe.TEST, params=params, **kw)
    frames = fillers.synthetic_frames(B * N, seed=1234)
    net.set_input_device("data", torch.from_numpy(frames).to(dev))
    for _ in range(3):
        net.forward_device()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        net.forward_device()
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / steps
    peak = PEAK_MFMA_TFLOPS[dtype]
    prof = net._engine.profile(2)
    floor_ms = sum(1e3 * max(p["flops"] / (peak * 1e12), p["bytes"] / (PEAK_HBM_GBS * 1e9)) for p in prof)
    useful_floor_ms = sum(1e3 * max(p.get("useful_flops", p["flops"]) / (peak * 1e12), p["bytes"] / (PEAK_HBM_GBS * 1e9))
                          for p in prof)
    executed = sum(p["flops"] for p in prof)
    fam = {}
    for p in prof:
        fam[p["kernel"].split("<")[0]] = fam.get(p["kernel"].split("<")[0], 0.0) + p["ms"]
    dom = max(fam.items(), key=lambda kv: kv[1])
    clips = [0, B // 3, (2 * B) // 3, B - 1] if dtype == "bf16" else [0, B - 1]
    clips = sorted(set(c for c in clips if 0 <= c < B))
    got_all = net.blobs[spec.outputs[0]].tensor.detach().float().cpu().numpy()
    got = got_all[clips]
    ref = reference_logits(gen, N, frames, params, clips=clips)
    # ... and EVERY clip of the batch against the reference's im2col + sgemm call sequence with the batch's images spread over the
    # host cores (the arithmetic of bench.py's `image_parallel` CPU leg, pinned bit-identically to the compiled ConvolutionLayer
    # class in tests/test_oracle_ref.py): one CPU forward of the whole batch, 15-30 s
    all_clips = None
    if all_parity:
        try:
            all_clips = parity_record(got_all, reference_logits_batch(gen, N, B, frames, params), dtype,
                                      "reference im2col (compiled from util/im2col.cpp) + OpenBLAS sgemm, the batch's images spread "
                                      "over host threads; all clips of the batch", list(range(B)))
        except Exception as e:  # the leg must survive a failing whole-batch reference (memory, missing oracle/_ref, ...)
            all_clips = {"error": f"{type(e).__name__}: {e}"}
    name = "Lite" if variant == "lite" else "Full"
    out = {"workload": "ECO-%s num_segments=%d batch=%d %s (%s)" % (name, N, B, dtype, baseline_config(variant, N, B, dtype, 1)),
           "steps": steps, "ms_per_step": round(ms, 3), "clips_per_s": round(B * 1e3 / ms, 1), "dtype": dtype,
           "timed_region": f"wall clock over {steps} steps between two device synchronisations after 3 warm-up steps",
           "launches_per_step": len(prof), "step_frac": round(floor_ms / ms, 4),
           "step_useful_frac": round(useful_floor_ms / ms, 4),
           "executed_frac_of_mfma_peak": round(executed / (ms * 1e-3) / 1e12 / peak, 4),
           "largest_kernel": {"name": dom[0], "ms_per_step": round(dom[1], 3)},
           "parity": parity_record(got, ref, dtype, "CPU oracle with every convolution through " + _ref_conv()[1], clips)}
    # what "top-1 logits vs CPU ref" came to, said at this record's top level and strictly (round-4 verdict): equality on
    # every checked clip; `..._or_reference_near_tie` beside it is the weaker statement the bf16 tolerance can support on
    # random-init logits (a differing class is within twice the clip's measured error of the reference's maximum)
    out["top1_equal"] = out["parity"]["top1_equal"]
    out["top1_equal_or_reference_near_tie"] = out["parity"]["top1_equal_or_reference_near_tie"]
    out["max_rel_err"] = out["parity"]["max_rel_err"]
    if all_clips is not None:
        out["parity_all_clips"] = all_clips
        if "max_rel_err" in all_clips:   # the top-level figures speak for the whole batch when it was checked
            out["clips_checked"] = all_clips["clips_checked"]
            out["max_rel_err"] = max(out["max_rel_err"], all_clips["max_rel_err"])
            out["top1_equal"] = bool(out["top1_equal"] and all_clips["top1_equal"])
            out["top1_equal_or_reference_near_tie"] = bool(out["top1_equal_or_reference_near_tie"] and
                                                            all_clips["top1_equal_or_reference_near_tie"])
    del net
    return out


def cpu_baseline(args, gen, N, frames, params, logits):
    """The CPU path of the reference timed on this box, on a bounded sample of the same workload, in three forms:

    * caffe_cost  -- the reference's own structure: images in sequence, per image the REFERENCE's compiled
      im2col (oracle/_ref, util/im2col.cpp) + one cblas_sgemm on all cores + bias sgemm
      (conv_layer.cpp:28-43, base_conv_layer.cpp:264-287, math_functions.cpp:12-21); BN / ReLU / pooling /
      eltwise as separate passes (NumPy restatement, oracle/eco_oracle.py).  This is `value`.
    * image_parallel -- the same arithmetic with the images of a layer spread over the cores (one BLAS thread
      each): how one would run the reference's CPU path for throughput.
    * torch_cpu -- torch.nn.functional.conv2d/conv3d in place of im2col+sgemm: a courtesy upper bound for this CPU.
    The first clips' logits of the caffe_cost run are the parity reference for the GPU's."""
    import numpy as np
    import torch
    from eco_amd.netspec import NetSpec
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import eco_oracle  # checker/baseline only; never on the product path
    import eco_ref     # compiled reference sources (oracle/_ref), test infrastructure
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count()
    spec1 = NetSpec.from_prototxt(gen(num_segments=N, num_clips=1))
    out_name = spec1.outputs[0]
    have_ref = eco_ref.available()

    def torch_conv(x, w, b, kernel, stride, pad):
        f = torch.nn.functional.conv2d if len(kernel) == 2 else torch.nn.functional.conv3d
        with torch.no_grad():
            return f(torch.from_numpy(x), torch.from_numpy(np.ascontiguousarray(w, np.float32)),
                     None if b is None else torch.from_numpy(np.ascontiguousarray(b, np.float32)), tuple(stride),
                     tuple(pad)).numpy()

    def run(conv_impl, budget_s, max_clips, clips_at_once=1):
        done, t_cpu, refs = 0, 0.0, []
        while done < max_clips:
            k = min(clips_at_once, max_clips - done)
            specs = spec1 if k == 1 else NetSpec.from_prototxt(gen(num_segments=N, num_clips=k))
            x1 = frames[done * N:(done + k) * N]
            t1 = time.perf_counter()
            refs.append(eco_oracle.forward(specs, params, {"data": x1}, conv_impl=conv_impl)[out_name])
            t_cpu += time.perf_counter() - t1
            done += k
            if not args.cpu_clips and budget_s > 0 and t_cpu >= budget_s:
                break
        return done, t_cpu, np.concatenate(refs, 0)

    variants = {}
    max_clips = min(args.cpu_clips or 8, len(frames) // N)     # (never more clips than the batch holds: --clips-per-gpu 2)
    blas_threads = min(cores, 64)   # SciPy's OpenBLAS is built for at most 64 threads
    if have_ref:
        eco_ref.set_blas_threads(blas_threads)
        conv_fn, conv_kind = _ref_conv()
        d, t, ref = run(conv_fn, 10.0, max_clips)
        variants["caffe_cost"] = dict(clips_per_s=round(d / t, 4), clips=d, seconds=round(t, 2), blas_threads=blas_threads,
                                      kind=conv_kind + ", images in sequence")
        # throughput form: a whole GPU batch of clips at once, its images spread over the cores (the 3-D trunk has
        # one "image" per clip, so fewer clips would leave most cores idle there)
        par = args.cpu_clips or min(32, len(frames) // N)
        img_threads = min(cores, 64)   # OpenBLAS keeps one buffer per concurrent caller, at most its NUM_THREADS (64)
        d2, t2, ref_all = run(lambda *a: eco_ref.convolution(*a, image_threads=img_threads), 0.0, par, clips_at_once=par)
        variants["image_parallel"] = dict(clips_per_s=round(d2 / t2, 4), clips=d2, seconds=round(t2, 2), blas_threads=1,
                                          image_threads=img_threads,
                                          kind="same arithmetic, one batch of clips at once, images spread over host threads")
    else:  # oracle/_ref not shipped: the NumPy restatement (np.matmul = OpenBLAS sgemm)
        d, t, ref = run(None, 10.0, max_clips)
        variants["caffe_cost"] = dict(clips_per_s=round(d / t, 4), clips=d, seconds=round(t, 2), blas_threads=blas_threads,
                                      kind="NumPy restatement (im2col_nd + np.matmul), images in sequence")
    torch.set_num_threads(cores)
    d3, t3, _ = run(torch_conv, 6.0, max_clips, clips_at_once=min(4, max_clips))
    variants["torch_cpu"] = dict(clips_per_s=round(d3 / t3, 4), clips=d3, seconds=round(t3, 2), threads=cores,
                                 kind="torch.nn.functional.conv2d/conv3d (courtesy upper bound, not the reference's structure)")
    cc = variants["caffe_cost"]
    flops_clip = spec1.conv_fc_flops()
    cpu = {"value": cc["clips_per_s"], "unit": "clips/sec", "cores": cores,
           "kind": "reference" if have_ref else "port",
           "sample": f"{cc['clips']} clip(s) of the same workload (num_segments={N}; the first clips of rank 0's batch), "
                     f"{cc['seconds']} s; conv = {cc['kind']} with {blas_threads} BLAS threads, other layers NumPy "
                     f"(oracle/eco_oracle.py); caffe_3d itself cannot be built here (DESIGN.md section 4)",
           "gflops": round(cc["clips_per_s"] * flops_clip / 1e9, 1), "variants": variants}
    done = cc["clips"]
    got = logits[:done].detach().float().cpu().numpy()
    parity = parity_record(got, ref, args.dtype, "caffe_cost CPU run above: " + cc["kind"], list(range(done)))
    if have_ref and d2 > done:
        # the image_parallel leg computed EVERY clip of the batch on the CPU (the reference's im2col + sgemm call sequence,
        # pinned bit-identically to the compiled ConvolutionLayer class in tests/test_oracle_ref.py): the line's parity
        # record covers them all; the record against the class itself (first clips) stays beside it
        first = parity
        parity = parity_record(logits[:d2].detach().float().cpu().numpy(), ref_all, args.dtype,
                               "image_parallel CPU run above (reference im2col, compiled from util/im2col.cpp, + OpenBLAS sgemm; "
                               "all clips of the batch)", list(range(d2)))
        parity["first_clips_vs_convolution_layer_class"] = first
    return cpu, parity


if __name__ == "__main__":
    main()
