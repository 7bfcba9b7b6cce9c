#!/usr/bin/env python
"""Condense rocprofv3 outputs (gpurun_out/<run>/...) into the small tracked summaries under
profiles/: per-kernel stats of `rocprofv3 --kernel-trace --stats -- python bench.py` and the
per-kernel HBM traffic of the two PMC passes (FETCH_SIZE, WRITE_SIZE collected separately:
TCC has 4 slots, FETCH_SIZE takes 3).

    python tools/summarize_profiles.py gpurun_out/r1 r01

HBM traffic correction (/opt/skills/guides/MI355X_MICROARCH.md, HBM section): rocprofv3 reports
FETCH_SIZE / WRITE_SIZE in KiB; on gfx950 FETCH_SIZE counts 128-B fabric read requests as 64 B,
i.e. half of the bytes of a wide coalesced stream, so read bytes = 2 * FETCH_SIZE * 1024.
WRITE_SIZE is taken as is (uncalibrated per the guide)."""
import collections
import csv
import json
import os
import sys


def short(name):
    return name.split("(")[0].replace("void ", "").strip()


def main():
    src, tag = sys.argv[1], sys.argv[2]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out_dir = os.path.join(root, "profiles")
    os.makedirs(out_dir, exist_ok=True)
    stats = list(csv.DictReader(open(os.path.join(src, "trace", "bench_kernel_stats.csv"))))
    with open(os.path.join(out_dir, f"{tag}_kernel_stats.csv"), "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline\n")
        f.write("kernel,calls,total_ms,avg_us,min_us,max_us,percent\n")
        for r in stats:
            f.write("%s,%s,%.3f,%.1f,%.1f,%.1f,%s\n" % (short(r["Name"]).replace(",", ";"), r["Calls"],
                    float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3,
                    float(r["MaxNs"]) / 1e3, r["Percentage"]))
    for sub, cmd in (("bf16", "--segments 32 --dtype bf16"), ("full", "--variant full")):
        bf = os.path.join(src, f"trace_{sub}", "bench_kernel_stats.csv")
        if os.path.exists(bf):
            with open(os.path.join(out_dir, f"{tag}_{sub}_kernel_stats.csv"), "w") as f:
                f.write(f"# rocprofv3 --kernel-trace --stats -- python bench.py {cmd} --steps 10 --warmup 3 --no-cpu-baseline\n")
                f.write("kernel,calls,total_ms,avg_us,min_us,max_us,percent\n")
                for r in csv.DictReader(open(bf)):
                    f.write("%s,%s,%.3f,%.1f,%.1f,%.1f,%s\n" % (short(r["Name"]).replace(",", ";"), r["Calls"],
                            float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3,
                            float(r["MaxNs"]) / 1e3, r["Percentage"]))
    for nm in ("bench_line.json", "bench_line_bf16.json", "bench_line_full.json", "bench_line_b1.json",
               "bench_line_b1_graph.json", "eco_time.txt", "eco_time_bf16.txt", "eco_time_full.txt"):
        p = os.path.join(src, nm)
        if os.path.exists(p) and os.path.getsize(p):
            with open(p) as fi, open(os.path.join(out_dir, f"{tag}_{nm}"), "w") as fo:
                fo.write("".join(l for l in fi if "amdgpu.ids" not in l))
    summary = summarize_traffic(src, "", os.path.join(out_dir, f"{tag}_pmc_hbm_traffic.csv"), "python bench.py")
    # the other profiled workloads (bench.py's key: variant/segments/clips per GPU/dtype): traffic per launch belongs to the
    # launch SIZE it was counted on -- a bench line of another workload gets null, not these figures
    others = {}
    if os.path.exists(os.path.join(src, "pmc_fetch_bf16")):
        others["lite/32/32/bf16"] = {"source": f"profiles/{tag}_bf16_pmc_hbm_traffic.csv",
                                     "kernels": summarize_traffic(src, "_bf16", os.path.join(out_dir, f"{tag}_bf16_pmc_hbm_traffic.csv"),
                                                                  "python bench.py --segments 32 --dtype bf16")}
    if os.path.exists(os.path.join(src, "pmc_fetch_full")):
        others["full/16/32/f32"] = {"source": f"profiles/{tag}_full_pmc_hbm_traffic.csv",
                                    "kernels": summarize_traffic(src, "_full", os.path.join(out_dir, f"{tag}_full_pmc_hbm_traffic.csv"),
                                                                 "python bench.py --variant full")}
    sha = os.path.join(src, "src.sha256")
    with open(os.path.join(out_dir, "hbm_traffic_latest.json"), "w") as f:
        json.dump({"source": f"profiles/{tag}_pmc_hbm_traffic.csv",
                   # what the profiled build was made FROM (eco_source_digest() of the library that ran): bench.py reports
                   # `roofline.traffic` while the running library reports the same digest -- a rebuild of identical
                   # sources keeps the field, an edited kernel drops it
                   "src_sha256": open(sha).read().strip().splitlines()[-1] if os.path.exists(sha) else None,
                   "workload": "lite/16/32/f32",
                   # steps each PMC pass ran (tools/profile_round.sh: bench.py --steps 2 --warmup 1 + the three per-launch
                   # profiling iterations): launches / pmc_steps = launches per step
                   "pmc_steps": 6,
                   "kernels": summary, "workloads": others}, f, indent=1)
    sq = os.path.join(src, "pmc_sq", "bench_counter_collection.csv")
    if os.path.exists(sq):
        summarize_sq(sq, os.path.join(out_dir, f"{tag}_pmc_sq.csv"))
    # configs[4] (bf16, N=32): the same three counter passes (traffic: above)
    sqb = os.path.join(src, "pmc_sq_bf16", "bench_counter_collection.csv")
    if os.path.exists(sqb):
        summarize_sq(sqb, os.path.join(out_dir, f"{tag}_bf16_pmc_sq.csv"), "python bench.py --segments 32 --dtype bf16")
    # configs[3] (ECO-Full)
    sqf = os.path.join(src, "pmc_sq_full", "bench_counter_collection.csv")
    if os.path.exists(sqf):
        summarize_sq(sqf, os.path.join(out_dir, f"{tag}_full_pmc_sq.csv"), "python bench.py --variant full")
    print("wrote", os.listdir(out_dir))


def summarize_traffic(src, suffix, out_path, cmd):
    traffic = collections.defaultdict(dict)
    for nm, key in (("pmc_fetch" + suffix, "fetch_kib"), ("pmc_write" + suffix, "write_kib")):
        p = os.path.join(src, nm, "bench_counter_collection.csv")
        if not os.path.exists(p):
            continue
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(p)):
            agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            traffic[k][key] = sum(v) / len(v)
            traffic[k][key + "_launches"] = len(v)
    summary = {}
    with open(out_path, "w") as f:
        f.write(f"# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- {cmd} --steps 2 --warmup 1\n")
        f.write("# hbm_bytes_per_launch = 2*FETCH_SIZE*1024 (gfx950 half-count correction) + WRITE_SIZE*1024\n")
        f.write("kernel,avg_FETCH_SIZE_KiB,avg_WRITE_SIZE_KiB,hbm_MB_per_launch_corrected\n")
        for k, v in sorted(traffic.items(), key=lambda kv: -kv[1].get("fetch_kib", 0)):
            fk, wk = v.get("fetch_kib", 0.0), v.get("write_kib", 0.0)
            hbm = 2 * fk * 1024 + wk * 1024
            summary[k] = {"fetch_kib": fk, "write_kib": wk, "hbm_bytes_per_launch": hbm,
                          "launches": int(v.get("fetch_kib_launches", v.get("write_kib_launches", 1)))}
            f.write("%s,%.1f,%.1f,%.2f\n" % (k.replace(",", ";"), fk, wk, hbm / 1e6))
    return summary


SQ_COUNTERS = ("SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY "
               "SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_MFMA")


def summarize_sq(path, out_path, cmd="python bench.py"):
    """Per-kernel shader-engine counters of one `rocprofv3 --pmc <SQ_COUNTERS>` pass.  Derived columns:
    clock = GRBM_GUI_ACTIVE / 8 XCDs / kernel time; MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (per-XCD cycles * 1024
    SIMDs); wait / active = share of SQ_WAVE_CYCLES; avg waves per SIMD = 4 * SQ_WAVE_CYCLES / (cycles * 1024)
    (the SQ wave-cycle counters tick once per 4 clocks)."""
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    dur = collections.defaultdict(float)
    seen = set()
    for r in csv.DictReader(open(path)):
        k = short(r["Kernel_Name"])
        per[k][r["Counter_Name"]] += float(r["Counter_Value"])
        did = r.get("Dispatch_Id", "")
        disp[k].add(did)
        if (k, did) not in seen and r.get("Start_Timestamp") and r.get("End_Timestamp"):
            seen.add((k, did))
            dur[k] += (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e6
    with open(out_path, "w") as f:
        f.write(f"# rocprofv3 --pmc {SQ_COUNTERS} -- {cmd} --steps 2 --warmup 1 --no-cpu-baseline\n")
        f.write("# clock = GRBM_GUI_ACTIVE/8 XCDs / kernel time; MfmaUtil = MFMA_BUSY / (cycles * 1024 SIMDs); "
                "wait/active = fraction of wave cycles\n")
        f.write("kernel,dispatches,total_ms,clock_GHz,MfmaUtil,wait_inst_frac,wait_any_frac,active_frac,"
                "avg_waves_per_SIMD,lds_bank_conflict_cycles,insts_mfma\n")
        for k, c in sorted(per.items(), key=lambda kv: -dur[kv[0]]):
            cyc = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
            wave = c.get("SQ_WAVE_CYCLES", 0.0)
            if cyc <= 0 or wave <= 0:
                continue
            ms = dur[k]
            f.write("%s,%d,%.3f,%.3f,%.3f,%.3f,%.3f,%.3f,%.2f,%.0f,%.0f\n" % (
                k.replace(",", ";"), len(disp[k]), ms, cyc / (ms * 1e6) if ms else 0.0,
                c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (cyc * 1024), c.get("SQ_WAIT_INST_ANY", 0.0) / wave,
                c.get("SQ_WAIT_ANY", 0.0) / wave, c.get("SQ_ACTIVE_INST_ANY", 0.0) / wave, 4.0 * wave / (cyc * 1024),
                c.get("SQ_LDS_BANK_CONFLICT", 0.0), c.get("SQ_INSTS_MFMA", 0.0)))


if __name__ == "__main__":
    main()
