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
    traffic = collections.defaultdict(dict)
    for nm, key in (("pmc_fetch", "fetch_kib"), ("pmc_write", "write_kib")):
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
    with open(os.path.join(out_dir, f"{tag}_pmc_hbm_traffic.csv"), "w") as f:
        f.write("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python bench.py --steps 2 --warmup 1\n")
        f.write("# hbm_bytes_per_launch = 2*FETCH_SIZE*1024 (gfx950 half-count correction) + WRITE_SIZE*1024\n")
        f.write("kernel,avg_FETCH_SIZE_KiB,avg_WRITE_SIZE_KiB,hbm_MB_per_launch_corrected\n")
        for k, v in sorted(traffic.items(), key=lambda kv: -kv[1].get("fetch_kib", 0)):
            fk, wk = v.get("fetch_kib", 0.0), v.get("write_kib", 0.0)
            hbm = 2 * fk * 1024 + wk * 1024
            summary[k] = {"fetch_kib": fk, "write_kib": wk, "hbm_bytes_per_launch": hbm}
            f.write("%s,%.1f,%.1f,%.2f\n" % (k.replace(",", ";"), fk, wk, hbm / 1e6))
    with open(os.path.join(out_dir, "hbm_traffic_latest.json"), "w") as f:
        json.dump({"source": f"profiles/{tag}_pmc_hbm_traffic.csv", "kernels": summary}, f, indent=1)
    print("wrote", os.listdir(out_dir))


if __name__ == "__main__":
    main()
