# Shader clock per kernel from SQ_BUSY_CYCLES (shader-clock domain) next to GRBM_GUI_ACTIVE, fp32 and bf16 bench steps
export TMPDIR=/tmp
OUT=gpurun_out/clk; mkdir -p $OUT
B="python $PWD/bench.py --no-cpu-baseline --no-extra-configs --steps 2 --warmup 1"
rocprofv3 --pmc SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES -d $PWD/$OUT/f32 -o bench --output-format csv -- $B > $OUT/f32.log 2>&1
rocprofv3 --pmc SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES -d $PWD/$OUT/bf16 -o bench --output-format csv -- $B --segments 32 --dtype bf16 > $OUT/bf16.log 2>&1
python - <<'PY'
import csv, collections, glob
for sub in ("f32", "bf16"):
    p = glob.glob(f"gpurun_out/clk/{sub}/**/bench_counter_collection.csv", recursive=True)
    if not p: print(sub, "no csv"); continue
    per = collections.defaultdict(lambda: collections.defaultdict(float)); dur = collections.defaultdict(float); seen = set()
    for r in csv.DictReader(open(p[0])):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()
        per[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (k, r["Dispatch_Id"])
        if key not in seen:
            seen.add(key); dur[k] += (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3   # us
    print("==", sub, "kernel, ms, SQ_BUSY/us, GRBM/8/us (GHz), MFMA_BUSY/(SQ_BUSY...)")
    for k, c in sorted(per.items(), key=lambda kv: -dur[kv[0]])[:12]:
        us = dur[k]
        print(f"{k[:44]:44s} {us/1e3:8.3f} ms  SQ_BUSY/us {c['SQ_BUSY_CYCLES']/us:10.1f}  GRBM/8 {c['GRBM_GUI_ACTIVE']/8/us/1e3:6.3f} GHz  MFMA_BUSY/us {c['SQ_VALU_MFMA_BUSY_CYCLES']/us:12.1f}")
PY
