#!/bin/bash
# PMC passes over tools/exp/w3d_layer.py for the wgemm launches: MFMA busy, waits, LDS conflicts, HBM bytes.
set -u
OUT=gpurun_out/${1:-pmc_gemm}
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python $PWD/tools/exp/w3d_layer.py ${2:-32}"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY -d $PWD/$OUT/p1 -o w3 --output-format csv -- $CMD > $OUT/p1.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $PWD/$OUT/p2 -o w3 --output-format csv -- $CMD > $OUT/p2.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $PWD/$OUT/p3 -o w3 --output-format csv -- $CMD > $OUT/p3.log 2>&1
python - <<PY
import csv, glob, collections
for p in ("p1", "p2", "p3"):
    fs = glob.glob("$OUT/%s/**/*counter_collection.csv" % p, recursive=True)
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for f in fs:
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:50] + " grid=" + r.get("Grid_Size", "?")
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[(k, r["Counter_Name"])] += 1
    for k, d in sorted(agg.items()):
        if "wgemm" in k:
            print(p, k, {c: round(v / cnt[(k, c)]) for c, v in d.items()})
PY
find $OUT -name "*agent_info*" -delete
