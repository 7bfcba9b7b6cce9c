#!/bin/bash
# PMC passes over the trunk-layer microbenchmark (tools/exp/w3d_layer.py): what bounds the 3-D Winograd transforms.
set -u
OUT=gpurun_out/${1:-pmc_w3}
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python $PWD/tools/exp/w3d_layer.py ${2:-32}"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE -d $PWD/$OUT/p1 -o w3 --output-format csv -- $CMD > $OUT/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY -d $PWD/$OUT/p2 -o w3 --output-format csv -- $CMD > $OUT/p2.log 2>&1
python - <<PY
import csv, glob, collections
for p in ("p1", "p2"):
    fs = glob.glob("$OUT/%s/**/*counter_collection.csv" % p, recursive=True)
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for f in fs:
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:60]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
            cnt[(k, r["Counter_Name"])] += 1
    for k, d in agg.items():
        if "wino3" in k or "wino_" in k:
            print(p, k, {c: round(v / cnt[(k, c)]) for c, v in d.items()})
PY
find $OUT -name "*agent_info*" -delete
