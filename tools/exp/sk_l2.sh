export TMPDIR=/tmp
mkdir -p gpurun_out/sk_l2
for v in sk split; do
  if [ $v = split ]; then export ECO_NO_STREAMK=1; else unset ECO_NO_STREAMK; fi
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum -d $PWD/gpurun_out/sk_l2/$v -o t --output-format csv -- python tools/eco_time.py --iterations 1 > gpurun_out/sk_l2/$v.log 2>&1
  python - <<PY
import csv, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
import glob
f = glob.glob("gpurun_out/sk_l2/$v/**/*counter_collection.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0]
    if "streamk" in k or "conv_mfma" in k or "splitk" in k:
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, c in agg.items():
    h, m = c.get("TCC_HIT_sum", 0), c.get("TCC_MISS_sum", 0)
    print("$v", k[:60], "hit %.3g miss %.3g hitrate %.3f rdreq %.3g" % (h, m, h / max(h + m, 1), c.get("TCC_EA0_RDREQ_sum", 0)))
PY
done
find gpurun_out/sk_l2 -name "*.csv" -delete
