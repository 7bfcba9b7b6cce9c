set -u
O=gpurun_out/${1:-quick}
mkdir -p $O
export TMPDIR=/tmp
python bench.py --no-cpu-baseline > $O/bench_f32.json 2> $O/bench_f32.err; echo "bench f32 rc=$?"
python tools/eco_time.py --iterations 5 > $O/time_f32.txt 2>&1
python bench.py --variant full --no-cpu-baseline > $O/bench_full.json 2> $O/bench_full.err; echo "bench full rc=$?"
python - <<PY
import json
for f in ("bench_f32", "bench_full"):
    d=json.load(open("$O/%s.json" % f))
    print(f, d["value"], d["unit"], d["ms_per_step"])
    for k,v in list(d["roofline"].get("per_kernel",{}).items())[:12]: print("   ", k, v)
PY
grep "conv_mfma" $O/time_f32.txt | awk -F'\t' '{print substr($1,1,60), $2}'
