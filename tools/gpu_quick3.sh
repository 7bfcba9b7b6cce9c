set -u
O=gpurun_out/${1:-quick}
mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
python bench.py --no-cpu-baseline --segments 32 --dtype bf16 > $O/bench_bf16.json 2> $O/bench_bf16.err; echo "bench bf16 rc=$?"
python tools/eco_time.py --iterations 5 --segments 32 --dtype bf16 > $O/time_bf16.txt 2>&1
python bench.py --no-cpu-baseline > $O/bench_f32.json 2> $O/bench_f32.err; echo "bench f32 rc=$?"
python - <<PY
import json
for f in ("bench_bf16", "bench_f32"):
    d=json.load(open("$O/%s.json" % f))
    print(f, d["value"], d["unit"], d["ms_per_step"])
    for k,v in list(d["roofline"].get("per_kernel",{}).items())[:10]: print("   ", k, v)
PY
grep " | " $O/time_bf16.txt | awk -F'\t' '{print substr($1,1,40), $2}'
