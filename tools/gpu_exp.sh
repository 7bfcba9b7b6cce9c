set -u
export TMPDIR=/tmp
O=gpurun_out/wf
mkdir -p $O
timeout 600 python -m pytest tests/test_wgemm.py -m gpu -x -q 2>&1 | tail -2
python bench.py --no-cpu-baseline > $O/bench_f32.json 2> $O/bench_f32.err; echo "bench rc=$?"
python tools/eco_time.py --iterations 5 > $O/time_f32.txt 2>&1
python - <<PY
import json
d=json.load(open("$O/bench_f32.json"))
print(d["value"], d["unit"], d["ms_per_step"])
for k,v in list(d["roofline"].get("per_kernel",{}).items())[:8]: print("   ", k, v)
PY
grep "wfused" $O/time_f32.txt | awk -F'\t' '{print substr($1,1,40), $2}'
