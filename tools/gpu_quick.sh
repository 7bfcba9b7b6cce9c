# scratch: stem tests + fp32 bench + per-launch table (gpurun from the repo root)
set -u
O=gpurun_out/${1:-quick}
mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests/test_stem.py tests/test_eco_full_size.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
python bench.py --no-cpu-baseline > $O/bench_f32.json 2> $O/bench_f32.err; echo "bench f32 rc=$?"
python tools/eco_time.py --iterations 5 > $O/time_f32.txt 2>&1
python - <<PY
import json
d=json.load(open("$O/bench_f32.json"))
print(d["value"], d["unit"], d["ms_per_step"])
for k,v in d["roofline"].get("per_kernel",{}).items(): print(k, v)
PY
grep -n "stem_kernel\|conv1_7x7" $O/time_f32.txt | head
