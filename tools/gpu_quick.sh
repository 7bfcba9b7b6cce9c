# scratch: GPU tests + fp32 / full / bf16 bench + per-launch table (gpurun from the repo root)
set -u
O=gpurun_out/${1:-quick}
mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
python bench.py --no-cpu-baseline > $O/bench_f32.json 2> $O/bench_f32.err; echo "bench f32 rc=$?"
python tools/eco_time.py --iterations 5 > $O/time_f32.txt 2>&1
python bench.py --variant full --no-cpu-baseline > $O/bench_full.json 2> $O/bench_full.err; echo "bench full rc=$?"
python tools/eco_time.py --iterations 5 --variant full > $O/time_full.txt 2>&1
python - <<PY
import json
for f in ("bench_f32", "bench_full"):
    d=json.load(open("$O/%s.json" % f))
    print(f, d["value"], d["unit"], d["ms_per_step"])
    for k,v in d["roofline"].get("per_kernel",{}).items(): print("   ", k, v)
PY
grep -n " | " $O/time_f32.txt | cut -c1-60,170-400 | head
