# GPU-box check used during development (through gpurun, from the repo root): the -m gpu suite, then bench lines at
# 1 / 4 / 32 clips (ECO-Lite) and 32 clips (ECO-Full), and the one-clip per-launch table.  Usage: bash tools/gpu_check.sh <tag>
set -u
O=gpurun_out/${1:-quick}
mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
python bench.py --clips-per-gpu 1 --no-cpu-baseline --steps 200 --warmup 20 > $O/bench_b1.json 2> $O/bench_b1.err
python bench.py --clips-per-gpu 1 --graph --no-cpu-baseline --steps 200 --warmup 20 > $O/bench_b1g.json 2> $O/bench_b1g.err
python bench.py --clips-per-gpu 4 --no-cpu-baseline --steps 100 --warmup 20 > $O/bench_b4.json 2> $O/bench_b4.err
python bench.py --no-cpu-baseline > $O/bench_f32.json 2> $O/bench_f32.err
python bench.py --variant full --no-cpu-baseline > $O/bench_full.json 2> $O/bench_full.err
python tools/eco_time.py --iterations 20 --clips 1 > $O/time_b1.txt 2>&1
python - <<PY
import json
for f in ("bench_b1", "bench_b1g", "bench_b4", "bench_f32", "bench_full"):
    d=json.load(open("$O/%s.json" % f))
    print(f, d["value"], d["unit"], d["ms_per_step"])
PY
