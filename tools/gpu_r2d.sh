set -u
O=gpurun_out/r2d
mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q -s > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
python bench.py --no-cpu-baseline > $O/bench_f32.json 2> $O/bench_f32.err; echo "bench f32 rc=$?"
python tools/eco_time.py --iterations 5 > $O/time_f32.txt 2>&1
python bench.py --variant full --no-cpu-baseline > $O/bench_full.json 2> $O/bench_full.err; echo "bench full rc=$?"
