set -u
mkdir -p gpurun_out/r2a
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q -s > gpurun_out/r2a/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2a/pytest.log
tail -5 gpurun_out/r2a/pytest.log
python bench.py > gpurun_out/r2a/bench_f32.json 2> gpurun_out/r2a/bench_f32.err; echo "bench f32 rc=$?"
python bench.py --segments 32 --dtype bf16 > gpurun_out/r2a/bench_bf16.json 2> gpurun_out/r2a/bench_bf16.err; echo "bench bf16 rc=$?"
python bench.py --dtype f32x3 --cpu-clips 1 > gpurun_out/r2a/bench_f32x3.json 2> gpurun_out/r2a/bench_f32x3.err; echo "bench f32x3 rc=$?"
python tools/eco_time.py --segments 32 --dtype bf16 --iterations 5 > gpurun_out/r2a/time_bf16.txt 2>&1
python tools/eco_time.py --dtype f32x3 --iterations 5 > gpurun_out/r2a/time_f32x3.txt 2>&1
python tools/eco_time.py --iterations 5 > gpurun_out/r2a/time_f32.txt 2>&1
python bench.py --variant full --no-cpu-baseline > gpurun_out/r2a/bench_full.json 2> gpurun_out/r2a/bench_full.err; echo "bench full rc=$?"
