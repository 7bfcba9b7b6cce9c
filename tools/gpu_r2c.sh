set -u
mkdir -p gpurun_out/r2c
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q -s > gpurun_out/r2c/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2c/pytest.log
tail -4 gpurun_out/r2c/pytest.log
python bench.py --segments 32 --dtype bf16 --no-cpu-baseline > gpurun_out/r2c/bench_bf16.json 2> gpurun_out/r2c/bench_bf16.err; echo "bench bf16 rc=$?"
python tools/eco_time.py --segments 32 --dtype bf16 --iterations 5 > gpurun_out/r2c/time_bf16.txt 2>&1
