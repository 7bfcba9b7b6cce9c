set -u
mkdir -p gpurun_out/r2b
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q -s > gpurun_out/r2b/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2b/pytest.log
tail -4 gpurun_out/r2b/pytest.log
python bench.py --segments 32 --dtype bf16 --no-cpu-baseline > gpurun_out/r2b/bench_bf16.json 2> gpurun_out/r2b/bench_bf16.err; echo "bench bf16 rc=$?"
python tools/eco_time.py --segments 32 --dtype bf16 --iterations 5 > gpurun_out/r2b/time_bf16.txt 2>&1
python bench.py --variant full --no-cpu-baseline > gpurun_out/r2b/bench_full.json 2> gpurun_out/r2b/bench_full.err; echo "bench full rc=$?"
python tools/eco_time.py --variant full --iterations 5 > gpurun_out/r2b/time_full.txt 2>&1
python bench.py --dtype bf16 --no-cpu-baseline > gpurun_out/r2b/bench_bf16_n16.json 2> gpurun_out/r2b/bench_bf16_n16.err; echo "bench bf16 n16 rc=$?"
