# the un-profiled bench lines of a profile round, again (after a bench.py change): into gpurun_out/$1
OUT=gpurun_out/${1:-lines}; mkdir -p $OUT
python bench.py > $OUT/bench_line.json 2> $OUT/bench_line.err
python bench.py --segments 32 --dtype bf16 > $OUT/bench_line_bf16.json 2> $OUT/bench_line_bf16.err
python bench.py --variant full > $OUT/bench_line_full.json 2> $OUT/bench_line_full.err
python bench.py --clips-per-gpu 1 --graph --no-cpu-baseline --steps 200 --warmup 20 > $OUT/bench_line_b1_graph.json 2> $OUT/bench_line_b1_graph.err
python bench.py --clips-per-gpu 1 --no-cpu-baseline --steps 200 --warmup 20 > $OUT/bench_line_b1.json 2> $OUT/bench_line_b1.err
for f in bench_line bench_line_bf16 bench_line_full bench_line_b1_graph bench_line_b1; do python -c "
import json
d=json.loads(open('$OUT/$f.json').read().strip().splitlines()[-1]); r=d['roofline']
print('$f', d['value'], d['ms_per_step'], r['kernel'], r['frac'], r.get('step_frac'), 'traffic', r.get('traffic'), r.get('traffic_unit','')[:70])
"; done
