#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   kernel-trace stats, HBM traffic (FETCH_SIZE and WRITE_SIZE in separate --pmc passes), SQ counters,
#   and the un-profiled bench line.  Outputs under gpurun_out/$1; condense with tools/summarize_profiles.py.
set -u
OUT=gpurun_out/${1:-prof}
mkdir -p $OUT
export TMPDIR=/tmp
B="python $PWD/bench.py --no-cpu-baseline --no-extra-configs"
# bench.py reports PMC traffic only while the running library was built from THESE sources (eco_source_digest())
python -c "import sys; sys.path.insert(0, '.'); from eco_amd import hip; print(hip.EcoLib(hip.LIB_PATH).source_digest())" > $OUT/src.sha256
rocprofv3 --kernel-trace --stats -d $PWD/$OUT/trace -o bench --output-format csv -- $B --steps 10 --warmup 3 > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $PWD/$OUT/pmc_fetch -o bench --output-format csv -- $B --steps 2 --warmup 1 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $PWD/$OUT/pmc_write -o bench --output-format csv -- $B --steps 2 --warmup 1 > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_MFMA -d $PWD/$OUT/pmc_sq -o bench --output-format csv -- $B --steps 2 --warmup 1 > $OUT/pmc_sq.log 2>&1
# the other BASELINE configurations: configs[4] (bf16, N=32) with its kernel trace, configs[3] (ECO-Full)
rocprofv3 --kernel-trace --stats -d $PWD/$OUT/trace_bf16 -o bench --output-format csv -- $B --segments 32 --dtype bf16 --steps 10 --warmup 3 > $OUT/trace_bf16.log 2>&1
# ... and its counters: SQ (MFMA busy, LDS bank conflicts, wait / active shares), FETCH_SIZE, WRITE_SIZE in separate passes
BB="$B --segments 32 --dtype bf16 --steps 2 --warmup 1"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_MFMA -d $PWD/$OUT/pmc_sq_bf16 -o bench --output-format csv -- $BB > $OUT/pmc_sq_bf16.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $PWD/$OUT/pmc_fetch_bf16 -o bench --output-format csv -- $BB > $OUT/pmc_fetch_bf16.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $PWD/$OUT/pmc_write_bf16 -o bench --output-format csv -- $BB > $OUT/pmc_write_bf16.log 2>&1
# configs[3] (ECO-Full): kernel trace + the same three counter passes
BF="$B --variant full"
rocprofv3 --kernel-trace --stats -d $PWD/$OUT/trace_full -o bench --output-format csv -- $BF --steps 10 --warmup 3 > $OUT/trace_full.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_MFMA -d $PWD/$OUT/pmc_sq_full -o bench --output-format csv -- $BF --steps 2 --warmup 1 > $OUT/pmc_sq_full.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $PWD/$OUT/pmc_fetch_full -o bench --output-format csv -- $BF --steps 2 --warmup 1 > $OUT/pmc_fetch_full.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $PWD/$OUT/pmc_write_full -o bench --output-format csv -- $BF --steps 2 --warmup 1 > $OUT/pmc_write_full.log 2>&1
# the un-profiled bench lines report `roofline.traffic` from profiles/hbm_traffic_latest.json when that file belongs to the
# sources of the running library and holds PMC passes of the line's workload: condense ALL of this run's PMC passes first
# (into this box's copy of profiles/), then bench
python tools/summarize_profiles.py $OUT ${2:-boxtmp} > /dev/null 2>&1   # (second argument: the tag the summaries will be committed under)
python bench.py > $OUT/bench_line.json 2> $OUT/bench_line.err
python bench.py --segments 32 --dtype bf16 > $OUT/bench_line_bf16.json 2> $OUT/bench_line_bf16.err
python bench.py --variant full > $OUT/bench_line_full.json 2> $OUT/bench_line_full.err
# online recognition: one clip per step, the launch list replayed as a hipGraph and submitted call by call
python bench.py --clips-per-gpu 1 --graph --no-cpu-baseline --steps 200 --warmup 20 > $OUT/bench_line_b1_graph.json 2> $OUT/bench_line_b1_graph.err
python bench.py --clips-per-gpu 1 --no-cpu-baseline --steps 200 --warmup 20 > $OUT/bench_line_b1.json 2> $OUT/bench_line_b1.err
python tools/eco_time.py --iterations 5 > $OUT/eco_time.txt 2>&1
python tools/eco_time.py --iterations 5 --segments 32 --dtype bf16 > $OUT/eco_time_bf16.txt 2>&1
python tools/eco_time.py --iterations 5 --variant full > $OUT/eco_time_full.txt 2>&1
find $OUT -name "*.csv" | head -20
# keep only what the summariser reads (the merge back is capped at 64 MiB)
find $OUT -name "*agent_info*" -delete
ls -la $OUT/*/ 2>/dev/null | head -40
