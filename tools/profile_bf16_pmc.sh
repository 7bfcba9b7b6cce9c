#!/bin/bash
# rocprofv3 counter passes for the blocked bf16 path (BASELINE.json configs[4], one GPU): SQ counters, FETCH_SIZE and
# WRITE_SIZE in separate --pmc passes (TCC has 4 slots; FETCH_SIZE takes 3), plus the kernel trace of the same command.
#   gpurun -- 'bash tools/profile_bf16_pmc.sh r03_bf16'   ->  gpurun_out/r03_bf16/…  -> tools/summarize_profiles.py --bf16
set -u
OUT=gpurun_out/${1:-bf16_pmc}
mkdir -p $OUT
export TMPDIR=/tmp
B="python $PWD/bench.py --no-cpu-baseline --segments 32 --dtype bf16"
rocprofv3 --kernel-trace --stats -d $PWD/$OUT/trace -o bench --output-format csv -- $B --steps 10 --warmup 3 > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_MFMA -d $PWD/$OUT/pmc_sq -o bench --output-format csv -- $B --steps 2 --warmup 1 > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $PWD/$OUT/pmc_fetch -o bench --output-format csv -- $B --steps 2 --warmup 1 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $PWD/$OUT/pmc_write -o bench --output-format csv -- $B --steps 2 --warmup 1 > $OUT/pmc_write.log 2>&1
python tools/eco_time.py --iterations 5 --segments 32 --dtype bf16 > $OUT/eco_time_bf16.txt 2>&1
find $OUT -name "*agent_info*" -delete
ls -la $OUT/*/ 2>/dev/null | head -40
