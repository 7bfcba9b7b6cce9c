mkdir -p gpurun_out/lean
python tools/eco_time.py --iterations 5 --segments 32 --dtype bf16 2>/dev/null | grep -v amdgpu > gpurun_out/lean/eco_time_bf16.txt
grep Average gpurun_out/lean/eco_time_bf16.txt
sed 's/forward://; s/GFLOP.*//' gpurun_out/lean/eco_time_bf16.txt | grep " ms" | awk '{n=$1; sub(/\+.*/,"",n); printf "%s %s | ", n, $(NF-3)}'; echo
timeout 900 python -m pytest tests/test_blocked.py tests/test_siblings.py tests/test_eco_full_size.py tests/test_reference_logits.py tests/test_advice_r3.py -m gpu -x -q 2>&1 | tail -3
