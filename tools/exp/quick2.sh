mkdir -p gpurun_out/q2
python tools/eco_time.py --iterations 8 --segments 32 --dtype bf16 2>/dev/null | grep -v amdgpu > gpurun_out/q2/eco_time_bf16.txt
grep Average gpurun_out/q2/eco_time_bf16.txt
grep -E "^ *(conv1_7x7|pool2|res5[ab]_[12]|res5a_down)" gpurun_out/q2/eco_time_bf16.txt | sed 's/+[a-z0-9_+]*//; s/forward://; s/GFLOP.*//' | awk '{printf "%s %s | ", $1, $2}'; echo
timeout 900 python -m pytest tests/test_stemb.py tests/test_blocked.py tests/test_siblings.py tests/test_eco_full_size.py tests/test_reference_logits.py tests/test_advice_r3.py -m gpu -x -q 2>&1 | tail -3
