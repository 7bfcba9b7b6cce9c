# A/B of one library build: per-layer bf16 timing at configs[4] + the blocked GPU tests
mkdir -p gpurun_out/frag
python tools/eco_time.py --iterations 5 --segments 32 --dtype bf16 2>/dev/null | grep -v amdgpu > gpurun_out/frag/eco_time_bf16.txt
grep Average gpurun_out/frag/eco_time_bf16.txt
grep -E "res3a_2|res4a_2|res4b_1|res5a_2|res5b_1|inception_3a/3x3 |inception_4a/3x3 " gpurun_out/frag/eco_time_bf16.txt | cut -c1-90
timeout 900 python -m pytest tests/test_blocked.py tests/test_siblings.py tests/test_eco_full_size.py -m gpu -x -q 2>&1 | tail -3
ECO_SPANP=0 timeout 600 python -m pytest tests/test_blocked.py -m gpu -x -q -k "span or tail or split" 2>&1 | tail -2
