# quick GPU check: selected tests + per-launch times of configs[1]
python -m pytest tests/test_wgemm.py tests/test_stem.py tests/test_pool_commute.py tests/test_siblings.py tests/test_kernels.py -m gpu -x -q 2>&1 | tail -2
python tools/eco_time.py --iterations 10 2>/dev/null | grep -v amdgpu > gpurun_out/eco_time_quick.txt
python tools/exp/summ_time.py gpurun_out/eco_time_quick.txt
