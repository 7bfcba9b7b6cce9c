# persistent span kernel: correctness on the GPU, then per-launch A/B against the per-tile kernel (ECO_SPANP=0)
mkdir -p gpurun_out/spanp
timeout 900 python -m pytest tests/test_blocked.py -m gpu -x -q -k "span or mini or sibling" > gpurun_out/spanp/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/spanp/pytest.log
for v in 1 0 1 0; do
  ECO_SPANP=$v python tools/eco_time.py --iterations 5 --segments 32 --dtype bf16 2>/dev/null | grep -v amdgpu > gpurun_out/spanp/eco_time_bf16_$v.txt
  echo "== ECO_SPANP=$v $(grep Average gpurun_out/spanp/eco_time_bf16_$v.txt | cut -c1-40)"
  grep -E "span" gpurun_out/spanp/eco_time_bf16_$v.txt | sed 's/+[a-z0-9_+]*//; s/forward://; s/GFLOP.*//' | awk '{printf "%s %s | ", $1, $2}'; echo
done
timeout 900 python -m pytest "tests/test_eco_full_size.py::test_eco_lite_c5_bf16_n32" tests/test_reference_logits.py -m gpu -x -q -s > gpurun_out/spanp/pytest2.log 2>&1; echo "pytest2 rc=$?"; grep -E "bf16 N=32|passed|failed" gpurun_out/spanp/pytest2.log | tail -5
