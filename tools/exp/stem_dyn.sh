for r in 1 2 3; do for d in 0 1; do
  echo "dynamic=$d $r $(ECO_STEMB_DYNAMIC=$d python tools/eco_time.py --iterations 8 --segments 32 --dtype bf16 2>/dev/null | grep -E 'conv1_7x7' | sed 's/.*forward://; s/GFLOP.*//')"
done; done
timeout 600 python -m pytest tests/test_stemb.py tests/test_eco_full_size.py -m gpu -x -q 2>&1 | tail -2
