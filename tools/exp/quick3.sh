timeout 900 python -m pytest tests/test_stemb.py tests/test_eco_full_size.py tests/test_reference_logits.py -m gpu -x -q 2>&1 | tail -3
python tools/eco_time.py --iterations 8 --segments 32 --dtype bf16 2>/dev/null | grep -E "Average|conv1_7x7" | cut -c1-120
