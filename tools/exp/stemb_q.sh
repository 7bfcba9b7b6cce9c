bash tools/exp/stemb_ts.sh | grep -E "stemb|patch [1-3]" | cut -c1-330 | head -8
timeout 600 python -m pytest tests/test_stemb.py -m gpu -x -q 2>&1 | tail -2
python tools/eco_time.py --iterations 8 --segments 32 --dtype bf16 2>/dev/null | grep -E "Average|conv1_7x7" | cut -c1-120
