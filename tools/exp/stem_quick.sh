python -m pytest tests/test_stem.py -m gpu -x -q 2>&1 | tail -2
for s in 0 4; do
echo "stagger $s: $(ECO_STEM_STAGGER=$s python tools/eco_time.py --iterations 10 2>/dev/null | grep -E 'stem_kernel|Average' | sed 's/.*forward://; s/GFLOP.*//' | tr '\n' ' ')"
done
