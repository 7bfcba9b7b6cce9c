#!/bin/bash
# experiment: stem kernel timing against the start offset of the second workgroup of each CU
mkdir -p gpurun_out/exp
python -m pytest tests/test_stem.py -m gpu -x -q 2>&1 | tail -3
for s in 0 1 2 3 4 6 8 12; do
  echo "stagger $s: $(ECO_STEM_STAGGER=$s python tools/eco_time.py --iterations 10 2>/dev/null | grep -E 'stem_kernel|Average' | sed 's/.*forward://; s/GFLOP.*//' | tr '\n' ' ')"
done 2>&1 | tee gpurun_out/exp/stem_stagger.txt
