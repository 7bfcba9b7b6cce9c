#!/bin/bash
# A/B two library builds on bench.py lines in ONE gpurun call: tools/exp/ab_bench.sh <variant .so> <bench args...>
V=$1; shift
P=eco-efficient-video-understanding_amd/libeco_hip.so
cp $P /tmp/eco_product.so
for r in 1 2; do
  cp /tmp/eco_product.so $P; python bench.py --no-cpu-baseline --no-extra-configs "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('product', d['ms_per_step'])"
  cp $V $P;                  python bench.py --no-cpu-baseline --no-extra-configs "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('variant', d['ms_per_step'])"
done
cp /tmp/eco_product.so $P
