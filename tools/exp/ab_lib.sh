#!/bin/bash
# A/B two builds of the library inside ONE gpurun call (boxes differ by a few per cent): tools/exp/ab_lib.sh <variant .so> [eco_time args]
# runs tools/eco_time.py alternately with the product library and with the variant copied over it, twice each, and prints the
# per-family summaries.  The product library is restored at the end (the GPU box works on a scratch copy anyway).
V=$1; shift
P=eco-efficient-video-understanding_amd/libeco_hip.so
mkdir -p gpurun_out/ab
cp $P /tmp/eco_product.so
for r in 1 2; do
  cp /tmp/eco_product.so $P; python tools/eco_time.py --iterations 10 "$@" 2>/dev/null | grep -v amdgpu > gpurun_out/ab/product_$r.txt
  cp $V $P;                  python tools/eco_time.py --iterations 10 "$@" 2>/dev/null | grep -v amdgpu > gpurun_out/ab/variant_$r.txt
done
cp /tmp/eco_product.so $P
for f in product_1 variant_1 product_2 variant_2; do echo "== $f"; python tools/exp/summ_time.py gpurun_out/ab/$f.txt | head -8; done
