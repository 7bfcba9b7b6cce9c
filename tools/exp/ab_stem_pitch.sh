# A/B of the fp32 stem's pooling-stage pitch (tools/exp/libeco_hip_p29.so / _p39.so built by build_variant.sh)
for v in p29 p39 p29 p39; do
  cp tools/exp/libeco_hip_$v.so eco-efficient-video-understanding_amd/libeco_hip.so
  echo VARIANT=$v
  python tools/eco_time.py --iterations 10 2>&1 | grep -E "stem_kernel|Average" | sed -E "s/.*forward: +([0-9.]+) ms.*/stem \1/; s/Average Forward pass: ([0-9.]+) ms.*/step \1/"
done
