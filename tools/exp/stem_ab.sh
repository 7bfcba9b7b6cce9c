PKG=eco-efficient-video-understanding_amd
cp $PKG/libeco_hip.so /tmp/libeco_hip_orig.so
for r in 1 2 3; do for v in orig "$@"; do
  if [ $v = orig ]; then cp /tmp/libeco_hip_orig.so $PKG/libeco_hip.so; else cp tools/exp/libeco_hip_$v.so $PKG/libeco_hip.so; fi
  echo "$v $r $(python tools/eco_time.py --iterations 8 --segments 32 --dtype bf16 2>/dev/null | grep -E 'conv1_7x7' | sed 's/.*forward://; s/GFLOP.*//')"
done; done
cp /tmp/libeco_hip_orig.so $PKG/libeco_hip.so
