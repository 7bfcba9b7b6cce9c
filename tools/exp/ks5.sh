PKG=eco-efficient-video-understanding_amd
cp $PKG/libeco_hip.so /tmp/libeco_hip_orig.so
cp tools/exp/libeco_hip_ks.so $PKG/libeco_hip.so
for v in 2 3 4 5 6 7 8 5; do
  echo "== ksplit $v $(ECO_CONVB_KSPLIT=$v python tools/eco_time.py --iterations 6 --segments 32 --dtype bf16 2>/dev/null | grep -E '^ *res5(a_2|b_1|b_2)' | sed 's/+[a-z0-9_+]*//; s/forward://; s/GFLOP.*//' | awk '{printf "%s %s | ", $1, $2}')"
done
cp /tmp/libeco_hip_orig.so $PKG/libeco_hip.so
