PKG=eco-efficient-video-understanding_amd
cp $PKG/libeco_hip.so /tmp/libeco_hip_orig.so
mkdir -p gpurun_out/epi2
for v in orig pp16 ppst orig; do
  if [ $v = orig ]; then cp /tmp/libeco_hip_orig.so $PKG/libeco_hip.so; else cp tools/exp/libeco_hip_$v.so $PKG/libeco_hip.so; fi
  python tools/eco_time.py --iterations 5 --segments 32 --dtype bf16 2>/dev/null | grep -v amdgpu > gpurun_out/epi2/eco_time_bf16_$v.txt
  echo "== $v $(grep Average gpurun_out/epi2/eco_time_bf16_$v.txt | cut -c1-40)"
  grep -E "spanp" gpurun_out/epi2/eco_time_bf16_$v.txt | sed 's/+[a-z0-9_+]*//; s/forward://; s/GFLOP.*//' | awk '{printf "%s %s | ", $1, $2}'; echo
done
cp /tmp/libeco_hip_orig.so $PKG/libeco_hip.so
