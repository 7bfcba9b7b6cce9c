PKG=eco-efficient-video-understanding_amd
cp $PKG/libeco_hip.so /tmp/libeco_hip_orig.so
for v in orig epi; do
  if [ $v = orig ]; then cp /tmp/libeco_hip_orig.so $PKG/libeco_hip.so; else cp tools/exp/libeco_hip_$v.so $PKG/libeco_hip.so; fi
  python tools/eco_time.py --iterations 10 --segments 32 --dtype bf16 2>/dev/null | grep -v amdgpu > gpurun_out/eco_time_bf16_$v.txt
  echo "== $v $(python tools/exp/summ_time.py gpurun_out/eco_time_bf16_$v.txt | grep -E 'span|Average')"
  grep -E "span_kernel" gpurun_out/eco_time_bf16_$v.txt | sed 's/+[a-z0-9_+]*//; s/forward://; s/GFLOP.*//' | awk '{printf "%s %s | ", $1, $2}'; echo
done 2>&1 | tee gpurun_out/exp_span_probe.txt
cp /tmp/libeco_hip_orig.so $PKG/libeco_hip.so
