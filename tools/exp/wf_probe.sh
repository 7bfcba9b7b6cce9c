PKG=eco-efficient-video-understanding_amd
cp $PKG/libeco_hip.so /tmp/libeco_hip_orig.so
for v in orig wfh3 wfh4 wfh5; do
  if [ $v = orig ]; then cp /tmp/libeco_hip_orig.so $PKG/libeco_hip.so; else cp tools/exp/libeco_hip_$v.so $PKG/libeco_hip.so; fi
  echo "$v: $(python tools/eco_time.py --iterations 10 2>/dev/null | grep -E 'wfused_kernel' | sed 's/.*forward://; s/ms.*//' | tr '\n' ' ')"
done 2>&1 | tee gpurun_out/exp_wf_probe.txt
cp /tmp/libeco_hip_orig.so $PKG/libeco_hip.so
