PKG=eco-efficient-video-understanding_amd
cp $PKG/libeco_hip.so /tmp/libeco_hip_orig.so
for v in orig sbprobe1 sbprobe2 sbprobe4 sbprobe8 sbprobe3; do
  if [ $v = orig ]; then cp /tmp/libeco_hip_orig.so $PKG/libeco_hip.so; else cp tools/exp/libeco_hip_$v.so $PKG/libeco_hip.so; fi
  echo "$v: $(python tools/eco_time.py --iterations 10 --segments 32 --dtype bf16 2>/dev/null | grep -E 'stemb_kernel' | sed 's/.*forward://; s/ms.*//' | tr '\n' ' ')"
done 2>&1 | tee gpurun_out/exp_sb_probe.txt
cp /tmp/libeco_hip_orig.so $PKG/libeco_hip.so
