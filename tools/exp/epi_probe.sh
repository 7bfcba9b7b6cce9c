PKG=eco-efficient-video-understanding_amd
cp $PKG/libeco_hip.so /tmp/libeco_hip_orig.so
for v in orig epic epicw; do
  if [ $v = orig ]; then cp /tmp/libeco_hip_orig.so $PKG/libeco_hip.so; else cp tools/exp/libeco_hip_$v.so $PKG/libeco_hip.so; fi
  python tools/eco_time.py --iterations 10 2>/dev/null | grep -v amdgpu > gpurun_out/eco_time_$v.txt
  echo "== $v"; python tools/exp/summ_time.py gpurun_out/eco_time_$v.txt | head -12
done 2>&1 | tee gpurun_out/exp_epi_probe.txt
cp /tmp/libeco_hip_orig.so $PKG/libeco_hip.so
