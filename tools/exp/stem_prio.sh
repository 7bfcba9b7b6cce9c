PKG=eco-efficient-video-understanding_amd
cp $PKG/libeco_hip.so /tmp/libeco_hip_orig.so
for v in stemprio0 stemprio1 stemprio3; do
  cp tools/exp/libeco_hip_$v.so $PKG/libeco_hip.so
  for s in 0 4; do
  echo "$v stagger $s: $(ECO_STEM_STAGGER=$s python tools/eco_time.py --iterations 10 2>/dev/null | grep -E 'stem_kernel' | sed 's/.*forward://; s/GFLOP.*//')"
  done
done 2>&1 | tee gpurun_out/exp_stem_prio.txt
cp tools/exp/libeco_hip_stemprio3ts.so $PKG/libeco_hip.so
ECO_STEM_STAGGER=0 python tools/exp/stem_ts.py 2>&1 | grep -v amdgpu.ids | head -4 | tee -a gpurun_out/exp_stem_prio.txt
cp /tmp/libeco_hip_orig.so $PKG/libeco_hip.so
