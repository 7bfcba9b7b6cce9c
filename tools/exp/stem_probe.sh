#!/bin/bash
# where the stem's time goes: the kernel without its reduction (1), without its epilogue (2), without both (3)
PKG=eco-efficient-video-understanding_amd
cp $PKG/libeco_hip.so /tmp/libeco_hip_orig.so
for v in orig stemprobe1 stemprobe2 stemprobe3; do
  if [ $v = orig ]; then cp /tmp/libeco_hip_orig.so $PKG/libeco_hip.so; else cp tools/exp/libeco_hip_$v.so $PKG/libeco_hip.so; fi
  echo "$v: $(ECO_STEM_STAGGER=0 python tools/eco_time.py --iterations 10 2>/dev/null | grep -E 'stem_kernel' | sed 's/.*forward://; s/GFLOP.*//')"
done 2>&1 | tee gpurun_out/exp_stem_probe.txt
cp /tmp/libeco_hip_orig.so $PKG/libeco_hip.so
