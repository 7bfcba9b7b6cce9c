# Per-launch times of the bf16 path (configs[4] size) for probe builds of convb_span_kernel (ECO_SPAN_PROBE bits:
# 2 no DMA after the prologue, 4 no tap masks, 8 no fragment reads, 16 no epilogue, 32 no MFMAs).
PKG=eco-efficient-video-understanding_amd
cp $PKG/libeco_hip.so /tmp/libeco_hip_orig.so
mkdir -p gpurun_out/sp2
for v in orig sp2 sp4 sp8 sp16 sp32 sp48 sp58 orig; do
  if [ $v = orig ]; then cp /tmp/libeco_hip_orig.so $PKG/libeco_hip.so; else cp tools/exp/libeco_hip_$v.so $PKG/libeco_hip.so; fi
  python tools/eco_time.py --iterations 5 --segments 32 --dtype bf16 2>/dev/null | grep -v amdgpu > gpurun_out/sp2/eco_time_bf16_$v.txt
  echo "== $v $(grep Average gpurun_out/sp2/eco_time_bf16_$v.txt | cut -c1-40)"
  grep -E "span_kernel" gpurun_out/sp2/eco_time_bf16_$v.txt | sed 's/+[a-z0-9_+]*//; s/forward://; s/GFLOP.*//' | awk '{printf "%s %s | ", $1, $2}'; echo
done 2>&1 | tee gpurun_out/sp2/summary.txt
cp /tmp/libeco_hip_orig.so $PKG/libeco_hip.so
