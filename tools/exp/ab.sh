# A/B of two builds of libeco_hip.so: the product and tools/exp/libeco_hip_$1.so, both dtypes, alternating
PKG=eco-efficient-video-understanding_amd
cp $PKG/libeco_hip.so /tmp/libeco_hip_orig.so
for v in orig $1 orig $1; do
  if [ $v = orig ]; then cp /tmp/libeco_hip_orig.so $PKG/libeco_hip.so; else cp tools/exp/libeco_hip_$v.so $PKG/libeco_hip.so; fi
  python tools/eco_time.py --iterations 10 --segments 32 --dtype bf16 2>/dev/null | grep -v amdgpu > gpurun_out/ab_bf16_$v.txt
  python tools/eco_time.py --iterations 10 2>/dev/null | grep -v amdgpu > gpurun_out/ab_f32_$v.txt
  echo "== $v bf16: $(python tools/exp/summ_time.py gpurun_out/ab_bf16_$v.txt | grep -E 'span|dma|Average' | awk '{print $(NF-1)}' | tr '\n' ' ')  f32: $(python tools/exp/summ_time.py gpurun_out/ab_f32_$v.txt | grep -E 'Average|conv_mfma|output_dm' | awk '{print $(NF-1)}' | tr '\n' ' ')"
done 2>&1 | tee gpurun_out/exp_ab.txt
cp /tmp/libeco_hip_orig.so $PKG/libeco_hip.so
