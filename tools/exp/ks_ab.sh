PKG=eco-efficient-video-understanding_amd
cp $PKG/libeco_hip.so /tmp/libeco_hip_orig.so
cp tools/exp/libeco_hip_ks.so $PKG/libeco_hip.so
mkdir -p gpurun_out/ks
for v in 1 2 3 4 5 6 8; do
  ECO_CONVB_KSPLIT=$v python tools/eco_time.py --iterations 5 --segments 32 --dtype bf16 2>/dev/null | grep -v amdgpu > gpurun_out/ks/eco_time_bf16_$v.txt
  echo "== ksplit $v $(grep Average gpurun_out/ks/eco_time_bf16_$v.txt | cut -c1-40)"
  grep -E "res4|res5" gpurun_out/ks/eco_time_bf16_$v.txt | sed 's/+[a-z0-9_+]*//; s/forward://; s/GFLOP.*//' | awk '{printf "%s %s | ", $1, $2}'; echo
done
cp /tmp/libeco_hip_orig.so $PKG/libeco_hip.so
