# Per-launch times of the bf16 path for probe builds of convb_spanp_kernel (ECO_SPANP_PROBE bits: 1 no barrier, 2 no DMA
# after the prologue, 4 no tap masks, 8 no fragment reads, 16 no epilogue, 32 no MFMAs, 64 no per-item index arithmetic).
PKG=eco-efficient-video-understanding_amd
cp $PKG/libeco_hip.so /tmp/libeco_hip_orig.so
mkdir -p gpurun_out/pp
for v in orig pp1 pp2 pp4 pp8 pp16 pp32 pp48 pp64 pp122 orig; do
  if [ $v = orig ]; then cp /tmp/libeco_hip_orig.so $PKG/libeco_hip.so; else cp tools/exp/libeco_hip_$v.so $PKG/libeco_hip.so; fi
  python tools/eco_time.py --iterations 5 --segments 32 --dtype bf16 2>/dev/null | grep -v amdgpu > gpurun_out/pp/eco_time_bf16_$v.txt
  echo "== $v $(grep Average gpurun_out/pp/eco_time_bf16_$v.txt | cut -c1-40)"
  grep -E "span" gpurun_out/pp/eco_time_bf16_$v.txt | sed 's/+[a-z0-9_+]*//; s/forward://; s/GFLOP.*//' | awk '{printf "%s %s | ", $1, $2}'; echo
done
cp /tmp/libeco_hip_orig.so $PKG/libeco_hip.so
