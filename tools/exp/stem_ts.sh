PKG=eco-efficient-video-understanding_amd
cp $PKG/libeco_hip.so /tmp/libeco_hip_orig.so
cp tools/exp/libeco_hip_stemprobe4.so $PKG/libeco_hip.so
for s in 0; do echo "== stagger $s"; ECO_STEM_STAGGER=$s python tools/exp/stem_ts.py 2>&1 | grep -v amdgpu.ids | head -3; done | tee gpurun_out/exp_stem_ts.txt
cp tools/exp/libeco_hip_stemprobe2.so $PKG/libeco_hip.so
echo "probe2 (no epilogue): $(ECO_STEM_STAGGER=0 python tools/eco_time.py --iterations 10 2>/dev/null | grep -E 'stem_kernel' | sed 's/.*forward://; s/GFLOP.*//')"
cp /tmp/libeco_hip_orig.so $PKG/libeco_hip.so
