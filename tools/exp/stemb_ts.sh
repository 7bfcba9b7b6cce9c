PKG=eco-efficient-video-understanding_amd
cp $PKG/libeco_hip.so /tmp/libeco_hip_orig.so
cp tools/exp/libeco_hip_sbts.so $PKG/libeco_hip.so
mkdir -p gpurun_out/ts
timeout 300 python tools/exp/stemb_ts.py > gpurun_out/ts/stemb.txt 2>&1
cp /tmp/libeco_hip_orig.so $PKG/libeco_hip.so
cat gpurun_out/ts/stemb.txt | cut -c1-300
