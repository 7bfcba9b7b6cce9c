PKG=eco-efficient-video-understanding_amd
cp $PKG/libeco_hip.so /tmp/libeco_hip_orig.so
cp tools/exp/libeco_hip_ts.so $PKG/libeco_hip.so
mkdir -p gpurun_out/ts
for cu in 0; do
echo "#### num_cu $cu" >> gpurun_out/ts/ts4.txt
TS_NUM_CU=$cu timeout 300 python tools/exp/spanp_ts.py inc3a_3x3 conv2_3x3 res3b_1 >> gpurun_out/ts/ts4.txt 2>&1
done
cp /tmp/libeco_hip_orig.so $PKG/libeco_hip.so
cat gpurun_out/ts/ts4.txt | cut -c1-20,300-520
