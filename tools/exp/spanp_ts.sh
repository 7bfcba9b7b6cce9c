PKG=eco-efficient-video-understanding_amd
cp $PKG/libeco_hip.so /tmp/libeco_hip_orig.so
cp tools/exp/libeco_hip_ts.so $PKG/libeco_hip.so
mkdir -p gpurun_out/ts
for m in RAND TS_RELU TS_ZERO; do
echo "#### $m" >> gpurun_out/ts/ts6.txt
env $m=1 timeout 300 python tools/exp/spanp_ts.py res3b_1 inc3a_3x3 >> gpurun_out/ts/ts6.txt 2>&1
done
cp /tmp/libeco_hip_orig.so $PKG/libeco_hip.so
grep -E "####|==|block 0: first" gpurun_out/ts/ts6.txt | cut -c1-200
