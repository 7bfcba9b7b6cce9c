PKG=eco-efficient-video-understanding_amd
cp $PKG/libeco_hip.so /tmp/libeco_hip_orig.so
cp tools/exp/libeco_hip_clk2.so $PKG/libeco_hip.so
mkdir -p gpurun_out/clk
python bench.py --no-cpu-baseline --no-extra-configs --steps 6 --warmup 4 --profile-iters 1 --segments 32 --dtype bf16 2>/dev/null | grep -E "^CLKB" > gpurun_out/clk/bench2_bf16.txt
cp /tmp/libeco_hip_orig.so $PKG/libeco_hip.so
tail -150 gpurun_out/clk/bench2_bf16.txt | grep spanp | tail -51 | head -51
