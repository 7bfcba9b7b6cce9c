# Shader clock per kernel family, measured inside the kernels (probe build -DECO_CLOCK_PROBE: s_memtime against the
# constant 100 MHz s_memrealtime over workgroup 0's lifetime), fp32 configs[1] and bf16 configs[4] steps
PKG=eco-efficient-video-understanding_amd
cp $PKG/libeco_hip.so /tmp/libeco_hip_orig.so
cp tools/exp/libeco_hip_clk.so $PKG/libeco_hip.so
mkdir -p gpurun_out/clk
python tools/eco_time.py --iterations 3 2>/dev/null | grep "^CLK" > gpurun_out/clk/f32.txt
python tools/eco_time.py --iterations 3 --segments 32 --dtype bf16 2>/dev/null | grep "^CLK" > gpurun_out/clk/bf16.txt
cp /tmp/libeco_hip_orig.so $PKG/libeco_hip.so
python - <<'PY'
import collections
for f in ("f32", "bf16"):
    acc = collections.defaultdict(lambda: [0, 0, 0])
    for l in open(f"gpurun_out/clk/{f}.txt"):
        p = l.split()
        if len(p) != 4: continue
        _, name, dt, dr = p
        a = acc[name]; a[0] += int(dt); a[1] += int(dr); a[2] += 1
    print("==", f, "(kernel, launches sampled, mean lifetime of workgroup 0 in us, shader clock over it)")
    for k, (dt, dr, n) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        print(f"  {k:12s} {n:5d} {dr / n / 100:9.1f} us  {dt / dr * 0.1:6.3f} GHz")
PY
