# as clock_probe.sh, but inside bench.py's back-to-back steps (the chip as warm as the bench line has it)
PKG=eco-efficient-video-understanding_amd
cp $PKG/libeco_hip.so /tmp/libeco_hip_orig.so
cp tools/exp/libeco_hip_clk.so $PKG/libeco_hip.so
mkdir -p gpurun_out/clk
python bench.py --no-cpu-baseline --no-extra-configs --steps 30 --warmup 10 --profile-iters 1 2>/dev/null | grep -E "^CLK|ms_per_step" > gpurun_out/clk/bench_f32.txt
python bench.py --no-cpu-baseline --no-extra-configs --steps 30 --warmup 10 --profile-iters 1 --segments 32 --dtype bf16 2>/dev/null | grep -E "^CLK|ms_per_step" > gpurun_out/clk/bench_bf16.txt
cp /tmp/libeco_hip_orig.so $PKG/libeco_hip.so
python - <<'PY'
import collections, json
for f in ("f32", "bf16"):
    acc = collections.defaultdict(lambda: [0, 0, 0])
    for l in open(f"gpurun_out/clk/bench_{f}.txt"):
        p = l.split()
        if l.startswith("{"):
            print("  bench line:", json.loads(l)["ms_per_step"], "ms per step (probe build)")
            continue
        if len(p) != 4: continue
        _, name, dt, dr = p
        a = acc[name]; a[0] += int(dt); a[1] += int(dr); a[2] += 1
    print("==", f, "(kernel, launches sampled, mean lifetime of workgroup 0 in us, shader clock over it)")
    for k, (dt, dr, n) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        print(f"  {k:12s} {n:5d} {dr / n / 100:9.1f} us  {dt / dr * 0.1:6.3f} GHz")
PY
