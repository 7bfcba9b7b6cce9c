# bf16 A/B in one call: the product build against tools/exp/libeco_hip_$1.so, alternating; per-layer table at the end
PKG=eco-efficient-video-understanding_amd
cp $PKG/libeco_hip.so /tmp/libeco_hip_orig.so
mkdir -p gpurun_out/abb
for r in 1 2; do for v in ${ORDER:-orig $1}; do
  if [ $v = orig ]; then cp /tmp/libeco_hip_orig.so $PKG/libeco_hip.so; else cp tools/exp/libeco_hip_$v.so $PKG/libeco_hip.so; fi
  python tools/eco_time.py --iterations 8 --segments 32 --dtype bf16 2>/dev/null | grep -v amdgpu > gpurun_out/abb/${v}_$r.txt
  echo "== $v $r $(grep Average gpurun_out/abb/${v}_$r.txt | cut -c1-40)"
done; done
cp /tmp/libeco_hip_orig.so $PKG/libeco_hip.so
python - "$1" <<'PY'
import re, sys
def load(p):
    d = {}
    for l in open(p):
        m = re.match(r"\s*(.*?)\s+forward:\s+([\d.]+) ms", l)
        if m: d[m.group(1).split('+')[0].split(' ')[0]] = float(m.group(2))
    return d
v = sys.argv[1]
a = [load(f"gpurun_out/abb/orig_{r}.txt") for r in (1, 2)]; b = [load(f"gpurun_out/abb/{v}_{r}.txt") for r in (1, 2)]
for k in a[0]:
    x = min(q[k] for q in a); y = min(q[k] for q in b)
    print(f"{k:34s} product {x:.4f}  {v} {y:.4f}  {100 * (x - y) / y:+.1f}%")
print("sum", sum(min(q[k] for q in a) for k in a[0]), sum(min(q[k] for q in b) for k in a[0]))
PY
