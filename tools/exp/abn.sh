# bf16 A/B/n in one call: the product build ("orig") and tools/exp/libeco_hip_<v>.so for every v given, two rounds, per-layer minima
PKG=eco-efficient-video-understanding_amd
cp $PKG/libeco_hip.so /tmp/libeco_hip_orig.so
mkdir -p gpurun_out/abn
for r in 1 2; do for v in orig "$@"; do
  if [ $v = orig ]; then cp /tmp/libeco_hip_orig.so $PKG/libeco_hip.so; else cp tools/exp/libeco_hip_$v.so $PKG/libeco_hip.so; fi
  python tools/eco_time.py --iterations 8 --segments 32 --dtype bf16 2>/dev/null | grep -v amdgpu > gpurun_out/abn/${v}_$r.txt
  echo "== $v $r $(grep Average gpurun_out/abn/${v}_$r.txt | cut -c1-40)"
done; done
cp /tmp/libeco_hip_orig.so $PKG/libeco_hip.so
python - orig "$@" <<'PY'
import re, sys
def load(p):
    d = {}
    for l in open(p):
        m = re.match(r"\s*(.*?)\s+forward:\s+([\d.]+) ms", l)
        if m: d[m.group(1).split('+')[0].split(' ')[0]] = float(m.group(2))
    return d
vs = sys.argv[1:]
t = {v: [load(f"gpurun_out/abn/{v}_{r}.txt") for r in (1, 2)] for v in vs}
keys = list(t["orig"][0])
print(f"{'':34s}" + "".join(f"{v:>10s}" for v in vs))
for k in keys:
    print(f"{k:34s}" + "".join(f"{min(q[k] for q in t[v]):10.4f}" for v in vs))
print(f"{'sum':34s}" + "".join(f"{sum(min(q[k] for q in t[v]) for k in keys):10.4f}" for v in vs))
PY
